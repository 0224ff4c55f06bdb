"""Experiment: the audio front-end of step k+1 as a side branch of step k's graph (features double-buffered), so that
the MFCC kernel is never in front of anything.  Prints ms/step beside the standard replayed step."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
from var_amd._lib import ptr
from var_amd.layout import N_PARAMS
B, HW = 256, 84
cfg = types.SimpleNamespace(img_dim=(3, HW, HW), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
pool = var_amd.SyntheticTripletPool(4096, hw=HW, seed=0, clips_per_class=64).freeze_pairs()
rows = 64
table = pool.index_table(B, rows, drop_last=True)[:rows].contiguous()

def timeit(step, n=300, warm=30):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

m = var_amd.VARPretextNet(cfg).to("cuda"); tr = var_amd.VARTrainer(m)
replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
print("standard replayed step      %.4f ms" % timeit(replay))
l_std = [float(replay().item()) for _ in range(3)]

m2 = var_amd.VARPretextNet(cfg).to("cuda"); tr2 = var_amd.VARTrainer(m2)
c, flat, dev = tr2.ctx, m2.flat_parameters(), tr2.dev
row_ints = 5 * B
tr2._g_table = table.clone(); tr2._g_cursor = torch.zeros(1, dtype=torch.int32, device=dev)
tr2._g_idx = torch.zeros(2 * row_ints, dtype=torch.int32, device=dev); tr2._device_scalars(B)
cur, nxt = tr2._g_idx[:row_ints], tr2._g_idx[row_ints:]
F = [torch.zeros(2 * B, 1, 100, 40, device=dev) for _ in range(2)]
adam = tr2._body_adam(tr2._g_table, rows, row_ints, 1)
images, pcm = pool.images, pool.clips
def front(dst):
    c.check(c.lib.var_mfcc(c.handle, c.stream(), ptr(pcm), ptr(nxt[3 * B:5 * B]), ptr(nxt[B:3 * B]), 2 * B, pcm.stride(0), 100, ptr(dst)), "var_mfcc")
def grad(src):
    tr2._bind()
    c.check(c.lib.var_arm_loss_grad_gather(c.handle, c.stream(), ptr(flat), ptr(images), 1, images.stride(0), ptr(cur[:B]), ptr(src), ptr(src[B:]),
                                           B, HW, 1.0, 1.0 / B, ptr(tr2.gbuf), tr2._loss_ptr(), None), "gather")
c.ensure_plan(B, HW)
side = torch.cuda.Stream()
def make(par):
    def body():
        s = torch.cuda.current_stream()
        side.wait_stream(s)
        with torch.cuda.stream(side):
            front(F[1 - par])
        grad(F[par])
        s.wait_stream(side)
        adam()
    return body
front(F[0])
graphs = [c.capture([[make(p)]])[0] for p in (0, 1)]
cur.copy_(table[0]); nxt.copy_(table[0]); front(F[0]); nxt.copy_(table[1]); tr2._g_cursor.zero_()
st = {"k": 0}
def step():
    graphs[st["k"] & 1](); st["k"] += 1
    return tr2.loss
l_pipe = [float(step().item()) for _ in range(3)]
print("front-end one step ahead    %.4f ms" % timeit(step, n=300, warm=29))   # keep parity/cursor consistent: 3 + 29 + 300
print("first losses standard", l_std, "pipelined", l_pipe)

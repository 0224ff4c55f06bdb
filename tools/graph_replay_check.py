"""The replay sequences that used to fail while the library zeroed buffers with hipMemsetAsync (memset nodes of a captured
graph lost their fill pattern across a synchronise on ROCm 7.2 / MI355X; see var_amd._lib.new_graph and
csrc/var_common.h: var_zero_async).

    python tools/graph_replay_check.py            # on a GPU box

Eight fresh processes per mode; in each, three iTHOR trainers in a row capture their step (plain torch.cuda.CUDAGraph) over the
same batch from the same weights and replay it three times, so every loss triple must be [0.9914, 0.9287, 0.9312]:
  sync      a hipStreamSynchronize(NULL) between the first and the second replay
  devsync   a torch.cuda.synchronize() there
  destroy   no synchronise; the older trainers' graphs are destroyed by reference counting while a later one is in use
With the memset nodes about half of the "sync" / "devsync" processes printed 1.0 (= the margin: the initial GRU state was
garbage, the gradient zero) from the second replay on; now all print the same triple."""
import ctypes
import os
import subprocess
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) == 1:
    for mode in ("sync", "devsync", "destroy"):
        res = set()
        for _ in range(8):
            out = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
            res.add(out.stdout.strip().replace("\n", " | ") if not out.returncode else "ERROR " + out.stderr.strip()[-200:])
        print("%-8s" % mode, "->", "8 x " + res.pop() if len(res) == 1 else "DIFFERENT RESULTS:\n  " + "\n  ".join(sorted(res)))
    sys.exit(0)

import torch
import var_amd

mode = sys.argv[1]
hip = ctypes.CDLL("libamdhip64.so")
cfg = types.SimpleNamespace(img_dim=(3, 96, 96), sound_dim=(1, 600, 40), representationDim=3)
pool = var_amd.SyntheticTripletPool(22, hw=96, seed=5, clips_per_class=3, n_samples=24000, ragged_lens=True).freeze_pairs()
B = 8
row = pool.index_table(B, 1, drop_last=True)[0]
img, pcm, lens = pool.images[row[:B].long()].contiguous(), pool.clips[row[B:3 * B].long()].contiguous(), row[3 * B:5 * B].contiguous()
torch.manual_seed(977)
sd = var_amd.IthorVARPretextNet(cfg).state_dict()
for k in range(3):
    m = var_amd.IthorVARPretextNet(cfg)
    m.load_state_dict(sd)
    m = m.to("cuda")
    tr = var_amd.IthorTrainer(m, lr=1e-3)
    replay = tr.capture_step(img.clone(), pcm.clone(), lens.clone(), _ctx=tr.ctx)     # (the warm-up without an optimiser step)
    out = [round(float(replay().item()), 4)]
    if mode == "sync":
        hip.hipStreamSynchronize(None)
    elif mode == "devsync":
        torch.cuda.synchronize()
    out += [round(float(replay().item()), 4) for _ in range(2)]
    print(out)

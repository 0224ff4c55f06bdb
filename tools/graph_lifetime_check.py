"""Why the package captures with var_amd._lib.new_graph() (keep_graph=True, graphs retired instead of destroyed).

    python tools/graph_lifetime_check.py            # on a GPU box

Three iTHOR trainers in a row in one process, each capturing its step over the same batch, then a
hipStreamSynchronize(NULL) (what var_ithor_plan and other set-up calls of the C ABI issue), then three replays; every
trainer starts from the same weights, so all loss triples must be identical.
  shipped         new_graph(): keep_graph=True + process-lifetime registry
  plain+sync      torch.cuda.CUDAGraph() kept alive, with the NULL-stream synchronise
  plain+destroy   torch.cuda.CUDAGraph(), older graphs destroyed by reference counting once a later one exists, no synchronise
  keep+destroy    keep_graph=True but not retained
Measured on ROCm 7.2 / torch 2.10 / MI355X: the two "plain" modes print loss 1.0 (= the margin: zero gradients) or a drifting
third value; "shipped" prints three equal triples."""
import ctypes
import os
import subprocess
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) == 1:
    for mode in ("shipped", "plain+sync", "plain+destroy", "keep+destroy"):
        out = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
        print("%-14s" % mode, "->", out.stdout.strip().replace("\n", " | "), out.stderr.strip()[-300:] if out.returncode else "")
    sys.exit(0)

import torch
import var_amd
from var_amd import _lib, ithor

mode = sys.argv[1]
if mode.startswith("plain"):
    ithor.new_graph = torch.cuda.CUDAGraph
elif mode.startswith("keep"):
    ithor.new_graph = lambda: torch.cuda.CUDAGraph(keep_graph=True)
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
    if not mode.endswith("destroy"):
        hip.hipStreamSynchronize(None)
    print([round(float(replay().item()), 4) for _ in range(3)])

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
pool = var_amd.SyntheticTripletPool(2048, hw=84, seed=0, clips_per_class=32).freeze_pairs()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
i, c, l = pool.next_batch_indices(B)
for _ in range(5): var_amd.mfcc(pool.clips, l, 100, c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): var_amd.mfcc(pool.clips, l, 100, c)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("VAR_HIP_LIB", "default"), "mfcc us/launch (incl. ~5 us of torch/ctypes per call):", e0.elapsed_time(e1) * 1e3 / 50)

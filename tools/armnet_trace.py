#!/usr/bin/env python3
"""Actor-critic forward (BASELINE config 5, B = 8) as a replayed graph, for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace -d gpurun_out/prof_arm -o arm -f csv -- python3 tools/armnet_trace.py run
    python tools/armnet_trace.py show gpurun_out/prof_arm      # per-launch durations of the last replay, in order"""
import csv, glob, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import var_amd

    class Box:
        def __init__(self, n):
            self.shape = (n,)
    acfg = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
    torch.manual_seed(453)
    B = 8
    ac = var_amd.ArmNetPolicy(None, Box(2), config=acfg, base='arm_VAR',
                              base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512,
                                           'actionHiddenSize': 128}).to("cuda")
    obs = {'image': torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda"),
           'image_feat': torch.randn(B, 3, device="cuda"), 'robot_pose': torch.randn(B, 2, device="cuda"),
           'goal_sound_feat': torch.randn(B, 3, device="cuda")}
    hxs, masks = torch.zeros(B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
    g = var_amd._lib.new_graph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ac._base_forward(obs, hxs, masks)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            ac._base_forward(obs, hxs, masks)
    torch.cuda.synchronize()
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()


def show(d, n_avg=40):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    ends = [i for i, n in enumerate(names) if "armnet_chain" in n]          # the chain kernel closes a forward
    steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])][-n_avg:]
    steps = [st for st in steps if len(st) == len(steps[-1])]
    k = len(steps[-1])
    mean = lambda xs: sorted(xs)[len(xs) // 2]           # the median: a replay that waited for the profiler's flush does not count
    start = [mean([(int(st[i]["Start_Timestamp"]) - int(st[0]["Start_Timestamp"])) / 1e3 for st in steps]) for i in range(k)]
    dur = [mean([(int(st[i]["End_Timestamp"]) - int(st[i]["Start_Timestamp"])) / 1e3 for st in steps]) for i in range(k)]
    for i in range(k):
        r = steps[-1][i]
        print(f"{start[i]:8.1f} {dur[i]:7.1f}  grid {r.get('Grid_Size_X', '?'):>7}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}  {r['Kernel_Name'][:110]}")
    span = mean([(int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])) / 1e3 for st in steps])
    print(f"span {span:.1f} us, {k} launches (medians of the last {len(steps)} replays)")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])

#!/usr/bin/env python3
"""bench.py -- pretext triplets/sec of the VAR contrastive-pretext step on MI355X.

One "step" = one pass of the hot path over one batch of synthetic triplets already resident in HBM:
  gather (u8 image 84x84x3, 2 x int16 16 kHz/1 s clips) -> MFCC front-end (HIP) -> image CNN + sound CNN
  + heads forward -> TripletMarginLoss -> backward -> [RCCL all-reduce of the flat gradient arena] -> Adam.
Workload = BASELINE.json configs[1]: Kuka + GoogleCommand shapes, per-GPU batch 256, fp32, weak scaling.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Rank 0 prints ONE JSON line."""
import argparse
import json
import math
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HW = 84
BATCH = 256
# 2*MAC per image of layer l's conv (fwd == wgrad == dgrad): Ho^2 * Cout * Cin * 9 * 2
_CH = [3, 32, 32, 64, 64, 64]


def layer_flops(hw):
    out, h = [], hw
    for l in range(5):
        h = (h - 1) // 2 + 1
        out.append(2 * h * h * _CH[l + 1] * _CH[l] * 9)
    return out


LAYER_FLOPS = layer_flops(84)
FLOPS_PER_TRIPLET = 58.407e6          # fwd+bwd @84, SURVEY.md section 8(d) (@96: 70.282e6)
F32_MFMA_PEAK = 157.3                 # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"


def host_cores():
    """CPU threads this process may actually use: affinity mask and cgroup quota, whichever is smaller."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)          # torch's intra-op pool does not scale past this on these small convs


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_run(batch, threads, seconds, min_steps):
    from oracle.torch_oracle import CPUTrainer
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (batch, 3, HW, HW), dtype=torch.uint8, generator=g)
    pos = torch.randn(batch, 1, 100, 40, generator=g) * 5
    neg = torch.randn(batch, 1, 100, 40, generator=g) * 5
    torch.manual_seed(453)
    tr = CPUTrainer(hw=HW)
    tr.step(img, pos, neg)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(img, pos, neg)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= min_steps:
            break
    return n, dt


def parity_vs_cpu(var_amd, model, pool, batch=64):
    """max |delta| of the three embeddings and of the loss between the HIP path and the CPU restatement of the reference
    on the same (image, audio) triplets (BASELINE.md section 3's last bullet; north_star: within 1e-3 fp32).  The MFCC
    features come from the HIP front-end for both sides (its own parity vs the numpy oracle is tests/test_gpu_parity.py's)."""
    from oracle.torch_oracle import KukaNetCPU
    row = pool.epoch_index_table(batch, drop_last=True)[0]
    img = pool.images[row[:batch].long()].contiguous()
    feats = var_amd.mfcc(pool.clips, row[3 * batch:], out_frames=100, clip_index=row[batch:3 * batch])
    pos, neg = feats[:batch].contiguous(), feats[batch:].contiguous()
    was_training = model.training
    model.eval()
    with torch.no_grad():
        d = model(img, pos, neg)
    model.train(was_training)
    net = KukaNetCPU(HW)
    net.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    with torch.no_grad():
        a, p, n = net((img.cpu() / 255.).float(), pos.cpu(), neg.cpu())
        crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)
        l_cpu = float(crit(a, p, n))
        l_hip = float(crit(d["image_feat"].cpu(), d["sound_feat_positive"].cpu(), d["sound_feat_negative"].cpu()))
    diff = max(float((d["image_feat"].cpu() - a).abs().max()), float((d["sound_feat_positive"].cpu() - p).abs().max()),
               float((d["sound_feat_negative"].cpu() - n).abs().max()))
    return {"batch": batch, "max_abs_embedding_diff": diff, "loss_diff": abs(l_hip - l_cpu), "tolerance": 1e-3}


def cpu_baseline(seconds=8.0):
    """The reference's step on the host cores: torch.nn CPU restatement (oracle/torch_oracle.py), same shapes, MFCC
    precomputed (as VARFineTuneDataset does), tensors in RAM.  `value` = batch 256 on every core the process may use
    (BASELINE configs[1]'s batch); `others` = batch 32 (configs[0], the reference's own CPU-runnable case) and a
    one-thread figure, as BASELINE.md section 3 asks.  About 20 s in all."""
    cores = host_cores()
    n, dt = _cpu_run(BATCH, cores, seconds, 5)
    n32, dt32 = _cpu_run(32, cores, 4.0, 10)
    n1, dt1 = _cpu_run(BATCH, 1, 6.0, 1)
    torch.set_num_threads(cores)
    return {"value": round(n * BATCH / dt, 1), "unit": "triplets/s", "cores": cores, "kind": "port",
            "cpu": cpu_model(),
            "sample": f"{n} steps of batch {BATCH} ({HW}x{HW} u8 images, precomputed f32 MFCC in RAM), "
                      f"torch.nn CPU restatement of the reference step, {cores} threads, {dt:.1f} s",
            "others": {"batch32_all_threads": {"value": round(n32 * 32 / dt32, 1), "steps": n32, "seconds": round(dt32, 1)},
                       "batch256_one_thread": {"value": round(n1 * BATCH / dt1, 1), "steps": n1, "seconds": round(dt1, 1)}}}


# ---- BASELINE.json configs[3] shape on ONE model replica per GPU: the iTHOR model (fp32 here; config 4 names bf16) ----
ITHOR_FLOPS_FWD = 3318e6
ITHOR_FLOPS_STEP = 3 * ITHOR_FLOPS_FWD - 2 * 96 * 96 * 27 * 32 - 2 * (2 * 300 * 20 * 121 * 64)   # no dX for the first convs
# the 11x5 stride-2 sound convolution (300x20 -> 150x13, pad 5), any of its 3 directions: 2 * 64 * 64 * the (output
# pixel, tap) pairs whose input pixel lies inside the map -- 1635 (rows) x 50 (columns) = 76 % of the 1950 x 55 pairs of
# the GEMM form; the padding products are not counted as work
_S2_ROWS = sum(sum(0 <= 2 * oy - 5 + ky < 300 for ky in range(11)) for oy in range(150))
_S2_COLS = sum(sum(0 <= 2 * ox - 5 + kx < 20 for kx in range(5)) for ox in range(13))
ITHOR_S2_FLOPS_PER_CLIP = 2 * 64 * 64 * _S2_ROWS * _S2_COLS


def ithor_cpu_baseline(seconds=12.0, batch=8):
    from oracle.torch_oracle import ithor_seeded
    cores = host_cores()
    torch.set_num_threads(cores)
    m = ithor_seeded(977)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    g = torch.Generator().manual_seed(0)
    img = (torch.randint(0, 256, (batch, 3, 96, 96), dtype=torch.uint8, generator=g) / 255.).float()
    pos = torch.randn(batch, 1, 600, 40, generator=g) * 6
    neg = torch.randn(batch, 1, 600, 40, generator=g) * 6

    def one():
        opt.zero_grad()
        a, p, n = m(img, pos, neg)
        crit(a, p, n).backward()
        opt.step()
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 2:
            break
    return {"value": round(n * batch / dt, 2), "unit": "triplets/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {batch} (96x96 images, precomputed (1,600,40) features in RAM), torch.nn CPU "
                      f"restatement of the reference's iTHOR step, {cores} threads, {dt:.1f} s"}


def main_ithor(args, rank, local_rank, world, dev):
    """`--workload ithor`: the same contract for the reference's second pretext model."""
    import var_amd
    from var_amd._lib import Context
    B = args.batch
    cfg = types.SimpleNamespace(img_dim=(3, 96, 96), sound_dim=(1, 600, 40), representationDim=3)
    torch.manual_seed(977)                                      # iTHOR pretextEnvSeed; identical weights on every rank
    model = var_amd.IthorVARPretextNet(cfg).to(dev).set_precision("bf16" if args.dtype == "bf16" else "fp32")
    if args.rehearse_one_device:
        model.set_gru_sequence(False)      # ranks share the card: the persistent GRU launches need their whole grid resident
    peak = 2500.0 if args.dtype == "bf16" else F32_MFMA_PEAK      # dense MFMA peak of the operand type, TFLOP/s
    tr = var_amd.IthorTrainer(model, lr=1e-4, weight_decay=1e-6, margin=1.0)
    g = torch.Generator(device=dev).manual_seed(rank)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device=dev, generator=g)
    pcm = torch.randint(-8000, 8000, (2 * B, 96000), dtype=torch.int16, device=dev, generator=g)
    lens = torch.randint(30000, 96001, (2 * B,), dtype=torch.int32, device=dev, generator=g)
    lens[::5] = 0                                               # 20 % "empty" class
    ctx = Context.get(local_rank)

    def step():
        tr.step_from_pcm(img, pcm, lens, global_batch=B * world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup // 2)):
        step()
    roof = None
    if not args.no_roofline:
        names = ctx.tag_names()
        best = None
        for tag in range(len(names)):
            if "11x5" not in names[tag]:
                continue
            ctx.profile_select(tag)
            for _ in range(3):
                step()
            ms, n = ctx.profile_read()
            if n and (best is None or ms / n > best[1]):
                best = (tag, ms / n, n)
        ctx.profile_select(-1)
        if best:
            flops = ITHOR_S2_FLOPS_PER_CLIP * 2 * B
            ach = flops / (best[1] * 1e-3) / 1e12
            traffic = None
            try:                                                # HBM bytes per launch from the committed PMC passes
                if B == 256 and args.dtype == "f32":             # tools/pmc_traffic.sh r03_ithor_f32 --workload ithor
                    with open(newest_profile("ithor_f32_pmc_hbm_traffic.json")) as f:
                        pm = json.load(f)
                    want = {"forward": "ConvFwdP<", "data gradient": "ConvDgradS2P<", "weight gradient": "ConvWgradP<"}[names[best[0]].split("s2 ")[1]]
                    traffic = max(v["hbm_bytes_fetch_x2_plus_write"] for kk, v in pm.items() if want in kk and "Geo<11, 5," in kk)
                elif B == 256:                                  # tools/pmc_traffic.sh r03_ithor_bf16 --workload ithor --dtype bf16
                    with open(newest_profile("ithor_bf16_pmc_hbm_traffic.json")) as f:
                        pm = json.load(f)
                    want = {"forward": ("snd_fwd_kernel", "Geo2"), "data gradient": ("snd_dgrad_kernel", "DGeo2"),
                            "weight gradient": ("snd_wgrad_kernel", "WGeo2")}[names[best[0]].split("s2 ")[1]]
                    traffic = [v["hbm_bytes_fetch_x2_plus_write"] for kk, v in pm.items() if want[0] in kk and want[1] in kk][0]
            except (OSError, KeyError, ValueError, IndexError, TypeError):
                pass
            roof = {"bound": "mfma", "kernel": names[best[0]], "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "avg_us": round(1e3 * best[1], 1), "launches": best[2], "flops_per_launch": flops}
    use_graph = not args.no_graph
    if use_graph:                                               # the whole step (~700 launches) as one replayed HIP graph
        replay = tr.capture_step(img, pcm, lens, global_batch=B * world)
        step = replay                                           # noqa: F811
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        value = args.steps * B * world / dt
        out = {"metric": "pretext triplets/sec (iTHOR model: 96x96 RGB + 16 kHz/6 s audio)", "value": round(value, 1),
               "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "ms_per_step_device": round(dev_ms / args.steps, 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "n_ranks_seen": args.n_ranks_seen,
               "config": {"workload": "iTHOR pretext step (BASELINE.json configs[3] shapes), batch per GPU as given: u8 "
                                      "96x96 image + 2 int16 clips of up to 6 s resident in HBM -> python_speech_features "
                                      "MFCC -> fwd + triplet loss + bwd + Adam",
                          "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "launch": "hip-graph replay" if use_graph else "eager",
                          "final_loss": round(float(tr.loss.item()), 6)},
               "mfma_frac_whole_step": round(value / world * ITHOR_FLOPS_STEP / 1e12 / peak, 4)}
        if args.dtype == "bf16":
            # the persistent GRU launches report expired hand-off waits here (0 = none; the step is NaN-poisoned otherwise)
            out["config"]["gru"] = "one launch per time step" if args.rehearse_one_device else "one persistent launch per pass"
            out["config"]["gru_handoff_status"] = model.gru_status(local_rank)
            if out["config"]["gru_handoff_status"] or not math.isfinite(out["config"]["final_loss"]):
                raise SystemExit("bench.py: a GRU hand-off wait expired (is another process using this GPU?)")
        if roof:
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = ithor_cpu_baseline()
        emit(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def tag_kernel(tag, hw):
    """The kernel a profiled conv tag stands for at image size `hw` (its name as rocprofv3 prints it)."""
    return {1: "img_head2_kernel<", 2: "img_mid3_kernel<", 7: "img_wgrad345_kernel<", 11: "img_tail2_kernel<",
            12: "img_chain_kernel", 15: "img_wgrad_reduce_kernel"}.get(tag)          # (the same kernels at 84 x 84 and 96 x 96)


def tag_flops(tag):
    """Algorithmic FLOPs per triplet of a profiled conv launch (csrc/api.hip kTagNames), both image sizes: tag 1 = conv 1 + conv 2
    forward, 2 = conv 3 + 4 + 5 forward (+ image head, not counted), 7 = the weight gradients of conv 3-5, 11 = data gradient
    of conv 2 + weight gradients of conv 2 and conv 1, 12 = the data gradients of conv 5, 4, 3.
    Halo recomputation inside the fused kernels is not counted."""
    L = LAYER_FLOPS
    if tag == 1:
        return L[0] + L[1]
    if tag == 2:
        return L[2] + L[3] + L[4]
    if tag in (7, 12):
        return L[2] + L[3] + L[4]
    return {11: 2 * L[1] + L[0]}.get(tag, 0)


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round NN present (the PMC passes are re-taken per round), or None."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d\d)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def pmc_traffic(tag, hw):
    """HBM bytes per launch of the dominant kernel, from the committed PMC pass of THIS round's kernels (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction: FETCH_SIZE x 2; tools/pmc_traffic.sh ->
    profiles/rNN_pmc_hbm_traffic.json | rNN_hw96_pmc_hbm_traffic.json, the newest round present; batch 256).
    PMC counters cannot be collected from inside this process, so this is the figure of that pass
    (same workload, same kernel); None when the file or the kernel is not in it."""
    path = newest_profile("pmc_hbm_traffic.json" if hw == 84 else "hw96_pmc_hbm_traffic.json")
    if hw not in (84, 96) or path is None:
        return None
    want = tag_kernel(tag, hw)
    if want is None or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    for name, row in table.items():
        if want in name:
            return int(row["hbm_bytes_fetch_x2_plus_write"])
    return None


def algorithmic_bytes(tag, B, hw):
    """Bytes a launch of the kernel family MUST move: inputs read once, outputs written once, every
    tensor at its own size (NCHW f32; the u8 image as bytes).  Layout padding of the build's own making -- act1's band tiles
    at 84 x 84 hold 67 328 floats per image for 56 448 real ones -- is NOT algorithmic: it shows up in `traffic`'s excess."""
    h = [hw]
    for _ in range(5):
        h.append((h[-1] - 1) // 2 + 1)
    act = [B * _CH[l] * h[l] * h[l] * 4 for l in range(6)]     # act[0] as f32; the u8 image is act[0] / 4
    img = act[0] // 4
    act1 = act[1]
    return {1: img + act1 + act[2],                          # image in; act1, act2 out
            2: act[2] + act[3] + act[4] + act[5],
            7: act[2] + 2 * (act[3] + act[4]) + act[5],      # x of conv 3-5 (act2-4) and their output gradients (gact3-5)
            11: act1 + act[2] + img,                         # act1 (ReLU gate + conv 2's x), gact2, image
            12: act[5] + 2 * (act[4] + act[3]) + 2 * act[2]  # gact5 in; act4, act3, act2 (ReLU gates) in; gact4, gact3, gact2 out
            }.get(tag)


_REAL_STDOUT = None


def quiet_stdout():
    """Keep stdout for the ONE JSON line: RCCL prints a banner (ROCm version / hostname / library path) to fd 1 when a
    communicator comes up, so fd 1 is pointed at stderr for the run and the result is written to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N
    bench.py <same arguments>` as a CHILD process (this process has not touched the GPU and never will), pass rank
    0's JSON line through on stdout and exit with the children's return code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4")))
    raise SystemExit(rc)


def dry_run_cpu(args, rank, world):
    """The measurement's host-side skeleton with nothing on a GPU: rendezvous, barrier, K timed "steps", barrier, MAX of the
    ranks' wall times, one JSON line from rank 0.  What can break an 8-GPU run before its first kernel breaks here too."""
    if world > 1 or "RANK" in os.environ:
        torch.distributed.init_process_group("gloo")
    seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        time.sleep(1e-3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        emit(json.dumps({"metric": f"pretext triplets/sec ({args.hw}x{args.hw} RGB + 16 kHz/1 s audio)",
                         "value": round(args.steps * args.batch * world / dt, 1), "unit": "triplets/s", "n_gpus": world,
                         "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                         "n_ranks_seen": seen, "dry_run": "CPU launch-logic rehearsal (a step is a 1 ms sleep): NOT a measurement",
                         "config": {"per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}"}}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--streams", type=int, default=None,
                    help="kuka workload: stream plan mask of var_set_streams (bit 0 sound forward, bit 1 sound backward on the "
                         "side stream, bit 4 MFCC stays on the caller's stream, bit 5 (32) the two-launch image forward; default 3)")
    ap.add_argument("--serial", action="store_true",
                    help="kuka workload: every kernel on ONE stream (var_set_streams(0)) -- per-kernel profiling runs")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="ithor workload only: operand precision of the products (bf16 = BASELINE config 4's; accumulation fp32)")
    ap.add_argument("--hw", type=int, choices=(84, 96), default=84,
                    help="kuka workload: image side; 84 = BASELINE's metric (default), 96 = the reference's default img_dim")
    ap.add_argument("--head", choices=("triplet", "inbatch"), default="triplet",
                    help="kuka workload: the reference's triplet loss (default, BASELINE's metric) or the in-batch-negatives "
                         "contrastive head of configs[2] (extension; replayed step with the MFCC front-end inside)")
    ap.add_argument("--workload", choices=("kuka", "ithor"), default="kuka",
                    help="kuka = BASELINE.json's metric (default); ithor = the reference's second pretext model")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="multi-rank REHEARSAL on a box with one GPU: every rank uses cuda:0 and the ranks exchange through "
                         "gloo (RCCL refuses two ranks on one device).  Exercises the N > 1 code path end to end; the "
                         "number it prints is not a scaling measurement and is labelled as such.")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launch-logic rehearsal WITHOUT a GPU: the ranks come up exactly as for a measurement (launcher, environment, "
                         "process group -- gloo --, barriers, MAX over ranks, ONE JSON line from rank 0) but a step is a 1 ms sleep; the "
                         "line is labelled and is not a measurement (tests/test_trainer_host.py runs it with 8 ranks on the CPU)")
    ap.add_argument("--pool", type=int, default=4096,
                    help="kuka workload: triplets in the HBM-resident synthetic pool (16384 = 1.4 GB > the 256 MB Infinity Cache)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as children, BEFORE this process touches
        # the GPU (a process that has initialised HIP must never exec another program on this pool)
        return launch_ranks(args.gpus)

    quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    n_dev = torch.cuda.device_count()                          # counting devices does not initialise HIP
    if args.rehearse_one_device:
        local_rank = 0
    if n_dev < 1 or (not args.rehearse_one_device and (n_dev < world or local_rank >= n_dev)):
        sys.stderr.write(f"bench.py: --gpus {world} needs {world} visible devices, found {n_dev} "
                         f"(rank {rank}, local rank {local_rank})\n")
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        if args.rehearse_one_device:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)
    args.n_ranks_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    if args.workload == "ithor":
        if args.steps == 200 and args.warmup == 20:           # the defaults are sized for the 0.37 ms Kuka step
            args.steps, args.warmup = 20, 3
        return main_ithor(args, rank, local_rank, world, dev)

    import var_amd
    from var_amd._lib import Context

    global HW, FLOPS_PER_TRIPLET, LAYER_FLOPS
    if args.hw != 84:
        HW, FLOPS_PER_TRIPLET = args.hw, 70.282e6             # SURVEY.md section 8(d), Kuka @96
        LAYER_FLOPS = layer_flops(args.hw)
    B = args.batch
    cfg = types.SimpleNamespace(img_dim=(3, HW, HW), sound_dim=(1, 100, 40), representationDim=3)
    torch.manual_seed(453)                                     # pretextEnvSeed; identical weights on every rank
    model = var_amd.VARPretextNet(cfg).to(dev)
    tr = var_amd.VARTrainer(model, lr=1e-4, weight_decay=1e-6, margin=1.0)
    pool = var_amd.SyntheticTripletPool(args.pool, hw=HW, seed=rank, clips_per_class=64, device=dev).freeze_pairs()
    ctx = Context.get(local_rank)
    ctx.ensure_plan(B, HW)
    if args.serial:
        ctx.set_streams(0)
    if args.streams is not None:
        ctx.set_streams(args.streams)

    use_graph = not args.no_graph
    if args.head == "inbatch":
        args.no_roofline = True
    state = {"tab": None, "row": 0}

    def next_row():
        if state["tab"] is None or state["row"] >= state["tab"].shape[0]:
            state["tab"], state["row"] = pool.epoch_index_table(B, drop_last=True), 0
        r = state["tab"][state["row"]]
        state["row"] += 1
        return r

    def eager_step():
        r = next_row()
        if args.head == "inbatch":
            feats = var_amd.mfcc(pool.clips, r[3 * B:], out_frames=100, clip_index=r[B:3 * B])
            tr.step_inbatch(pool.images[r[:B].long()], feats[:B], feats[B:], tau=0.1)
            return
        tr.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3 * B], r[3 * B:], global_batch=B * world)

    step = eager_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(2, args.warmup // 2)):                  # eager warm-up: lazy kernel attributes, workspace plan
        eager_step()
    # Capture FIRST, measure the eager legs after it: the capture leaves the GPU idle for ~9 ms, and the clocks then take ~40
    # steps (13 ms) to come back (tools/ramp_check.py: 0.330 -> 0.309 ms per replay over the first 40 replays after a capture,
    # with a zero learning rate too; 0.308 from the second replay on when eager steps ran in between).  With the eager legs
    # between capture and replays the timed region starts at the sustained clock, as a training run is after its first
    # milliseconds.
    if use_graph:
        # the captured step walks a device-resident table of shuffled epochs by itself (no host copy per step);
        # the host installs the next epochs' table when this one is used up
        trows = 256
        if args.head == "inbatch":
            replay, load_table = tr.capture_inbatch_epoch_steps(pool.images, pool.clips, B,
                                                                pool.index_table(B, trows)[:trows].contiguous(), tau=0.1)
        else:
            replay, load_table = tr.capture_epoch_steps(pool.images, pool.clips, B, pool.index_table(B, trows)[:trows].contiguous(),
                                                        global_batch=B * world)
        gstate = {"left": trows}

        def step():
            if gstate["left"] == 0:
                load_table(pool.index_table(B, trows)[:trows].contiguous())
                gstate["left"] = trows
            gstate["left"] -= 1
            replay()
    # pick the dominant conv kernel family by the kernel's OWN duration (three profiled eager steps per candidate with every
    # launch on one stream: what the rocprofv3 kernel-stats average reads too; beside the side stream's kernels the event
    # bracket of whichever kernel runs next to the MFCC reads longest, which says nothing about that kernel)
    dom_tag = None
    if not args.no_roofline:
        best = -1.0
        sel_mask = ctx.set_streams(0)
        for tag in list(range(0, 10)) + list(range(11, 15)):
            ctx.profile_select(tag)
            for _ in range(3):
                eager_step()
                torch.cuda.synchronize()
            ms, n = ctx.profile_read()
            if n and ms / n > best:
                best, dom_tag = ms / n, tag
        ctx.profile_select(-1)
        ctx.set_streams(sel_mask)
    # roofline leg: HIP events (var_profile_select) around every launch of the dominant kernel, on the
    # stream it is launched on, over eagerly launched steps of the same workload (events cannot be
    # read back from inside a replayed graph)
    roof_ms, roof_n, iso_ms, iso_n = 0.0, 0, 0.0, 0
    if dom_tag is not None:
        ctx.profile_select(dom_tag)
        for _ in range(min(args.steps, 100)):
            eager_step()
            torch.cuda.synchronize()     # one step in flight at a time, as inside the replayed graph
        roof_ms, roof_n = ctx.profile_read()
        # the same kernel alone on the GPU (every launch of the step on one stream): what the kernel itself
        # achieves, without the sound CNN running beside it
        old_mask = ctx.set_streams(0)
        ctx.profile_select(dom_tag)
        for _ in range(min(args.steps, 100)):
            eager_step()
            torch.cuda.synchronize()
        iso_ms, iso_n = ctx.profile_read()
        ctx.set_streams(old_mask)
        ctx.profile_select(-1)

    # north_star's own number: MFMA fraction of the image CNN forward + backward = its algorithmic FLOPs / the summed durations of
    # its kernels.  Two legs, both with HIP events around each kernel family in turn over eager steps: `alone` -- every launch of
    # the step on one stream, i.e. each kernel's own duration (what the rocprofv3 kernel stats of the serial run read too) -- and
    # `beside_sound` -- under the step's own stream plan, the sound branch running beside them.  Inside the REPLAYED graph the
    # kernels take less than the second leg reads (eager launches and the event brackets stretch whatever runs beside the MFCC):
    # the rocprofv3 timeline of the replayed step under profiles/ is the reference for the in-step figure.
    img_us, img_us_side = {}, {}
    if not args.no_roofline and HW in (84, 96):
        tags = (1, 2, 12, 7, 11, 15)
        for serial, dst in ((True, img_us), (False, img_us_side)):
            old_mask = ctx.set_streams(0) if serial else None
            for tag in tags:
                ctx.profile_select(tag)
                for _ in range(20):
                    eager_step()
                    torch.cuda.synchronize()
                ms, n = ctx.profile_read()
                if n:
                    dst[tag_kernel(tag, HW).rstrip("<,")] = round(1e3 * ms / 20, 2)      # per step (a family may launch twice)
            ctx.profile_select(-1)
            if serial:
                ctx.set_streams(old_mask)

    if use_graph:
        tr.sync_device_scalars()                     # (the eager legs above advanced the optimiser's step count)
    for _ in range(args.warmup):
        step()

    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)          # the same K steps between two HIP events on the launch stream: no host launch / sync cost

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(tr.loss.item())

    if rank == 0:
        value = args.steps * B * world / dt
        out = {
            "metric": f"pretext triplets/sec ({HW}x{HW} RGB + 16 kHz/1 s audio)",
            "value": round(value, 1), "unit": "triplets/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "ms_per_step_device": round(dev_ms / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "n_ranks_seen": args.n_ranks_seen,
            **({"rehearsal": "all ranks on ONE device, gloo exchange: code-path check, not a scaling number"}
               if args.rehearse_one_device else {}),
            "config": {"workload": "Kuka+GoogleCommand pretext step, batch 256 per GPU on MI355X, fp32: "
                                   f"u8 {HW}x{HW} image + 2 int16 1 s clips per triplet resident in HBM -> MFCC -> "
                                   "fwd + triplet loss + bwd + Adam (BASELINE.json configs[1])",
                       "per_gpu_batch": B, "global_batch": B * world, "image": [3, HW, HW], "audio": "16 kHz x 1 s int16 x 2",
                       "parallelism": f"dp{world}", "final_loss": round(loss, 6)},
            "mfma_frac_whole_step": round(value / world * FLOPS_PER_TRIPLET / 1e12 / F32_MFMA_PEAK, 4),
        }
        out["config"]["launch"] = "hip-graph replay" if use_graph else "eager"
        if getattr(tr, "dp_graphs_per_step", None):              # data-parallel step: 1 = the collective is captured in the step's graph
            out["config"]["dp_graph_launches_per_step"] = tr.dp_graphs_per_step
        out["config"]["head"] = args.head
        if dom_tag is not None and roof_n and iso_n:
            # `achieved` is priced on the kernel's own duration (HIP events around it with every launch of the step
            # on one stream): that is what the committed rocprofv3 summary averages to as well.  With the sound CNN
            # running beside it on the side stream the same event pair reads longer (`with_side_stream_us`) -- part of
            # that is the event bracket itself, inside the replayed graph the kernel takes about its own time.
            names = ctx.tag_names()
            flops = tag_flops(dom_tag) * B
            us = 1e3 * iso_ms / iso_n
            ach = flops / (us * 1e-6) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": (tag_kernel(dom_tag, HW) or "").rstrip("<,") + " [" + names[dom_tag] + "]", "achieved": round(ach, 2),
                               "peak": F32_MFMA_PEAK, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK, 4),
                               "traffic": pmc_traffic(dom_tag, HW) if B == 256 else None,
                               "algorithmic_bytes": algorithmic_bytes(dom_tag, B, HW),
                               "avg_us": round(us, 2), "launches": iso_n,
                               "with_side_stream_us": round(1e3 * roof_ms / roof_n, 2),
                               "flops_per_launch": flops}
        if img_us:
            img_flops = sum(LAYER_FLOPS) * 3 * B          # forward + data gradient + weight gradient of the five convolutions
            img_flops -= LAYER_FLOPS[0] * B               # (conv 1 has no data gradient)
            tot, tot_side = sum(img_us.values()), sum(img_us_side.values())
            out["image_cnn_mfma_frac"] = round(img_flops / (tot * 1e-6) / 1e12 / F32_MFMA_PEAK, 4)
            out["image_cnn"] = {"flops_per_step": img_flops, "kernels_us_alone": img_us, "sum_us_alone": round(tot, 1),
                                "kernels_us_beside_sound": img_us_side, "sum_us_beside_sound": round(tot_side, 1),
                                "mfma_frac_beside_sound": round(img_flops / (tot_side * 1e-6) / 1e12 / F32_MFMA_PEAK, 4),
                                "how": "HIP events around each kernel family over eager steps: alone = every launch on one stream; "
                                       "beside_sound = the step's stream plan (eager: an upper bound of the in-graph durations, see the "
                                       "rocprofv3 timeline under profiles/)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            out["parity_vs_cpu"] = parity_vs_cpu(var_amd, model, pool)
        emit(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

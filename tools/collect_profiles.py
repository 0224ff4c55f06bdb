#!/usr/bin/env python3
"""Copy the summaries of a tools/checkpoint_profiles.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python tools/collect_profiles.py TAG"""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
stats = glob.glob(f"{go}/prof_{tag}/**/*kernel_stats.csv", recursive=True)
shutil.copy(stats[0], f"{pr}/{tag}_kernel_stats.csv")
run = lambda *a: subprocess.run([sys.executable] + list(a), capture_output=True, text=True, cwd=root).stdout
open(f"{pr}/{tag}_timeline.txt", "w").write(run("tools/timeline.py", f"{go}/prof_{tag}"))
open(f"{pr}/{tag}_pmc_mfma_lds.txt", "w").write(run("tools/pmc_summary.py", f"{go}/pmc_{tag}"))
shutil.copy(f"{go}/{tag}_pmc_hbm_traffic.json", f"{pr}/{tag}_pmc_hbm_traffic.json")
shutil.copy(f"{go}/{tag}_pmc_hbm_traffic.json", f"{pr}/{tag[:3]}_pmc_hbm_traffic.json")     # the round's current pass: what bench.py reads
for src, dst in ((f"bench_{tag}.log", f"{tag}_bench.json"), (f"bench20_{tag}.log", f"{tag}_bench_steps20.json"),
                 (f"bench_prof_{tag}.log", f"{tag}_bench_under_rocprof.json")):
    lines = [l for l in open(f"{go}/{src}") if l.startswith("{")]
    open(f"{pr}/{dst}", "w").write(lines[-1])
print(open(f"{pr}/{tag}_timeline.txt").read())
print(open(f"{pr}/{tag}_pmc_mfma_lds.txt").read())

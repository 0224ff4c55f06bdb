#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 *_kernel_stats.csv: tools/kstats.py <dir> <steps>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r"\s+", " ", r["Name"])
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)[:90]
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step {int(r['Calls']) / steps:7.1f} calls {float(r['AverageNs']) / 1e3:9.1f} us  {n}")
print(f"{tot / steps / 1e6:.3f} ms/step in kernels")

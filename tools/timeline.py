#!/usr/bin/env python3
"""Timeline of one replayed step from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration and the gap to
the previous kernel's end (all streams), averaged over the steps between two adam_pack_dev_kernel dispatches.
usage: timeline.py <dir with *_kernel_trace.csv> [skip_steps]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
# split into steps at adam_pack_dev_kernel
steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'adam_pack_dev_kernel' in r['Kernel_Name']:
        steps.append(cur); cur = []
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = [s for s in steps[skip:] if len(s) == len(steps[-1])]
n = len(steps)
print("steps used", n, "kernels per step", len(steps[-1]))
acc = collections.OrderedDict()
tot = 0
for s in steps:
    t0 = int(s[0]['Start_Timestamp'])
    for i, r in enumerate(s):
        st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        a = acc.setdefault(i, [name(r), 0, 0, 0])
        a[1] += st; a[2] += en - st; a[3] += en
    tot += int(s[-1]['End_Timestamp']) - t0
print("step span (first start .. adam end) us: %.1f" % (tot / n / 1e3))
prev_end = 0
for i, (nm, st, du, en) in acc.items():
    print("%2d %-60s start %7.1f dur %6.1f end %7.1f" % (i, nm, st / n / 1e3, du / n / 1e3, en / n / 1e3))
# period between consecutive adam ends
ends = [int(s[-1]['End_Timestamp']) for s in steps]
if len(ends) > 1:
    print("mean period us: %.1f" % ((ends[-1] - ends[0]) / (len(ends) - 1) / 1e3))

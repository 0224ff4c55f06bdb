#!/bin/bash
# HBM traffic per kernel of one training step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md, HBM section: the two do not fit one pass; FETCH_SIZE x 2 on gfx950), eager launches.
# usage (on the GPU box, from the repo root): bash tools/pmc_traffic.sh r02     -> gpurun_out/r02_pmc_hbm_traffic.json
TAG=${1:-r02}; shift; EXTRA="$*"      # further bench.py arguments, e.g. --workload ithor --dtype bf16
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --streams 67 $EXTRA > $R/gpurun_out/pmc_$C.log 2>&1 || exit 1
done
python3 - "$R" "$TAG" <<'PY'
import csv, glob, json, sys, collections
R, tag = sys.argv[1], sys.argv[2]
out = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "at::" in k or "rocprim" in k or "amd_rocclr" in k:
            continue
        out.setdefault(k, {})[c + "_KiB"] = round(sum(v) / len(v), 1)
for k, d in out.items():
    d["hbm_bytes_fetch_x2_plus_write"] = int(1024 * (2 * d.get("FETCH_SIZE_KiB", 0.0) + d.get("WRITE_SIZE_KiB", 0.0)))
json.dump(out, open(f"{R}/gpurun_out/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_fetch_x2_plus_write"])[:12]:
    print("%-60s %8.1f MB" % (k.replace("void ", "")[:60], d["hbm_bytes_fetch_x2_plus_write"] / 1e6))
PY
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE

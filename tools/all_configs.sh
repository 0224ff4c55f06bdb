#!/bin/bash
# Every bench configuration at one commit on one GPU box (from the repo root, ON the box):
#   bash tools/all_configs.sh r04_final   ->  gpurun_out/<tag>_all_configs.txt, gpurun_out/<tag>_config5_latency.txt, and the
#   full JSON lines of the iTHOR runs (gpurun_out/<tag>_ithor_{f32,bf16}_bench.json)
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_all_configs.txt
: > $OUT
run() {   # label, args...
  local label="$1"; shift
  echo "== bench.py $label" >> $OUT
  timeout -k 10 500 python bench.py "$@" > gpurun_out/_cfg.log 2>&1 || { echo "FAILED" >> $OUT; return; }
  grep '^{' gpurun_out/_cfg.log | tail -1 > gpurun_out/_cfg.json
  python3 - >> $OUT <<'PY'
import json
d = json.load(open("gpurun_out/_cfg.json"))
keep = {k: d[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_device", "steps", "dtype")}
print(json.dumps(keep), d["config"]["workload"][:90])
PY
}
run "" --steps 200 --warmup 20 --no-roofline --no-cpu-baseline
run "--hw 96" --hw 96 --steps 200 --warmup 20 --no-roofline --no-cpu-baseline
run "--head inbatch" --head inbatch --steps 200 --warmup 20 --no-roofline --no-cpu-baseline
run "--rehearse-one-device" --rehearse-one-device --steps 200 --warmup 20 --no-roofline --no-cpu-baseline
run "--workload ithor" --workload ithor --no-cpu-baseline
cp gpurun_out/_cfg.json gpurun_out/${TAG}_ithor_f32_bench.json
run "--workload ithor --dtype bf16" --workload ithor --dtype bf16 --no-cpu-baseline
cp gpurun_out/_cfg.json gpurun_out/${TAG}_ithor_bf16_bench.json
timeout -k 10 300 python tools/infer_latency.py > gpurun_out/${TAG}_config5_latency.txt 2>&1
cat $OUT gpurun_out/${TAG}_config5_latency.txt

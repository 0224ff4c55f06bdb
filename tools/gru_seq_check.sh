#!/bin/bash
# one-off: the persistent GRU kernels -- parity test first, then the bench with and without them
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ithor_bf16.py -x -q -k "gru" > gpurun_out/gru_seq_test.log 2>&1 || { tail -30 gpurun_out/gru_seq_test.log; exit 1; }
tail -3 gpurun_out/gru_seq_test.log
timeout -k 10 300 python bench.py --workload ithor --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/gru_seq_bench.json 2> gpurun_out/gru_seq_bench.err || { tail -20 gpurun_out/gru_seq_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/gru_seq_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'])
PY

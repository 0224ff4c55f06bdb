#!/bin/bash
# usage: run_gpu.sh TAG   (tests + rocprof stats + bench)
TAG=$1
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out && timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/test_$TAG.log 2>&1; tail -5 gpurun_out/test_$TAG.log; cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 \$GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > \$GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.log 2>&1; cd \$GRAFT_REPO_ROOT && timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-400"

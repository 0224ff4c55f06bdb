#!/bin/bash
# usage: tools/gpu_profile.sh TAG   -- GPU suite, the default bench under rocprofv3 --kernel-trace --stats, then the plain default bench
TAG=$1
/usr/local/graft/bin/gpurun --timeout 1100 -- "mkdir -p gpurun_out && timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/test_$TAG.log 2>&1 && tail -2 gpurun_out/test_$TAG.log && cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 \$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > \$GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.log 2>&1 && cd \$GRAFT_REPO_ROOT && timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1 && tail -1 gpurun_out/bench_$TAG.log | cut -c1-600"

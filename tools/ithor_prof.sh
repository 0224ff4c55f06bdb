#!/bin/bash
# per-kernel times of the iTHOR step (eager) under rocprofv3: tools/ithor_prof.sh <tag> [bench args]
TAG=$1; shift
/usr/local/graft/bin/gpurun --timeout 900 -- "python -m pytest tests/test_gpu_ithor_bf16.py -m gpu -q 2>&1 | tail -3 && cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 \$GRAFT_REPO_ROOT/bench.py --workload ithor --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-roofline $* > \$GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1; tail -1 \$GRAFT_REPO_ROOT/gpurun_out/$TAG.log | cut -c1-300"

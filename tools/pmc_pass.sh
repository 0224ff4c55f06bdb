#!/bin/bash
# usage: run_pmc.sh TAG "COUNTERS"
TAG=$1; CNT=$2
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && export VAR_SERIAL=1 && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -- python3 \$GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1; tail -1 \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log | cut -c1-150; ls \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/*/"

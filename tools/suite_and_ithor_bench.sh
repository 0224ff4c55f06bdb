#!/bin/bash
# the full GPU suite, then the iTHOR bf16 bench line (output under gpurun_out/)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1 || { tail -30 gpurun_out/gpu_suite.log; exit 1; }
tail -2 gpurun_out/gpu_suite.log
timeout -k 10 300 python bench.py --workload ithor --dtype bf16 --no-cpu-baseline > gpurun_out/ithor_bf16_bench.json 2> gpurun_out/ithor_bf16_bench.err || { tail -20 gpurun_out/ithor_bf16_bench.err; exit 1; }
tail -1 gpurun_out/ithor_bf16_bench.json | cut -c1-400

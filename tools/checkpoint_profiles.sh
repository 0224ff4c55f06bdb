#!/bin/bash
# One GPU-box call that produces everything a profiles/<TAG>_* checkpoint holds for the Kuka step:
#   kernel stats + timeline of the default bench under rocprofv3, the plain default bench, the driver-style short bench,
#   one PMC pass for MFMA-busy / LDS conflicts, the two HBM-traffic PMC passes.
# usage (here):  bash tools/checkpoint_profiles.sh r03_mid   then   python tools/collect_profiles.py r03_mid
TAG=$1
/usr/local/graft/bin/gpurun --timeout 1100 -- "mkdir -p gpurun_out && R=\$GRAFT_REPO_ROOT && cd /tmp && export TMPDIR=/tmp && rm -rf \$R/gpurun_out/prof_$TAG \$R/gpurun_out/pmc_$TAG && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d \$R/gpurun_out/prof_$TAG -- python3 \$R/bench.py --no-cpu-baseline > \$R/gpurun_out/bench_prof_$TAG.log 2>&1 && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d \$R/gpurun_out/pmc_$TAG -- python3 \$R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > \$R/gpurun_out/pmc_$TAG.log 2>&1 && cd \$R && bash tools/pmc_traffic.sh $TAG > gpurun_out/traffic_$TAG.log 2>&1 && timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1 && timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench20_$TAG.log 2>&1 && tail -1 gpurun_out/bench20_$TAG.log | cut -c1-400"

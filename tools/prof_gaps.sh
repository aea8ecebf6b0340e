#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_gaps.sh <tag> [workload]  -> rocprofv3 kernel trace of a few train steps + tools/gap_analysis.py (GPU idle time between kernels)
TAG=$1; WL=${2:-unet}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_gaps_$TAG -o p -- python3 $ROOT/bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_gaps_$TAG.log 2>&1 || true
cd $ROOT
python3 tools/gap_analysis.py gpurun_out/prof_gaps_$TAG/p_kernel_trace.csv
rm -f gpurun_out/prof_gaps_$TAG/p_kernel_trace.csv

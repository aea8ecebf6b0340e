#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_ws.sh <tag>  -> rocprofv3 kernel stats of the default raygun-segment route (bench --workload watershed)
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_ws -o p -- python3 $ROOT/bench.py --workload watershed --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_ws.log 2>&1 || true
cd $ROOT
python3 tools/prof_summary.py stats gpurun_out/prof_${TAG}_ws/p_kernel_stats.csv gpurun_out/${TAG}_watershed256_kernel_stats.csv

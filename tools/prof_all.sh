#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_all.sh <tag>   -> kernel stats for both workloads + PMC fetch/write passes (UNet)
set -e
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_unet -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_${TAG}_unet.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_cg -o p -- python3 $ROOT/bench.py --workload cyclegan --steps 4 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_cg.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_fetch -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_write -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_${TAG}_write.log 2>&1
cd $ROOT
python3 tools/prof_summary.py stats gpurun_out/prof_${TAG}_unet/p_kernel_stats.csv gpurun_out/${TAG}_unet196_kernel_stats.csv
python3 tools/prof_summary.py stats gpurun_out/prof_${TAG}_cg/p_kernel_stats.csv gpurun_out/${TAG}_cyclegan128_kernel_stats.csv
python3 tools/prof_summary.py pmc gpurun_out/prof_${TAG}_fetch/p_counter_collection.csv gpurun_out/${TAG}_unet196_pmc_fetch_size.csv
python3 tools/prof_summary.py pmc gpurun_out/prof_${TAG}_write/p_counter_collection.csv gpurun_out/${TAG}_unet196_pmc_write_size.csv
ls gpurun_out/*${TAG}*csv

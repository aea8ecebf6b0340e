#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_aux.sh <tag>  -> rocprofv3 kernel stats of the predict and segment bench workloads
set -e
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_seg -o p -- python3 $ROOT/bench.py --workload segment --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_seg.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_pred -o p -- python3 $ROOT/bench.py --workload predict --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_pred.log 2>&1 || true
cd $ROOT
python3 tools/prof_summary.py stats gpurun_out/prof_${TAG}_seg/p_kernel_stats.csv gpurun_out/${TAG}_segment256_kernel_stats.csv
python3 tools/prof_summary.py stats gpurun_out/prof_${TAG}_pred/p_kernel_stats.csv gpurun_out/${TAG}_predict404_kernel_stats.csv
ls gpurun_out/*${TAG}*csv
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_segfetch -o p -- python3 $ROOT/bench.py --workload segment --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_segfetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_segwrite -o p -- python3 $ROOT/bench.py --workload segment --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_${TAG}_segwrite.log 2>&1 || true
cd $ROOT
python3 tools/prof_summary.py pmc gpurun_out/prof_${TAG}_segfetch/p_counter_collection.csv gpurun_out/${TAG}_segment256_pmc_fetch_size.csv
python3 tools/prof_summary.py pmc gpurun_out/prof_${TAG}_segwrite/p_counter_collection.csv gpurun_out/${TAG}_segment256_pmc_write_size.csv

#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_epi.sh [RG_GEMM5 = 1]  -> device-side duration of whole-tile igemm5 (1) / igemm4 (-1) launches against K
# (1^3 / 2^3 / 3^3 taps x 256 channels, 250 tiles of 256 x 256 = one round): slope = time per K-tile, intercept = fixed cost of a tile (launch, prologue, epilogue)
G5=${1:-1}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_epi_$k -o p -- python3 $ROOT/tools/bench_sk.py 256 256 $((39+k)) $k -1:$G5 > /dev/null 2>&1 || true
  echo "k=$k"; grep igemm $ROOT/gpurun_out/prof_epi_$k/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
  rm -f $ROOT/gpurun_out/prof_epi_$k/p_kernel_trace.csv
done

#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/prof_unet.sh <tag> [workload]
set -e
TAG=$1; WL=${2:-unet}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o p -- python3 $ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $ROOT
python3 tools/prof_summary.py stats gpurun_out/prof_$TAG/p_kernel_stats.csv gpurun_out/prof_$TAG/summary.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$TAG/summary.csv")))
tot=sum(float(r['total_ms']) for r in rows)
print("total kernel ms/step", tot/7)
for r in rows[:45]:
    print(f"{r['kernel'][:64]:64s} {r['calls']:>6s} {float(r['total_ms'])/7:8.3f} {r['avg_us']:>9s}")
PY

#!/bin/bash
# usage: bash tools/prof_seg.sh <tag> [sizes...]  -> rocprofv3 kernel stats of tests/tools/bench_seg.py
TAG=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_seg_$TAG -o p -- python3 $ROOT/tests/tools/bench_seg.py "$@" > $ROOT/gpurun_out/prof_seg_$TAG.log 2>&1
cd $ROOT
grep "n=" gpurun_out/prof_seg_$TAG.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_seg_$TAG/p_kernel_stats.csv")))
for r in rows[:12]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY

#!/bin/bash
# usage: bash tools/pmc_conv.sh <tag> <shape list> <counters...>
set -e
TAG=$1; SH=$2; shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_$TAG -o p -- python3 $ROOT/tools/bench_conv.py $SH > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
cd $ROOT
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open("gpurun_out/pmc_$TAG/p_counter_collection.csv")):
    k = r["Kernel_Name"][:60]
    if not any(s in k for s in ("igemm", "wgrad_kw", "wgrad_line")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print(f"    {c:32s} {v:.4g}")
PY

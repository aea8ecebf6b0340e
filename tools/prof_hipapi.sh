#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_hipapi.sh [workload] [ms]  -> HIP API calls longer than `ms` (default 0.5) during a short bench run, and
# how often each synchronising call was made (what stalls the launch queue inside a step)
WL=${1:-cyclegan}; MS=${2:-0.5}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --output-format csv -d $ROOT/gpurun_out/prof_hipapi -o p -- python3 $ROOT/bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_hipapi.log 2>&1 || true
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_hipapi/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 40 % of the run = steady-state timed steps
t_end = int(rows[-1]["End_Timestamp"]); t0 = int(rows[0]["Start_Timestamp"]); cut = t0 + (t_end - t0) * 0.75
tail = [r for r in rows if int(r["Start_Timestamp"]) > cut]
c = collections.Counter(); tt = collections.Counter()
for r in tail:
    c[r["Function"]] += 1; tt[r["Function"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("last quarter of the run: calls / total ms per HIP function")
for k, v in sorted(tt.items(), key=lambda kv: -kv[1])[:14]: print(f"  {k:40s} x{c[k]:6d} {v / 1e6:9.2f} ms")
print("long calls there (>= $MS ms):")
for r in tail:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d >= $MS * 1e6: print(f"  t={(int(r['Start_Timestamp']) - t0) / 1e6:10.1f} ms  {d / 1e6:8.2f} ms  {r['Function']}")
PY
rm -rf gpurun_out/prof_hipapi

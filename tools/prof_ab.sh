#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_ab.sh "<ENV=val>" "<ENV=val>"  -> rocprofv3 kernel-trace of the U-Net step per variant on ONE box:
# per-step sum of kernel durations, wall time per step from the trace (first kernel start to last kernel end of the timed steps), top kernels
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf $ROOT/gpurun_out/prof_ab_$i
  env $v rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_ab_$i -o p -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/prof_ab_$i.log 2>&1
  python3 - <<PY
import csv, collections, glob, json
f = glob.glob("$ROOT/gpurun_out/prof_ab_$i/**/p_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = the last 6 of 9 steps: find step boundaries by the Adam kernels' bursts is fragile; use the last 6/9 of the kernel list by count
n = len(rows); per = n // 9
sel = rows[n - 6 * per:]
wall = (sel[-1][1] - sel[0][0]) / 6e6
busy = sum(e - s for s, e, _ in sel) / 6e6
gaps = sum(max(0, sel[k + 1][0] - sel[k][1]) for k in range(len(sel) - 1)) / 6e6
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    a = agg[k[:60]]; a[0] += 1; a[1] += e - s
line = open("$ROOT/gpurun_out/prof_ab_$i.log").read().strip().splitlines()[-1]
try: ms = json.loads(line)["ms_per_step"]
except Exception: ms = None
print("[$v] bench ms/step", ms, "| trace: wall %.3f ms/step, sum of kernel durations %.3f, idle gaps %.3f, kernels/step %d" % (wall, busy, gaps, per))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %8.3f ms/step x%-3d %s" % (t / 6e6, c // 6, k))
PY
done

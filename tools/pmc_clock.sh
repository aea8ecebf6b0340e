#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_clock.sh <tag>  -> effective shader clock per kernel of the U-Net step
# (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / dispatch duration, MI355X_MICROARCH.md "DVFS give-back")
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_clk_$TAG -o p -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/pmc_clk_$TAG.log 2>&1 || true
cd $ROOT
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc_clk_$TAG/p_counter_collection.csv")))
print(list(rows[0].keys()))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"][:70]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += dur
out = open("gpurun_out/${TAG}_unet196_clock.csv", "w")
out.write("kernel,dispatches,total_ms,ghz\n")
for k, (n, c, d) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    line = f'"{k}",{n},{d/1e6:.3f},{c/8/d:.3f}'
    print(line); out.write(line + "\n")
PY

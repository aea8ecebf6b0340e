#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_mfma.sh <tag> [workload]
# MFMA utilisation per kernel of the train step from SQ counters (own pass: --pmc only, no trace domains):
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles an MFMA pipe is busy (summed over SIMDs / XCDs as rocprofv3 reports it)
#   SQ_BUSY_CYCLES            cycles the SQ has waves (per shader engine)
#   SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F16   MFMA operations (units of 512 FLOP per the counter definition)
#   SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY  where the waves' cycles go (quad-cycles)
# GRBM_GUI_ACTIVE in the same pass gives the effective clock (sum over 8 XCDs / 8 / duration).
TAG=$1; WL=${2:-unet}
ROOT=$(pwd)
EXTRA="--no-secondary"
[ "$WL" = "cyclegan" ] && EXTRA=""
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d $ROOT/gpurun_out/pmc_mfma_$TAG -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $ROOT/gpurun_out/pmc_mfma_$TAG.log 2>&1
rc=$?
cd $ROOT
[ $rc -ne 0 ] && { tail -5 gpurun_out/pmc_mfma_$TAG.log; exit $rc; }
python3 - <<PY
import csv, collections
rows = csv.DictReader(open("gpurun_out/pmc_mfma_$TAG/p_counter_collection.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"][:90]
    a = agg[k]
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a["_n"] += 1
        a["_dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
with open("gpurun_out/${TAG}_${WL}_pmc_mfma.csv", "w", newline="") as f:
    w = csv.writer(f)
    # mfma_busy_frac: MFMA-busy cycles / (GRBM_GUI_ACTIVE/8 cycles x 1024 SIMDs); tflops: MOPS x 512 FLOP / duration
    w.writerow(["kernel", "dispatches", "total_ms", "ghz"] + names + ["mfma_busy_frac_of_simd_cycles", "mops_tflops"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["_dur_ns"])[:60]:
        d = a["_dur_ns"]
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0
        tf = a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / d / 1e3 if d else 0
        w.writerow([k, int(a["_n"]), f"{d / 1e6:.3f}", f"{cyc / d:.3f}" if d else ""] + [f"{a[n]:.6g}" for n in names] + [f"{busy:.4f}", f"{tf:.1f}"])
print(open("gpurun_out/${TAG}_${WL}_pmc_mfma.csv").read()[:6000])
PY

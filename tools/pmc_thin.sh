#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_thin.sh  -> instruction-mix counters of the thin conv kernels in the UNet step
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_thin_$tag -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $ROOT/gpurun_out/pmc_thin_$tag.log 2>&1 || true
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_thin_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm_thin" in k or "igemm4_kernel<4, 9, true, false>" in k:
            key = (k[:60], r["Counter_Name"])
            agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:60s} {c:32s} {n:3d} {v/n:14.0f}")
PY

#!/bin/bash
# usage (GPU box): bash tools/pmc_wgrad.sh  -> SQ / LDS / TCC counters of the wide weight-gradient kernels (kwd vs kwp) on the 2592 x 2592 layer
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_wg1 -o p -- python3 $ROOT/tools/bench_wgrad.py 0,-1 > $ROOT/gpurun_out/pmc_wg1.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $ROOT/gpurun_out/pmc_wg2 -o p -- python3 $ROOT/tools/bench_wgrad.py 0,-1 > $ROOT/gpurun_out/pmc_wg2.log 2>&1
cd $ROOT
python3 - <<PY
import csv, collections, glob
for d in ("pmc_wg1", "pmc_wg2"):
    f = glob.glob("gpurun_out/%s/**/p_counter_collection.csv" % d, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "wgrad_kw" not in k: continue
        key = (k, r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "FETCH_SIZE"): n[key] += 1
    for key, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
        print(d, key, "dispatches", n[key], {k: "%.4g" % (v / max(1, n[key])) for k, v in c.items()})
PY

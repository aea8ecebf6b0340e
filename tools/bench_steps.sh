#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_steps.sh <tag> [ENV=val ...]  -> one quick U-Net bench line (ms/step + top kernels) under the given env
TAG=$1; shift
env "$@" python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bs_$TAG.json 2> gpurun_out/bs_$TAG.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/bs_$TAG.json").read().strip().splitlines()[-1])
print("$TAG", d["ms_per_step"], "ms;", " ".join(f"{k.split('_kernel')[0]}{k.split('_kernel')[1]}={v['ms_per_step']:.2f}" for k,v in list(d["kernels"].items())[:6]))
PY

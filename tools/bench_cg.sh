#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_cg.sh <tag> [ENV=val ...]  -> CycleGAN 128^3 step time under the given env
TAG=$1; shift
env "$@" python3 bench.py --workload cyclegan --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/cg_$TAG.json 2> gpurun_out/cg_$TAG.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/cg_$TAG.json").read().strip().splitlines()[-1])
print("$TAG", d["ms_per_step"], "ms")
PY

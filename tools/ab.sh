#!/bin/bash
# usage (GPU box): bash tools/ab.sh "ENV_A=.." "ENV_B=.." [n]  -> alternating U-Net bench runs under the two environments
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do for e in "$A" "$B"; do
  env $e python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'])"
done; done

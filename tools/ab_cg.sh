#!/bin/bash
# usage (GPU box): bash tools/ab_cg.sh "<ENV=val>" ...  -> CycleGAN 128^3 link step per variant, same box, interleaved twice
for rnd in 1 2; do
  for v in "$@"; do
    env $v python3 bench.py --workload cyclegan --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernels']
top=sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:7]
print('round $rnd [$v] %.2f ms/step  executed %.1f TF/s | ' % (d['ms_per_step'], d['executed_tflops']) + ' | '.join('%s x%d %.2f ms %.0f TF/s' % (n[:34], v['launches'], v['ms_per_step'], v['tflops']) for n,v in top))
"
  done
done

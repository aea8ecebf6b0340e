#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_step.sh "<ENV=val ...>" "<ENV=val ...>" ...   -> U-Net step time per variant, SAME box, interleaved twice
# (devices of the pool differ by several per cent on MFMA-dense kernels: never compare numbers of different gpurun calls)
for rnd in 1 2; do
  for v in "$@"; do
    env $v python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
k=d['kernels']
def tot(f): return sum(v['ms_per_step'] for n,v in k.items() if f(n))
print('round $rnd  [$v]  %.3f ms/step | fwd/dgrad wide (igemm4 non-flat, igemm5, winograd) %.3f | wgrad_kwp<9,9> %.3f | other wgrad %.3f | flat/thin/halo %.3f | all conv kernels %.3f' % (
  d['ms_per_step'], tot(lambda n: ('igemm5' in n or 'winograd' in n or ('igemm4' in n and 'flat' not in n))), tot(lambda n: n.startswith('wgrad_kwp_kernel<9,9>')),
  tot(lambda n: n.startswith('wgrad') and not n.startswith('wgrad_kwp_kernel<9,9>')), tot(lambda n: 'flat' in n or 'thin' in n or 'halo' in n or 'igemm2' in n or 'igemm3' in n), tot(lambda n: True)))
"
  done
done

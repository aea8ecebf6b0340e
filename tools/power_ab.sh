#!/bin/bash
# usage (GPU box): bash tools/power_ab.sh "<ENV=val>" ...  -> step time + sampled board power / clocks while the timed steps run
for v in "$@"; do
  ( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; sleep 0.4; done ) > gpurun_out/power_$$.txt &
  SM=$!
  env $v python3 bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v] %.3f ms/step over 300 steps' % d['ms_per_step'])"
  wait $SM
  sed -n '8,30p' gpurun_out/power_$$.txt | awk '{print "   " $0}' | cut -c1-260 | sed -n '1~4p'
done

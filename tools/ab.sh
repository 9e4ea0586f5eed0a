#!/bin/bash
# Diagnostic: time the A/B builds of the library (csrc/Makefile: liblgstep_ab{0,1}.so) alternately in one GPU session.
n=${1:-3}
root=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $n); do
  for v in ${AB_VARIANTS:-0 1}; do
    LGSTEP_LIB=$root/extended_legged_gym_amd/csrc/liblgstep_ab$v.so python $root/bench.py --no-cpu-baseline --steps ${AB_STEPS:-600} --warmup ${AB_WARMUP:-400} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('AB=$v  ms_per_step %.4f  physics %.2f us  post %.2f us  event overhead %.2f us' % (d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['post_kernel_ms'], 1e3*r['hip_event_pair_overhead_ms']))"
  done
done

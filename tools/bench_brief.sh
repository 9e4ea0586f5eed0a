#!/bin/bash
# Diagnostic: run bench.py N times and print only ms_per_step and the three kernel times.
n=${1:-2}
for i in $(seq $n); do
  python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step %.4f  physics %.2f us  post %.2f us  event overhead %.2f us  value %.3e' % (d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['post_kernel_ms'], 1e3*r['hip_event_pair_overhead_ms'], d['value']))"
done

#!/bin/bash
# Diagnostic: headline workload at several env counts (one JSON line of bench.py each, reduced to throughput / step time).
for n in ${NSWEEP:-1024 2048 4096 8192 16384 65536}; do
  python bench.py --envs-per-gpu $n --steps 300 --warmup 150 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%6d envs  %.3e env-steps/s  %.4f ms/step' % (d['config']['num_envs_per_gpu'], d['value'], d['ms_per_step']))"
done

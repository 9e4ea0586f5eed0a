#!/bin/bash
# A/B of library builds on the triangle-mesh configs (config 3, the registered trimesh task) in ONE GPU session: tools/ab_mesh.sh libA.so libB.so [rounds]
D=$(cd "$(dirname "$0")/.." && pwd)/extended_legged_gym_amd/csrc
for i in $(seq ${3:-2}); do
  for L in $1 $2; do
    LGSTEP_LIB=$D/$L timeout -k 10 300 python tools/bench_configs.py 3 trimesh 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line); print('$L', d['config'][:60], '%.4f ms' % d['ms_per_step'])"
  done
done

#!/bin/bash
# A/B in ONE GPU session over (library, environment) variants: each argument is "label:lib.so[:VAR=VALUE...]" (library relative to csrc/).
# usage: AB_ROUNDS=3 tools/ab_env.sh "plain:liblgstep.so:LG_CAPS=0" "caps:liblgstep.so" "ab21:liblgstep_ab21.so"
D=$(cd "$(dirname "$0")/.." && pwd)/extended_legged_gym_amd/csrc
for i in $(seq ${AB_ROUNDS:-3}); do
  for spec in "$@"; do
    IFS=':' read -r label lib rest <<< "$spec"
    envs=$(echo "$rest" | tr ':' ' ')
    v=$(env $envs LGSTEP_LIB=$D/$lib timeout -k 10 300 python bench.py --steps ${AB_STEPS:-4000} --warmup 1000 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.5f ms  %.3e" % (d["ms_per_step"], d["value"]))')
    echo "$label  $v"
  done
done

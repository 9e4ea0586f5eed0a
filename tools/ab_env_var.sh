#!/bin/bash
# A/B of values of ONE environment switch on a config of tools/bench_configs.py in ONE GPU session: tools/ab_env_var.sh <config> <VAR> <value> [<value> ...]
CFG=$1; VAR=$2; shift; shift
for i in 1 2; do
  for D in "$@"; do
    v=$(env $VAR=$D python tools/bench_configs.py $CFG 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d.get("ms_per_step_without_camera",0),4))')
    echo "$CFG $VAR=$D  $v"
  done
done

#!/bin/bash
# A/B of builds of liblgstep.so on a config of tools/bench_configs.py in ONE GPU session: tools/ab_cfg_libs.sh <config> <lib.so> [<lib.so> ...]   (libs relative to extended_legged_gym_amd/csrc)
CFG=$1; shift
for i in 1 2 3; do
  for L in "$@"; do
    v=$(LGSTEP_LIB=$PWD/extended_legged_gym_amd/csrc/$L python tools/bench_configs.py $CFG 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d.get("ms_per_step_without_camera",0),4))')
    echo "$CFG $L  $v"
  done
done

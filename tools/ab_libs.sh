#!/bin/bash
# A/B of two builds of liblgstep.so in ONE GPU session (clocks differ between boxes: only same-session pairs compare).
# usage: tools/ab_libs.sh <libA.so> <libB.so> [rounds]   (paths relative to extended_legged_gym_amd/csrc)
A=$1; B=$2; R=${3:-3}
D=extended_legged_gym_amd/csrc
for i in $(seq $R); do
  for L in $A $B; do
    v=$(LGSTEP_LIB=$PWD/$D/$L timeout -k 10 200 python bench.py --steps ${AB_STEPS:-20000} --warmup 1000 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.5f ms  %.3e" % (d["ms_per_step"], d["value"]))')
    echo "$L  $v"
  done
done

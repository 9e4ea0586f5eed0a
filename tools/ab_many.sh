#!/bin/bash
# A/B/C... of several builds of liblgstep.so in ONE GPU session.  usage: tools/ab_many.sh <rounds> <lib.so> <lib.so> ...   (names in extended_legged_gym_amd/csrc)
R=$1; shift
D=extended_legged_gym_amd/csrc
for i in $(seq $R); do
  for L in "$@"; do
    v=$(LGSTEP_LIB=$PWD/$D/$L timeout -k 10 200 python bench.py --steps ${AB_STEPS:-10000} --warmup 1000 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.5f ms  %.3e" % (d["ms_per_step"], d["value"]))')
    echo "$L  $v"
  done
done

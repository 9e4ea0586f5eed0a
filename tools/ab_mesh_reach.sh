#!/bin/bash
# A/B of the distance-cache reach of the triangle-mesh contact queries in ONE GPU session: tools/ab_mesh_reach.sh <config of tools/bench_configs.py> <reach> [<reach> ...]
CFG=$1; shift
for i in 1 2; do
  for D in "$@"; do
    v=$(LG_MESH_REACH=$D python tools/bench_configs.py $CFG 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d.get("ms_per_step_without_camera",0),4))')
    echo "$CFG reach $D  $v"
  done
done

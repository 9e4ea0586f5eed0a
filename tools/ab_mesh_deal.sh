#!/bin/bash
# A/B of contact-query deals on a triangle-mesh config in ONE GPU session: tools/ab_mesh_deal.sh <config of tools/bench_configs.py> <deal> [<deal> ...]   (deal = LG_MESH_DEAL digits)
CFG=$1; shift
for i in 1 2 3 4; do
  for D in "$@"; do
    v=$(LG_MESH_DEAL=$D python tools/bench_configs.py $CFG 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d.get("ms_per_step_without_camera",0),4))')
    echo "$CFG deal $D  $v"
  done
done

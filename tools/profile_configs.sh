#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel-trace statistics + separate FETCH_SIZE / WRITE_SIZE passes for the secondary configs
# (tools/bench_configs.py 3 / 4 / 5 / hexapod / cassie) and the rollout-collection benchmark (tools/bench_rollout.py: policy_act_kernel, lg_compute_returns).
# Usage: tools/profile_configs.sh <tag>     (outputs under gpurun_out/prof_<tag>_cfg*/ ; tools/collect_config_profiles.py <tag> summarises)
set -u
TAG=${1:-r03}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for W in 3 4 5 hexapod cassie rollout; do
  OUT=$REPO/gpurun_out/prof_${TAG}_cfg$W
  mkdir -p $OUT
  if [ "$W" = "rollout" ]; then CMD="python3 $REPO/tools/bench_rollout.py"; else CMD="python3 $REPO/tools/bench_configs.py $W"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/line.json 2> $OUT/trace.log
  echo "cfg $W trace done" >&2
  if [ "$W" != "5" ] && [ "$W" != "hexapod" ] && [ "$W" != "cassie" ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.log
    echo "cfg $W pmc done" >&2
  fi
done
cd $REPO

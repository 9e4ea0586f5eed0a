#!/bin/bash
# Run on the GPU box through gpurun: kernel-trace stats + separate PMC passes for HBM traffic of the bench command.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python3 $REPO/bench.py --steps 200 --warmup 100 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.log
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
# Diagnostic (GPU box, via gpurun): SQ counters of every kernel of an arbitrary command, one small counter group per pass (tools/pmc_sq.sh is this for bench.py).
# Usage: tools/pmc_cmd.sh <tag> <kernel-name filter> <python script and its arguments>      (outputs under gpurun_out/pmc_<tag>/)
set -u
TAG=$1; FILT=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 "$@" > $OUT/g$i.json 2> $OUT/g$i.log
done
cd $REPO
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, sys, collections
out, filt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print("==", k)
    for c in sorted(acc[k]):
        print("   %-32s %16.1f per launch (%d launches)" % (c, acc[k][c] / max(cnt[k][c], 1), cnt[k][c]))
PY

"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{row.get('Name','?')[:70]:70s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} "
                  f"avg_ns={row.get('AverageNs')} pct={row.get('Percentage')}")
print("== per-dispatch resources (first dispatch of each kernel) ==")
seen = set()
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row.get("Kernel_Name", "?")
            if n in seen:
                continue
            seen.add(n)
            print(f"{n[:60]:60s} grid={row.get('Grid_Size_X', row.get('Grid_Size'))} wg={row.get('Workgroup_Size_X', row.get('Workgroup_Size'))} "
                  f"vgpr={row.get('VGPR_Count')} accum={row.get('Accum_VGPR_Count')} sgpr={row.get('SGPR_Count')} lds={row.get('LDS_Block_Size')} "
                  f"scratch={row.get('Scratch_Size', row.get('Private_Segment_Size'))}")
for tag, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = defaultdict(lambda: [0.0, 0])
    for f in find(f"{tag}/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != cname:
                    continue
                k = row.get("Kernel_Name", "?")
                agg[k][0] += float(row.get("Counter_Value", 0))
                agg[k][1] += 1
    print(f"== {cname} per dispatch (KiB units as reported; gfx950: FETCH_SIZE of wide coalesced reads under-counts 2x) ==")
    for k, (v, n) in agg.items():
        print(f"{k[:70]:70s} dispatches={n} mean={v / max(n, 1):.1f}")

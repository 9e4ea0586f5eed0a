"""Turn the raw outputs of tools/profile_round.sh <tag> and tools/pmc_sq.sh <tag> (under gpurun_out/) into the committed
summaries profiles/<tag>_*: rocprofv3 kernel statistics, HBM traffic of the kernels from the separate FETCH_SIZE / WRITE_SIZE
passes, the SQ counter table, the bench lines; then name <tag> in profiles/LATEST (bench.py reads its traffic / issue figures
from the summaries named there).     usage: python tools/collect_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prof = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
pmc = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
dst = os.path.join(ROOT, "profiles")


def out(name):
    return os.path.join(dst, f"{tag}_{name}")


if os.path.exists(os.path.join(prof, "summary.txt")):
    shutil.copy(os.path.join(prof, "summary.txt"), out("rocprofv3_summary.txt"))
stats = sorted(glob.glob(os.path.join(prof, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)   # (newest: gpurun_out/ keeps earlier runs)
if stats:
    shutil.copy(stats[0], out("kernel_stats.csv"))
for src, name in (("trace_bench.json", "bench_under_rocprof.json"),):
    if os.path.exists(os.path.join(prof, src)):
        shutil.copy(os.path.join(prof, src), out(name))
kern = defaultdict(dict)
for sub, cname, key in (("pmc_fetch", "FETCH_SIZE", "FETCH_SIZE_KiB_per_launch"), ("pmc_write", "WRITE_SIZE", "WRITE_SIZE_KiB_per_launch")):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(prof, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == cname:
                k = row["Kernel_Name"].split("(")[0]
                agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items():
        kern[k][key] = round(v / max(n, 1), 1)
if kern:
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 200 --warmup 100 (tools/profile_round.sh {tag})",
               "note": "FETCH_SIZE as reported (gfx950 tallies 128-B requests at 64 B: wide coalesced reads need x2; this kernel's loads are narrow and uncalibrated); WRITE_SIZE exact for wide stores",
               "kernels": kern}, open(out("pmc.json"), "w"), indent=1)
sq = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}.txt")
if os.path.exists(sq):
    shutil.copy(sq, out("sq_counters.txt"))
for src, name in ((f"bench_{tag}.json", "bench.json"), (f"configs_{tag}.jsonl", "other_configs.jsonl"), (f"rollout_{tag}.json", "rollout_collection.json")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", src)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", src), out(name))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha256: bench.py quotes these counters only next to timings of the same sources)
try:
    import subprocess
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
except Exception:
    head = None
json.dump({"kernel_source_sha256": bench.kernel_source_sha256(), "git_head_at_collection": head}, open(out("build.json"), "w"), indent=1)
open(os.path.join(dst, "LATEST"), "w").write(tag + "\n")
print("profiles/%s_* written:" % tag, sorted(os.path.basename(f) for f in glob.glob(out("*"))))

"""Summarise gpurun_out/prof_<tag>_cfg{3,4,5,hexapod,cassie,rollout}/ (tools/profile_configs.sh) into profiles/<tag>_other_kernels.json: per kernel the
rocprofv3 average duration and call count, HBM traffic per launch from the separate FETCH_SIZE / WRITE_SIZE passes, and a roofline object
from the algorithmic bytes (or flops) one launch processes.      usage: python tools/collect_config_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
HBM, MFMA_F32 = 8000.0, 157.0          # GB/s, TFLOP/s dense fp32 matrix (MI355X_MICROARCH.md)
N = 4096
# algorithmic bytes per launch of the kernels that dominate each config (DESIGN.md s4 / s8 / s9 / s11 state the per-unit figures)
ALGO = {
    # config 3: A1 (PD), whole fused step: 8696 B of the ANYmal-C row minus the LSTM state (3072 B) plus the persisted closest-point
    # cache of 32 spheres (16 B read + 16 B written each); mesh / BVH traversal traffic is what the counters add on top
    ("3", "physics_kernel<0, true"): dict(bytes=(8696 - 3072 + 32 * 32) * N, unit="env-step", note="A1, confined OBJ mesh (SDF contacts)"),
    ("3", "sdf_bodies_kernel"): dict(bytes=5 * (12 + 4 + 12 + 12) * N, unit="5 body points per env"),
    # config 4: trimesh ANYmal-C rough step (LSTM) + grid-mesh contact queries; depth camera: 2 x 28 x 56 floats of FIFO written per env
    ("4", "physics_kernel<0, true"): dict(bytes=(8696 + 32 * 32) * N, unit="env-step", note="ANYmal-C rough, slope-corrected trimesh via grid-mesh queries"),
    ("4", "depth_kernel"): dict(bytes=(2 * 28 * 56 * 4 + 52) * N, unit="1800 rays per env -> (2, 28, 56) depth FIFO"),
    ("5", "physics_kernel<0, false"): dict(bytes=(8696 - 3072 - 1688) * N, unit="rollout env-step", note="PD, plane, 48 observations, no height scan"),
    ("rollout", "policy_act_kernel"): dict(flops=2.0 * N * (235 * 512 + 512 * 256 + 256 * 128 + 128 * 12 + 235 * 512 + 512 * 256 + 256 * 128 + 128), unit="PPO.act on 4096 rows"),
}
out = {}
for W in ("3", "4", "5", "hexapod", "cassie", "rollout"):
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_cfg{W}")
    stats = sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)   # (newest: gpurun_out/ keeps earlier runs)
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_cfg{W}_kernel_stats.csv"))
    kern = {}
    for row in csv.DictReader(open(stats[0])):
        kern[row["Name"]] = dict(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3, total_pct=float(row["Percentage"]))
    traffic = defaultdict(dict)
    for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        agg = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == cname:
                    agg[row["Kernel_Name"]][0] += float(row["Counter_Value"]); agg[row["Kernel_Name"]][1] += 1
        for k, (v, n) in agg.items():
            traffic[k][cname + "_KiB_per_launch"] = round(v / max(n, 1), 1)
    line = None
    try:
        line = json.loads([l for l in open(os.path.join(d, "line.json")).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        pass
    rows = {}
    for name, k in sorted(kern.items(), key=lambda kv: -kv[1]["total_pct"])[:8]:
        r = dict(k)
        short = name.split("(")[0]
        for tk, tv in traffic.items():
            if tk.split("(")[0] == short:
                r.update(tv)
                rd, wr = tv.get("FETCH_SIZE_KiB_per_launch"), tv.get("WRITE_SIZE_KiB_per_launch")
                if rd is not None and wr is not None:
                    r["traffic_bytes_per_launch_as_reported"] = (rd + wr) * 1024.0
        for (w, pat), a in ALGO.items():
            if w == W and pat in name:
                if "bytes" in a:
                    ach = a["bytes"] / (k["avg_us"] * 1e-6) / 1e9
                    r["roofline"] = dict(bound="hbm", achieved=ach, peak=HBM, unit="GB/s", frac=ach / HBM, algorithmic_bytes_per_launch=a["bytes"],
                                         traffic=r.get("traffic_bytes_per_launch_as_reported"), per=a["unit"], note=a.get("note", ""))
                else:
                    ach = a["flops"] / (k["avg_us"] * 1e-6) / 1e12
                    r["roofline"] = dict(bound="mfma", achieved=ach, peak=MFMA_F32, unit="TFLOP/s", frac=ach / MFMA_F32, flops_per_launch=a["flops"],
                                         traffic=r.get("traffic_bytes_per_launch_as_reported"), per=a["unit"])
        rows[name] = r
    out["config " + W] = dict(bench_line=line, kernels=rows)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_other_kernels.json"), "w"), indent=1)
for c, v in out.items():
    print(c, {k: v["bench_line"].get(k) for k in ("ms_per_step", "env_steps_per_s", "step_rollout_ms", "ppo_act_native_ms") if v["bench_line"] and k in v["bench_line"]})
    for n, r in v["kernels"].items():
        rf = r.get("roofline")
        print("   %-60s %6d calls %10.1f us %5.1f %%  %s" % (n[:60], r["calls"], r["avg_us"], r["total_pct"], ("%s frac %.4f" % (rf["bound"], rf["frac"])) if rf else ""))

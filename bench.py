"""Headline benchmark: env-steps/s of `LeggedRobot.step()` for ANYmal-C on rough heightfield terrain, 4096 envs per
MI355X (BASELINE.json configs[1]); the reference's own measurement loop is `legged_gym/tests/test_env_simstep_time.py:10-17`
(random-normal actions, mean wall time of env.step).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; envs shard by rank (weak scaling: 4096 envs per GPU) with no data-path collective; after the timed
region every rank contributes its episode statistics to one RCCL all-gather.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # cpu_baseline: idle OpenMP threads must not spin on a shared host

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
SCLK_GHZ = 2.4                 # MI355X peak engine clock (MI355X_MICROARCH.md): converts the VALU issue floor from cycles to time

# Algorithmic HBM bytes per env per policy step for the tensors the kernels actually materialise:
# each is read once and/or written once per lg_step because the 4 substeps run on-chip.
PHYSICS_BYTES = dict(
    read=48 + 52 + 96 + 48 + 1536 + 4 + 4,                 # actions, root, dof, last_dof_vel, LSTM h+c, friction, payload
    write=48 + 52 + 96 + 48 + 1536 + 17 * 12 + 17 * 52,    # clipped actions, root, dof, torques, LSTM h+c, contact forces, rigid-body state
)
POST_BYTES = dict(
    read=52 + 96 + 24 + 24 + 16 + 204 + 4 * 52 + 48 + 48 + 48 + 48 + 8 + 32 + 4 + 36 + 187 * 3 * 2 + 16 + 16,
    write=748 + 940 + 4 + 2 + 8 + 36 + 24 + 48 + 48 + 24 + 32 + 4 + 36 + 16 + 4 + 16,
)


def build_env(rank, world, num_envs, pd_control=False, solver=None, mesh_type="heightfield"):
    from extended_legged_gym_amd.envs import Anymal, AnymalCRoughCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params, set_seed
    from extended_legged_gym_amd.utils.sharding import shard_env_cfg
    cfg = AnymalCRoughCfg()
    cfg.terrain.mesh_type = mesh_type              # BASELINE config 2: "heightfield", collide against the 900x900 int16 grid (diagnostic tools pass "trimesh")
    cfg.seed = 1
    shard_env_cfg(cfg, rank, world, num_envs)      # global terrain-column indexing + private Philox stream per shard
    if pd_control:                                 # diagnostic only (not the headline workload): PD law instead of the LSTM
        cfg.control.use_actuator_network = False
    if solver == "pgs":                            # diagnostic only: round 2's solver (the headline runs sim.physx.solver_type = 1, TGS)
        cfg.sim.physx.solver_type, cfg.sim.physx.friction_model = 0, "cone"
    args = get_args([])
    args.sim_device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        set_seed(1)                                # identical terrain (np.random seed 1) on every rank
    sp = parse_sim_params(args, {"sim": class_to_dict(cfg.sim)})
    return Anymal(cfg, sp, args.physics_engine, args.sim_device, True), cfg


def cpu_baseline(env, actions_pool, budget_s=15.0):
    """The oracle (a scalar C++ port of the same step, OpenMP over envs) timed on this host, on a bounded sample of
    the same workload: the same 4096 envs continued from the GPU env's current state for as many policy steps as fit
    in ~budget_s seconds."""
    from oracle.oracle_lib import OracleEnv, lib
    o = OracleEnv(env.setup)
    for name in ["root_states", "dof_state", "friction_coeffs", "base_mass_added", "terrain_levels", "terrain_types",
                 "env_origins", "commands", "last_actions", "last_dof_vel", "last_root_vel", "episode_length_buf",
                 "sea_hidden_state", "sea_cell_state", "feet_air_time", "feet_contact_time", "last_contacts",
                 "episode_sums", "gait_idx", "gait_foot_z", "base_lin_acc", "base_ang_acc", "step_counters"]:
        o.t[name][...] = env.core.t[name].cpu().numpy()
    acts = [a.cpu().numpy() for a in actions_pool[:8]]
    # the box advertises every host CPU but may only grant a share of them: probe a few OpenMP team sizes and keep the fastest
    # (oversubscribed teams are far slower than one thread).  Per candidate: one untimed step (thread start-up, first touch), then the
    # MEDIAN of five -- a single step per candidate picked 64 threads on one box and 128 on the next for a 1.5x different number
    avail = len(os.sched_getaffinity(0))
    o.step(acts[0])
    best, cores = None, 1
    for th in sorted({8, 16, 32, 64, 128, avail}):
        if th > avail:
            continue
        lib().lgo_set_threads(th)
        o.step(acts[1])
        ts = []
        for k in range(5):
            t0 = time.perf_counter()
            o.step(acts[(2 + k) % len(acts)])
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        if best is None or t < best:
            best, cores = t, th
    lib().lgo_set_threads(cores)
    per_step = best
    n = int(max(4, min(400, budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for i in range(n):
        o.step(acts[i % len(acts)])
    dt = time.perf_counter() - t0
    val = env.num_envs * n / dt
    o.close()
    return dict(value=val, unit="env-steps/s", cores=int(cores), kind="port",
                sample=f"{n} policy steps x {env.num_envs} envs of the same workload, oracle/lg_oracle.cpp -- an UNOPTIMISED scalar port (-O2, dense "
                       f"18 x 18 Cholesky per env and substep; the checker of the parity tests, not a tuned CPU simulator) -- with OpenMP on {cores} "
                       f"threads (fastest of the probed team sizes by the median of 5 steps; {avail} CPUs visible), {dt:.1f} s")


def kernel_source_sha256():
    """Hash of the kernel sources the library is built from: counter summaries are only quoted next to timings of the same code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "extended_legged_gym_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "lgstep.h"), "rb").read())
    return h.hexdigest()


def _profile_tag():
    try:
        return open(os.path.join(ROOT, "profiles", "LATEST")).read().strip() or "r01_z"
    except OSError:
        return "r01_z"


def _latest_profile(suffix):
    """Counter summary `profiles/<tag>_<suffix>` of the build named in `profiles/LATEST` (written by tools/collect_profiles.py
    when the summaries of a build are committed)."""
    return os.path.join(ROOT, "profiles", f"{_profile_tag()}_{suffix}")


def committed_counters_match_this_build():
    """True when `profiles/<tag>_build.json` (tools/collect_profiles.py) names the sources this library was built from and the
    run uses the default one-launch step: PMC / SQ figures of another build, or of the split path, are not quoted."""
    if os.environ.get("LG_FUSE", "1") == "0" or os.environ.get("LG_SPLIT") or os.environ.get("LG_GRID_MESH"):
        return False
    try:
        with open(_latest_profile("build.json")) as f:
            return json.load(f)["kernel_source_sha256"] == kernel_source_sha256()
    except Exception:
        return False


def sq_issue(N):
    """VALU busy fraction of the physics kernel's waves from the committed SQ counter passes (`tools/pmc_sq.sh`): the kernel
    is bound by instruction issue of one wave per SIMD, not by bytes, so this is the utilisation figure that moves."""
    path = _latest_profile("sq_counters.txt")
    if not os.path.exists(path) or N != 4096 or not committed_counters_match_this_build():
        return {}
    try:
        vals, on = {}, False
        for line in open(path):
            if line.startswith("=="):
                on = "physics_kernel" in line
            elif on and "per launch" in line:
                k, v = line.split()[:2]
                vals[k] = float(v)
        # the second ceiling: a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles, the launch is one wave per SIMD on
        # 256 CUs x 4 SIMDs, so the instruction stream as it stands cannot finish faster than insts x 4 / 1024 cycles
        floor_us = vals["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (SCLK_GHZ * 1e3)
        return {"valu_busy_frac": vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"], "valu_insts_per_launch": vals["SQ_INSTS_VALU"],
                "issue_source": os.path.relpath(path, ROOT), "_valu_floor_us": floor_us}
    except Exception:
        return {}


def pmc_traffic(N):
    """HBM bytes per launch of the physics kernel from the committed rocprofv3 PMC passes (`tools/profile_round.sh`,
    separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command).  bench.py cannot collect counters
    itself; the figure is only reported when the committed passes were taken at the same env count."""
    files = [_latest_profile("pmc.json")]     # the passes taken on the most recent build of the kernels
    if not os.path.exists(files[0]) or N != 4096:
        return {"traffic": None}
    if not committed_counters_match_this_build():
        sys.stderr.write("bench.py: profiles/%s_* were taken on other kernel sources (or another launch mode): roofline.traffic = null\n" % _profile_tag())
        return {"traffic": None, "traffic_note": "committed counter passes belong to another build"}
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"]
        name = [n for n in k if "physics_kernel" in n][0]
        rd, wr = k[name]["FETCH_SIZE_KiB_per_launch"] * 1024.0, k[name]["WRITE_SIZE_KiB_per_launch"] * 1024.0
        return {"traffic": rd + wr, "traffic_read_bytes_as_reported": rd, "traffic_write_bytes": wr,
                "traffic_source": os.path.relpath(files[-1], ROOT) + " (committed profile of this build, not measured by this run)",
                "profile_tag": _profile_tag()}
    except Exception:
        return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pd-control", action="store_true", help="diagnostic: PD actuators instead of the LSTM net")
    ap.add_argument("--solver", choices=["tgs", "pgs"], default="tgs", help="diagnostic: pgs = round 2's solver instead of the configured TGS")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process has not touched the GPU yet, so it may start the N ranks as
        # children (one process per GPU, rendezvous on the loopback address) and hand their exit code on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    # LG_REHEARSE_ONE_GPU=1 (tests only; never the driver's run): the N ranks of `--gpus N` share cuda:0 and the collectives go over gloo -- RCCL refuses two
    # ranks on one device.  Everything else is the multi-GPU path as the driver runs it: the launcher, the rendezvous, shard configs, barriers, the MAX over
    # ranks, the all-gathers.  The line says so ("rehearsal": true).
    rehearse = os.environ.get("LG_REHEARSE_ONE_GPU") == "1" and world > 1
    if rehearse:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torchrun the group is formed even for one rank (same code path)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if a.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    env, cfg = build_env(rank, world, a.envs_per_gpu, a.pd_control, a.solver)
    N = env.num_envs
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    pool = [torch.randn(N, 12, generator=gen).to(dev) for _ in range(64)]   # resident in HBM before timing starts

    env.reset()
    # Untimed pre-warm in front of the --warmup steps, whatever --warmup says: a short run (the driver's --steps 20 --warmup 5) otherwise times
    # a GPU that is still ramping its clocks and a host whose launch path is cold (0.0784 ms per step in a 20-step window against 0.0752 after
    # 1000 steps, same library).  The timed region below is untouched; the count is reported as "pre_warm_steps".
    pre_warm_steps = max(0, 256 - a.warmup)
    for i in range(pre_warm_steps + a.warmup):
        env.step(pool[i % len(pool)])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()

    # the timed region: exactly --steps calls of env.step, nothing else (no events, no profiler)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        env.step(pool[i % len(pool)])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    # a separate, untimed pass for the per-kernel durations of the roofline block: HIP events on the stream the kernels
    # run on, every 4th step sampled, at least 64 samples whatever --steps was
    samples = 128
    env.core.profile_begin(samples, 4)
    for i in range(4 * samples):
        env.step(pool[i % len(pool)])
    torch.cuda.synchronize(dev)
    prof = env.core.profile_end()

    # episode statistics of every shard: one all-gather (RCCL over xGMI when world > 1)
    from extended_legged_gym_amd.utils.sharding import gather_episode_stats
    if dist is not None:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    _, totals = gather_episode_stats(env.core.t["episode_stats"].cpu().clone() if rehearse else env.core.t["episode_stats"].clone(), dist)
    stats_all = totals.cpu().numpy()
    # what each shard drew for itself (per-shard domain randomisation: disjoint Philox streams, shard-seeded host draws)
    fr = env.core.t["friction_coeffs"].double()
    mine = torch.stack([fr.mean(), fr[0], env.core.t["base_mass_added"].double().mean()])
    if rehearse:
        mine = mine.cpu()
    if dist is not None:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    else:
        every = [mine]
    shards = [dict(rank=i, friction_mean=float(v[0]), friction_first=float(v[1]), payload_mean=float(v[2])) for i, v in enumerate(every)]
    finite = bool(torch.isfinite(env.obs_buf).all().item() and torch.isfinite(env.root_states).all().item())

    if rank == 0:
        value = world * N * a.steps / elapsed
        # lg_step is ONE launch: the post-physics step runs as the tail of the physics kernel (LG_FUSE=0 keeps the two-launch
        # path), so the dominant kernel moves the bytes of the whole step
        fused = os.environ.get("LG_FUSE", "1") != "0"
        post_bytes = POST_BYTES["read"] + POST_BYTES["write"]
        phys_bytes = PHYSICS_BYTES["read"] + PHYSICS_BYTES["write"] + (post_bytes if fused else 0)
        # the kernel's own duration: the HIP-event interval minus what two events back to back measure (the interval alone is longer than ms_per_step;
        # the net figure is what rocprofv3 reports for the kernel, profiles/)
        net_ms = max(prof["physics_ms"] - prof["finalize_ms"], 0.0)
        achieved = phys_bytes * N / (net_ms * 1e-3) / 1e9 if net_ms > 0 else 0.0
        issue = sq_issue(N)
        floor_us = issue.pop("_valu_floor_us", None)
        kernel_us = max(prof["physics_ms"] - prof["finalize_ms"], 0.0) * 1e3
        # the ceiling that actually binds: VALU issue of the instruction stream as it stands (one wave per SIMD), next to the HBM one
        issue_obj = ({"valu_floor_us": floor_us, "kernel_us": kernel_us, "frac": floor_us / kernel_us if kernel_us > 0 else None,
                      "note": "floor = VALU wave-instructions per launch x 4 cycles / 1024 SIMDs at 2.4 GHz; frac = floor / measured kernel time"}
                     if floor_us else {"valu_floor_us": None, "note": "no SQ counter pass of this build is committed (tools/pmc_sq.sh)"})
        out = {
            "metric": "env-steps/sec, ANYmal-C rough 4096 envs/GPU", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "pre_warm_steps": pre_warm_steps, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ANYmal-C rough heightfield terrain (8x8 tiles, 900x900 int16 grid, seed 1), "
                                   f"{N} envs/GPU, LSTM actuator net, 235-dim obs, noise+pushes+curriculum on, "
                                   "actions N(0,1), one step = 4 physics substeps (TGS contact solver, 4 sub-intervals each: sim.physx.solver_type = 1) + post-physics",
                       "num_envs_per_gpu": N, "decimation": 4, "sim_dt": 0.005, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "limited_by": "instruction issue / dependent latency of one heavy wave per SIMD (roofline.issue), not bytes: the HBM fraction is reported because the contract asks for it",
                         "kernel": "physics_kernel<0> (4 substeps + fused post-physics tail)" if fused else "physics_kernel<0>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, **pmc_traffic(N), **issue, "issue": issue_obj,
                         "algorithmic_bytes_per_env_step": phys_bytes, "kernel_ms": prof["physics_ms"],
                         "kernel_ms_net_of_event_overhead": net_ms,   # what `achieved` and `frac` divide by, and what rocprofv3 reports (profiles/)
                         "post_kernel_ms": prof["post_ms"], "hip_event_pair_overhead_ms": prof["finalize_ms"],   # two events back to back: what every event interval above carries on top of its kernel
                         "hip_event_samples": prof["samples"],
                         "whole_step_bytes_per_env_step": PHYSICS_BYTES["read"] + PHYSICS_BYTES["write"] + post_bytes, "launches_per_step": 1 if fused else 2},
            "episode_stats": {"sum_return": float(stats_all[0]), "sum_length": float(stats_all[1]),
                              "episodes": float(stats_all[2]), "env_steps": float(stats_all[3])},
            "finite": finite, "shards": shards,
        }
        if rehearse:
            out["rehearsal"] = True          # ranks shared one GPU over gloo: a functional run of the N > 1 path, not a measurement
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(env, pool)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

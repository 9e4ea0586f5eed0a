"""bench.py's output contract, and its distributed code path on one GPU.

The multi-GPU scaling run belongs to the driver (N = 2, 4, 8 on a whole node); what can be checked on a one-GPU box is
that the SAME code path works: under `torch.distributed.run` a 1-rank `nccl` (= RCCL) process group is formed, the
barriers and the episode-statistics all-gather run on it, and rank 0 prints one JSON line with the required keys.  The
timed region must carry no instrumentation: `hip_event_samples` comes from the separate pass and does not depend on --steps."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_source_keeps_the_profiler_out_of_the_timed_region():
    src = open(os.path.join(ROOT, "bench.py")).read()
    timed = src[src.index("t0 = time.perf_counter()\n    for i in range(a.steps)"):src.index("elapsed = time.perf_counter() - t0")]
    assert "profile_begin" not in timed and "Event" not in timed
    assert src.index("profile_begin") > src.index("elapsed = time.perf_counter() - t0")


@pytest.mark.gpu
def test_bench_under_torchrun_forms_a_one_rank_rccl_group():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "20",
           "--envs-per-gpu", "1024", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json_line(r.stdout)
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["finite"] is True
    assert out["roofline"]["hip_event_samples"] >= 64            # the instrumented pass is separate from the 20 timed steps
    assert out["episode_stats"]["env_steps"] > 0                 # went through the all-gather of the 1-rank group
    assert out["value"] > 0 and abs(out["ms_per_step"] * 1e-3 * out["value"] - 1024) < 1.0


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The driver's scaling run, rehearsed at N = 2 through bench.py's own child-process launcher (`python bench.py --gpus 2`): one
    process per GPU, RCCL group, per-shard randomisation, the episode-statistics all-gather.  Skipped on the one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "20", "--envs-per-gpu", "1024",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["finite"] is True
    # every env-step of both shards arrived through the all-gather: reset() step + warm-up + timed + the instrumented pass, on 2 x 1024 envs
    per_rank_steps = out["episode_stats"]["env_steps"] / (2 * 1024)
    assert per_rank_steps == int(per_rank_steps) and per_rank_steps >= 70
    assert abs(out["ms_per_step"] * 1e-3 * out["value"] - 2 * 1024) < 2.0
    a, b = out["shards"]
    assert a["rank"] == 0 and b["rank"] == 1 and a["friction_first"] != b["friction_first"] and a["friction_mean"] != b["friction_mean"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the one-GPU box: bench.py's own launcher starts two ranks (torch.distributed.run, rendezvous on 127.0.0.1), each
    builds its shard (`shard_env_cfg`), steps it, and the barriers / MAX over ranks / all-gathers run -- over gloo with both ranks on cuda:0
    (`LG_REHEARSE_ONE_GPU=1`: RCCL refuses two ranks on one device).  What this leaves unproven of the driver's scaling run is RCCL over xGMI itself."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LG_REHEARSE_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "10", "--envs-per-gpu", "512", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["finite"] is True and out["rehearsal"] is True
    per_rank_steps = out["episode_stats"]["env_steps"] / (2 * 512)
    assert per_rank_steps == int(per_rank_steps) and per_rank_steps >= 40
    assert abs(out["ms_per_step"] * 1e-3 * out["value"] - 2 * 512) < 2.0
    a, b = out["shards"]
    assert a["rank"] == 0 and b["rank"] == 1 and a["friction_first"] != b["friction_first"] and a["friction_mean"] != b["friction_mean"]


@pytest.mark.gpu
def test_sharded_main_rollout_driver_rehearsed_on_one_gpu():
    """`python tools/bench_configs.py --gpus 2 5` end to end on the one-GPU box (`LG_REHEARSE_ONE_GPU=1`, see above): two ranks, each with its 128 mains x 32
    rollouts (`shard_main_rollout_cfg`), barriers around the timed loops, the slowest rank's time, the episode-statistics all-gather; rank 0 prints the job's line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LG_REHEARSE_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "--gpus", "2", "5"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["rehearsal"] is True and out["first_global_main"] == 0
    assert out["episode_stats"]["ranks"] == 2 and out["episode_stats"]["env_steps"] > 0
    assert out["rollout_env_steps_per_s"] > 0 and out["rollout_batch_H16_ms"] > 0

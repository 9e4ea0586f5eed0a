"""Diagnostic: where the host time of `RobotBatchRollout.step_rollout` goes (cProfile over 2000 calls of config 5)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs
env = bench_configs.config5_env()
a = torch.randn(128 * 32, 12, device="cuda")
for _ in range(100):
    env.step_rollout(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    env.step_rollout(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time per call %.1f us; with the final sync %.1f us" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    env.step_rollout(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

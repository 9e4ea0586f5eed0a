"""Diagnostic: run the bench workload on the -DLG_STAMPS build and print the share of shader cycles per phase of
physics_kernel (lane 0 of workgroup 0).  Shares only — never quote this build's run time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LGSTEP_LIB"] = os.path.join(ROOT, "extended_legged_gym_amd", "csrc", os.environ.get("LG_STAMPS_LIB", "liblgstep_stamps.so"))
sys.path.insert(0, ROOT)
import torch
import bench
pd = "--pd" in sys.argv
if "--config3" in sys.argv:                       # A1 on the confined OBJ mesh (BVH contacts), tools/bench_configs.py config 3
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    env = bench_configs.config3_env()
elif "--config5" in sys.argv:                     # main-rollout env (tools/bench_configs.py config 5): the ROLLOUT step of 128 x 32 envs
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    env = bench_configs.config5_env()
else:
    env, cfg = bench.build_env(0, 1, 4096, pd, mesh_type="trimesh" if "--trimesh" in sys.argv else "heightfield")   # --trimesh: anymal_c_rough as registered
    env.reset()
g = torch.Generator().manual_seed(0)
pool = [torch.randn(4096, 12, generator=g).cuda() for _ in range(16)]
for i in range(300):
    if "--config5" in sys.argv:
        env.step_rollout(pool[i % 16])
    else:
        env.step(pool[i % 16])
lib = env.core.lib
out = (C.c_ulonglong * 64)()
lib.lg_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
rc = lib.lg_debug_read_stamps(env.core.ctx, out)
import extended_legged_gym_amd.native as nat
print('lib', nat.LIB_PATH, 'rc', rc)
STEPS = 301
sub = {15: "substep prologue", 24: "publish q/qd (LDS writes)", 25: "barrier (A)", 0: "inline actuator | copies", 1: "kinematics", 2: "bias set-up | mesh terrains: this wave's contact queries", 3: "CRBA+Schur+chol",
       29: "own contact detection", 5: "wait at rendezvous (A2) + slot mask", 6: "contact pass B (setup)", 4: "wait for torques (barrier B)",
       7: "unconstrained + solver passes", 8: "limits+forces+integrate", 9: "fault guard", 10: "write-back + final FK"}
tail = {39: "publish final state + wait at (F)", 11: "TAIL: main part 1 (rows, features, rotations)", 19: "TAIL: wait at (G1)", 20: "TAIL: serial part (callback, rewards, reset)",
        21: "TAIL: wait at (G2)", 12: "TAIL: state stores", 32: "TAIL wb: row stores", 33: "TAIL wb: stats + ticket", 49: "TAIL wb: obs rows: entry table loads", 50: "TAIL wb: obs rows: entries (+ noise) -> LDS rows", 34: "TAIL wb: obs rows: LDS rows -> global stores",
        13: "TAIL: (call overhead)", 14: "TAIL: arrival + finalize"}
helper = {22: "HELPER: wait at (A3)", 23: "HELPER: recurrent half of the network (3 of 4 substeps)", 40: "HELPER: wait at (A) [the main wave's sweeps]", 41: "HELPER: state fetch + kinematics", 42: "HELPER: bias / detection loads",
          43: "HELPER: LSTM input half + detection finish", 44: "HELPER: wait at (A2)", 45: "HELPER: contact set-up share"}
whole = out[36] / STEPS
print(f"main wave, whole kernel: {whole:9.0f} cycles per step")
ssum = 0
for k, n in sub.items():
    ssum += out[k]
    print(f"  {n:48s} per substep {out[k] / (STEPS * 4):8.0f}   {100.0 * out[k] / max(out[36], 1):5.1f} %")
print(f"  {'substeps, total':48s} per step    {ssum / STEPS:8.0f}   {100.0 * ssum / max(out[36], 1):5.1f} %")
tsum = 0
for k, n in tail.items():
    tsum += out[k]
    print(f"  {n:48s} per step    {out[k] / STEPS:8.0f}   {100.0 * out[k] / max(out[36], 1):5.1f} %")
print(f"  {'tail, total':48s} per step    {tsum / STEPS:8.0f}   {100.0 * tsum / max(out[36], 1):5.1f} %")
print(f"  entry -> first barrier (config / model block -> LDS)  per step {out[37] / STEPS:8.0f}")
print(f"  first barrier -> state rows in registers             per step {out[38] / STEPS:8.0f}")
print(f"  unaccounted (entry, loads, exit)                 per step    {(out[36] - ssum - tsum) / STEPS:8.0f}")
for k, n in {46: "HELPER, last substep: set-up share end -> (A3) passed", 47: "HELPER, last substep: prefetch for the tail", 48: "HELPER: wait at (F)", 35: "HELPER: feet rows + height scan", 51: "HELPER: rigid-body rows", 52: "HELPER: wait at (G1) / (G2) for the serial part"}.items():
    print(f"  {n:64s} per step    {out[k] / STEPS:8.0f}")
for k, n in {58: "SERIAL: input batch", 59: "SERIAL: callback (resample, heading, push)", 60: "SERIAL: termination + flag stores", 61: "SERIAL: reward terms + config-order sum",
             62: "SERIAL: reward stores, reset", 63: "SERIAL: result row, episode sums"}.items():
    print(f"  {n:64s} per step    {out[k] / STEPS:8.0f}")
print(f"  MAIN: barrier (A) of the FIRST substep                          per step    {out[26] / STEPS:8.0f}")
print(f"  HELPER: (-DLG_SCAN_TWICE: the height scan a second time, warm) | LSTM state + first recurrent half  per step    {out[27] / STEPS:8.0f}")
for k, n in {53: "HELPER first substep: kinematics", 54: "HELPER first substep: bias / detection loads", 55: "HELPER first substep: input half + detection finish", 56: "HELPER first substep: wait at (A2)", 57: "HELPER first substep: set-up share"}.items():
    print(f"  {n:64s} per step    {out[k] / STEPS:8.0f}")
for k, n in helper.items():
    print(f"  {n:64s} per substep (of the other three) {out[k] / (STEPS * 3):8.0f}")
print('active slots per 16-lane group (4 envs) per substep', out[28] / max(out[17], 1) / 4)
print('wave-substeps with >= 4 active slots: %.1f %%, >= 5: %.1f %%' % (100.0 * out[30] / max(out[17], 1), 100.0 * out[31] / max(out[17], 1)))
print('active slots per wave-substep', out[16] / max(out[17], 1), ' active contacts per wave-substep', out[18] / max(out[17], 1), '(of', 64 * 7, 'lane-slots)')
if "--config3" in sys.argv:
    calls = max(out[57], 1)
    print("mesh contact queries, wave 2 of workgroup 0: %d detection calls; per call: %.1f queries issued (of 128), %.1f BVH nodes visited summed over lanes, %.1f by the busiest lane; %.0f cycles inside the traversal"
          % (calls, out[54] / calls, out[55] / calls, out[56] / calls, out[53] / calls))

"""Diagnostic: run the bench workload on the -DLG_STAMPS build and print the share of shader cycles per phase of
physics_kernel (lane 0 of workgroup 0).  Shares only — never quote this build's run time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LGSTEP_LIB"] = os.path.join(ROOT, "extended_legged_gym_amd", "csrc", os.environ.get("LG_STAMPS_LIB", "liblgstep_stamps.so"))
sys.path.insert(0, ROOT)
import torch
import bench
pd = "--pd" in sys.argv
env, cfg = bench.build_env(0, 1, 4096, pd)
env.reset()
g = torch.Generator().manual_seed(0)
pool = [torch.randn(4096, 12, generator=g).cuda() for _ in range(16)]
for i in range(300):
    env.step(pool[i % 16])
lib = env.core.lib
out = (C.c_ulonglong * 32)()
lib.lg_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
rc = lib.lg_debug_read_stamps(env.core.ctx, out)
import extended_legged_gym_amd.native as nat
print('lib', nat.LIB_PATH, 'rc', rc)
names = {15: "substep prologue", 0: "publish q/qd | inline actuator", 1: "kinematics", 2: "bias (RNEA)", 3: "CRBA+Schur+chol",
         29: "own contact detection", 5: "wait at rendezvous (A2) + slot mask", 6: "contact pass B (setup)", 4: "wait for torques (barrier B)", 7: "unconstrained + PGS",
         8: "limits+forces+integrate", 9: "fault guard", 10: "write-back + final FK",
         11: "TAIL: main part 1 (rows, features, rotations)", 19: "TAIL: wait at (G1)", 20: "TAIL: serial part (callback, rewards, reset)",
         21: "TAIL: wait at (G2)", 12: "TAIL: state stores", 13: "TAIL: write-back + obs rows", 14: "TAIL: arrival + finalize"}
tot = sum(out[:16]) + sum(out[19:22]) + out[29]
hn = {22: "TAIL wb: row stores (main wave; helper stamp 22-24 unused in this build)", 23: "TAIL wb: stats + store drain + ticket", 24: "TAIL wb: obs rows", 25: "HELPER: LSTM joint", 26: "HELPER: wait at (A2)", 27: "HELPER: contact set-up share"}
for k, n in hn.items():
    print(f"{n:34s} per substep {out[k] / (301 * 4):9.0f} cycles")
print('active slots per 16-lane group (4 envs) per substep', out[28] / max(out[17], 1) / 4)
print('wave-substeps with >= 4 active slots: %.1f %%, >= 5: %.1f %%' % (100.0 * out[30] / max(out[17], 1), 100.0 * out[31] / max(out[17], 1)))
print('active slots per wave-substep', out[16] / max(out[17], 1), ' active contacts per wave-substep', out[18] / max(out[17], 1), '(of', 64 * 7, 'lane-slots)')
for k, n in names.items():
    print(f"{n:28s} {out[k]:14d} cycles  {100.0 * out[k] / max(tot, 1):5.1f} %   per substep-call {out[k] / (301 * 4):9.0f}")
print("total cycles per step", tot / 301)

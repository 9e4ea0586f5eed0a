"""Diagnostic: phase stamps of physics_kernel<0, true> (triangle-mesh contacts) on the config-3 workload."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LGSTEP_LIB"] = os.path.join(ROOT, "extended_legged_gym_amd", "csrc", "liblgstep_stamps.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs

orig_timeit = bench_configs.timeit
holder = {}


def timeit(fn, warm, steps):
    holder.setdefault("fn", fn)
    return orig_timeit(fn, min(warm, 50), min(steps, 50))


bench_configs.timeit = timeit
import extended_legged_gym_amd.native as nat
cores = []
orig_init = nat.NativeCore.__init__


def init(self, *a, **k):
    orig_init(self, *a, **k)
    cores.append(self)


nat.NativeCore.__init__ = init
print(bench_configs.config3())
core = cores[0]
out = (C.c_ulonglong * 32)()
core.lib.lg_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
core.lib.lg_debug_read_stamps(core.ctx, out)
names = {15: "substep prologue", 0: "publish | PD torques", 1: "kinematics", 2: "bias", 3: "CRBA+Schur+chol", 5: "wait (A2) + slot mask",
         6: "contact set-up share", 4: "torques", 7: "unconstrained + PGS", 8: "limits+forces+integrate", 9: "fault guard", 10: "write-back",
         22: "HELPER w2: wait at (A)", 23: "HELPER: kinematics", 24: "HELPER: mesh contact detect (4 slots)", 25: "HELPER: (no LSTM)",
         26: "HELPER: wait at (A2)", 27: "HELPER: contact set-up share"}
calls = max(out[17], 1)
for k, n in names.items():
    print(f"{n:40s} per substep {out[k] / calls:10.0f} cycles")
print("active slots per wave-substep", out[16] / calls)
pairs = max(out[31], 1)
print(f"wave 2 of workgroup 0, per paired query: {out[28] / pairs:.1f} of 128 lane-queries issued, traversal steps: sum over lanes "
      f"{out[29] / pairs:.0f}, max over lanes {out[30] / pairs:.1f}")

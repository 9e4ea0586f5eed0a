"""Diagnostic: contact queries on a mesh terrain (task anymal_c_rough as registered: trimesh) with the -DLG_STAMPS build: queries
issued and cells / BVH nodes visited per wave-substep by wave 2 of workgroup 0 (slots 4-5), plus the phase shares of the main wave."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LGSTEP_LIB"] = os.path.join(ROOT, "extended_legged_gym_amd", "csrc", os.environ.get("LG_STAMPS_LIB", "liblgstep_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from extended_legged_gym_amd.envs import Anymal, AnymalCRoughCfg
from tools.bench_configs import sim_params
cfg = AnymalCRoughCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
np.random.seed(1)
env = Anymal(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
env.reset()
a = torch.randn(4096, 12, device="cuda")
n = 200
for i in range(n):
    env.step(a)
lib = env.core.lib
out = (C.c_ulonglong * 64)()
lib.lg_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.lg_debug_read_stamps(env.core.ctx, out)
calls = max(out[57], 1)
print("wave 2 of workgroup 0: %d detection calls; per call: %.1f queries issued (of 128), %.1f cells or nodes visited summed over lanes, %.1f by the busiest lane"
      % (calls, out[54] / calls, out[55] / calls, out[56] / calls))
print("cycles inside the queries per call (wave 2):", out[53] / calls, " clearance stage", out[22] / calls, " centre cell", out[23] / calls, " window scan: loads", out[25] / calls, " box tests", out[26] / calls, " exact tests + loop ends", out[24] / calls)
for k in list(range(16)) + [29]:
    print(k, out[k] // (n + 1) // 4, "cycles per substep")

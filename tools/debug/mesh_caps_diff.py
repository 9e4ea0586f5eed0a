"""Debug: HIP against the oracle on the staircase states of tests/test_mesh_capsules.py, with and without the capsule segments: which envs / bodies differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_mesh_capsules import stairs_setup, stairs_states, load_oracle
from extended_legged_gym_amd.native import NativeCore
from oracle.oracle_lib import OracleEnv
n = 256
cfg, ter, s, model = stairs_setup(n)
root, dof, geom = stairs_states(s, model, n, seed=2)
for caps in (0, 1):
    os.environ["LG_MESH_CAPS"] = str(caps)
    core = NativeCore(s, "cuda:0")
    core.t["friction_coeffs"].fill_(1.0)
    core.t["root_states"].copy_(torch.from_numpy(root)); core.t["dof_state"].copy_(torch.from_numpy(dof.reshape(tuple(core.t["dof_state"].shape)))); core.t["torques"].zero_()
    o = OracleEnv(s); o.L.lgo_set_mesh_caps(o.ctx, caps); load_oracle(o, root, dof)
    o.simulate(); core.simulate(); torch.cuda.synchronize()
    a = core.t["contact_forces"].cpu().numpy().reshape(n, -1, 3); b = o.t["contact_forces"].reshape(n, -1, 3)
    la, lb = np.linalg.norm(a, axis=2) > 1e-3, np.linalg.norm(b, axis=2) > 1e-3
    d = (la != lb)
    print(f"caps={caps}: envs with different loaded bodies {d.any(1).sum()} of {n}; per body (hip only / oracle only):")
    for bi, name in enumerate(model["body_names"]):
        if d[:, bi].any():
            print(f"   {name:10s} hip only {int((la & ~lb)[:, bi].sum()):4d}   oracle only {int((lb & ~la)[:, bi].sum()):4d}")
    err = np.abs(a - b).reshape(n, -1).max(1)
    same = ~d.any(1)
    print("   max |dF| among envs with the same bodies:", float(err[same].max()), " median", float(np.median(err[same])))
    core.close(); o.close()

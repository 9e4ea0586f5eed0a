"""Diagnostic (GPU box): error levels HIP vs oracle after ONE substep and after one full step from synced state, per solver."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.test_hip_vs_oracle import build, init_oracle, COPY, env_rows, contact_pattern
from extended_legged_gym_amd.native import NativeCore
from oracle.oracle_lib import OracleEnv

def levels(kind, solver, fric, n=256, seed=11):
    import tests.test_hip_vs_oracle as T
    cfg, s, terrain = build(kind, n, seed)
    s.cfg.solver_type = solver; s.cfg.friction_model = fric
    o = OracleEnv(s); core = NativeCore(s, "cuda:0")
    rng = init_oracle(o, cfg, s, terrain, n, seed)
    out = {}
    for it in range(40):
        act = rng.normal(size=(n, 12)).astype(np.float32)
        if it % 10 == 9:
            for name in COPY: core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            # one substep on copies of the state
            keep = {k: o.t[k].copy() for k in COPY}
            o.compute_torques(act); o.simulate()
            core.compute_torques(torch.from_numpy(act).cuda()); core.simulate(); torch.cuda.synchronize()
            differ = (contact_pattern(core.t["contact_forces"].cpu().numpy(), n) != contact_pattern(o.t["contact_forces"], n)).any(1)
            for name in ["root_states", "dof_state", "rigid_body_state", "contact_forces", "torques"]:
                a = env_rows(name, core.t[name].cpu().numpy(), n); b = env_rows(name, o.t[name], n)
                err = (np.abs(a - b) / np.maximum(1.0, np.abs(b)))[~differ]
                out.setdefault("sub_" + name, []).append([float(np.median(err)), float(np.quantile(err, 0.995)), float(err.max())])
            out.setdefault("sub_differ", []).append(int(differ.sum()))
            for k in COPY: o.t[k][...] = keep[k]
            for name in COPY: core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            o.step(act); core.step(torch.from_numpy(act).cuda()); torch.cuda.synchronize()
            differ = (contact_pattern(core.t["contact_forces"].cpu().numpy(), n) != contact_pattern(o.t["contact_forces"], n)).any(1)
            for name in ["root_states", "dof_state", "rigid_body_state", "contact_forces", "torques", "obs_buf"]:
                a = env_rows(name, core.t[name].cpu().numpy(), n); b = env_rows(name, o.t[name], n)
                err = (np.abs(a - b) / np.maximum(1.0, np.abs(b)))[~differ]
                out.setdefault("full_" + name, []).append([float(np.median(err)), float(np.quantile(err, 0.995)), float(err.max())])
            out.setdefault("full_differ", []).append(int(differ.sum()))
        else:
            o.step(act)
    core.close(); o.close()
    return out

if __name__ == "__main__":
    for kind in ("flat_pd", "flat_lstm", "rough_lstm"):
        for solver, fric in ((0, 0), (1, 0), (1, 1)):
            r = levels(kind, solver, fric)
            print(kind, "solver", solver, "fric", fric)
            for k, v in r.items():
                if k.endswith("differ"): print("   ", k, v)
                else:
                    v = np.array(v); print(f"    {k:24s} median {v[:,0].max():.2e}  q99.5 {v[:,1].max():.2e}  max {v[:,2].max():.2e}")

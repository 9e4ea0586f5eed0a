"""Diagnostic (test infrastructure, CPU oracle): play the reference's PhysX-trained walking policy and print gait statistics."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.helpers import ANYMAL_GAIT, sim_params_for
from tests.test_walk_policy import play_cfg, numpy_actor, load_policy_fixture, CMDS
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from oracle.oracle_lib import OracleEnv


def physx_overrides():
    """WALK_SOLVER=pgs|tgs, WALK_FRICTION=cone|pyramid, WALK_ITERS=n, WALK_ERP=x select the physics variant (cfg.sim.physx)."""
    o = {}
    if "WALK_SOLVER" in os.environ: o["solver_type"] = {"pgs": 0, "tgs": 1}[os.environ["WALK_SOLVER"]]
    if "WALK_FRICTION" in os.environ: o["friction_model"] = os.environ["WALK_FRICTION"]
    if "WALK_ITERS" in os.environ: o["num_position_iterations"] = int(os.environ["WALK_ITERS"])
    if "WALK_ERP" in os.environ: o["penetration_recovery"] = float(os.environ["WALK_ERP"])
    return o


def run(n=96, steps=400, payload=0.0, mu=1.0, trace_env=None, verbose=True, physx=None):
    cfg = play_cfg(n)
    for k, v in (physx_overrides() if physx is None else physx).items():
        setattr(cfg.sim.physx, k, v)
    setup = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=1, gait=ANYMAL_GAIT)
    o = OracleEnv(setup)
    o.t["friction_coeffs"][:] = mu
    o.t["base_mass_added"][:] = payload
    o.reset_idx(np.arange(n))
    act = numpy_actor(load_policy_fixture())
    vx_cmd = np.array(CMDS, np.float32)[np.arange(n) % len(CMDS)]
    cmd = np.zeros((n, 4), np.float32); cmd[:, 0] = vx_cmd
    o.t["commands"][:] = cmd
    o.step(np.zeros((n, 12), np.float32))
    fallen = np.zeros(n, bool); fall_steps = []
    vx = np.zeros((steps, n), np.float32)
    feet = setup.feet_indices if hasattr(setup, "feet_indices") else None
    contact = np.zeros((steps, n, 4), bool); tq = np.zeros((steps, n, 12), np.float32); bz = np.zeros((steps, n), np.float32); pg = np.zeros((steps, n, 3), np.float32)
    fi = [4, 8, 12, 16]
    slip = []; fxs = []; slipv = []
    for it in range(steps):
        o.t["commands"][:] = cmd
        obs = o.t["obs_buf"].copy()
        obs[:, 9:12] = cmd[:, :3] * np.array([2.0, 2.0, 0.25], np.float32)
        o.step(act(obs))
        vx[it] = o.t["base_lin_vel"][:, 0]
        term = (o.t["reset_buf"] != 0) & (o.t["time_out_buf"] == 0)
        if it >= 100:
            fallen |= term
        fall_steps += [(it, int(e)) for e in np.nonzero(term)[0]]
        cf = o.t["contact_forces"].reshape(n, -1, 3)
        contact[it] = cf[:, fi, 2] > 1.0
        rb = o.t['rigid_body_state'].reshape(n, -1, 13)
        if it >= 100:
            sp = np.linalg.norm(rb[:, fi, 7:9], axis=-1); slip.append(sp[contact[it]].mean()); slipv.append(sp[contact[it] & contact[it-1] & contact[it-2]]); fxs.append((np.linalg.norm(cf[:, fi, :2], axis=-1) / np.maximum(cf[:, fi, 2], 1e-3))[contact[it]].mean())
        tq[it] = o.t["torques"]; bz[it] = o.t["root_states"][:, 2]; pg[it] = o.t["projected_gravity"]
    res = dict(frac_fallen=float(fallen.mean()), slip_speed=float(np.mean(slip)), slip_q=np.quantile(np.concatenate(slipv), [0.1, 0.25, 0.5, 0.75, 0.9]).round(3).tolist(), ft_over_fn=float(np.mean(fxs)), per_cmd_fallen=[float(fallen[vx_cmd == c].mean()) for c in CMDS], early_fallen=float(len(set(f[1] for f in fall_steps if f[0] < 100)) / n), late_fall_steps=sorted(f[0] for f in fall_steps if f[0] >= 100), track_err=float(np.abs(vx[100:] - vx_cmd).mean()),
               duty=contact[100:].mean(axis=(0, 1)).round(3).tolist(), tq_absmax=np.abs(tq[100:]).max(axis=(0, 1)).round(1).tolist(),
               tq_rms=np.sqrt((tq[100:] ** 2).mean(axis=(0, 1))).round(1).tolist(), base_z=float(bz[100:].mean()), pitch_gx=float(pg[100:, :, 0].mean()))
    if verbose:
        print(res)
    if trace_env is not None:
        e = trace_env
        for it in range(150, 200):
            print(it, "".join("X" if c else "." for c in contact[it, e]), np.round(tq[it, e], 0).astype(int).tolist(), round(float(bz[it, e]), 3), np.round(pg[it, e], 2).tolist(), round(float(vx[it, e]), 2))
    o.close()
    return res


if __name__ == "__main__":
    payload = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    mu = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    t0 = time.time()
    run(payload=payload, mu=mu, trace_env=2)
    print("sec", time.time() - t0)

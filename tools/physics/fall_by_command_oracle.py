"""Diagnostic (CPU, the oracle's physics): tools/physics/fall_by_command.py on the scalar restatement, so that a change of the contact model can be looked at
before it is written as a kernel.  Task anymal_c_flat as registered, the reference's PhysX-trained checkpoint played deterministically, no observation noise,
no pushes; steady-state (>= 100 steps after a reset) contact terminations binned by the command in force.

    python tools/physics/fall_by_command_oracle.py [envs] [steps] [key=value ...]      (keys: sim.physx.* attributes, e.g. friction_anchors=1)
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(n=1024, steps=800, over=()):
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
    from oracle.oracle_lib import OracleEnv
    from tests.helpers import ANYMAL_GAIT, sim_params_for
    from tests.test_walk_policy import numpy_actor, load_policy_fixture
    cfg = AnymalCFlatCfg()
    cfg.env.num_envs = n; cfg.seed = 1
    cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False
    anchors = 0
    for kv in over:
        k, v = kv.split("=")
        if k == "friction_anchors":
            anchors = int(v); continue
        old = getattr(cfg.sim.physx, k, None)
        setattr(cfg.sim.physx, k, v if isinstance(old, str) else (int(float(v)) if isinstance(old, (int, bool)) or old is None else float(v)))
    setup = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=1, gait=ANYMAL_GAIT)
    o = OracleEnv(setup)
    o.L.lgo_set_friction_anchors(o.ctx, anchors)       # the oracle's experiment switch (oracle/lg_oracle.cpp, lgo_ctx::friction_anchors)
    rng = np.random.default_rng(1)
    if cfg.domain_rand.randomize_friction:
        lo, hi = cfg.domain_rand.friction_range
        o.t["friction_coeffs"][:] = rng.uniform(lo, hi, n).astype(np.float32)
    if cfg.domain_rand.randomize_base_mass:
        lo, hi = cfg.domain_rand.added_mass_range
        o.t["base_mass_added"][:] = rng.uniform(lo, hi, n).astype(np.float32)
    o.reset_idx(np.arange(n))
    act = numpy_actor(load_policy_fixture())
    o.step(np.zeros((n, 12), np.float32))
    names = ("fwd", "back", "lateral", "yaw", "lateral+yaw", "stand")
    bins = {k: [0, 0] for k in names}
    age = np.zeros(n, np.int64)
    verr = []
    for t in range(steps):
        cmd = o.t["commands"].copy()
        o.step(act(o.t["obs_buf"].copy()))
        done = o.t["reset_buf"] != 0
        term = done & (o.t["time_out_buf"] == 0) & (age >= 100)
        vx, vy, w = cmd[:, 0], cmd[:, 1], cmd[:, 2]
        lat, yaw = np.abs(vy) >= 0.3, np.abs(w) >= 0.5
        zero = (vx == 0) & (vy == 0)
        sel = {"fwd": ~lat & ~yaw & (vx > 0) & ~zero, "back": ~lat & ~yaw & (vx < 0) & ~zero, "lateral": lat & ~yaw, "yaw": yaw & ~lat,
               "lateral+yaw": lat & yaw, "stand": zero & ~yaw}
        steady = age >= 100
        for k, m in sel.items():
            bins[k][0] += int((m & steady).sum()); bins[k][1] += int((m & term).sum())
        bv = o.t["base_lin_vel"]
        verr.append(float(np.sqrt(((bv[steady, :2] - cmd[steady, :2]) ** 2).sum(1)).mean()) if steady.any() else 0.0)
        age = np.where(done, 0, age + 1)
    out = {k: dict(env_steps=v[0], falls=v[1], falls_per_env_step=round(v[1] / max(v[0], 1), 5)) for k, v in bins.items()}
    tot = [sum(v[0] for v in bins.values()), sum(v[1] for v in bins.values())]
    out["all"] = dict(env_steps=tot[0], falls=tot[1], falls_per_env_step=round(tot[1] / max(tot[0], 1), 5))
    out["mean_xy_tracking_error"] = round(float(np.mean(verr[100:])), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if "=" not in x]
    main(int(a[0]) if a else 1024, int(a[1]) if len(a) > 1 else 800, [x for x in sys.argv[1:] if "=" in x])

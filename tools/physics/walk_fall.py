import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.helpers import ANYMAL_GAIT, sim_params_for
from tests.test_walk_policy import play_cfg, numpy_actor, load_policy_fixture, CMDS
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from oracle.oracle_lib import OracleEnv
n=int(os.environ.get("WN","24")); steps=400; only=int(os.environ.get("WCMD","-1"))
payload=float(sys.argv[1]); mu=float(sys.argv[2])
cfg = play_cfg(n)
setup = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=1, gait=ANYMAL_GAIT)
o = OracleEnv(setup)
o.t["friction_coeffs"][:] = mu; o.t["base_mass_added"][:] = payload
o.reset_idx(np.arange(n))
act = numpy_actor(load_policy_fixture())
vx_cmd = np.array(CMDS, np.float32)[np.arange(n) % len(CMDS)]
cmd = np.zeros((n, 4), np.float32); cmd[:, 0] = vx_cmd
o.t["commands"][:] = cmd
o.step(np.zeros((n, 12), np.float32))
hist=[]
first_fall=None
for it in range(steps):
    o.t["commands"][:] = cmd
    obs = o.t["obs_buf"].copy(); obs[:, 9:12] = cmd[:, :3] * np.array([2.0, 2.0, 0.25], np.float32)
    a = act(obs)
    o.step(a)
    term = (o.t["reset_buf"] != 0) & (o.t["time_out_buf"] == 0)
    cf = o.t["contact_forces"].reshape(n, -1, 3)
    hist.append((cf[:, [4,8,12,16], 2].copy(), o.t["torques"].copy(), o.t["root_states"][:, 2].copy(), o.t["projected_gravity"].copy(), o.t["base_lin_vel"][:,0].copy(), a.copy(), o.t["dof_state"].reshape(n,12,2)[:,:,0].copy(), cf[:, [3,7,11,15],2].copy()))
    if only >= 0: term = term & (np.arange(n) % 3 == only)
    if it>100 and term.any() and first_fall is None:
        first_fall=(it, int(np.nonzero(term)[0][0])); break
print("first fall", first_fall)
if first_fall:
    it0,e=first_fall
    for it in range(max(0,it0-70), it0):
        h=hist[it]
        print(it, "Fz", np.round(h[0][e]).astype(int).tolist(), "shank", np.round(h[7][e]).astype(int).tolist(), "tq", np.round(h[1][e]).astype(int).tolist(), "z %.3f"%h[2][e], "g", np.round(h[3][e],2).tolist(), "vx %.2f"%h[4][e])

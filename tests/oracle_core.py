"""TEST INFRASTRUCTURE: the CPU oracle behind `NativeCore`'s Python surface.

The env classes talk to the step through `NativeCore` (ctypes -> liblgstep.so, GPU only: the product has no CPU path).  A few
things can only be checked in the build container, where the reference tree is importable but no GPU exists -- above all whether the
reference's own rsl_rl `OnPolicyRunner` accepts the env classes as they are.  For those tests `OracleCore` stands in for `NativeCore`
(monkeypatched into `envs.base.legged_robot`): same tensor names, same calls, the oracle's arrays wrapped zero-copy as torch-CPU tensors.
Never imported by the product (tests/test_abi.py checks that)."""
import numpy as np
import torch

from oracle.oracle_lib import OracleEnv


class OracleCore:
    def __init__(self, setup, device):
        if torch.device(device).type != "cpu":
            raise RuntimeError("OracleCore is the CPU checker; the product path is NativeCore on a GPU")
        self.setup, self.device = setup, torch.device("cpu")
        self.collision_mesh = None
        self.o = OracleEnv(setup)
        self.t = {name: torch.from_numpy(a) for name, a in self.o.t.items()}      # views of the oracle's own buffers

    @staticmethod
    def _np(x):
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    def step(self, actions):
        self.o.step(self._np(actions))

    def compute_torques(self, actions=None):
        self.o.compute_torques(None if actions is None else self._np(actions))

    def simulate(self):
        self.o.simulate()

    def post_physics_step(self):
        self.o.post_physics_step()

    def reset_idx(self, env_ids, update_curriculum=0):
        self.o.reset_idx(self._np(env_ids), int(update_curriculum))

    def set_reward_terms(self, term_ids, scales):
        self.o.set_reward_terms(term_ids, scales)

    def close(self):
        self.o.close()

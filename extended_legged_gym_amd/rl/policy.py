"""`NativeActorCritic`: the feed-forward `ActorCritic` of the vendored rsl_rl (`modules/actor_critic.py:16-136`) with its two
MLPs evaluated by `lg_mlp_forward` / `lg_policy_act` (include/lgpolicy.h): all layers of a network in one launch on the fp32
matrix cores, and `PPO.act` (`algorithms/ppo.py:147-159`: sample, value, log-prob, mean, sigma) as ONE launch instead of
~25.  Inference only: the weights come from a trained / initialised torch `ActorCritic` (its `state_dict`), gradients
stay in PyTorch."""
import ctypes as C

import numpy as np
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.native import load_library


def _lib():
    lib = load_library()
    if not getattr(lib, "_policy_declared", False):
        abi.declare_policy(lib)
        lib._policy_declared = True
    return lib


class NativeMLP:
    """nn.Sequential(Linear, act, ..., Linear) on the GPU.  `layers`: list of (weight (out, in), bias (out))."""

    def __init__(self, layers, activation="elu", device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("the policy kernels run on the GPU only (no CPU path)")
        self.lib, self.device = _lib(), dev
        ws = [np.ascontiguousarray(np.asarray(w, dtype=np.float32)) for w, _ in layers]
        bs = [np.ascontiguousarray(np.asarray(b, dtype=np.float32)) for _, b in layers]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for i, w in enumerate(ws):
            assert w.shape == (dims[i + 1], dims[i]) and bs[i].shape == (dims[i + 1],)
        L = len(ws)
        fp = C.POINTER(C.c_float)
        wp = (fp * L)(*[w.ctypes.data_as(fp) for w in ws])
        bp = (fp * L)(*[b.ctypes.data_as(fp) for b in bs])
        self.dims = dims
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.handle = self.lib.lg_mlp_create(L, (C.c_int32 * (L + 1))(*dims), wp, bp, abi.ACTIVATIONS[activation], index)
        if not self.handle:
            raise RuntimeError("lg_mlp_create failed: " + (self.lib.lg_mlp_last_error(None) or b"").decode())

    @classmethod
    def from_sequential_state(cls, state, prefix, activation="elu", device="cuda:0"):
        """Layers `prefix.0.weight`, `prefix.2.weight`, ... of an nn.Sequential state dict."""
        idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in state if k.startswith(prefix + ".") and k.endswith(".weight")})
        layers = [(state[f"{prefix}.{i}.weight"].detach().cpu().numpy(), state[f"{prefix}.{i}.bias"].detach().cpu().numpy()) for i in idx]
        return cls(layers, activation, device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __call__(self, x):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        assert x.dim() == 2 and x.shape[1] == self.dims[0]
        y = torch.empty(x.shape[0], self.dims[-1], device=self.device)
        rc = self.lib.lg_mlp_forward(self.handle, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(y.data_ptr()), self._stream())
        if rc != abi.LG_OK:
            raise RuntimeError("lg_mlp_forward failed: " + (self.lib.lg_mlp_last_error(self.handle) or b"").decode())
        return y

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.lg_mlp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeActorCritic:
    """Same surface as `ActorCritic` for rollout collection: `act`, `act_inference`, `evaluate`, `get_actions_log_prob`,
    `action_mean`, `action_std`, `entropy` (`actor_critic.py:96-136`)."""
    is_recurrent = False

    def __init__(self, state_dict, activation="elu", noise_std_type="scalar", device="cuda:0", seed=0):
        self.device = torch.device(device)
        self.actor = NativeMLP.from_sequential_state(state_dict, "actor", activation, device)
        self.critic = NativeMLP.from_sequential_state(state_dict, "critic", activation, device)
        self.noise_std_type = noise_std_type
        if noise_std_type == "scalar":
            self.std = state_dict["std"].detach().to(self.device, torch.float32).contiguous()
        elif noise_std_type == "log":
            self.std = torch.exp(state_dict["log_std"].detach().to(self.device, torch.float32)).contiguous()
        else:
            raise ValueError(f"Unknown standard deviation type: {noise_std_type}. Should be 'scalar' or 'log'")
        self.num_actions = self.actor.dims[-1]
        self.seed, self._call = int(seed), 0
        self._mean = self._actions = self._logp = self._values = None

    def reset(self, dones=None):
        pass

    def _run(self, obs, critic_obs, deterministic):
        lib = self.actor.lib
        obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
        cobs = obs if critic_obs is None else critic_obs.to(device=self.device, dtype=torch.float32).contiguous()
        n = obs.shape[0]
        self._actions = torch.empty(n, self.num_actions, device=self.device)
        self._mean = torch.empty(n, self.num_actions, device=self.device)
        self._logp = torch.empty(n, device=self.device)
        self._values = torch.empty(n, self.critic.dims[-1], device=self.device)
        self._call += 1
        rc = lib.lg_policy_act(self.actor.handle, self.critic.handle, C.c_void_p(obs.data_ptr()), C.c_void_p(cobs.data_ptr()), n,
                               C.c_void_p(self.std.data_ptr()), self.seed, self._call, int(deterministic),
                               C.c_void_p(self._actions.data_ptr()), C.c_void_p(self._mean.data_ptr()),
                               C.c_void_p(self._logp.data_ptr()), C.c_void_p(self._values.data_ptr()), self.actor._stream())
        if rc != abi.LG_OK:
            raise RuntimeError("lg_policy_act failed: " + (lib.lg_mlp_last_error(self.actor.handle) or b"").decode())

    def act_and_evaluate(self, obs, critic_obs=None):
        """`PPO.act` in one launch: (actions, values, actions_log_prob, action_mean, action_sigma)."""
        self._run(obs, critic_obs, False)
        return self._actions, self._values, self._logp, self._mean, self.action_std

    def act(self, observations, **kwargs):
        self._run(observations, None, False)
        return self._actions

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)

    def get_actions_log_prob(self, actions):
        if actions is self._actions:
            return self._logp
        sd = self.std
        return (-((actions - self._mean) ** 2) / (2 * sd * sd) - torch.log(sd) - 0.9189385332046727).sum(dim=-1)

    @property
    def action_mean(self):
        return self._mean

    @property
    def action_std(self):
        return self.std.expand_as(self._mean)

    @property
    def entropy(self):
        return (0.5 + 0.9189385332046727 + torch.log(self.std)).sum().expand(self._mean.shape[0])

"""`RolloutStorage.compute_returns` (`rsl_rl/storage/rollout_storage.py:145-167`) as one kernel (one lane per env, the
24-step recursion in registers) plus a deterministic normalisation pass, instead of a Python loop over (N, 1) tensors."""
import ctypes as C

import torch

from extended_legged_gym_amd import abi
from .policy import _lib


def compute_returns(rewards, dones, values, last_values, gamma, lam, normalize_advantage=True):
    """rewards / dones / values: (T, N, 1) or (T, N); last_values (N, 1) or (N).  Returns (returns, advantages) shaped
    like `values`."""
    lib = _lib()
    shape = values.shape
    T, N = shape[0], shape[1]
    dev = values.device
    r = rewards.reshape(T, N).to(torch.float32).contiguous()
    d = dones.reshape(T, N).to(torch.float32).contiguous()
    v = values.reshape(T, N).to(torch.float32).contiguous()
    lv = last_values.reshape(N).to(torch.float32).contiguous()
    ret, adv = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.lg_compute_returns(C.c_void_p(r.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(lv.data_ptr()),
                                T, N, float(gamma), float(lam), int(bool(normalize_advantage)), C.c_void_p(ret.data_ptr()),
                                C.c_void_p(adv.data_ptr()), stream)
    if rc != abi.LG_OK:
        raise RuntimeError(f"lg_compute_returns failed ({rc})")
    return ret.reshape(shape), adv.reshape(shape)

"""Loader and thin ctypes binding of the HIP env-step library (`csrc/liblgstep.so`, C ABI in `include/lgstep.h`).

PyTorch is plumbing here: it owns the device arena (one uint8 tensor) and hands out zero-copy typed views of it, and
its current HIP stream is the stream every library call is enqueued on.  There is NO CPU path: if the library is
missing, or no GPU is present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np
import torch

from extended_legged_gym_amd import abi

LIB_PATH = os.environ.get("LGSTEP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblgstep.so")
_TORCH_DTYPE = {abi.LG_F32: torch.float32, abi.LG_I64: torch.int64, abi.LG_U8: torch.uint8, abi.LG_I16: torch.int16,
                abi.LG_I32: torch.int32, abi.LG_F64: torch.float64}
_lib = None


def load_library():
    """dlopen liblgstep.so and check that its struct layouts are the ones this Python side was written against."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (or `make -C extended_legged_gym_amd/csrc`). There is no CPU fallback.")
    lib = abi.declare_product(C.CDLL(LIB_PATH))
    sizes = (C.c_int32 * 4)()
    lib.lg_abi_sizes(sizes)
    want = [abi.LG_ABI_VERSION, C.sizeof(abi.lg_config), C.sizeof(abi.lg_robot_model), C.sizeof(abi.lg_terrain)]
    if list(sizes) != want:
        raise RuntimeError(f"ABI mismatch between liblgstep.so {list(sizes)} and extended_legged_gym_amd/abi.py {want}")
    _lib = lib
    return lib


class NativeCore:
    """One env-step context on one GPU; `self.t[name]` are torch views of the library's tensors."""

    def __init__(self, setup, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"the env step runs on an MI355X GPU only (got sim_device='{device}'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to PyTorch-ROCm; the env step has no CPU path")
        self.lib = load_library()
        self.setup = setup
        self.device = dev
        self.device_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.collision_mesh = None
        if setup.terrain.mesh_type == abi.LG_MESH_TRIMESH:
            # the triangle soup the reference hands to gym.add_triangle_mesh; BVH built once, owned by this core
            from extended_legged_gym_amd.utils.mesh import DeviceMesh
            self.collision_mesh = DeviceMesh(setup.collision_vertices, setup.collision_triangles, device=dev)
            setup.terrain.collision_mesh = self.collision_mesh.handle
        nbytes = self.lib.lg_arena_bytes(C.byref(setup.cfg), C.byref(setup.model), C.byref(setup.terrain))
        if nbytes == 0:
            raise ValueError("lg_arena_bytes rejected the configuration: " + self._err(None))
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        self.ctx = self.lib.lg_create(C.byref(setup.cfg), C.byref(setup.model), C.byref(setup.terrain),
                                      self.device_index, C.c_void_p(self.arena.data_ptr()))
        if not self.ctx:
            raise RuntimeError("lg_create failed: " + self._err(None))
        self.t = {}
        base = self.arena.data_ptr()
        for name, tid in abi.TENSOR_ID.items():
            p, shp, nd, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
            self._check(self.lib.lg_get_tensor(self.ctx, tid, C.byref(p), shp, C.byref(nd), C.byref(dt)))
            shape = tuple(shp[i] for i in range(nd.value))
            tdt = _TORCH_DTYPE[dt.value]
            n = int(np.prod(shape)) * torch.empty((), dtype=tdt).element_size()
            off = p.value - base
            self.t[name] = self.arena[off:off + n].view(tdt).view(shape)

    # ------------------------------------------------------------------ helpers
    def _err(self, ctx):
        s = self.lib.lg_last_error(ctx)
        return s.decode() if s else ""

    def _check(self, rc):
        if rc != abi.LG_OK:
            raise RuntimeError(f"liblgstep call failed ({rc}): {self._err(self.ctx)}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _f32(t):
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        return t

    # ------------------------------------------------------------------ ABI calls
    def step(self, actions):
        a = self._f32(actions)
        self._check(self.lib.lg_step(self.ctx, C.c_void_p(a.data_ptr()), self._stream()))

    def set_extra_obs(self, rows):
        """Bind the (N, num_extra_obs) float32 device tensor appended to every observation row."""
        assert rows.is_contiguous() and rows.dtype == torch.float32
        self._extra_obs = rows
        self._check(self.lib.lg_set_extra_obs(self.ctx, C.c_void_p(rows.data_ptr())))

    def set_extra_termination(self, flags):
        """Bind the (N,) uint8 device tensor of per-env flags that the next post-physics steps OR into their contact terminations (None: unbind)."""
        assert flags is None or (flags.is_contiguous() and flags.dtype == torch.uint8)
        self._extra_term = flags
        self._check(self.lib.lg_set_extra_termination(self.ctx, C.c_void_p(flags.data_ptr() if flags is not None else 0)))

    def compute_torques_and_simulate(self, actions):
        """The physics half of `lg_step` (clip actions, `decimation` x (actuators + one dt)), without post-physics."""
        a = self._f32(actions)
        self._check(self.lib.lg_step_physics(self.ctx, C.c_void_p(a.data_ptr()), self._stream()))

    def set_reward_terms(self, term_ids, scales):
        """New active reward terms (evaluation order, dt-scaled scales); zeroes every episode sum."""
        n = len(term_ids)
        ids = (C.c_int32 * max(n, 1))(*[int(i) for i in term_ids])
        sc = (C.c_float * max(n, 1))(*[float(x) for x in scales])
        self._check(self.lib.lg_set_reward_terms(self.ctx, n, ids, sc, self._stream()))

    def set_async_gait(self, weights, foot_z_align):
        """`AsyncGaitScheduler` term: stage weights (dof_align, dof_nominal_pos, reward_foot_z_align) and the spawn-pose constant."""
        w = (C.c_float * 3)(*[float(x) for x in weights])
        self._check(self.lib.lg_set_async_gait(self.ctx, w, float(foot_z_align), self._stream()))

    def step_subset_physics(self, actions, env_ids_i32):
        a = self._f32(actions)
        self._check(self.lib.lg_step_subset_physics(self.ctx, C.c_void_p(a.data_ptr()), C.c_void_p(env_ids_i32.data_ptr()),
                                                    int(env_ids_i32.numel()), self._stream()))

    def post_physics_subset(self, env_ids_i32, rollout_mode):
        self._check(self.lib.lg_post_physics_subset(self.ctx, C.c_void_p(env_ids_i32.data_ptr()), int(env_ids_i32.numel()),
                                                    int(rollout_mode), self._stream()))

    def step_subset(self, actions, env_ids_i32, rollout_mode):
        a = self._f32(actions)
        self._check(self.lib.lg_step_subset(self.ctx, C.c_void_p(a.data_ptr()), C.c_void_p(env_ids_i32.data_ptr()),
                                            int(env_ids_i32.numel()), int(rollout_mode), self._stream()))

    def _step_row_buffers(self, n):
        dev = self.t["obs_buf"].device
        return (torch.empty((n, int(self.t["obs_buf"].shape[1])), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
                torch.empty(n, dtype=self.t["reset_buf"].dtype, device=dev), torch.empty(n, dtype=self.t["time_out_buf"].dtype, device=dev))

    def gather_step_rows(self, env_ids_i32):
        """Fresh dense copies (obs, rew, reset, time_outs) of the listed envs' rows, one launch (`lg_gather_step_rows`)."""
        n = int(env_ids_i32.numel())
        obs, rew, reset, tout = self._step_row_buffers(n)
        self._check(self.lib.lg_gather_step_rows(self.ctx, C.c_void_p(env_ids_i32.data_ptr()), n, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()),
                                                 C.c_void_p(reset.data_ptr()), C.c_void_p(tout.data_ptr()), self._stream()))
        return obs, rew, reset.view(torch.bool), tout.view(torch.bool)

    def step_subset_rows(self, actions, env_ids_i32, rollout_mode):
        """`step_subset` + `gather_step_rows` as one library call (`lg_step_subset_rows`): a rollout step hands the rows back from its one launch."""
        a = self._f32(actions)
        n = int(env_ids_i32.numel())
        obs, rew, reset, tout = self._step_row_buffers(n)
        self._check(self.lib.lg_step_subset_rows(self.ctx, C.c_void_p(a.data_ptr()), C.c_void_p(env_ids_i32.data_ptr()), n, int(rollout_mode),
                                                 C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(reset.data_ptr()), C.c_void_p(tout.data_ptr()),
                                                 self._stream()))
        return obs, rew, reset.view(torch.bool), tout.view(torch.bool)

    def sync_main_to_rollout(self, rollouts_per_main, pos_drift=0.0):
        self._check(self.lib.lg_sync_main_to_rollout(self.ctx, int(rollouts_per_main), float(pos_drift), self._stream()))

    def rollout_batch(self, all_us, env_ids_i32, rollouts_per_main, pos_drift=0.0):
        """(n, H, num_dof) plan -> (n, H) rewards: sync, H rollout steps, sync, enqueued by one library call."""
        a = self._f32(all_us)
        n, H = int(a.shape[0]), int(a.shape[1])
        nd = int(self.t["actions"].shape[1])
        if n != int(env_ids_i32.numel()) or a.dim() != 3 or a.shape[2] != nd:
            raise ValueError(f"expected a plan of shape ({int(env_ids_i32.numel())}, H, {nd}), got {tuple(a.shape)}")
        rew = torch.empty(n, H, dtype=torch.float32, device=self.device)
        self._check(self.lib.lg_rollout_batch(self.ctx, C.c_void_p(a.data_ptr()), H, C.c_void_p(env_ids_i32.data_ptr()), n,
                                              int(rollouts_per_main), float(pos_drift), C.c_void_p(rew.data_ptr()), self._stream()))
        return rew

    def compute_torques(self, actions=None):
        if actions is None:
            self._check(self.lib.lg_compute_torques(self.ctx, None, self._stream()))
        else:
            a = self._f32(actions)
            self._check(self.lib.lg_compute_torques(self.ctx, C.c_void_p(a.data_ptr()), self._stream()))

    def simulate(self):
        self._check(self.lib.lg_simulate(self.ctx, self._stream()))

    def post_physics_step(self):
        self._check(self.lib.lg_post_physics_step(self.ctx, self._stream()))

    def reset_idx(self, env_ids, update_curriculum=0):
        ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
        self._check(self.lib.lg_reset_idx(self.ctx, C.c_void_p(ids.data_ptr()), int(ids.numel()), int(update_curriculum),
                                          self._stream()))

    def set_state_indexed(self, env_ids, root_states=None, dof_state=None):
        """Teleport the listed envs (`gym.set_actor_root_state_tensor_indexed` / `set_dof_state_tensor_indexed`): rows of the
        given FULL tensors (default: the library's own, i.e. whatever the caller wrote into the views) become the state, and
        the rigid-body states of those envs are recomputed from it."""
        ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
        r = None if root_states is None else self._f32(root_states)
        d = None if dof_state is None else self._f32(dof_state)
        self._check(self.lib.lg_set_state_indexed(self.ctx, C.c_void_p(r.data_ptr()) if r is not None else None,
                                                  C.c_void_p(d.data_ptr()) if d is not None else None,
                                                  C.c_void_p(ids.data_ptr()), int(ids.numel()), self._stream()))

    def profile_begin(self, max_samples=256, stride=1):
        self._check(self.lib.lg_profile_begin(self.ctx, int(max_samples), int(stride)))

    def profile_end(self):
        ms, n = (C.c_float * 3)(), C.c_int32()
        self._check(self.lib.lg_profile_end(self.ctx, ms, C.byref(n)))
        return dict(physics_ms=ms[0], post_ms=ms[1], finalize_ms=ms[2], samples=n.value)

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize(self.device)
            self.lib.lg_destroy(self.ctx)
            self.ctx = None
        if getattr(self, "collision_mesh", None) is not None:
            self.collision_mesh.close()
            self.collision_mesh = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Device triangle mesh (BVH on the GPU) — the stand-in for `wp.Mesh` / `convert_to_warp_mesh`
(reference `utils/ray_caster.py:23-42`).  Thin ctypes wrapper over `lg_mesh_*` of include/lgstep.h."""
import ctypes as C

import numpy as np
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.native import load_library


class DeviceMesh:
    def __init__(self, vertices, triangles, device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("mesh queries run on the GPU only (no CPU path)")
        self.lib = load_library()
        self.device = dev
        v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32).reshape(-1, 3))
        t = np.ascontiguousarray(np.asarray(triangles, dtype=np.int32).reshape(-1, 3))
        self.num_vertices, self.num_triangles = len(v), len(t)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.handle = self.lib.lg_mesh_create(v.ctypes.data_as(C.POINTER(C.c_float)), len(v),
                                              t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), index)
        if not self.handle:
            raise RuntimeError("lg_mesh_create failed: " + (self.lib.lg_mesh_last_error(None) or b"").decode())
        info = (C.c_int64 * 2)()
        self.lib.lg_mesh_info(self.handle, info)
        self.num_bvh_nodes = int(info[1])
        lat = (C.c_int32 * 2)()
        self.lib.lg_mesh_ray_lattice(self.handle, lat)
        self.ray_lattice = (int(lat[0]), int(lat[1]))      # (0, 0): rays walk the BVH
        self.lib.lg_mesh_contact_lattice(self.handle, lat)
        self.contact_lattice = (int(lat[0]), int(lat[1]))  # (0, 0): the physics kernel's contact queries walk the BVH

    def _check(self, rc):
        if rc != abi.LG_OK:
            raise RuntimeError(f"mesh query failed ({rc}): " + (self.lib.lg_mesh_last_error(self.handle) or b"").decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.lg_mesh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convert_to_warp_mesh(vertices, triangles, device="cuda:0"):
    """Name kept from the reference (`ray_caster.py:23-42`); returns a `DeviceMesh`."""
    return DeviceMesh(vertices, triangles, device)


def plane_mesh(size=100.0):
    """The two-triangle ground plane the reference builds for `mesh_type='plane'` (`legged_robot_raycast.py:198-213`)."""
    v = np.array([[-size, -size, 0.0], [size, -size, 0.0], [size, size, 0.0], [-size, size, 0.0]], dtype=np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    return v, t

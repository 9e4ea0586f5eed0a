"""Signed-distance queries against a triangle mesh with the reference's interface (`utils/mesh_sdf.py:118-336`):
`MeshSDFCfg`, `MeshSDFData`, `MeshSDF.query / nearest_points`.  The Warp kernel + numpy round trip (`:268-293`) is
`lg_mesh_query_sdf`: one launch for any number of points, device-resident, no byte-string cache needed."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Tuple

import torch

from extended_legged_gym_amd.utils.mesh import DeviceMesh


@dataclass
class MeshSDFCfg:
    mesh_paths: List[str] = field(default_factory=list)
    vertices: torch.Tensor = None
    triangles: torch.Tensor = None
    default_sdf_value: float = 1000.0
    max_distance: float = 100.0
    enable_caching: bool = False


@dataclass
class MeshSDFData:
    sdf_values: torch.Tensor = None
    sdf_gradients: torch.Tensor = None


class MeshSDF:
    def __init__(self, cfg: MeshSDFCfg, device: str = "cuda:0", mesh: DeviceMesh = None):
        self.cfg, self.device = cfg, device
        self.meshes = {}
        if mesh is not None:
            self.meshes["custom_mesh"] = mesh
        elif cfg.mesh_paths:
            from extended_legged_gym_amd.utils.obj_io import load_obj
            for path in cfg.mesh_paths:
                v, t = load_obj(path)
                self.meshes[path] = DeviceMesh(v, t, device)
        elif cfg.vertices is not None and cfg.triangles is not None:
            self.meshes["custom_mesh"] = DeviceMesh(torch.as_tensor(cfg.vertices).cpu().numpy(),
                                                    torch.as_tensor(cfg.triangles).cpu().numpy(), device)
        else:
            raise ValueError("No mesh paths or vertices/triangles provided for SDF calculation.")
        self._data = MeshSDFData()
        self._is_initialized = True

    def query(self, points: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """SDF values and unit gradients for points of shape (B, P, 3) or (P, 3)."""
        if points.dim() not in (2, 3):
            raise ValueError(f"Expected points to have rank 2 or 3, got {points.dim()}")
        mesh = next(iter(self.meshes.values()))
        shape = points.shape
        p = points.reshape(-1, 3).to(device=mesh.device, dtype=torch.float32).contiguous()
        sdf = torch.empty(p.shape[0], device=mesh.device)
        grad = torch.empty_like(p)
        mesh._check(mesh.lib.lg_mesh_query_sdf(mesh.handle, C.c_void_p(p.data_ptr()), p.shape[0], float(self.cfg.max_distance),
                                               C.c_void_p(sdf.data_ptr()), C.c_void_p(grad.data_ptr()), mesh._stream()))
        self._data.sdf_values, self._data.sdf_gradients = sdf.reshape(shape[:-1]), grad.reshape(shape)
        return self._data.sdf_values, self._data.sdf_gradients

    def query_bodies(self, rigid_body_state, num_bodies, body_indices_i32, sphere_offsets, sdf_rows, gradients=None,
                     nearest=None, env_ids_i32=None):
        """Fused `_update_sdf_values` (`robot_batch_rollout_percept.py:384-440`): for the listed envs (all when None) and
        every query body, point = body position + body rotation * offset; writes `sdf_rows[e, b]` (a strided
        (N, n_query) view is fine) and optionally `gradients` / `nearest` (N, n_query, 3).  One launch, no host sync."""
        mesh = next(iter(self.meshes.values()))
        rb = rigid_body_state if rigid_body_state.is_contiguous() else rigid_body_state.contiguous()
        nq = int(body_indices_i32.numel())
        assert sdf_rows.dtype == torch.float32 and sdf_rows.stride(1) == 1 and sdf_rows.shape[1] == nq
        N = sdf_rows.shape[0]
        n = N if env_ids_i32 is None else int(env_ids_i32.numel())

        def ptr(t):
            return C.c_void_p(None) if t is None else C.c_void_p(t.data_ptr())
        mesh._check(mesh.lib.lg_sdf_bodies_update(mesh.handle, ptr(rb), int(num_bodies), ptr(body_indices_i32), ptr(sphere_offsets),
                                                  nq, ptr(env_ids_i32), n, float(self.cfg.max_distance), ptr(sdf_rows),
                                                  int(sdf_rows.stride(0)), ptr(gradients), ptr(nearest), mesh._stream()))

    def nearest_points(self, query_points: torch.Tensor) -> torch.Tensor:
        sdf, grad = self.query(query_points)
        return query_points - sdf.unsqueeze(-1) * grad

    def clear_cache(self):
        pass

    @property
    def data(self) -> MeshSDFData:
        return self._data

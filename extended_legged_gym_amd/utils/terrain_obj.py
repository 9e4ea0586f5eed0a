"""Terrain from a Wavefront OBJ mesh — the reference's `TerrainObj` (`legged_gym/utils/terrain_obj.py:19-283`) with the same
public attributes (`vertices`, `triangles`, `heightsamples`, `tot_rows`, `tot_cols`, `env_origins`, `env_length`,
`border_size`) and the same placement rule: the mesh is centred in XY and then shifted by `border_size =
max(size_x, size_y) / 2`, so that `gym.add_triangle_mesh(transform.p = -border_size)` puts its centre on the world origin.

Differences by design: the OBJ is parsed by `utils/obj_io.py` (no `trimesh` dependency) and the height queries
(`get_height`, `get_heights_batch`, reference `:141-283`, a CPU `RayMeshIntersector`) are first-hit ray casts on the
GPU BVH (`lg_raycast_mesh`).  `heightsamples` stays all-zero as in the reference (`:116`): the height scan of a
mesh terrain reads zeros, contacts run against the triangles (`collide_as_mesh`)."""
import os

import numpy as np
import torch

from extended_legged_gym_amd import LEGGED_GYM_ROOT_DIR as ROOT_DIR
from extended_legged_gym_amd.utils.obj_io import load_obj


class TerrainObj:
    collide_as_mesh = True     # NativeSetup: LG_MESH_TRIMESH contacts instead of the height grid

    def __init__(self, cfg, verbose=False):
        self.cfg = cfg
        self.type = cfg.mesh_type
        self.verbose = verbose
        if self.type != "trimesh":
            raise ValueError("Only trimesh terrains are supported")
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border

        path = cfg.terrain_file
        if not os.path.isfile(path):
            path = os.path.join(ROOT_DIR, cfg.terrain_file)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Terrain mesh file not found: {cfg.terrain_file}")
        v, t = load_obj(path)
        v = v.astype(np.float64)
        lo, hi = v.min(0), v.max(0)
        size_x, size_y = hi[0] - lo[0], hi[1] - lo[1]
        center = (lo + hi) / 2
        self.border_size = max(size_x, size_y) / 2
        self.cfg.border_size = self.border_size          # read back by the env when it places the mesh
        v[:, 0] += self.border_size - center[0]          # centre in XY (Z untouched), then shift the corner to the origin
        v[:, 1] += self.border_size - center[1]
        self.bounds = np.stack([v.min(0), v.max(0)])
        self.vertices = v.astype(np.float32)
        self.triangles = t.astype(np.uint32)

        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        for i in range(cfg.num_rows):
            for j in range(cfg.num_cols):
                self.env_origins[i, j] = [j * self.env_length - 0.5 * self.border_size,
                                          i * self.env_width - 0.5 * self.border_size, 10.0]
        self.heightsamples = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self._device_mesh = None

    # ------------------------------------------------------------------ height queries (first hit of a vertical ray)
    def _mesh(self, device=None):
        if self._device_mesh is None:
            from extended_legged_gym_amd.utils.mesh import DeviceMesh
            self._device_mesh = DeviceMesh(self.vertices, self.triangles.astype(np.int32), device or "cuda:0")
        return self._device_mesh

    def get_heights_batch(self, positions, max_height=10.0, cast_dir=-1, device=None):
        """Heights (z of the first hit) at world (x, y) positions; 0 where the vertical ray misses (`:194-283`).
        cast_dir -1: from above, downwards (highest surface); +1: from below, upwards (lowest surface)."""
        from extended_legged_gym_amd.utils.ray_caster import raycast_mesh
        if isinstance(positions, torch.Tensor):
            positions = positions.detach().cpu().numpy()
        positions = np.asarray(positions, dtype=np.float64)
        if positions.ndim == 1:
            positions = positions.reshape(1, -1)
        if positions.shape[1] < 2:
            raise ValueError("positions must have at least 2 dimensions (x,y)")
        mesh = self._mesh(device)
        n = positions.shape[0]
        xy = positions[:, :2] + self.border_size
        z0 = self.bounds[1][2] + max_height if cast_dir < 0 else self.bounds[0][2] - max_height
        o = torch.tensor(np.column_stack([xy, np.full(n, z0)]), dtype=torch.float32, device=mesh.device)
        d = torch.tensor([0.0, 0.0, float(cast_dir)], device=mesh.device).expand(n, 3).contiguous()
        span = float(self.bounds[1][2] - self.bounds[0][2]) + 2.0 * max_height + 1.0
        hits, found = raycast_mesh(o, d, max_dist=span, mesh=mesh)
        h = torch.where(found, hits[:, 2], torch.zeros_like(hits[:, 2]))
        return h.cpu().numpy().astype(np.float64)

    def get_height(self, x, y, cast_dir=-1):
        """Single query (`:141-192`): 0 outside the XY bounds of the mesh or on a miss."""
        mx, my = x + self.border_size, y + self.border_size
        b = self.bounds
        if mx < b[0][0] or mx > b[1][0] or my < b[0][1] or my > b[1][1]:
            return 0.0
        return float(self.get_heights_batch(np.array([[x, y]]), max_height=10.0, cast_dir=cast_dir)[0])

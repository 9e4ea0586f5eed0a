"""Procedural heightfield terrain: the int16 grid that is both the height-scan source and, in the native step, the
collision surface.  Behavioural contract = the reference's `utils/terrain.py:39-198` (grid size and borders `:54-61`,
curriculum / random / selected layouts `:82-114`, difficulty → generator parameters `:116-154`, tile placement and
`env_origins` `:156-173`, gap / pit generators `:176-198`).  With the same `np.random` seed and the same sub-terrain
generators it produces the same grid, bit for bit (tests/test_terrain.py)."""
import numpy as np

from . import terrain_utils


def gap_terrain(terrain, gap_size, platform_size=1.):
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - plat) // 2
    y1 = (terrain.width - plat) // 2
    x2, y2 = x1 + gap, y1 + gap
    terrain.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    terrain.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.):
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    cx, cy = terrain.length // 2, terrain.width // 2
    terrain.height_field_raw[cx - half:cx + half, cy - half:cy + half] = -d


class Terrain:
    def __init__(self, cfg, num_robots, device=None) -> None:
        """`device` (e.g. "cuda:0", passed by the env): build on the GPU -- `cfg.device_generation` = "mesh" (default): tiles by the host
        generators (numpy's stream, the reference's grids bit for bit), the heightfield -> triangle-mesh conversion as a kernel
        (bit-exact); "all": every `make_terrain` tile type as a kernel too (slopes, stairs, gaps and pits integer-identical; rough slopes, discrete
        obstacles and stepping stones with Philox draws instead of numpy's: same distributions, other samples); "off" or `device=None` (tools, CPU tests): host numpy throughout."""
        self.cfg = cfg
        self._device = device
        self._mode = getattr(cfg, "device_generation", "mesh") if device is not None else "off"
        if self._mode not in ("off", "mesh", "all"):
            raise ValueError(f"terrain.device_generation must be 'off', 'mesh' or 'all', got {self._mode!r}")
        self._mesh_dev = None
        self._mesh_host = None
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ["none", "plane"]:
            return
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))

        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)

        scale = getattr(cfg, "difficulty_scale", 1.0)
        if self._mode == "all" and not cfg.selected:
            self._generate_on_device(scale)
        elif cfg.curriculum:
            self.curiculum(scale)
        elif cfg.selected:
            self.selected_terrain()
        else:
            self.randomized_terrain(scale)

        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":
            if self._mode == "off":
                self._mesh_host = terrain_utils.convert_heightfield_to_trimesh(
                    self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale, cfg.slope_treshold)
            else:
                from . import terrain_device
                self._mesh_dev = terrain_device.heightfield_to_trimesh(self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale,
                                                                       cfg.slope_treshold, device)

    # the triangle mesh: device tensors when built on the GPU (`mesh_device`), host arrays on demand (the BVH builder and the
    # reference's attribute names `vertices` / `triangles` want numpy)
    @property
    def mesh_device(self):
        return self._mesh_dev

    def _host_mesh(self):
        if self._mesh_host is None:
            if self._mesh_dev is None:
                raise AttributeError("this terrain has no triangle mesh (mesh_type is not 'trimesh')")
            v, t = self._mesh_dev
            self._mesh_host = (v.cpu().numpy(), t.cpu().numpy().view(np.uint32))
        return self._mesh_host

    @property
    def vertices(self):
        return self._host_mesh()[0]

    @vertices.setter
    def vertices(self, v):
        self._mesh_host = (v, self._mesh_host[1] if self._mesh_host is not None else None)
        self._mesh_dev = None

    @property
    def triangles(self):
        return self._host_mesh()[1]

    @triangles.setter
    def triangles(self, t):
        self._mesh_host = (self._mesh_host[0] if self._mesh_host is not None else None, t)
        self._mesh_dev = None

    def _generate_on_device(self, difficulty_scale=1.0):
        """The curriculum / random layouts (`terrain.py:82-101`) with the tiles written by `lg_terrain_generate`."""
        from . import terrain_device
        cfg = self.cfg
        tiles = [[None] * cfg.num_cols for _ in range(cfg.num_rows)]
        seed0 = int(np.random.randint(0, 2 ** 31 - 1))        # one draw from the seeded host stream keys every tile
        for j in range(cfg.num_cols):
            for i in range(cfg.num_rows):
                if cfg.curriculum:
                    difficulty, choice = i / cfg.num_rows * difficulty_scale, j / cfg.num_cols + 0.001
                else:
                    choice = np.random.uniform(0, 1)
                    difficulty = np.random.choice([0.5, 0.75, 0.9]) * difficulty_scale
                tiles[i][j] = terrain_device.tile_spec_for(choice, difficulty, self.proportions, self.width_per_env_pixels,
                                                           cfg.horizontal_scale, cfg.vertical_scale, seed0 + 7919 * (i * cfg.num_cols + j))
        H, org = terrain_device.generate(cfg, tiles, self._device)
        self.height_field_raw = H.cpu().numpy()
        self.env_origins = org.cpu().numpy().astype(np.float64)

    # ---- layouts
    def randomized_terrain(self, difficulty_scale=1.0):
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9]) * difficulty_scale
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self, difficulty_scale=1.0):
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                difficulty = i / self.cfg.num_rows * difficulty_scale
                choice = j / self.cfg.num_cols + 0.001
                self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def selected_terrain(self):
        kwargs = dict(self.cfg.terrain_kwargs)
        gen = getattr(terrain_utils, kwargs.pop("type").split(".")[-1])
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            tile = self._blank_tile()
            gen(tile, **kwargs)
            self.add_terrain_to_map(tile, i, j)

    def _blank_tile(self):
        return terrain_utils.SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                                        vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)

    # ---- one tile
    def make_terrain(self, choice, difficulty):
        tile = self._blank_tile()
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        obstacle_height = 0.05 + difficulty * 0.2
        stone_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        p = self.proportions
        if choice < p[0]:
            if choice < p[0] / 2:
                slope *= -1
            terrain_utils.pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
        elif choice < p[1]:
            terrain_utils.pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
            terrain_utils.random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            if choice < p[2]:
                step_height *= -1
            terrain_utils.pyramid_stairs_terrain(tile, step_width=0.31, step_height=step_height, platform_size=3.)
        elif choice < p[4]:
            terrain_utils.discrete_obstacles_terrain(tile, obstacle_height, 1., 2., 20, platform_size=3.)
        elif choice < p[5]:
            terrain_utils.stepping_stones_terrain(tile, stone_size=stone_size, stone_distance=stone_distance,
                                                  max_height=0., platform_size=4.)
        elif choice < p[6]:
            gap_terrain(tile, gap_size=1. * difficulty, platform_size=3.)
        else:
            pit_terrain(tile, depth=1. * difficulty, platform_size=4.)
        return tile

    def add_terrain_to_map(self, terrain, row, col):
        x0 = self.border + row * self.length_per_env_pixels
        y0 = self.border + col * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = terrain.height_field_raw

        hs = terrain.horizontal_scale
        x1, x2 = int((self.env_length / 2. - 1) / hs), int((self.env_length / 2. + 1) / hs)
        y1, y2 = int((self.env_width / 2. - 1) / hs), int((self.env_width / 2. + 1) / hs)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]

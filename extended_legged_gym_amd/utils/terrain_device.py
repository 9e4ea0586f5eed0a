"""Terrain construction on the GPU (include/lgstep.h: lg_terrain_generate, lg_heightfield_to_trimesh): the procedural `Terrain`'s tile
generators and its heightfield -> triangle-mesh conversion as kernels instead of host numpy (reference `utils/terrain.py:39-173`,
`isaacgym.terrain_utils`).  The host functions in `terrain_utils.py` stay as the CPU checker."""
import ctypes as C

import numpy as np
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.native import load_library


def _lib():
    return load_library()


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None, device="cuda:0"):
    """`convert_heightfield_to_trimesh` on the device, bit for bit: returns (vertices (n, 3) float32, triangles (m, 3) uint32 stored as
    int32) as torch tensors on `device`.  `height_field_raw`: int16 numpy array or device tensor."""
    dev = torch.device(device)
    H = height_field_raw if torch.is_tensor(height_field_raw) else torch.from_numpy(np.ascontiguousarray(height_field_raw, dtype=np.int16))
    H = H.to(dev).contiguous()
    rows, cols = int(H.shape[0]), int(H.shape[1])
    # np.linspace(0, (n - 1) * hs, n): step = stop / (n - 1) in float64, the last sample is `stop` itself
    hs, vs = float(horizontal_scale), float(vertical_scale)
    stop_x, stop_y = (rows - 1) * hs, (cols - 1) * hs
    step_x, step_y = stop_x / (rows - 1), stop_y / (cols - 1)
    thr = -1.0 if slope_threshold is None else float(slope_threshold) * hs / vs
    verts = torch.empty(rows * cols, 3, device=dev, dtype=torch.float32)
    tris = torch.empty(2 * (rows - 1) * (cols - 1), 3, device=dev, dtype=torch.int32)
    rc = _lib().lg_heightfield_to_trimesh(C.c_void_p(H.data_ptr()), rows, cols, step_x, step_y, stop_x, stop_y, hs, vs, thr,
                                          C.c_void_p(verts.data_ptr()), C.c_void_p(tris.data_ptr()), _stream(dev))
    if rc != abi.LG_OK:
        raise RuntimeError(f"lg_heightfield_to_trimesh failed ({rc})")
    return verts, tris


def tile_spec_for(choice, difficulty, proportions, tile_px, horizontal_scale, vertical_scale, seed):
    """`Terrain.make_terrain` (`terrain.py:116-154`) as a kernel argument: the integers the host generators derive from (choice,
    difficulty), computed here with the host generators' own expressions."""
    t = abi.lg_tile_spec()
    t.seed = int(seed) & 0xFFFFFFFF
    hs, vs = horizontal_scale, vertical_scale
    slope = difficulty * 0.4
    step_height = 0.05 + 0.18 * difficulty
    obstacle_height = 0.05 + difficulty * 0.2
    p = proportions
    W = tile_px

    def pyramid(slope_, platform=3.):
        t.kind = abi.LG_TILE_PYRAMID_SLOPE
        t.max_height = int(slope_ * (hs / vs) * (W / 2))
        cx = int(W / 2)
        half = int(platform / hs / 2)
        x1 = W // 2 - half
        f = (cx - abs(cx - x1)) / cx
        edge = int(np.array(t.max_height * f * f).astype(np.int16))
        t.clip_lo, t.clip_hi = min(edge, 0), max(edge, 0)

    if choice < p[0]:
        pyramid(-slope if choice < p[0] / 2 else slope)
    elif choice < p[1]:
        pyramid(slope)
        lo, hi, st = int(-0.05 / vs), int(0.05 / vs), int(0.005 / vs)
        levels = np.arange(lo, hi + st, st)
        t.noise_lo, t.noise_step, t.noise_levels = int(levels[0]), int(st), int(len(levels))
        t.noise_coarse = max(1, int(round(0.2 / hs)))
    elif choice < p[3]:
        if choice < p[2]:
            step_height *= -1
        t.kind = abi.LG_TILE_PYRAMID_STAIRS
        t.step_width, t.step_height = int(0.31 / hs), int(step_height / vs)
        platform = int(3. / hs)
        x0, x1, n = 0, W, 0
        while (x1 - x0) > platform:
            x0 += t.step_width; x1 -= t.step_width; n += 1
        t.num_steps = n
    elif len(p) <= 4 or choice < p[4]:
        t.kind = abi.LG_TILE_DISCRETE_OBSTACLES
        t.max_height = int(obstacle_height / vs)
        t.rect_min, t.rect_max, t.rect_count, t.platform = int(1. / hs), int(2. / hs), 20, int(3. / hs)
    elif choice < p[5]:          # terrain.py:145-147
        stone_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        t.kind = abi.LG_TILE_STEPPING_STONES
        t.rect_min, t.rect_max = int(stone_size / hs), int(stone_distance / hs)
        t.max_height, t.clip_lo, t.platform = int(0. / vs), int(-10 / vs), int(4. / hs)
    elif choice < p[6]:          # gap_terrain(tile, gap_size=1. * difficulty, platform_size=3.)
        t.kind = abi.LG_TILE_GAP
        t.rect_min, t.platform = int(1. * difficulty / hs), int(3. / hs)
    else:                        # pit_terrain(tile, depth=1. * difficulty, platform_size=4.)
        t.kind = abi.LG_TILE_PIT
        t.max_height, t.platform = int(1. * difficulty / vs), int(4. / hs / 2)
    return t


def generate(cfg, tiles, device="cuda:0"):
    """tiles: (num_rows, num_cols) nested list of lg_tile_spec.  Returns (height grid (tot_rows, tot_cols) int16 tensor, origins
    (num_rows, num_cols, 3) float32 tensor) on `device`."""
    dev = torch.device(device)
    nr, nc = cfg.num_rows, cfg.num_cols
    L, W = int(cfg.terrain_length / cfg.horizontal_scale), int(cfg.terrain_width / cfg.horizontal_scale)
    border = int(cfg.border_size / cfg.horizontal_scale)
    H = torch.empty(nr * L + 2 * border, nc * W + 2 * border, dtype=torch.int16, device=dev)
    org = torch.empty(nr, nc, 3, dtype=torch.float32, device=dev)
    arr = (abi.lg_tile_spec * (nr * nc))(*[tiles[i][j] for i in range(nr) for j in range(nc)])
    rc = _lib().lg_terrain_generate(arr, nr, nc, L, W, border, float(cfg.horizontal_scale), float(cfg.vertical_scale), float(cfg.terrain_length),
                                    float(cfg.terrain_width), C.c_void_p(H.data_ptr()), C.c_void_p(org.data_ptr()), _stream(dev))
    if rc != abi.LG_OK:
        raise RuntimeError(f"lg_terrain_generate failed ({rc})")
    return H, org

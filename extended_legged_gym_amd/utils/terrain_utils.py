"""Procedural sub-terrain primitives (restatement of `isaacgym.terrain_utils`).

The reference builds its int16 heightfield by calling these generators
(`legged_gym/utils/terrain.py:107-148`), but the module itself ships inside
the closed `isaacgym` package and is not in the reference tree.  They are
restated here from the published behaviour (parameter meaning, discretisation
to int16 grid units, platform carving).  Tile *contents* that depend on
`np.random` are statistically, not bitwise, equivalent to Isaac Gym's: parity
is therefore unpinned for this file; the layout arithmetic
that consumes the tiles is pinned by `tests/test_terrain.py` against the
reference `Terrain` class running on top of these same generators.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _bilinear_resample(coarse, n_rows, n_cols):
    """Linear interpolation of a coarse grid onto (n_rows, n_cols), both spanning the same extent."""
    r_src = np.linspace(0.0, 1.0, coarse.shape[0])
    c_src = np.linspace(0.0, 1.0, coarse.shape[1])
    r_dst = np.linspace(0.0, 1.0, n_rows)
    c_dst = np.linspace(0.0, 1.0, n_cols)
    tmp = np.empty((coarse.shape[0], n_cols))
    for i in range(coarse.shape[0]):
        tmp[i] = np.interp(c_dst, c_src, coarse[i])
    out = np.empty((n_rows, n_cols))
    for j in range(n_cols):
        out[:, j] = np.interp(r_dst, r_src, tmp[:, j])
    return out


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Add uniform noise sampled on a coarse grid and linearly upsampled."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo = int(min_height / terrain.vertical_scale)
    hi = int(max_height / terrain.vertical_scale)
    st = int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    n_r = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    n_c = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(levels, (n_r, n_c))
    fine = np.rint(_bilinear_resample(coarse.astype(np.float64), terrain.width, terrain.length))
    terrain.height_field_raw += fine.astype(np.int16)
    return terrain


def sloped_terrain(terrain, slope=1):
    ramp = np.arange(terrain.width).reshape(terrain.width, 1)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    terrain.height_field_raw[:, np.arange(terrain.length)] += \
        (max_height * ramp / terrain.width).astype(terrain.height_field_raw.dtype)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.):
    cx = int(terrain.width / 2)
    cy = int(terrain.length / 2)
    fx = ((cx - np.abs(cx - np.arange(terrain.width))) / cx).reshape(terrain.width, 1)
    fy = ((cy - np.abs(cy - np.arange(terrain.length))) / cy).reshape(1, terrain.length)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * fx * fy).astype(terrain.height_field_raw.dtype)

    half = int(platform_size / terrain.horizontal_scale / 2)
    x1 = terrain.width // 2 - half
    y1 = terrain.length // 2 - half
    edge = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.):
    max_height = int(max_height / terrain.vertical_scale)
    min_size = int(min_size / terrain.horizontal_scale)
    max_size = int(max_size / terrain.horizontal_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    n_i, n_j = terrain.height_field_raw.shape
    heights = [-max_height, -max_height // 2, max_height // 2, max_height]
    sizes = range(min_size, max_size, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes)
        l = np.random.choice(sizes)
        i0 = np.random.choice(range(0, n_i - w, 4))
        j0 = np.random.choice(range(0, n_j - l, 4))
        terrain.height_field_raw[i0:i0 + w, j0:j0 + l] = np.random.choice(heights)
    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def wave_terrain(terrain, num_waves=1, amplitude=1.):
    amplitude = int(0.5 * amplitude / terrain.vertical_scale)
    if num_waves > 0:
        div = terrain.length / (num_waves * np.pi * 2)
        xx = np.arange(terrain.width).reshape(terrain.width, 1)
        yy = np.arange(terrain.length).reshape(1, terrain.length)
        terrain.height_field_raw += (amplitude * np.cos(yy / div) + amplitude * np.sin(xx / div)).astype(
            terrain.height_field_raw.dtype)
    return terrain


def stairs_terrain(terrain, step_width, step_height):
    step_width = int(step_width / terrain.horizontal_scale)
    step_height = int(step_height / terrain.vertical_scale)
    n_steps = terrain.width // step_width
    h = step_height
    for i in range(n_steps):
        terrain.height_field_raw[i * step_width:(i + 1) * step_width, :] += h
        h += step_height
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.):
    step_width = int(step_width / terrain.horizontal_scale)
    step_height = int(step_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    h = 0
    x0, x1 = 0, terrain.width
    y0, y1 = 0, terrain.length
    while (x1 - x0) > platform_size and (y1 - y0) > platform_size:
        x0 += step_width
        x1 -= step_width
        y0 += step_width
        y1 -= step_width
        h += step_height
        terrain.height_field_raw[x0:x1, y0:y1] = h
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1., depth=-10):
    stone_size = int(stone_size / terrain.horizontal_scale)
    stone_distance = int(stone_distance / terrain.horizontal_scale)
    max_height = int(max_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    heights = np.arange(-max_height - 1, max_height, step=1)

    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        y = 0
        while y < terrain.length:
            y_stop = min(terrain.length, y + stone_size)
            x = np.random.randint(0, stone_size)
            x_stop = max(0, x - stone_distance)
            terrain.height_field_raw[0:x_stop, y:y_stop] = np.random.choice(heights)
            while x < terrain.width:
                x_stop = min(terrain.width, x + stone_size)
                terrain.height_field_raw[x:x_stop, y:y_stop] = np.random.choice(heights)
                x += stone_size + stone_distance
            y += stone_size + stone_distance
    else:
        x = 0
        while x < terrain.width:
            x_stop = min(terrain.width, x + stone_size)
            y = np.random.randint(0, stone_size)
            y_stop = max(0, y - stone_distance)
            terrain.height_field_raw[x:x_stop, 0:y_stop] = np.random.choice(heights)
            while y < terrain.length:
                y_stop = min(terrain.length, y + stone_size)
                terrain.height_field_raw[x:x_stop, y:y_stop] = np.random.choice(heights)
                y += stone_size + stone_distance
            x += stone_size + stone_distance

    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Regular-grid triangulation of a heightfield, two triangles per cell.

    Cell (i, j) with corners v0=(i,j), v1=(i,j+1), v2=(i+1,j), v3=(i+1,j+1) gives
    triangles (v0, v3, v1) and (v0, v2, v3), i.e. the diagonal runs v0→v3.  With a
    slope threshold, vertices next to a step steeper than the threshold are
    shifted by one cell so the step becomes a vertical wall.
    """
    hf = height_field_raw
    n_r, n_c = hf.shape
    ys = np.linspace(0, (n_c - 1) * horizontal_scale, n_c)
    xs = np.linspace(0, (n_r - 1) * horizontal_scale, n_r)
    yy, xx = np.meshgrid(ys, xs)

    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        mx = np.zeros((n_r, n_c))
        my = np.zeros((n_r, n_c))
        mc = np.zeros((n_r, n_c))
        mx[:n_r - 1, :] += (hf[1:n_r, :] - hf[:n_r - 1, :] > thr)
        mx[1:n_r, :] -= (hf[:n_r - 1, :] - hf[1:n_r, :] > thr)
        my[:, :n_c - 1] += (hf[:, 1:n_c] - hf[:, :n_c - 1] > thr)
        my[:, 1:n_c] -= (hf[:, :n_c - 1] - hf[:, 1:n_c] > thr)
        mc[:n_r - 1, :n_c - 1] += (hf[1:n_r, 1:n_c] - hf[:n_r - 1, :n_c - 1] > thr)
        mc[1:n_r, 1:n_c] -= (hf[:n_r - 1, :n_c - 1] - hf[1:n_r, 1:n_c] > thr)
        xx += (mx + mc * (mx == 0)) * horizontal_scale
        yy += (my + mc * (my == 0)) * horizontal_scale

    vertices = np.zeros((n_r * n_c, 3), dtype=np.float32)
    vertices[:, 0] = xx.flatten()
    vertices[:, 1] = yy.flatten()
    vertices[:, 2] = hf.flatten() * vertical_scale

    idx = np.arange(n_r * n_c, dtype=np.uint32).reshape(n_r, n_c)
    v0 = idx[:-1, :-1].reshape(-1)
    v1 = idx[:-1, 1:].reshape(-1)
    v2 = idx[1:, :-1].reshape(-1)
    v3 = idx[1:, 1:].reshape(-1)
    triangles = np.empty((2 * v0.size, 3), dtype=np.uint32)
    triangles[0::2] = np.stack([v0, v3, v1], axis=1)
    triangles[1::2] = np.stack([v0, v2, v3], axis=1)
    return vertices, triangles

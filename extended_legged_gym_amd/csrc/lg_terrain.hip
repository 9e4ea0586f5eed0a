// lg_terrain.hip — terrain construction on the device (include/lgstep.h: lg_terrain_generate, lg_heightfield_to_trimesh).
//
// Replaces the host numpy of the reference's Terrain class (utils/terrain.py:39-173) and isaacgym.terrain_utils' generators /
// convert_heightfield_to_trimesh (closed third party, restated in extended_legged_gym_amd/utils/terrain_utils.py, which is also the
// CPU checker of these kernels): one lane per height sample / vertex / cell.  The rough-terrain task's grid is 900 x 900 samples,
// its mesh 810 000 vertices and 1 616 402 triangles: 1.6 MB + 9.7 MB + 19.4 MB written once, HBM-bound, microseconds.
#include "../../include/lgstep.h"
#include "lg_device.h"
#include <hip/hip_runtime.h>

namespace {

struct GenParams { int num_rows, num_cols, L, W, border, tot_rows, tot_cols; };

LG_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// height (in vertical units) of sample (i, j) of a tile, exactly the integers of the host generators for the deterministic kinds
LG_DEV int tile_height(const lg_tile_spec& t, int i, int j, int L, int W) {
  int h = 0;
  if (t.kind == LG_TILE_PYRAMID_SLOPE) {
    // terrain_utils.pyramid_sloped_terrain: (max_height * fx * fy).astype(int16), then the platform clip
    const int cx = W / 2, cy = L / 2;      // (the generator's "width" is axis 0, "length" axis 1; tiles are square)
    const double fx = (double)(cx - abs(cx - i)) / (double)cx, fy = (double)(cy - abs(cy - j)) / (double)cy;
    const double v = (double)t.max_height * fx * fy;
    h = (int)v;                            // astype(int16) truncates toward zero
    h = min(max(h, t.clip_lo), t.clip_hi);
  } else if (t.kind == LG_TILE_PYRAMID_STAIRS) {
    // terrain_utils.pyramid_stairs_terrain: nested squares [k sw, n - k sw) set to k * step_height, k = 1 .. num_steps
    const int sw = t.step_width;
    int k = min(min(i / sw, (W - 1 - i) / sw), min(j / sw, (L - 1 - j) / sw));
    k = min(k, t.num_steps);
    h = k * t.step_height;
  } else if (t.kind == LG_TILE_DISCRETE_OBSTACLES) {
    // terrain_utils.discrete_obstacles_terrain: rect_count rectangles, later ones overwrite earlier ones; sizes and corners on a
    // 4-pixel lattice, heights from {-H, -H // 2, H // 2, H}; a flat platform in the middle
    const int H = t.max_height;
    const int hs4[4] = {-H, (-H - ((-H) % 2 != 0 ? 1 : 0)) / 2 /* floor division like Python's // */, H / 2, H};
    const int nsz = max(1, (t.rect_max - t.rect_min + 3) / 4);
    for (int r = 0; r < t.rect_count; ++r) {
      uint32_t o[4], o2[4];
      philox4((uint32_t)r, 0u, 0u, 11u, t.seed, 0x7e55u, o);
      philox4((uint32_t)r, 1u, 0u, 11u, t.seed, 0x7e55u, o2);
      const int w = t.rect_min + 4 * (int)(o[0] % (uint32_t)nsz), l = t.rect_min + 4 * (int)(o[1] % (uint32_t)nsz);
      const int ni = max(1, (W - w + 3) / 4), nj = max(1, (L - l + 3) / 4);
      const int i0 = 4 * (int)(o[2] % (uint32_t)ni), j0 = 4 * (int)(o[3] % (uint32_t)nj);
      if (i >= i0 && i < i0 + w && j >= j0 && j < j0 + l) h = hs4[o2[0] & 3u];
    }
    const int x1 = (W - t.platform) / 2, x2 = (W + t.platform) / 2, y1 = (L - t.platform) / 2, y2 = (L + t.platform) / 2;
    if (i >= x1 && i < x2 && j >= y1 && j < y2) h = 0;
  } else if (t.kind == LG_TILE_STEPPING_STONES) {
    // terrain_utils.stepping_stones_terrain (square tiles: the `length >= width` branch): strips of stones along axis 1, every strip with its
    // own random offset along axis 0 and a leading partial stone [0, offset - distance); a height per stone from arange(-max_height - 1,
    // max_height); everything else at `depth`; a flat platform in the middle.  rect_min = stone size, rect_max = stone distance (pixels),
    // clip_lo = depth (vertical units).
    const int sz = max(1, t.rect_min), pitch = sz + max(0, t.rect_max), nh = max(1, 2 * t.max_height + 1);
    h = t.clip_lo;
    const int strip = j / pitch;
    if (j - strip * pitch < sz) {
      uint32_t o[4];
      philox4((uint32_t)strip, 0u, 0u, 17u, t.seed, 0x7e55u, o);
      const int x0 = (int)(o[0] % (uint32_t)sz);
      int stone = -2;                                         // -1: the leading partial stone, m >= 0: the m-th full stone of the strip
      if (i < max(0, x0 - t.rect_max)) stone = -1;
      else if (i >= x0 && (i - x0) % pitch < sz) stone = (i - x0) / pitch;
      if (stone > -2) {
        philox4((uint32_t)strip, (uint32_t)(stone + 1), 1u, 17u, t.seed, 0x7e55u, o);
        h = -t.max_height - 1 + (int)(o[0] % (uint32_t)nh);
      }
    }
    const int x1 = (W - t.platform) / 2, x2 = (W + t.platform) / 2, y1 = (L - t.platform) / 2, y2 = (L + t.platform) / 2;
    if (i >= x1 && i < x2 && j >= y1 && j < y2) h = 0;
  } else if (t.kind == LG_TILE_GAP) {
    // terrain.py:gap_terrain: a square moat of width rect_min pixels at -1000 units around the central platform
    const int cx = L / 2, cy = W / 2, x1 = (L - t.platform) / 2, y1 = (W - t.platform) / 2, x2 = x1 + t.rect_min, y2 = y1 + t.rect_min;
    if (i >= cx - x2 && i < cx + x2 && j >= cy - y2 && j < cy + y2) h = -1000;
    if (i >= cx - x1 && i < cx + x1 && j >= cy - y1 && j < cy + y1) h = 0;
  } else if (t.kind == LG_TILE_PIT) {
    // terrain.py:pit_terrain: the central square of half-width `platform` pixels lowered by max_height units
    const int cx = L / 2, cy = W / 2;
    if (i >= cx - t.platform && i < cx + t.platform && j >= cy - t.platform && j < cy + t.platform) h = -t.max_height;
  }
  if (t.noise_levels > 0) {
    // terrain_utils.random_uniform_terrain: levels drawn on a coarse grid, bilinear up-sampling, rint
    const int c = max(1, t.noise_coarse);
    const float fi = ((float)i + 0.5f) / (float)c - 0.5f, fj = ((float)j + 0.5f) / (float)c - 0.5f;
    const int nci = max(1, W / c), ncj = max(1, L / c);
    const int i0 = max(0, min((int)floorf(fi), nci - 1)), j0 = max(0, min((int)floorf(fj), ncj - 1));
    const int i1 = min(i0 + 1, nci - 1), j1 = min(j0 + 1, ncj - 1);
    const float a = fminf(fmaxf(fi - (float)i0, 0.f), 1.f), b = fminf(fmaxf(fj - (float)j0, 0.f), 1.f);
    auto lvl = [&](int ci, int cj) {
      uint32_t o[4];
      philox4((uint32_t)ci, (uint32_t)cj, 0u, 13u, t.seed, 0x7e55u, o);
      return (float)(t.noise_lo + t.noise_step * (int)(o[0] % (uint32_t)t.noise_levels));
    };
    const float v = (1.f - a) * ((1.f - b) * lvl(i0, j0) + b * lvl(i0, j1)) + a * ((1.f - b) * lvl(i1, j0) + b * lvl(i1, j1));
    h += (int)rintf(v);
  }
  return h;
}

__global__ __launch_bounds__(256) void terrain_generate_kernel(const lg_tile_spec* __restrict__ tiles, GenParams g, int16_t* __restrict__ H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)g.tot_rows * g.tot_cols) return;
  const int r = (int)(idx / g.tot_cols), c = (int)(idx % g.tot_cols);
  const int ti = (r - g.border) / g.L, tj = (c - g.border) / g.W;
  int h = 0;
  if (r >= g.border && c >= g.border && ti < g.num_rows && tj < g.num_cols)
    h = tile_height(tiles[ti * g.num_cols + tj], r - g.border - ti * g.L, c - g.border - tj * g.W, g.L, g.W);
  H[idx] = (int16_t)h;
}

// origins (terrain.py:166-173): z = max of the tile's central window [x1, x2) x [y1, x2) * vertical_scale; one wave per tile
__global__ __launch_bounds__(64) void terrain_origins_kernel(const int16_t* __restrict__ H, GenParams g, int x1, int x2, int y1, int y2,
                                                             float vs, float env_length, float env_width, float* __restrict__ origins) {
  const int t = blockIdx.x, ti = t / g.num_cols, tj = t % g.num_cols, lane = threadIdx.x;
  const int r0 = g.border + ti * g.L, c0 = g.border + tj * g.W;
  int mx = -32768;
  const int nx = x2 - x1, ny = y2 - y1;
  for (int k = lane; k < nx * ny; k += 64) mx = max(mx, (int)H[(size_t)(r0 + x1 + k / ny) * g.tot_cols + c0 + y1 + k % ny]);
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
  if (lane == 0) {
    origins[3 * t] = (float)(((double)ti + 0.5) * (double)env_length);
    origins[3 * t + 1] = (float)(((double)tj + 0.5) * (double)env_width);
    origins[3 * t + 2] = (float)((double)mx * (double)vs);
  }
}

struct MeshParams { int rows, cols; double step_x, step_y, stop_x, stop_y, hs, vs, thr; };

// convert_heightfield_to_trimesh: float64 arithmetic as numpy's, one float32 rounding at the store
__global__ __launch_bounds__(256) void trimesh_vertices_kernel(const int16_t* __restrict__ H, MeshParams m, float* __restrict__ V) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)m.rows * m.cols) return;
  const int i = (int)(idx / m.cols), j = (int)(idx % m.cols);
  auto hf = [&](int a, int b) { return (int)H[(size_t)a * m.cols + b]; };
  double x = i == m.rows - 1 ? m.stop_x : (double)i * m.step_x;        // np.linspace(0, stop, n): start + i * step, the last sample = stop
  double y = j == m.cols - 1 ? m.stop_y : (double)j * m.step_y;
  if (m.thr >= 0.0) {
    const int h = hf(i, j);
    int mx = 0, my = 0, mc = 0;
    if (i + 1 < m.rows && (double)(hf(i + 1, j) - h) > m.thr) mx += 1;
    if (i > 0 && (double)(hf(i - 1, j) - h) > m.thr) mx -= 1;
    if (j + 1 < m.cols && (double)(hf(i, j + 1) - h) > m.thr) my += 1;
    if (j > 0 && (double)(hf(i, j - 1) - h) > m.thr) my -= 1;
    if (i + 1 < m.rows && j + 1 < m.cols && (double)(hf(i + 1, j + 1) - h) > m.thr) mc += 1;
    if (i > 0 && j > 0 && (double)(hf(i - 1, j - 1) - h) > m.thr) mc -= 1;
    x += (double)(mx + (mx == 0 ? mc : 0)) * m.hs;
    y += (double)(my + (my == 0 ? mc : 0)) * m.hs;
  }
  V[3 * idx] = (float)x; V[3 * idx + 1] = (float)y; V[3 * idx + 2] = (float)((double)H[idx] * m.vs);
}

__global__ __launch_bounds__(256) void trimesh_triangles_kernel(int rows, int cols, uint32_t* __restrict__ T) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)(rows - 1) * (cols - 1)) return;
  const int i = (int)(idx / (cols - 1)), j = (int)(idx % (cols - 1));
  const uint32_t v0 = (uint32_t)(i * cols + j), v1 = v0 + 1, v2 = v0 + (uint32_t)cols, v3 = v2 + 1;
  uint32_t* t = T + 6 * idx;
  t[0] = v0; t[1] = v3; t[2] = v1; t[3] = v0; t[4] = v2; t[5] = v3;
}

int device_of_ptr(const void* p) {
  hipPointerAttribute_t pa;
  return hipPointerGetAttributes(&pa, p) == hipSuccess ? pa.device : -1;
}
struct Scope {
  int prev = -1;
  explicit Scope(int d) { (void)hipGetDevice(&prev); if (d != prev) (void)hipSetDevice(d); }
  ~Scope() { int cur = -1; (void)hipGetDevice(&cur); if (cur != prev && prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

int lg_terrain_generate(const lg_tile_spec* tiles_host, int32_t num_rows, int32_t num_cols, int32_t tile_len_px, int32_t tile_wid_px,
                        int32_t border_px, float hs, float vs, float env_length, float env_width, int16_t* heights, float* origins,
                        void* stream) {
  if (!tiles_host || !heights || !origins || num_rows <= 0 || num_cols <= 0 || tile_len_px <= 0 || tile_wid_px <= 0 || border_px < 0) return LG_ERR_INVALID;
  const int dev = device_of_ptr(heights);
  if (dev < 0) return LG_ERR_INVALID;
  Scope sc(dev);
  hipStream_t st = (hipStream_t)stream;
  GenParams g{num_rows, num_cols, tile_len_px, tile_wid_px, border_px, num_rows * tile_len_px + 2 * border_px, num_cols * tile_wid_px + 2 * border_px};
  lg_tile_spec* d_tiles = nullptr;
  const size_t nb = sizeof(lg_tile_spec) * (size_t)num_rows * num_cols;
  if (hipMallocAsync((void**)&d_tiles, nb, st) != hipSuccess) return LG_ERR_HIP;
  if (hipMemcpyAsync(d_tiles, tiles_host, nb, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipFreeAsync(d_tiles, st); return LG_ERR_HIP; }
  const int64_t total = (int64_t)g.tot_rows * g.tot_cols;
  hipLaunchKernelGGL(terrain_generate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_tiles, g, heights);
  // the central window of add_terrain_to_map (terrain.py:166-170)
  const int x1 = (int)((env_length / 2.f - 1.f) / hs), x2 = (int)((env_length / 2.f + 1.f) / hs);
  const int y1 = (int)((env_width / 2.f - 1.f) / hs), y2 = (int)((env_width / 2.f + 1.f) / hs);
  hipLaunchKernelGGL(terrain_origins_kernel, dim3((unsigned)(num_rows * num_cols)), dim3(64), 0, st, heights, g, x1, x2, y1, y2, vs, env_length, env_width, origins);
  (void)hipFreeAsync(d_tiles, st);
  (void)hipStreamSynchronize(st);              // tiles_host may go away when we return
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

int lg_heightfield_to_trimesh(const int16_t* heights, int32_t rows, int32_t cols, double step_x, double step_y, double stop_x, double stop_y,
                              double hs, double vs, double thr, float* vertices, uint32_t* triangles, void* stream) {
  if (!heights || !vertices || !triangles || rows < 2 || cols < 2) return LG_ERR_INVALID;
  const int dev = device_of_ptr(heights);
  if (dev < 0) return LG_ERR_INVALID;
  Scope sc(dev);
  hipStream_t st = (hipStream_t)stream;
  MeshParams m{rows, cols, step_x, step_y, stop_x, stop_y, hs, vs, thr};
  const int64_t nv = (int64_t)rows * cols, nc = (int64_t)(rows - 1) * (cols - 1);
  hipLaunchKernelGGL(trimesh_vertices_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, heights, m, vertices);
  hipLaunchKernelGGL(trimesh_triangles_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, rows, cols, triangles);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

}  // extern "C"

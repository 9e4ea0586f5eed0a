// lg_mesh.hip — triangle-mesh queries on gfx950: BVH build (host), ray casting, closest point / signed distance,
// and the two fused sensor kernels built on them (RayCaster, depth camera).
//
// Replaces NVIDIA Warp in the reference: wp.Mesh(points, indices) BVH (utils/ray_caster.py:39-42, utils/mesh_sdf.py:32-35),
// wp.mesh_query_ray in raycast_mesh_kernel (ray_caster.py:45-92) and wp.mesh_query_point_sign_normal in
// query_sdf_kernel (mesh_sdf.py:38-116).  Warp itself is closed-source-adjacent third party code that is not in the
// reference tree; semantics restated: closest two-sided hit with 0 <= t <= max_dist; closest point on the surface with
// the sign taken from the normal of the closest feature's face.
//
// Layout: a binned-SAH binary tree built on the host, collapsed into 128-byte 4-wide nodes that carry their children's
// boxes (lg_bvh.h); triangles re-ordered by leaf and stored as three float4 (48 B) so one leaf is a short contiguous
// burst; everything stays resident in HBM/L2.  One ray / point per lane, nearest-first traversal with a per-lane stack.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "lg_device.h"
#include "../../include/lgstep.h"

#include "lg_bvh.h"


static thread_local std::string g_mesh_err;

// ------------------------------------------------------------------------------------------------ host: BVH build
namespace {
struct BuildTri { float v[9]; float c[3]; };
struct Builder {
  std::vector<BuildTri>& t; std::vector<BvhNode>& nodes;
  void bounds(int first, int count, float* mn, float* mx) {
    for (int k = 0; k < 3; ++k) { mn[k] = 1e30f; mx[k] = -1e30f; }
    for (int i = first; i < first + count; ++i)
      for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], t[i].v[3 * v + k]); mx[k] = std::max(mx[k], t[i].v[3 * v + k]); }
  }
  void build(int node, int first, int count) {
    BvhNode& n = nodes[node];
    bounds(first, count, n.bmin, n.bmax);
    if (count <= 4) { n.left_first = first; n.count = count; return; }
    // binned SAH on centroids, 8 bins per axis
    float cmin[3] = {1e30f, 1e30f, 1e30f}, cmax[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = first; i < first + count; ++i) for (int k = 0; k < 3; ++k) { cmin[k] = std::min(cmin[k], t[i].c[k]); cmax[k] = std::max(cmax[k], t[i].c[k]); }
    int best_axis = -1; float best_cost = 1e30f, best_split = 0.f;
    const int NB = 8;
    for (int ax = 0; ax < 3; ++ax) {
      float ext = cmax[ax] - cmin[ax];
      if (!(ext > 1e-9f)) continue;
      int cnt[NB] = {0}; float bmn[NB][3], bmx[NB][3];
      for (int b = 0; b < NB; ++b) for (int k = 0; k < 3; ++k) { bmn[b][k] = 1e30f; bmx[b][k] = -1e30f; }
      for (int i = first; i < first + count; ++i) {
        int b = std::min(NB - 1, (int)((t[i].c[ax] - cmin[ax]) / ext * NB));
        cnt[b]++;
        for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) { bmn[b][k] = std::min(bmn[b][k], t[i].v[3 * v + k]); bmx[b][k] = std::max(bmx[b][k], t[i].v[3 * v + k]); }
      }
      float lmn[3], lmx[3], area_l[NB], area_r[NB]; int nl[NB], nr[NB];
      auto area = [](const float* a, const float* b) { float x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2]; return x * y + y * z + z * x; };
      for (int k = 0; k < 3; ++k) { lmn[k] = 1e30f; lmx[k] = -1e30f; }
      int acc = 0;
      for (int b = 0; b < NB - 1; ++b) {
        for (int k = 0; k < 3; ++k) { lmn[k] = std::min(lmn[k], bmn[b][k]); lmx[k] = std::max(lmx[k], bmx[b][k]); }
        acc += cnt[b]; nl[b] = acc; area_l[b] = acc ? area(lmn, lmx) : 0.f;
      }
      for (int k = 0; k < 3; ++k) { lmn[k] = 1e30f; lmx[k] = -1e30f; }
      acc = 0;
      for (int b = NB - 1; b > 0; --b) {
        for (int k = 0; k < 3; ++k) { lmn[k] = std::min(lmn[k], bmn[b][k]); lmx[k] = std::max(lmx[k], bmx[b][k]); }
        acc += cnt[b]; nr[b - 1] = acc; area_r[b - 1] = acc ? area(lmn, lmx) : 0.f;
      }
      for (int b = 0; b < NB - 1; ++b) {
        if (nl[b] == 0 || nr[b] == 0) continue;
        float cost = area_l[b] * nl[b] + area_r[b] * nr[b];
        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = cmin[ax] + ext * (b + 1) / NB; }
      }
    }
    int mid;
    if (best_axis < 0) {      // all centroids coincide: split in the middle
      mid = first + count / 2;
    } else {
      auto it = std::partition(t.begin() + first, t.begin() + first + count, [&](const BuildTri& a) { return a.c[best_axis] < best_split; });
      mid = (int)(it - t.begin());
      if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int left = (int)nodes.size();
    nodes.push_back(BvhNode()); nodes.push_back(BvhNode());
    nodes[node].left_first = left; nodes[node].count = 0;
    build(left, first, mid - first);
    build(left + 1, mid, first + count - mid);
  }
};
// binary tree -> 4-wide nodes that hold their children's boxes (see lg_bvh.h)
struct Collapser {
  const std::vector<BvhNode>& n2; std::vector<BvhNode4>& n4;
  static float area(const BvhNode& n) { float x = n.bmax[0] - n.bmin[0], y = n.bmax[1] - n.bmin[1], z = n.bmax[2] - n.bmin[2]; return x * y + y * z + z * x; }
  int emit(int root2) {
    int kids[4]; int nk = 0;
    if (n2[root2].count > 0) kids[nk++] = root2;                       // the whole mesh is one leaf
    else { kids[nk++] = n2[root2].left_first; kids[nk++] = n2[root2].left_first + 1; }
    while (nk < 4) {                                                   // open the inner child with the largest box
      int best = -1; float ba = -1.f;
      for (int i = 0; i < nk; ++i) if (n2[kids[i]].count == 0 && area(n2[kids[i]]) > ba) { ba = area(n2[kids[i]]); best = i; }
      if (best < 0) break;
      const int c = kids[best];
      kids[best] = n2[c].left_first; kids[nk++] = n2[c].left_first + 1;
    }
    const int me = (int)n4.size();
    n4.push_back(BvhNode4());
    for (int k = 0; k < 4; ++k) {
      BvhNode4& N = n4[me];
      if (k >= nk) { N.minx[k] = N.miny[k] = N.minz[k] = 1e30f; N.maxx[k] = N.maxy[k] = N.maxz[k] = -1e30f; N.child[k] = BVH4_EMPTY; N.pad[k] = 0; continue; }
      const BvhNode& c = n2[kids[k]];
      // Boxes a micrometre (and two ulps) wider than their triangles: a ray without motion along an axis that runs exactly down a box face (a vertical ray on a lattice line
      // of a heightfield mesh) then lies INSIDE the box -- the slab test's 0 * 1e12 for that axis rejected the box on the ray's - side, and with it
      // triangles that reach the ray with an edge.  Wider boxes change no result (the triangle tests decide), only which leaves are looked at.
      // (an absolute part as well: two ulps of a coordinate near zero are denormal, and 1e-45 * 1e12 is still nothing)
      auto lo = [](float v) { return nextafterf(nextafterf(v - 1e-6f, -INFINITY), -INFINITY); };
      auto hi = [](float v) { return nextafterf(nextafterf(v + 1e-6f, INFINITY), INFINITY); };
      N.minx[k] = lo(c.bmin[0]); N.miny[k] = lo(c.bmin[1]); N.minz[k] = lo(c.bmin[2]);
      N.maxx[k] = hi(c.bmax[0]); N.maxy[k] = hi(c.bmax[1]); N.maxz[k] = hi(c.bmax[2]);
      N.pad[k] = 0;
      if (c.count > 0) N.child[k] = ~((c.left_first << 3) | (c.count - 1));
      else { const int idx = emit(kids[k]); n4[me].child[k] = idx; }   // (n4 may have been reallocated: index, not reference)
    }
    return me;
  }
};

// ---- ray lattice (lg_bvh.h: trace_ray_grid).  Built when the triangle vertices sit on a rectilinear lattice: few distinct x and y values compared
// with the number of triangles (a heightfield mesh of R x C vertices has R and C of them, slope-corrected or not).  Cell (ix, iy) is
// [xb[ix], xb[ix+1]] x [yb[iy], yb[iy+1]] between consecutive distinct coordinates, so every triangle edge lies ON cell lines and a triangle is listed
// exactly in the cells its xy bounding box overlaps with positive area (a box of zero width -- a vertical face -- in the cells on both sides of its line).
struct RayGridHost {
  std::vector<float> xb, yb; std::vector<int2> cells; std::vector<float2> zr; std::vector<float4> tris;
  bool build(const std::vector<BuildTri>& t) {
    const size_t T = t.size();
    std::vector<float> xs, ys; xs.reserve(3 * T); ys.reserve(3 * T);
    for (const BuildTri& b : t) for (int v = 0; v < 3; ++v) { xs.push_back(b.v[3 * v]); ys.push_back(b.v[3 * v + 1]); }
    for (const BuildTri& b : t) for (int k = 0; k < 9; ++k) if (!std::isfinite(b.v[k])) return false;
    std::sort(xs.begin(), xs.end()); xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
    std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
    if (xs.size() < 2 || ys.size() < 2 || xs.size() + ys.size() > 6000) return false;        // the boundary tables must fit LDS next to a depth image
    const size_t nx = xs.size() - 1, ny = ys.size() - 1;
    if (nx * ny > 4 * T + 64) return false;                                                     // not a lattice mesh: the cells would mostly be empty
    auto range = [](const std::vector<float>& b, float lo, float hi, int& c0, int& c1) {         // cells [c0, c1] the interval [lo, hi] is listed in
      const int n = (int)b.size() - 1;
      const int ilo = (int)(std::lower_bound(b.begin(), b.end(), lo) - b.begin()), ihi = (int)(std::lower_bound(b.begin(), b.end(), hi) - b.begin());   // b[ilo] == lo, b[ihi] == hi: vertex coordinates ARE the boundaries
      if (ilo == ihi) { c0 = std::max(0, ilo - 1); c1 = std::min(n - 1, ilo); }                  // zero width: the cells on both sides of the line
      else { c0 = ilo; c1 = ihi - 1; }
    };
    auto bbox = [&](const BuildTri& b, int& x0, int& x1, int& y0, int& y1) {
      const float lx = std::min({b.v[0], b.v[3], b.v[6]}), hx = std::max({b.v[0], b.v[3], b.v[6]});
      const float ly = std::min({b.v[1], b.v[4], b.v[7]}), hy = std::max({b.v[1], b.v[4], b.v[7]});
      range(xs, lx, hx, x0, x1); range(ys, ly, hy, y0, y1);
    };
    std::vector<uint32_t> count(nx * ny, 0);
    size_t total = 0;
    for (const BuildTri& b : t) { int x0, x1, y0, y1; bbox(b, x0, x1, y0, y1); for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) { ++count[(size_t)y * nx + x]; ++total; } }
    if (total > 8 * T || total >= (1ull << 31)) return false;
    cells.assign(nx * ny, make_int2(0, 0));
    size_t acc = 0;
    for (size_t c = 0; c < nx * ny; ++c) { cells[c].x = (int)acc; acc += count[c]; count[c] = 0; }
    tris.resize(total * 3);
    std::vector<float> zlo(nx * ny, 1e30f), zhi(nx * ny, -1e30f);
    for (const BuildTri& b : t) {
      int x0, x1, y0, y1; bbox(b, x0, x1, y0, y1);
      const float lz = std::min({b.v[2], b.v[5], b.v[8]}), hz = std::max({b.v[2], b.v[5], b.v[8]});
      for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
        const size_t c = (size_t)y * nx + x, at = (size_t)cells[c].x + count[c]++;
        for (int v = 0; v < 3; ++v) tris[3 * at + v] = make_float4(b.v[3 * v], b.v[3 * v + 1], b.v[3 * v + 2], 0.f);
        zlo[c] = std::min(zlo[c], lz); zhi[c] = std::max(zhi[c], hz);
      }
    }
    zr.resize(nx * ny);
    for (size_t c = 0; c < nx * ny; ++c) { cells[c].y = (int)count[c]; zr[c] = make_float2(zlo[c], zhi[c]); }     // (an empty cell keeps +1e30, -1e30: no ray reaches it)
    xb = xs; yb = ys;
    return true;
  }
};
}  // namespace


// ------------------------------------------------------------------------------------------------ kernels
// boundary tables of the ray lattice -> LDS (every thread of the workgroup calls this; no-op without a lattice)
LG_DEV void raygrid_stage(const RayGrid& G, float* tb) {
  for (int i = threadIdx.x; i <= G.nx; i += blockDim.x) tb[i] = G.xb[i];
  for (int i = threadIdx.x; i <= G.ny; i += blockDim.x) tb[G.nx + 1 + i] = G.yb[i];
  __syncthreads();
}
// GRID: the instance for meshes with a ray lattice -- a separate kernel, because the tree walk's per-lane stack lives in scratch and scratch limits the
// resident waves of every lane of a kernel that MAY take that path
template <bool GRID>
LG_DEV float trace_any(const MeshView& M, const RayGrid& G, const float* tb, V3 o, V3 d, float max_dist) {
  if (GRID) return trace_ray_grid_tb(G, tb, o, d, max_dist);
  return trace_ray(M, o, d, max_dist);
}
static RayGrid ray_grid_of(const lg_mesh* m) { return RayGrid{m->d_gxb, m->d_gyb, m->gnx, m->gny, m->d_gzr, m->d_gcells, m->d_gtris, m->d_gzb, m->gnbx, m->gnby}; }
static size_t ray_grid_lds(const lg_mesh* m) { return m->d_gcells ? (size_t)(m->gnx + m->gny + 2 + 4) * sizeof(float) : 0; }   // (+ 4: a workgroup's reduction scratch, depth_kernel)
// raycast_mesh (ray_caster.py:95-167): hit point o + t d, or the ray end point o + d max_dist and found = 0
template <bool GRID>
__global__ __launch_bounds__(256) void raycast_kernel(MeshView M, RayGrid G, const float* __restrict__ o, const float* __restrict__ d, int64_t n,
                                                      float max_dist, float* __restrict__ hits, uint8_t* __restrict__ found) {
  extern __shared__ float rg_tab[];
  if (GRID) raygrid_stage(G, rg_tab);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  V3 ro = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
  float t = trace_any<GRID>(M, G, rg_tab, ro, rd, max_dist);
  bool hit = t >= 0.f;
  V3 h = ro + (hit ? t : max_dist) * rd;
  hits[3 * i] = h.x; hits[3 * i + 1] = h.y; hits[3 * i + 2] = h.z;
  found[i] = hit ? 1 : 0;
}

// query_sdf_kernel (mesh_sdf.py:38-116): signed distance, unit gradient (pointing away from the surface outside, flipped
// inside), face-normal fallback on the surface, max_distance / zero gradient when nothing is within range
__global__ __launch_bounds__(256) void sdf_kernel(MeshView M, const float* __restrict__ pts, int64_t n, float max_dist,
                                                  float* __restrict__ sdf, float* __restrict__ grad) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  V3 p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  V3 cp, fn;
  if (!closest_point(M, p, max_dist, &cp, &fn)) {
    sdf[i] = max_dist; grad[3 * i] = 0.f; grad[3 * i + 1] = 0.f; grad[3 * i + 2] = 0.f;
    return;
  }
  V3 diff = p - cp; float dist = norm(diff);
  float fl = norm(fn); V3 nrm = fl > 0.f ? (1.f / fl) * fn : v3(0, 0, 1);
  float sign = dot(diff, nrm) < 0.f ? -1.f : 1.f;
  V3 g;
  if (dist > 1e-6f) g = (sign / dist) * diff; else g = sign * nrm;
  sdf[i] = sign * dist;
  grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
}

// RayCaster._update_ray_casting + LeggedRobotRayCast._get_raycast_distances (ray_caster.py:558-594,
// legged_robot_raycast.py:262-297): one lane per (listed env, ray); outputs are indexed by env id, the distance
// observation with a row stride so that it can live inside a wider extra-observation row
template <bool GRID>
__global__ __launch_bounds__(256) void raycaster_kernel(MeshView M, RayGrid G, const float* __restrict__ root /* (N,13) */, const float* __restrict__ ray_o,
                                                        const float* __restrict__ ray_d, const int32_t* __restrict__ ids, int n_ids, int R,
                                                        float max_dist, int yaw_only, float* __restrict__ hits, uint8_t* __restrict__ found,
                                                        float* __restrict__ dist, int dist_stride) {
  extern __shared__ float rg_tab[];
  if (GRID) raygrid_stage(G, rg_tab);
  int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (int64_t)n_ids * R) return;
  const int kq = (int)(gi / R), r = (int)(gi - (int64_t)kq * R);
  const int e = ids ? ids[kq] : kq;
  const int64_t i = (int64_t)e * R + r;
  const float* rs = root + (size_t)e * 13;
  float q[4] = {rs[3], rs[4], rs[5], rs[6]};
  if (yaw_only) {   // math_utils.quat_apply_yaw: zero x, y and renormalise
    float nrm = fmaxf(sqrtf(q[2] * q[2] + q[3] * q[3]), 1e-9f);
    q[0] = 0.f; q[1] = 0.f; q[2] /= nrm; q[3] /= nrm;
  }
  V3 pos = v3(rs[0], rs[1], rs[2]);
  V3 o = quat_apply(q, v3(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2])) + pos;
  V3 d = quat_apply(q, v3(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
  float t = trace_any<GRID>(M, G, rg_tab, o, d, max_dist);
  bool hit = t >= 0.f;
  V3 h = o + (hit ? t : max_dist) * d;
  hits[3 * i] = h.x; hits[3 * i + 1] = h.y; hits[3 * i + 2] = h.z;
  found[i] = hit ? 1 : 0;
  // distance from the ROBOT BASE position, not from the ray origin (legged_robot_raycast.py:278-285)
  float dd = norm(h - pos);
  float nd = 1.f - fminf(fmaxf(dd / max_dist, 0.f), 1.f);
  dist[(size_t)e * dist_stride + r] = hit ? nd : 0.f;
}

// RobotBatchRolloutPercept._update_sdf_values (robot_batch_rollout_percept.py:384-440): for every listed env and every query
// body, the collision-sphere centre = body position + body rotation * offset, then signed distance, unit gradient and the
// nearest surface point p - sdf * grad (mesh_sdf.py:295-336) — all bodies of all envs in one launch (the reference runs
// two Warp queries per body).  sdf rows have a stride so they can be the tail of the extra-observation row.
// Round 6: on a lattice mesh (the OBJ meshes of the confined-space / heightfield converters: lg_mesh.d_gcz) a query with a cached bound is answered from the
// cells around the body by a group of 16 lanes (closest_point_lattice_row16, lg_bvh.h) instead of one lane's tree walk -- the walk paid ~30 dependent
// 128-byte node fetches per query and re-fetched 14 x its algorithmic bytes.  Same per-face arithmetic and tie rule: the same answer to rounding
// (tests/test_hip_config3.py).  A query whose bound is wider than LATTICE_SDF_CELLS cell widths (no cache entry yet; a body far above the surface) keeps the
// tree, walked by the group's first lane.  LG_SDF_LATTICE=0: always the tree (the A/B switch and the tests' checker).
#define LATTICE_SDF_CELLS 12.f
__global__ __launch_bounds__(256) void sdf_bodies_kernel(MeshView M, LatticeView L, const float* __restrict__ rb /* (N,B,13) */, int B,
                                                         const int32_t* __restrict__ body_idx, const float* __restrict__ offsets, int nb,
                                                         const int32_t* __restrict__ ids, int n_ids, float max_dist,
                                                         float* __restrict__ sdf, int sdf_stride, float* __restrict__ grad,
                                                         float* __restrict__ nearest, float4* __restrict__ cache, int body_major) {
  const int k = threadIdx.x & 15;
  int64_t gi = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);                 // one query per 16 lanes
  const bool live = gi < (int64_t)n_ids * nb;
  if (!live) gi = (int64_t)n_ids * nb - 1;                                      // (a spare group repeats the last query and stores nothing: the group functions want all lanes)
  int kq = (int)(gi / nb), b = (int)(gi - (int64_t)kq * nb);
  if (body_major) { b = (int)(gi / n_ids); kq = (int)(gi - (int64_t)b * n_ids); }      // a wave's four groups = the same body of four envs (trunks with trunks: their windows are ~50 cells, the feet's a few)
  const int e = ids ? ids[kq] : kq;
  const float* s = rb + ((size_t)e * B + body_idx[b]) * 13;
  const float q[4] = {s[3], s[4], s[5], s[6]};
  V3 p = v3(s[0], s[1], s[2]);
  if (offsets) p = p + quat_apply(q, v3(offsets[3 * b], offsets[3 * b + 1], offsets[3 * b + 2]));
  V3 cp = p, fn = v3(0, 0, 1); float sd = max_dist; V3 g = v3(0, 0, 0);
  // The closest point this slot found last time is a point of the surface, so its distance from the new position bounds the new
  // distance from above -- whatever the body did in between: the search starts with that radius instead of max_dist (10 m in the
  // reference's configs) and opens a handful of nodes / cells instead of ~100.  Exact: every face within the true distance is still visited.
  float md = max_dist;
  const float4 c4 = cache[gi];
  if (c4.w == 1.f) md = fminf(max_dist, norm(p - v3(c4.x, c4.y, c4.z)) * (1.f + 1e-4f) + 1e-5f);
  bool found = false, tree = L.cell == nullptr;
  if (!tree) {
    // the group's search starts with a radius of a few cells and doubles it until something is found or the bound is reached: a body whose cache entry is
    // stale (a reset teleported it) or missing does not search its whole bound, and a body on the ground finds the surface in the first probe.  Exact: a
    // search with radius R returns the closest face whenever one lies within R.  Beyond LATTICE_SDF_CELLS cell widths the tree takes over.
    const float h = fminf(L.hx, L.hy);
    float R = fminf(md, 3.f * h);
    while (true) {                                                               // (R, md: the same on the 16 lanes of a group)
      found = closest_point_lattice_row16(L, p, R, k, &cp, &fn);
      if (found || R >= md) break;
      if (2.f * R > LATTICE_SDF_CELLS * h) { tree = true; break; }
      R = fminf(2.f * R, md);
    }
  }
  if (k != 0 || !live) return;
  if (tree) found = closest_point(M, p, md, &cp, &fn);
  if (!found && md < max_dist) found = closest_point(M, p, max_dist, &cp, &fn);     // (rounding at the boundary of the reduced radius)
  cache[gi] = found ? make_float4(cp.x, cp.y, cp.z, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (found) {
    V3 diff = p - cp; float dist = norm(diff);
    float sign = dot(diff, fn) < 0.f ? -1.f : 1.f;
    g = dist > 1e-6f ? (sign / dist) * diff : sign * fn;
    sd = sign * dist;
  }
  const size_t o = (size_t)e * nb + b;
  sdf[(size_t)e * sdf_stride + b] = sd;
  if (grad) { grad[3 * o] = g.x; grad[3 * o + 1] = g.y; grad[3 * o + 2] = g.z; }
  if (nearest) { V3 np_ = p - sd * g; nearest[3 * o] = np_.x; nearest[3 * o + 1] = np_.y; nearest[3 * o + 2] = np_.z; }
}

LG_DEV float cubic_w(float x) {   // Keys kernel, a = -0.75 (torch / torchvision bicubic)
  const float a = -0.75f; x = fabsf(x);
  if (x <= 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a;
  return 0.f;
}

// DepthCameraWarp.update + update_depth_buffer + process_depth_image (depth_camera.py:402-566, 84-138, 56-69):
// one workgroup per env; the raw H x W depth image lives in LDS between the ray pass and the resize pass.
// GRID: 0 the tree walk, 1 the lattice walk (boundary tables in LDS, cell records from global memory).
template <int GRID>
__global__ __launch_bounds__(256) void depth_kernel(MeshView M, RayGrid G, const float* __restrict__ root, const float* __restrict__ ray_d /* (H*W,3) */,
                                                    const int64_t* __restrict__ ep_len, int W, int H, int TW, int TH, int RW, int RH, int buffer_len,
                                                    float near_clip, float far_clip, float px, float py, float pz,
                                                    float qx, float qy, float qz, float qw, const float* __restrict__ env_noise,
                                                    float* __restrict__ cam_pos, float* __restrict__ cam_rot, float* __restrict__ depth_buffer, int getenv_skip, int rs_off) {
  extern __shared__ float img[];
  float* const rg_tab = img + W * H;       // GRID: x boundaries [nx + 1] | y boundaries [ny + 1] | 4 wave maxima
  if (GRID == 1) raygrid_stage(G, rg_tab);
  const int e = blockIdx.x, tid = threadIdx.x;
  const float* rs = root + (size_t)e * 13;
  const float bq[4] = {rs[3], rs[4], rs[5], rs[6]};
  // camera pose: position = base + R(base) offset; rotation = quat_mul(base, offset) in Isaac Gym's xyzw convention
  V3 cpos = v3(rs[0], rs[1], rs[2]) + quat_apply(bq, v3(px, py, pz));
  float cq[4];
  {
    float x1 = bq[0], y1 = bq[1], z1 = bq[2], w1 = bq[3], x2 = qx, y2 = qy, z2 = qz, w2 = qw;
    float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz, qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    cq[3] = qq - ww + (z1 - y1) * (y2 - z2); cq[0] = qq - xx + (x1 + w1) * (x2 + w2);
    cq[1] = qq - yy + (w1 - x1) * (y2 + z2); cq[2] = qq - zz + (z1 + y1) * (w2 - x2);
  }
  if (tid == 0) {
    cam_pos[3 * e] = cpos.x; cam_pos[3 * e + 1] = cpos.y; cam_pos[3 * e + 2] = cpos.z;
    cam_rot[4 * e] = cq[0]; cam_rot[4 * e + 1] = cq[1]; cam_rot[4 * e + 2] = cq[2]; cam_rot[4 * e + 3] = cq[3];
  }
  const float noise = env_noise ? env_noise[e] : 0.f;
  float ztop1 = 3.0e38f;
  if (GRID == 1 && getenv_skip) {
    // (round 6) the highest triangle the camera's rays can reach: block maxima over the square [cpos - R, cpos + R], R = far_clip * 1.001 (a ray longer than
    // that -- |d| > 1.001: the pattern's rays are unit vectors -- does not use it).  The walk of a ray starts where it comes down to that height
    // (trace_ray_grid); a ray that never does misses without a walk.
    float* const red = rg_tab + G.nx + G.ny + 2;
    const float R = far_clip * 1.001f * (1.f + 1e-4f) + 1e-4f;
    const RayTables A{rg_tab, &G, 3.0e38f, false};
    const float xlo = A.x_lo(), xhi = A.x_hi(), ylo = A.y_lo(), yhi = A.y_hi();
    const float ux = (float)G.nx / (xhi - xlo), uy = (float)G.ny / (yhi - ylo);
    const int bx0 = raygrid_locate(A, true, G.nx, cpos.x - R, (int)((cpos.x - R - xlo) * ux)) / RAY_BLK, bx1 = raygrid_locate(A, true, G.nx, cpos.x + R, (int)((cpos.x + R - xlo) * ux)) / RAY_BLK;
    const int by0 = raygrid_locate(A, false, G.ny, cpos.y - R, (int)((cpos.y - R - ylo) * uy)) / RAY_BLK, by1 = raygrid_locate(A, false, G.ny, cpos.y + R, (int)((cpos.y + R - ylo) * uy)) / RAY_BLK;
    const int bw = bx1 - bx0 + 1, nb = bw * (by1 - by0 + 1);
    float zt = -3.0e38f;
    for (int i = tid; i < nb; i += 256) { const int ry = i / bw, rx = i - ry * bw; zt = fmaxf(zt, G.zb[(size_t)(by0 + ry) * G.nbx + bx0 + rx]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zt = fmaxf(zt, __shfl_xor(zt, o));
    if ((tid & 63) == 0) red[tid >> 6] = zt;
    __syncthreads();
    ztop1 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (getenv_skip == 9) ztop1 = -3.0e37f;        // (timing probe: every ray misses without a walk -- what is left is staging, resize and the FIFO)
  }
  // (Measured and dropped, round 6: the rays sorted first -- spans for all, the ones with cells to cross queued in LDS, the walk over the queue -- 0.565 against
  // 0.490 ms per step of config 4: the kernel is bound by the vector instructions of the walks themselves (SQ counters: 142 M per launch, 66 % of their lanes
  // active), and a queue's neighbours are no longer the pixels of one tile, whose walks are equally long.)
  // a wave takes a TW x TH tile of pixels (TW * TH <= 64, chosen by the host), not 64 consecutive pixels of a row: the rays of a tile cross the same
  // cells (or tree nodes) and finish together
  const int lane = tid & 63, tcols = (W + TW - 1) / TW, ntiles = tcols * ((H + TH - 1) / TH);
  const int ly = lane / TW, lx = lane - ly * TW;
  for (int tile = tid >> 6; tile < ntiles; tile += 4) {
    const int tyi = tile / tcols, txi = tile - tyi * tcols, pxl = txi * TW + lx, pyl = tyi * TH + ly;
    if (ly >= TH || pxl >= W || pyl >= H) continue;
    const int p = pyl * W + pxl;
    V3 d = quat_apply(cq, v3(ray_d[3 * p], ray_d[3 * p + 1], ray_d[3 * p + 2]));
    float t = GRID == 1 ? trace_ray_grid_tb(G, rg_tab, cpos, d, far_clip, dot(d, d) <= 1.002f ? ztop1 : 3.0e38f, getenv_skip >= 2) : trace_ray(M, cpos, d, far_clip);
    float depth = t >= 0.f ? -(t * norm(d)) : -far_clip;
    depth += noise;
    img[p] = fminf(fmaxf(depth, -far_clip), -near_clip);
  }
  // The bicubic weights of an output pixel depend on its column and its row only: 4 + 4 weights and two source indices per column / row, formed ONCE per
  // workgroup (round 6).  In the loop below they were 20 evaluations of the cubic per output pixel -- a quarter of the kernel's vector instructions
  // (~380 of them per output, six outputs per thread).  Same expressions, same order of the sums: the same image bit for bit.
  float* const rsx = img + rs_off;            // [RW][5]: source column of tap -1 (unclamped) as float bits, then the four x weights
  float* const rsy = rsx + 5 * RW;            // [RH][5]
  const float sx = (float)W / (float)RW, sy = (float)H / (float)RH;
  const bool resize = !(RW == W && RH == H);
  if (resize) {
    for (int i = tid; i < RW + RH; i += 256) {
      const bool isx = i < RW;
      const int o = isx ? i : i - RW;
      const float f = (o + 0.5f) * (isx ? sx : sy) - 0.5f;
      const int i0 = (int)floorf(f);
      const float t = f - i0;
      float* dst = (isx ? rsx : rsy) + 5 * o;
      dst[0] = __int_as_float(i0);
#pragma unroll
      for (int k = -1; k <= 2; ++k) dst[2 + k] = cubic_w(t - k);
    }
  }
  __syncthreads();
  const bool init = ep_len[e] <= 1;
  float* buf = depth_buffer + (size_t)e * buffer_len * RW * RH;
  for (int p = tid; p < RW * RH; p += 256) {
    int oy = p / RW, ox = p - oy * RW;
    float v;
    if (!resize) v = img[p];
    else {   // bicubic, align_corners = False, no antialias
      const int ix = __float_as_int(rsx[5 * ox]), iy = __float_as_int(rsy[5 * oy]);
      float acc = 0.f;
#pragma unroll
      for (int m = -1; m <= 2; ++m) {
        int yy = min(max(iy + m, 0), H - 1); float wy = rsy[5 * oy + 2 + m];
        float row = 0.f;
#pragma unroll
        for (int k = -1; k <= 2; ++k) { int xx = min(max(ix + k, 0), W - 1); row += rsx[5 * ox + 2 + k] * img[yy * W + xx]; }
        acc += wy * row;
      }
      v = acc;
    }
    v = (v * -1.f - near_clip) / (far_clip - near_clip) - 0.5f;   // normalize_depth_image
    if (init) { for (int k = 0; k < buffer_len; ++k) buf[(size_t)k * RW * RH + p] = v; }
    else {
      for (int k = 0; k + 1 < buffer_len; ++k) buf[(size_t)k * RW * RH + p] = buf[(size_t)(k + 1) * RW * RH + p];
      buf[(size_t)(buffer_len - 1) * RW * RH + p] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* lg_mesh_last_error(lg_mesh* m) { return m ? m->err.c_str() : g_mesh_err.c_str(); }

void lg_mesh_destroy(lg_mesh* m) {
  if (!m) return;
  DeviceScope ds_(m->device);
  if (m->d_nodes) (void)hipFree(m->d_nodes);
  if (m->d_tris) (void)hipFree(m->d_tris);
  if (m->d_sdf_cache) (void)hipFree(m->d_sdf_cache);
  if (m->d_gxb) (void)hipFree(m->d_gxb);
  if (m->d_gyb) (void)hipFree(m->d_gyb);
  if (m->d_gcells) (void)hipFree(m->d_gcells);
  if (m->d_gzr) (void)hipFree(m->d_gzr);
  if (m->d_gzb) (void)hipFree(m->d_gzb);
  if (m->d_gcz) (void)hipFree(m->d_gcz);
  if (m->d_gcr) (void)hipFree(m->d_gcr);
  if (m->d_gtris) (void)hipFree(m->d_gtris);
  delete m;
}

lg_mesh* lg_mesh_create(const float* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles, int device_id) {
  if (!vertices || !triangles || n_vertices <= 0 || n_triangles <= 0) { g_mesh_err = "empty mesh"; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_mesh_err = "no HIP device: mesh queries have no CPU path"; return nullptr; }
  if (device_id < 0 || device_id >= ndev) { g_mesh_err = "bad device"; return nullptr; }
  DeviceScope ds_(device_id);
  if (!ds_.ok) { g_mesh_err = "bad device"; return nullptr; }
  std::vector<BuildTri> t((size_t)n_triangles);
  for (int64_t i = 0; i < n_triangles; ++i) {
    for (int v = 0; v < 3; ++v) {
      int32_t idx = triangles[3 * i + v];
      if (idx < 0 || idx >= n_vertices) { g_mesh_err = "triangle index out of range"; return nullptr; }
      for (int k = 0; k < 3; ++k) t[i].v[3 * v + k] = vertices[3 * (int64_t)idx + k];
    }
    for (int k = 0; k < 3; ++k) t[i].c[k] = (t[i].v[k] + t[i].v[3 + k] + t[i].v[6 + k]) / 3.f;
  }
  std::vector<BvhNode> nodes; nodes.reserve((size_t)n_triangles); nodes.push_back(BvhNode());
  Builder b{t, nodes};
  b.build(0, 0, (int)n_triangles);
  std::vector<float4> packed((size_t)n_triangles * 3);
  for (int64_t i = 0; i < n_triangles; ++i)
    for (int v = 0; v < 3; ++v) packed[3 * i + v] = make_float4(t[i].v[3 * v], t[i].v[3 * v + 1], t[i].v[3 * v + 2], 0.f);
  if (n_triangles >= (1ll << 28)) { g_mesh_err = "mesh too large for the leaf encoding (2^28 triangles)"; return nullptr; }
  std::vector<BvhNode4> nodes4; nodes4.reserve(nodes.size() / 2 + 1);
  Collapser col{nodes, nodes4};
  col.emit(0);
  lg_mesh* m = new lg_mesh();
  m->device = device_id; m->n_tris = n_triangles; m->n_nodes = (int64_t)nodes4.size();
  for (int k = 0; k < 3; ++k) { m->bmin[k] = nodes[0].bmin[k]; m->bmax[k] = nodes[0].bmax[k]; }
  if (hipMalloc((void**)&m->d_nodes, nodes4.size() * sizeof(BvhNode4)) != hipSuccess ||
      hipMalloc((void**)&m->d_tris, packed.size() * sizeof(float4)) != hipSuccess ||
      hipMemcpy(m->d_nodes, nodes4.data(), nodes4.size() * sizeof(BvhNode4), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->d_tris, packed.data(), packed.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
    g_mesh_err = "device allocation / upload of the BVH failed"; lg_mesh_destroy(m); return nullptr;
  }
  // rays over a lattice mesh walk its cells instead of the tree (LG_RAY_GRID=0: always the tree -- the A/B switch, and how the tests compare the two)
  const char* rg = getenv("LG_RAY_GRID");
  RayGridHost G;
  if (!(rg && rg[0] == '0') && G.build(t)) {
    if (hipMalloc((void**)&m->d_gxb, G.xb.size() * 4) != hipSuccess || hipMalloc((void**)&m->d_gyb, G.yb.size() * 4) != hipSuccess ||
        hipMalloc((void**)&m->d_gcells, G.cells.size() * sizeof(int2)) != hipSuccess || hipMalloc((void**)&m->d_gzr, G.zr.size() * sizeof(float2)) != hipSuccess ||
        hipMemcpy(m->d_gzr, G.zr.data(), G.zr.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess || hipMalloc((void**)&m->d_gtris, G.tris.size() * sizeof(float4)) != hipSuccess ||
        hipMemcpy(m->d_gxb, G.xb.data(), G.xb.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_gyb, G.yb.data(), G.yb.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_gcells, G.cells.data(), G.cells.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_gtris, G.tris.data(), G.tris.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
      g_mesh_err = "device allocation / upload of the ray lattice failed"; lg_mesh_destroy(m); return nullptr;
    }
    m->gnx = (int)G.xb.size() - 1; m->gny = (int)G.yb.size() - 1;
    {   // block maxima of the cells' z ranges (depth_kernel: where a camera's rays start their walk)
      m->gnbx = (m->gnx + RAY_BLK - 1) / RAY_BLK; m->gnby = (m->gny + RAY_BLK - 1) / RAY_BLK;
      std::vector<float> zb((size_t)m->gnbx * m->gnby, -3.0e38f);
      for (int iy = 0; iy < m->gny; ++iy) for (int ix = 0; ix < m->gnx; ++ix) {
        float& b = zb[(size_t)(iy / RAY_BLK) * m->gnbx + ix / RAY_BLK];
        b = std::max(b, G.zr[(size_t)iy * m->gnx + ix].y);
      }
      if (hipMalloc((void**)&m->d_gzb, zb.size() * 4) != hipSuccess || hipMemcpy(m->d_gzb, zb.data(), zb.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        g_mesh_err = "device allocation / upload of the ray lattice failed"; lg_mesh_destroy(m); return nullptr;
      }
    }
    // evenly spaced boundaries: the cell table of the closest-point queries (LG_LATTICE_CP=0: none -- contact queries walk the tree; the A/B switch and the tests' checker)
    const char* lc = getenv("LG_LATTICE_CP");
    const float hx = (G.xb.back() - G.xb.front()) / (float)m->gnx, hy = (G.yb.back() - G.yb.front()) / (float)m->gny;
    bool even = !(lc && lc[0] == '0') && hx > 0.f && hy > 0.f;
    for (int i = 0; even && i <= m->gnx; ++i) even = std::fabs(G.xb[i] - (G.xb[0] + (float)i * hx)) <= 0.5f * LATTICE_TOL * hx;
    for (int i = 0; even && i <= m->gny; ++i) even = std::fabs(G.yb[i] - (G.yb[0] + (float)i * hy)) <= 0.5f * LATTICE_TOL * hy;
    if (even) {
      // the faces of a cell by height (the ray walk does not care about their order), split at the largest gap between the faces below and the face above
      std::vector<float4> cz(G.cells.size()); std::vector<uint2> cr(G.cells.size());
      struct Face { float lo, hi; float4 v[3]; };
      std::vector<Face> fs;
      bool ok = true;
      for (size_t c = 0; c < cz.size() && ok; ++c) {
        const int first = G.cells[c].x, cnt = G.cells[c].y;
        ok = cnt <= 255 && G.tris.size() / 3 < (1u << 25) - 1u;     // (closest_point_lattice_pair: a group fits an empty table; 25-bit face numbers)
        fs.resize((size_t)cnt);
        for (int i = 0; i < cnt; ++i) {
          Face& f = fs[(size_t)i];
          for (int v = 0; v < 3; ++v) f.v[v] = G.tris[3 * (size_t)(first + i) + v];
          f.lo = std::min({f.v[0].z, f.v[1].z, f.v[2].z}); f.hi = std::max({f.v[0].z, f.v[1].z, f.v[2].z});
        }
        std::stable_sort(fs.begin(), fs.end(), [](const Face& x, const Face& y) { return x.lo < y.lo; });
        int split = cnt; float gap = 0.f, top = -1e30f;
        for (int i = 0; i < cnt; ++i) {
          if (i > 0 && fs[(size_t)i].lo - top > gap) { gap = fs[(size_t)i].lo - top; split = i; }
          top = std::max(top, fs[(size_t)i].hi);
        }
        float z[4] = {1e30f, -1e30f, 1e30f, -1e30f};
        for (int i = 0; i < cnt; ++i) {
          const int g = i < split ? 0 : 1;
          z[2 * g] = std::min(z[2 * g], fs[(size_t)i].lo); z[2 * g + 1] = std::max(z[2 * g + 1], fs[(size_t)i].hi);
          for (int v = 0; v < 3; ++v) G.tris[3 * (size_t)(first + i) + v] = fs[(size_t)i].v[v];
        }
        cz[c] = make_float4(z[0], z[1], z[2], z[3]);
        cr[c] = make_uint2((uint32_t)first, (uint32_t)split | ((uint32_t)(cnt - split) << 16));
        m->gmaxrun = std::max(m->gmaxrun, cnt);
      }
      if (ok) {
        if (hipMalloc((void**)&m->d_gcz, cz.size() * sizeof(float4)) != hipSuccess || hipMalloc((void**)&m->d_gcr, cr.size() * sizeof(uint2)) != hipSuccess ||
            hipMemcpy(m->d_gcz, cz.data(), cz.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->d_gcr, cr.data(), cr.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->d_gtris, G.tris.data(), G.tris.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
          g_mesh_err = "device allocation / upload of the closest-point cell table failed"; lg_mesh_destroy(m); return nullptr;
        }
      }
      m->gx0 = G.xb[0]; m->gy0 = G.yb[0]; m->ghx = hx; m->ghy = hy;
    }
  }
  return m;
}


int lg_mesh_info(lg_mesh* m, int64_t out[2]) { if (!m) return LG_ERR_INVALID; out[0] = m->n_tris; out[1] = m->n_nodes; return LG_OK; }
int lg_mesh_contact_lattice(lg_mesh* m, int32_t out[2]) { if (!m || !out) return LG_ERR_INVALID; out[0] = m->d_gcz ? m->gnx : 0; out[1] = m->d_gcz ? m->gny : 0; return LG_OK; }
int lg_mesh_ray_lattice(lg_mesh* m, int32_t out[2]) { if (!m || !out) return LG_ERR_INVALID; out[0] = m->d_gcells ? m->gnx : 0; out[1] = m->d_gcells ? m->gny : 0; return LG_OK; }

#define MESH_TRY(m, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (m)->err = std::string(#expr) + ": " + hipGetErrorString(_e); return LG_ERR_HIP; } } while (0)

int lg_raycast_mesh(lg_mesh* m, const float* origins, const float* dirs, int64_t n_rays, float max_dist, float* hits, uint8_t* found, void* stream) {
  if (!m || !origins || !dirs || !hits || !found || n_rays < 0) return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  if (n_rays == 0) return LG_OK;
  MeshView M{m->d_nodes, m->d_tris};
  if (m->d_gcells) hipLaunchKernelGGL(raycast_kernel<true>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), ray_grid_lds(m), (hipStream_t)stream, M, ray_grid_of(m), origins, dirs, n_rays, max_dist, hits, found);
  else hipLaunchKernelGGL(raycast_kernel<false>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, ray_grid_of(m), origins, dirs, n_rays, max_dist, hits, found);
  MESH_TRY(m, hipGetLastError());
  return LG_OK;
}

int lg_mesh_query_sdf(lg_mesh* m, const float* points, int64_t n, float max_dist, float* sdf, float* grad, void* stream) {
  if (!m || !points || !sdf || !grad || n < 0) return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  if (n == 0) return LG_OK;
  MeshView M{m->d_nodes, m->d_tris};
  hipLaunchKernelGGL(sdf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, points, n, max_dist, sdf, grad);
  MESH_TRY(m, hipGetLastError());
  return LG_OK;
}

int lg_raycaster_update_subset(lg_mesh* m, const float* root_states, const float* ray_origins, const float* ray_dirs, int32_t num_rays,
                               float max_dist, int32_t attach_yaw_only, const int32_t* env_ids, int32_t n, float* ray_hits,
                               uint8_t* hits_found, float* raycast_distances, int32_t distance_stride, void* stream) {
  if (!m || !root_states || !ray_origins || !ray_dirs || !ray_hits || !hits_found || !raycast_distances || n <= 0 || num_rays <= 0 ||
      distance_stride < num_rays)
    return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  MeshView M{m->d_nodes, m->d_tris};
  int64_t tot = (int64_t)n * num_rays;
  if (m->d_gcells)
    hipLaunchKernelGGL(raycaster_kernel<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), ray_grid_lds(m), (hipStream_t)stream, M, ray_grid_of(m), root_states, ray_origins,
                       ray_dirs, env_ids, n, num_rays, max_dist, attach_yaw_only, ray_hits, hits_found, raycast_distances, distance_stride);
  else
    hipLaunchKernelGGL(raycaster_kernel<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, ray_grid_of(m), root_states, ray_origins,
                       ray_dirs, env_ids, n, num_rays, max_dist, attach_yaw_only, ray_hits, hits_found, raycast_distances, distance_stride);
  MESH_TRY(m, hipGetLastError());
  return LG_OK;
}

int lg_raycaster_update(lg_mesh* m, const float* root_states, const float* ray_origins, const float* ray_dirs, int32_t num_envs,
                        int32_t num_rays, float max_dist, int32_t attach_yaw_only, float* ray_hits, uint8_t* hits_found,
                        float* raycast_distances, void* stream) {
  return lg_raycaster_update_subset(m, root_states, ray_origins, ray_dirs, num_rays, max_dist, attach_yaw_only, nullptr, num_envs,
                                    ray_hits, hits_found, raycast_distances, num_rays, stream);
}

int lg_sdf_bodies_update(lg_mesh* m, const float* rigid_body_state, int32_t num_bodies, const int32_t* body_indices,
                         const float* sphere_offsets, int32_t num_query_bodies, const int32_t* env_ids, int32_t n, float max_dist,
                         float* sdf_values, int32_t sdf_stride, float* sdf_gradients, float* nearest_points, void* stream) {
  if (!m || !rigid_body_state || !body_indices || !sdf_values || num_bodies <= 0 || num_query_bodies <= 0 || n <= 0 ||
      sdf_stride < num_query_bodies)
    return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  MeshView M{m->d_nodes, m->d_tris};
  int64_t tot = (int64_t)n * num_query_bodies;
  if (tot > m->sdf_cache_n) {                      // (first call, or a larger query set: the cache starts empty)
    if (m->d_sdf_cache) (void)hipFree(m->d_sdf_cache);
    m->d_sdf_cache = nullptr; m->sdf_cache_n = 0;
    MESH_TRY(m, hipMalloc((void**)&m->d_sdf_cache, (size_t)tot * sizeof(float4)));
    m->sdf_cache_n = tot;
    MESH_TRY(m, hipMemsetAsync(m->d_sdf_cache, 0, (size_t)tot * sizeof(float4), (hipStream_t)stream));
  }
  const char* sl = getenv("LG_SDF_LATTICE");
  LatticeView L{nullptr, nullptr, nullptr, 0, 0, 0.f, 0.f, 1.f, 1.f, LATP_CAP};
  if (m->d_gcz && m->d_gcr && !(sl && sl[0] == '0')) L = LatticeView{m->d_gcz, m->d_gcr, m->d_gtris, m->gnx, m->gny, m->gx0, m->gy0, m->ghx, m->ghy, LATP_CAP};
  // (round 6, measured on config 3, three rounds: 256 threads + one env's five bodies side by side 0.3435 ms per step; 128 threads + the same body of eight envs per
  //  workgroup 0.3373: a wave no longer waits for the one trunk among its four queries.  LG_SDF_BLOCK / LG_SDF_ORDER: A/B)
  int sdf_block = 128, sdf_order = 1;
  if (const char* ev = getenv("LG_SDF_BLOCK")) { const int v = atoi(ev); if (v == 64 || v == 128 || v == 256) sdf_block = v; }
  if (const char* ev = getenv("LG_SDF_ORDER")) sdf_order = atoi(ev) != 0;
  const int qpb = sdf_block / 16;
  hipLaunchKernelGGL(sdf_bodies_kernel, dim3((unsigned)((tot + qpb - 1) / qpb)), dim3(sdf_block), 0, (hipStream_t)stream, M, L, rigid_body_state, num_bodies,
                     body_indices, sphere_offsets, num_query_bodies, env_ids, n, max_dist, sdf_values, sdf_stride, sdf_gradients, nearest_points, m->d_sdf_cache, sdf_order);
  MESH_TRY(m, hipGetLastError());
  return LG_OK;
}

int lg_depth_camera_update(lg_mesh* m, const lg_depth_params* p, const float* root_states, const float* ray_dirs, const int64_t* episode_length_buf,
                           int32_t num_envs, const float* env_noise, float* camera_pos, float* camera_rot, float* depth_buffer, void* stream) {
  if (!m || !p || !root_states || !ray_dirs || !episode_length_buf || !camera_pos || !camera_rot || !depth_buffer || num_envs <= 0)
    return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  if (p->width <= 0 || p->height <= 0 || p->resized_width <= 0 || p->resized_height <= 0 || p->buffer_len <= 0) return LG_ERR_INVALID;
  size_t lds = (size_t)p->width * p->height * sizeof(float);
  if (lds > 64 * 1024) { m->err = "depth image too large for the LDS-staged resize"; return LG_ERR_UNSUPPORTED; }
  // the lattice instance keeps its boundary tables in LDS next to the image; together they must stay within the 64 KB a launch gets without opting in
  const char* rs_ = getenv("LG_RAY_SKIP"); const int skip = rs_ ? atoi(rs_) : 1;      // (A/B switch; 0: every ray walks from the camera, 2: also the coarse walk over blocks -- measured: +-1 %)
  int mode = 0;
  const size_t rs_lds = (size_t)5 * (p->resized_width + p->resized_height) * sizeof(float);      // the resize's per-column / per-row weights (depth_kernel)
  if (m->d_gcells && m->gnx >= 1 && m->gny >= 1 && lds + ray_grid_lds(m) + rs_lds <= 64 * 1024) { mode = 1; lds += ray_grid_lds(m); }
  if (lds + rs_lds > 64 * 1024) { m->err = "depth image too large for the LDS-staged resize"; return LG_ERR_UNSUPPORTED; }
  const int rs_off = (int)(lds / sizeof(float));
  lds += rs_lds;
  MeshView M{m->d_nodes, m->d_tris};
  // pixel tile of a wave: the TW x TH <= 64 that covers the image with the fewest tiles, the squarest of those
  int TW = 8, TH = 8, best_tiles = 1 << 30;
  for (int th = 1; th <= 64; ++th)
    for (int tw = 1; tw * th <= 64; ++tw) {
      const int nt = ((p->width + tw - 1) / tw) * ((p->height + th - 1) / th);
      if (nt < best_tiles || (nt == best_tiles && abs(tw - th) < abs(TW - TH))) { best_tiles = nt; TW = tw; TH = th; }
    }
#define LG_DEPTH_LAUNCH(MODE) hipLaunchKernelGGL(depth_kernel<MODE>, dim3(num_envs), dim3(256), lds, (hipStream_t)stream, M, ray_grid_of(m), root_states, ray_dirs, \
                     episode_length_buf, p->width, p->height, TW, TH, p->resized_width, p->resized_height, p->buffer_len, p->near_clip, p->far_clip, \
                     p->position[0], p->position[1], p->position[2], p->quat_offset[0], p->quat_offset[1], p->quat_offset[2], \
                     p->quat_offset[3], env_noise, camera_pos, camera_rot, depth_buffer, skip, rs_off)
  if (mode == 1) LG_DEPTH_LAUNCH(1); else LG_DEPTH_LAUNCH(0);
#undef LG_DEPTH_LAUNCH
  MESH_TRY(m, hipGetLastError());
  return LG_OK;
}

}  // extern "C"

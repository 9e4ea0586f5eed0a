// lg_bvh.h — BVH node / triangle layout and the two device traversals (closest ray hit, closest surface point) shared by
// the mesh-query kernels (lg_mesh.hip) and the triangle-mesh contact detection of the physics kernel (lg_step.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "lg_device.h"

// Host-side binary node (builder only) ...
struct BvhNode {          // 32 B
  float bmin[3]; int32_t left_first;   // inner: index of left child (right = left + 1); leaf: first triangle
  float bmax[3]; int32_t count;        // 0 = inner node, > 0 = number of triangles in the leaf
};

// ... collapsed for the device into 4-wide nodes that carry their CHILDREN's boxes: one 128-byte fetch per visited
// node decides all four children, so a traversal pays one dependent memory round trip per level of a tree half as
// deep (the binary layout paid two: the node, then its two children).
//   child[k] >= 0            inner node index
//   child[k] == BVH4_EMPTY   no child
//   otherwise                leaf: ~child[k] = (first_triangle << 3) | (count - 1), count <= 8
#define BVH4_EMPTY ((int32_t)0x80000000)
struct BvhNode4 {         // 128 B, structure of arrays over the four children
  float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4];
  int32_t child[4];
  int32_t pad[4];
};

struct lg_mesh {
  int device = 0;
  int64_t n_tris = 0, n_nodes = 0;
  BvhNode4* d_nodes = nullptr;
  float4* d_tris = nullptr;            // 3 float4 per triangle: v0, v1, v2 (w unused)
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};   // bounding box of the mesh
  // ray lattice (rays only; built by lg_mesh_create when the vertices sit on a rectilinear lattice in x and y, i.e. a heightfield-derived mesh)
  float* d_gxb = nullptr; float* d_gyb = nullptr; int gnx = 0, gny = 0;   // cell boundaries: gnx + 1 and gny + 1 ascending coordinates
  float2* d_gzr = nullptr;             // per cell (iy * gnx + ix): min z, max z of the triangles listed in it
  float* d_gzb = nullptr; int gnbx = 0, gnby = 0;   // per block of RAY_BLK x RAY_BLK cells: the highest z of a triangle listed in one of its cells
  int2* d_gcells = nullptr;            // per cell: first triangle of its run in d_gtris, count
  float4* d_gtris = nullptr;           // triangles in cell order (a triangle that overlaps k cells is stored k times)
  // closest-point queries over the same cells (closest_point_lattice, lg_physics.h): present when the boundaries are evenly spaced (a heightfield-derived mesh
  // read back from an OBJ file), so that the cells a ball reaches are an index range computed without the boundary tables
  // A cell's faces are sorted by height and split at their largest gap into a lower and an upper group (floor and ceiling of a two-layer terrain, the
  // foot and the top of a wall): a ball between the layers is within reach of a cell's whole z range and of neither group's.
  float4* d_gcz = nullptr;             // per cell: min z, max z of the lower group, min z, max z of the upper group (an empty group: +1e30, -1e30)
  uint2* d_gcr = nullptr;              // per cell: first triangle of its run in d_gtris, (faces in the lower group) | (faces in the upper group) << 16
  float gx0 = 0.f, gy0 = 0.f, ghx = 0.f, ghy = 0.f;   // boundary i of an axis = g?0 + i * gh? to within LATTICE_TOL * gh?
  int gmaxrun = 0;                     // most faces listed in one cell (the least a query table must hold: closest_point_lattice_pair)
  float4* d_sdf_cache = nullptr;       // lg_sdf_bodies_update: last closest surface point per query slot (xyz, w = 1 when set)
  int64_t sdf_cache_n = 0;
  std::string err;
};

// ------------------------------------------------------------------------------------------------ device: traversal
struct MeshView { const BvhNode4* __restrict__ nodes; const float4* __restrict__ tris; };
#define LATTICE_TOL 1e-3f
struct LatticeView { const float4* __restrict__ cell; const uint2* __restrict__ run; const float4* __restrict__ tris; int nx, ny; float x0, y0, hx, hy;
                     int cap; /* entries of a wave's query table (<= LATP_CAP; LG_LATTICE_CAP shrinks it: the tests' way to the refill path) */ };
#define LATP_PER 13
#define LATP_CAP (128 * LATP_PER)
#define BVH_STACK 40      // <= 3 pushes per level of a 4-wide tree

struct Node4Regs { float4 minx, miny, minz, maxx, maxy, maxz; int4 child; };
LG_DEV Node4Regs load_node4(const BvhNode4* __restrict__ nodes, int idx) {
  const float4* p = (const float4*)(nodes + idx);
  Node4Regs n;
  n.minx = p[0]; n.miny = p[1]; n.minz = p[2]; n.maxx = p[3]; n.maxy = p[4]; n.maxz = p[5];
  n.child = *(const int4*)(p + 6);
  return n;
}
#define F4(v, k) ((k) == 0 ? (v).x : (k) == 1 ? (v).y : (k) == 2 ? (v).z : (v).w)

// Nearest of up to four candidates (inner node indices or leaf codes; BVH4_EMPTY = none) becomes `cur`; the others go
// on the stack farthest first, so they pop nearest first.  Leaves travel through the stack like inner nodes: strict
// nearest-first order is what keeps the search radius (or the ray interval) shrinking as early as possible.
LG_DEV bool descend4(int cand[4], float key[4], int* stack_i, float* stack_k, int& sp, int& cur) {
  int nvalid = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) nvalid += cand[k] != BVH4_EMPTY ? 1 : 0;
  if (nvalid == 0) return false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {                 // push the largest key while more than one candidate is left
    if (nvalid > 1) {
      int bk = -1, bi = BVH4_EMPTY; float bv = -1.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (cand[k] != BVH4_EMPTY && key[k] > bv) { bv = key[k]; bk = k; bi = cand[k]; }
      if (sp < BVH_STACK) { stack_i[sp] = bi; stack_k[sp] = bv; ++sp; }
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == bk) cand[k] = BVH4_EMPTY;
      --nvalid;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (cand[k] != BVH4_EMPTY) cur = cand[k];
  return true;
}

// A ray that runs within rounding of an edge two triangles share must hit one of them: the barycentric tests accept a band of RAY_EDGE_EPS (the rounding of
// u, v for origins metres away from a decimetre triangle is ~2e-6; strict tests let 9 % of the vertical rays that start exactly on a lattice line of a
// heightfield mesh fall through the crack).  The plane of either neighbour gives the same t to rounding.  The oracle's brute-force scan uses the same band.
#define RAY_EDGE_EPS 1e-5f

// One triangle against a ray (Moller-Trumbore, two-sided, barycentric band RAY_EDGE_EPS): updates the closest hit.  ONE body for the tree walk and the lattice
// walk, compiled without floating-point contraction: with the default contraction a product and a sum fuse or not depending on the code AROUND the
// expression, and "the same arithmetic" in two walks gave hits that differed in the last bit once one of the loops was unrolled.
LG_DEV void ray_triangle(V3 o, V3 d, float4 a, float4 b, float4 cc, float& best, bool& hit) {
#pragma clang fp contract(off)
  const V3 v0 = v3(a.x, a.y, a.z), e1 = v3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = v3(cc.x - a.x, cc.y - a.y, cc.z - a.z);
  const V3 p = v3(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
  const float det = (e1.x * p.x + e1.y * p.y) + e1.z * p.z;
  if (fabsf(det) < 1e-20f) return;
  const float idet = 1.f / det;
  const V3 s = v3(o.x - v0.x, o.y - v0.y, o.z - v0.z);
  const float u = ((s.x * p.x + s.y * p.y) + s.z * p.z) * idet;
  if (u < -RAY_EDGE_EPS || u > 1.f + RAY_EDGE_EPS) return;
  const V3 q = v3(s.y * e1.z - s.z * e1.y, s.z * e1.x - s.x * e1.z, s.x * e1.y - s.y * e1.x);
  const float v = ((d.x * q.x + d.y * q.y) + d.z * q.z) * idet;
  if (v < -RAY_EDGE_EPS || u + v > 1.f + RAY_EDGE_EPS) return;
  const float t = ((e2.x * q.x + e2.y * q.y) + e2.z * q.z) * idet;
  if (t >= 0.f && t <= best) { best = t; hit = true; }
}

// closest two-sided hit with 0 <= t <= max_dist (Moller-Trumbore); returns t or -1
LG_DEV float trace_ray(const MeshView& M, V3 o, V3 d, float max_dist) {
  const V3 inv = v3(1.f / (fabsf(d.x) > 1e-12f ? d.x : copysignf(1e-12f, d.x)), 1.f / (fabsf(d.y) > 1e-12f ? d.y : copysignf(1e-12f, d.y)),
                    1.f / (fabsf(d.z) > 1e-12f ? d.z : copysignf(1e-12f, d.z)));
  float best = max_dist; bool hit = false;
  int stack_i[BVH_STACK]; float stack_k[BVH_STACK]; int sp = 0;
  int cur = 0;
  // "while-while" form (see closest_point_pair): inner nodes until every lane of the wave stands on a leaf, then the leaf faces together
  bool done = false;
  auto pop = [&]() -> bool {
    while (sp > 0) { --sp; if (stack_k[sp] <= best) { cur = stack_i[sp]; return true; } }
    return false;
  };
  while (!done) {
    while (cur >= 0) {
      const Node4Regs n = load_node4(M.nodes, cur);
      int cand[4]; float key[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = F4(n.child, k);
        float tx1 = (F4(n.minx, k) - o.x) * inv.x, tx2 = (F4(n.maxx, k) - o.x) * inv.x;
        float ty1 = (F4(n.miny, k) - o.y) * inv.y, ty2 = (F4(n.maxy, k) - o.y) * inv.y;
        float tz1 = (F4(n.minz, k) - o.z) * inv.z, tz2 = (F4(n.maxz, k) - o.z) * inv.z;
        float tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), 0.f));
        float tmx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), best));
        key[k] = tmin;
        cand[k] = (c != BVH4_EMPTY && tmin <= tmx) ? c : BVH4_EMPTY;
      }
      if (!descend4(cand, key, stack_i, stack_k, sp, cur) && !pop()) { done = true; break; }
    }
    if (done) break;
    {
      const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
      for (int i = 0; i < cnt; ++i) {
        const float4* T = M.tris + (size_t)(first + i) * 3;
        ray_triangle(o, d, T[0], T[1], T[2], best, hit);
      }
    }
    if (!pop()) done = true;
  }
  return hit ? best : -1.f;
}

// ---- ray lattice: closest hit by walking the cells of a rectilinear xy lattice under the ray (heightfield-derived meshes)
// A tree walk costs a ray ~15 node visits of seven 16-byte loads and ~110 instructions each; the vertices of a heightfield mesh sit on a lattice
// (also the ones the slope correction moved: they land on the neighbouring lattice line), so "which triangles can the ray meet while it is over
// cell (ix, iy)" is a table: one 16-byte record per cell (run of triangles whose xy extent overlaps the cell, their z range).  The walk visits
// the cells in the order the ray crosses them, skips a cell whose z range the ray does not reach while it is over it, and tests the listed
// triangles with the SAME arithmetic as trace_ray, so a hit has the same t, bit for bit.  It stops one cell AFTER the cell a hit was found in:
// the cell the walk believes the ray is in and the cell the hit point really lies in can differ by rounding at the line between two cells.
struct RayGrid { const float* __restrict__ xb; const float* __restrict__ yb; int nx, ny;
                 const float2* __restrict__ zr;    // per cell (iy * nx + ix): min z, max z of the triangles listed in it (empty: +1e30, -1e30) -- all the walk reads per cell
                 const int2* __restrict__ run;     // per cell: first triangle of its run in `tris`, count -- read only for cells whose z range the ray reaches
                                                   // (measured and dropped, round 6: both in ONE 16-byte record and a run's triangles fetched four at a time: 0.627 against 0.589 ms per step of config 4)
                 const float4* __restrict__ tris;
                 const float* __restrict__ zb; int nbx, nby; };   // per block of RAY_BLK x RAY_BLK cells: max z (a camera's rays start where they come down to the highest block in reach)
#define RAY_BLK 8
// How the walk reads the boundaries: the tables of the whole lattice in LDS (x boundaries (nx + 1) followed by y boundaries (ny + 1)).  ztop / blocks: what the
// depth camera knows about its rays (they start at the camera and end within far_clip of it): no triangle in reach is higher than ztop, and the block
// maxima may be walked in front of the cells.  (Measured and dropped, round 6: the cells' z ranges of the camera's square staged in LDS -- 18 KB and three
// barriers per workgroup cost more than the L1-resident loads they replaced: 0.65 against 0.59 ms per step of config 4.)
struct RayTables {
  const float* tb; const RayGrid* G; float ztop; bool blocks;
  LG_DEV float xb(int i) const { return tb[i]; }
  LG_DEV float yb(int i) const { return tb[G->nx + 1 + i]; }
  LG_DEV float x_lo() const { return tb[0]; }
  LG_DEV float x_hi() const { return tb[G->nx]; }
  LG_DEV float y_lo() const { return tb[G->nx + 1]; }
  LG_DEV float y_hi() const { return tb[G->nx + 1 + G->ny]; }
  LG_DEV float z_top() const { return ztop; }       // no triangle the ray can reach is higher than this (3e38: unknown)
  LG_DEV bool coarse() const { return blocks; }     // walk the blocks in front of the cells
};
template <class Acc>
LG_DEV int raygrid_locate(const Acc& A, bool xaxis, int n, float x, int guess) {
  int i = guess < 0 ? 0 : (guess > n - 1 ? n - 1 : guess);
  while (i > 0 && x < (xaxis ? A.xb(i) : A.yb(i))) --i;
  while (i < n - 1 && x >= (xaxis ? A.xb(i + 1) : A.yb(i + 1))) ++i;
  return i;
}
// The stretch [t0, t1] of a ray its walk has to cover: [0, max_dist] cut to the part over the lattice and, where the caller knows a ceiling of the reachable
// triangles (A.z_top()), to the part below that ceiling.  False: nothing to walk -- a miss.  (Its own function since round 6: the depth kernel sorts its rays
// by this answer before it walks any.)
struct RaySpan { V3 inv; bool mvx, mvy; float t0, t1; };
template <class Acc>
LG_DEV bool ray_span(const Acc& A, V3 o, V3 d, float max_dist, RaySpan& S) {
  S.inv = v3(1.f / (fabsf(d.x) > 1e-12f ? d.x : copysignf(1e-12f, d.x)), 1.f / (fabsf(d.y) > 1e-12f ? d.y : copysignf(1e-12f, d.y)),
             1.f / (fabsf(d.z) > 1e-12f ? d.z : copysignf(1e-12f, d.z)));
  const V3 inv = S.inv;
  // a ray that does not move along an axis (|d| <= 1e-12: a vertical ray) never crosses that axis's lines: its boundary times are -inf / +inf while it
  // is between the outer lines (those included), not the 0 * 1e12 a start exactly ON a line gives
  const bool mvx = fabsf(d.x) > 1e-12f, mvy = fabsf(d.y) > 1e-12f;
  S.mvx = mvx; S.mvy = mvy;
  const float xlo = A.x_lo(), xhi = A.x_hi(), ylo = A.y_lo(), yhi = A.y_hi();
  // the part of the ray over the lattice
  float t0 = 0.f, t1 = max_dist;
  {
    const bool inx = o.x >= xlo && o.x <= xhi, iny = o.y >= ylo && o.y <= yhi;
    const float ax = mvx ? (xlo - o.x) * inv.x : (inx ? -3.0e38f : 3.0e38f), bx = mvx ? (xhi - o.x) * inv.x : (inx ? 3.0e38f : -3.0e38f);
    const float ay = mvy ? (ylo - o.y) * inv.y : (iny ? -3.0e38f : 3.0e38f), by = mvy ? (yhi - o.y) * inv.y : (iny ? 3.0e38f : -3.0e38f);
    t0 = fmaxf(t0, fmaxf(fminf(ax, bx), fminf(ay, by)));
    t1 = fminf(t1, fminf(fmaxf(ax, bx), fmaxf(ay, by)));
  }
  {
    // while the ray is above every triangle it can reach, no cell is a candidate: the walk starts where the ray comes down to that height (a ray that never
    // does misses) -- the cells in front of that point would all have been skipped one by one
    const float ztop = A.z_top();
    if (ztop < 3.0e37f && o.z > ztop) {
      const float pad = 2e-5f * fabsf(t1) + 1e-6f;
      const float ts = d.z < 0.f ? (ztop - o.z) * inv.z - pad * (1.f + fabsf(inv.z)) - 1e-4f : 3.0e38f;      // o.z + t d.z <= ztop from ts on (taken early by the walk's own padding and more)
      t0 = fmaxf(t0, ts);
    }
  }
  S.t0 = t0; S.t1 = t1;
  return t0 <= t1;
}
template <class Acc>
LG_DEV float trace_ray_grid(const RayGrid& G, const Acc& A, V3 o, V3 d, float max_dist) {
  RaySpan S;
  if (!ray_span(A, o, d, max_dist, S)) return -1.f;
  const V3 inv = S.inv; const bool mvx = S.mvx, mvy = S.mvy;
  float t0 = S.t0; const float t1 = S.t1;
  const float xlo = A.x_lo(), xhi = A.x_hi(), ylo = A.y_lo(), yhi = A.y_hi();
  const float px = o.x + t0 * d.x, py = o.y + t0 * d.y;
  const float ux = (float)G.nx / (xhi - xlo), uy = (float)G.ny / (yhi - ylo);
  int ix = raygrid_locate(A, true, G.nx, px, (int)((px - xlo) * ux)), iy = raygrid_locate(A, false, G.ny, py, (int)((py - ylo) * uy));
  const int sx = inv.x >= 0.f ? 1 : -1, sy = inv.y >= 0.f ? 1 : -1;
  const int ox = sx > 0 ? 1 : 0, oy = sy > 0 ? 1 : 0;     // the boundary AHEAD of cell ix / iy is boundary ix + ox / iy + oy
  // (the boundary AHEAD of a cell on such an axis: +inf as well -- the cell was left before it was looked at: a miss where the tree walk hits)
  // ... and a ray without motion across the lines of an axis that starts exactly ON one of them runs down the border of two columns of cells: triangles
  // of the - side column reach it only with an edge (they are not listed in the + side cells), and where the slope correction folded the surface the
  // nearest hit may be one of theirs.  Such rays (a measure-zero set; tested wave-wide, so other waves pay three ballots) walk the - side column(s) too.
  bool lx = !mvx && ix > 0 && px == A.xb(ix), ly = !mvy && iy > 0 && py == A.yb(iy);
  if (A.coarse() && A.z_top() < 3.0e37f && G.zb != nullptr && !lx && !ly) {
    // (round 6; callers that know a ceiling of the reachable triangles -- the depth camera -- also get the coarse walk)  Blocks of RAY_BLK x RAY_BLK cells
    // with the highest z of their triangles: the ray crosses the blocks in front of it one by one until it comes down to a block's top; the cell walk
    // starts where it enters THAT block (eight cells further on per step; a ray that stays above every block it crosses misses without a cell walk).
    // Conservative like the cell walk's own z test (same padding), so the cells left out would all have been skipped: the hit is the same.
    const float padb = 2e-5f * fabsf(t1) + 1e-6f;
    int bx = ix / RAY_BLK, by = iy / RAY_BLK;
    auto edge_x = [&](int b) { return mvx ? (A.xb(min((b + ox) * RAY_BLK, G.nx)) - o.x) * inv.x : 3.0e38f; };
    auto edge_y = [&](int b) { return mvy ? (A.yb(min((b + oy) * RAY_BLK, G.ny)) - o.y) * inv.y : 3.0e38f; };
    float tbx = edge_x(bx), tby = edge_y(by), tc = t0;
    bool reach = false;
    while (true) {
      const float tn = fminf(tbx, tby);
      const float top = G.zb[(size_t)by * G.nbx + bx];
      const float za = o.z + (tc - padb) * d.z, zb_ = o.z + (fminf(tn, t1) + padb) * d.z;
      if (fminf(za, zb_) <= top) { reach = true; break; }
      if (!(tn < t1)) break;
      const bool stx = tbx <= tby;
      if (stx) { bx += sx; if (bx < 0 || bx >= G.nbx) break; tbx = edge_x(bx); }
      else { by += sy; if (by < 0 || by >= G.nby) break; tby = edge_y(by); }
      tc = tn;
    }
    if (!reach) return -1.f;
    if (tc > t0) {
      t0 = tc;
      const float qx = o.x + t0 * d.x, qy = o.y + t0 * d.y;
      ix = raygrid_locate(A, true, G.nx, qx, (int)((qx - xlo) * ux)); iy = raygrid_locate(A, false, G.ny, qy, (int)((qy - ylo) * uy));
    }
  }
  const int ix0 = ix, iy0 = iy;
  float best = max_dist; bool hit = false;
  for (int var = 0; var < 4; ++var) {
  const bool use = var == 0 || (((var & 1) == 0 || lx) && ((var & 2) == 0 || ly));
  if (var > 0 && __ballot(use) == 0ull) continue;
  // (a lane that does not take part keeps its own cell: ix0 - 1 would be -1 for a lane in the first column -- an index below the tables)
  ix = use ? ix0 - (var & 1) : ix0; iy = use ? iy0 - (var >> 1) : iy0;
  float tmx = mvx ? (A.xb(ix + ox) - o.x) * inv.x : 3.0e38f, tmy = mvy ? (A.yb(iy + oy) - o.y) * inv.y : 3.0e38f;
  float tcur = t0;
  // the ends of a cell's stretch of the ray are rounded: the z range is taken a little beyond either end
  const float padc = 2e-5f * fabsf(t1) + 1e-6f;
  bool done = !use;
  // one cell on, without branches: the axis whose boundary comes first, the boundary ahead of the new cell
  auto advance = [&]() {
    const bool stx = tmx <= tmy;
    const int inew = stx ? ix + sx : iy + sy;
    const bool out = inew < 0 || inew >= (stx ? G.nx : G.ny);
    ix = stx ? inew : ix; iy = stx ? iy : inew;
    const int ib = out ? 0 : inew + (stx ? ox : oy);
    const float bnd = stx ? A.xb(ib) : A.yb(ib);
    const float tn = (stx ? mvx : mvy) ? (bnd - (stx ? o.x : o.y)) * (stx ? inv.x : inv.y) : 3.0e38f;
    tmx = stx ? tn : tmx; tmy = stx ? tmy : tn;
    done = done || out;
  };
  // "while-while", like the tree walk: every lane skips cells until it stands on one whose z range it reaches (or its ray is over), THEN the wave
  // tests triangles together -- with the test inside the walk, nearly every step of the wave had some lane in the (long, load-bound) triangle loop
  while (!done) {
    bool cand = false;
    float tnext = 0.f;
    while (!done && !cand) {
      tnext = fminf(tmx, tmy);
      const float2 z = G.zr[(size_t)iy * G.nx + ix];
      const float za = o.z + (tcur - padc) * d.z, zb = o.z + (fminf(tnext, t1) + padc) * d.z;
      cand = !(fminf(za, zb) > z.y || fmaxf(za, zb) < z.x);
      if (!cand) {
        // found before this cell began: every cell the hit point can lie in has been seen; or the ray ends / leaves the lattice in this cell
        if ((hit && best <= tcur) || !(tnext < t1)) done = true; else advance();
        tcur = tnext;
      }
    }
    if (cand) {
      const int2 c = G.run[(size_t)iy * G.nx + ix];
      for (int i = 0; i < c.y; ++i) {
        const float4* T = G.tris + (size_t)(c.x + i) * 3;
        ray_triangle(o, d, T[0], T[1], T[2], best, hit);
      }
      if ((hit && best <= tcur) || !(tnext < t1)) done = true; else advance();
      tcur = tnext;
    }
  }
  }
  return hit ? best : -1.f;
}
// tb: the boundary tables in LDS, x boundaries (nx + 1) followed by y boundaries (ny + 1)
LG_DEV float trace_ray_grid_tb(const RayGrid& G, const float* tb, V3 o, V3 d, float max_dist, float ztop = 3.0e38f, bool blocks = false) { return trace_ray_grid(G, RayTables{tb, &G, ztop, blocks}, o, d, max_dist); }

// closest point on triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5)
LG_DEV V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
  V3 ab = b - a, ac = c - a, ap = p - a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) return a;
  V3 bp = p - b; float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.f && d4 <= d3) return b;
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); return a + v * ab; }
  V3 cp = p - c; float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.f && d5 <= d6) return c;
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { float w = d2 / (d2 - d6); return a + w * ac; }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); return b + w * (c - b); }
  float denom = 1.f / (va + vb + vc);
  return a + (vb * denom) * ab + (vc * denom) * ac;
}

// squared distance from p to the bounding box of triangle (a, b, c): a lower bound of the distance to the triangle.  A leaf holds up to 8
// faces and every one of them used to get the full closest-point test (~150 instructions, twice in the pair traversal) although after the
// first hit most lie beyond the current best distance: the box test (20 instructions) rejects those.  Exact: a face whose box is farther
// than the acceptance limit holds no point within it.
LG_DEV float tri_box_dist2(V3 p, V3 a, V3 b, V3 c) {
  const float dx = fmaxf(fmaxf(fminf(fminf(a.x, b.x), c.x) - p.x, 0.f), p.x - fmaxf(fmaxf(a.x, b.x), c.x));
  const float dy = fmaxf(fmaxf(fminf(fminf(a.y, b.y), c.y) - p.y, 0.f), p.y - fmaxf(fmaxf(a.y, b.y), c.y));
  const float dz = fmaxf(fmaxf(fminf(fminf(a.z, b.z), c.z) - p.z, 0.f), p.z - fmaxf(fmaxf(a.z, b.z), c.z));
  return dx * dx + dy * dy + dz * dz;
}

// closest point within max_dist; outputs the point and the unit normal of the face that decides the sign.  When several
// faces are equally close (the closest feature is a shared edge or vertex) the face whose plane is farthest from the
// query point decides: that rule is independent of traversal order, so the BVH and a brute-force scan agree.
LG_DEV bool closest_point(const MeshView& M, V3 p, float max_dist, V3* cp_out, V3* fn_out) {
  float best2 = max_dist * max_dist; bool found = false; float bestabs = -1.f;
  V3 bestp = p, bestn = v3(0, 0, 1);
  int stack_i[BVH_STACK]; float stack_k[BVH_STACK]; int sp = 0; int cur = 0;
  // "while-while" form (see closest_point_pair)
  bool done = false;
  auto pop = [&]() -> bool {
    while (sp > 0) { --sp; if (stack_k[sp] <= best2 * (1.f + 1e-5f) + 1e-12f) { cur = stack_i[sp]; return true; } }
    return false;
  };
  while (!done) {
    while (cur >= 0) {
      const Node4Regs n = load_node4(M.nodes, cur);
      int cand[4]; float key[4];
      const float lim = best2 * (1.f + 1e-5f) + 1e-12f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = F4(n.child, k);
        float dx = fmaxf(fmaxf(F4(n.minx, k) - p.x, 0.f), p.x - F4(n.maxx, k));
        float dy = fmaxf(fmaxf(F4(n.miny, k) - p.y, 0.f), p.y - F4(n.maxy, k));
        float dz = fmaxf(fmaxf(F4(n.minz, k) - p.z, 0.f), p.z - F4(n.maxz, k));
        key[k] = dx * dx + dy * dy + dz * dz;
        cand[k] = (c != BVH4_EMPTY && key[k] <= lim) ? c : BVH4_EMPTY;
      }
      if (!descend4(cand, key, stack_i, stack_k, sp, cur) && !pop()) { done = true; break; }
    }
    if (done) break;
    {
      const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
      // The triangles of a leaf (up to 8) are fetched four at a time, every load issued before the first test: one memory round trip per
      // four faces instead of one per face (a lane tests its faces one after the other and the wave waits for its slowest lane).
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float4 ta[4], tb[4], tc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4* T = M.tris + (size_t)(first + min(i0 + u, cnt - 1)) * 3;
          ta[u] = T[0]; tb[u] = T[1]; tc[u] = T[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u >= cnt) continue;
          V3 a = v3(ta[u].x, ta[u].y, ta[u].z), b = v3(tb[u].x, tb[u].y, tb[u].z), cc = v3(tc[u].x, tc[u].y, tc[u].z);
          if (!(tri_box_dist2(p, a, b, cc) <= best2 * (1.f + 1e-5f) + 1e-12f)) continue;
          // zero-area faces (slope-corrected height-field meshes are full of them) are skipped: their points belong to
          // the edges of their neighbours, and the barycentric arithmetic below is 0/0 on them
          V3 fn = cross(b - a, cc - a); float fl = norm(fn);
          if (!(fl > 1e-10f)) continue;
          V3 q = closest_on_triangle(p, a, b, cc);
          V3 dq = p - q; float d2 = dot(dq, dq);
          if (!(d2 <= best2 * (1.f + 1e-5f) + 1e-12f)) continue;
          const bool strictly = !found || d2 < best2 * (1.f - 1e-5f) - 1e-12f;
          if (strictly) { bestabs = -1.f; bestn = v3(0, 0, 1); }
          {
            V3 nh = (1.f / fl) * fn;
            float sd = dot(dq, nh);
            float ab = fabsf(sd) * (sd > 0.f ? 1.001f : 1.f);   // coincident faces of opposite orientation: outside wins
            if (ab > bestabs) { bestn = nh; bestabs = ab; }
          }
          if (!found || d2 < best2) { best2 = d2; bestp = q; }
          found = true;
        }
      }
    }
    if (!pop()) done = true;
  }
  *cp_out = bestp; *fn_out = bestn;
  return found;
}

// Two closest-point queries in ONE traversal (the collision spheres a wave tests in a substep come in neighbouring
// pairs - two spheres of one link, or of adjacent links - whose searches visit almost the same nodes): a child is opened
// when either query still needs it, each leaf triangle is tested against both, and each query keeps its own radius and
// result exactly as `closest_point` would have (same tolerances, same tie rule), so the outcome per query is the
// solo outcome.  A query with on = false is skipped.
struct ClosestQuery { V3 p; float max_dist; bool on; bool found; V3 cp, fn;
                      float range, lb; };   // grid meshes: `range` = beyond this distance only a lower bound is wanted, returned in `lb` when nothing is found
#ifndef LG_PAIR_CHUNK
#define LG_PAIR_CHUNK 4
#endif
// EXT: the traversal stack lives in memory the caller provides (ext_k / ext_i, BVH_STACK entries each, per lane) -- LDS in the physics
// kernel.  As private arrays the stacks are scratch memory (dynamic indexing): every pop was a dependent load through L1 / L2.
template <bool EXT>
LG_DEV void closest_point_pair_t(const MeshView& M, ClosestQuery& A, ClosestQuery& B, float* ext_k, int* ext_i, int* visits) {
  float bestA = A.max_dist * A.max_dist, bestB = B.max_dist * B.max_dist, absA = -1.f, absB = -1.f;
  bool fA = false, fB = false;
  V3 pA = A.p, pB = B.p, cA = A.p, cB = B.p, nA = v3(0, 0, 1), nB = v3(0, 0, 1);
  const bool onA = A.on, onB = B.on;
  int loc_i[EXT ? 1 : BVH_STACK]; float loc_k[EXT ? 1 : BVH_STACK];
  int* const stack_i = EXT ? ext_i : loc_i; float* const stack_k = EXT ? ext_k : loc_k;
  int sp = 0; int cur = 0;
  // "while-while" form: every lane first walks inner nodes until it stands on a leaf (lanes that already do wait), then the whole wave
  // tests leaf faces together.  As one loop whose body handled "inner node or leaf", each of the wave's ~26 iterations paid for the
  // leaf branch (up to 8 faces, the expensive part) as soon as ANY lane stood on a leaf; now it is paid once per round of leaves.
  bool done = !(onA || onB);
  auto pop = [&]() -> bool {
    const float lim = fmaxf(onA ? bestA * (1.f + 1e-5f) + 1e-12f : -1.f, onB ? bestB * (1.f + 1e-5f) + 1e-12f : -1.f);
    while (sp > 0) { --sp; if (stack_k[sp] <= lim) { cur = stack_i[sp]; return true; } }
    return false;
  };
  while (!done) {
    const float limA = onA ? bestA * (1.f + 1e-5f) + 1e-12f : -1.f, limB = onB ? bestB * (1.f + 1e-5f) + 1e-12f : -1.f;
    while (cur >= 0) {
      if (visits) ++*visits;
      const Node4Regs n = load_node4(M.nodes, cur);
      int cand[4]; float key[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = F4(n.child, k);
        const float x0 = F4(n.minx, k), x1 = F4(n.maxx, k), y0 = F4(n.miny, k), y1 = F4(n.maxy, k), z0 = F4(n.minz, k), z1 = F4(n.maxz, k);
        float dx = fmaxf(fmaxf(x0 - pA.x, 0.f), pA.x - x1), dy = fmaxf(fmaxf(y0 - pA.y, 0.f), pA.y - y1), dz = fmaxf(fmaxf(z0 - pA.z, 0.f), pA.z - z1);
        const float kA = dx * dx + dy * dy + dz * dz;
        dx = fmaxf(fmaxf(x0 - pB.x, 0.f), pB.x - x1); dy = fmaxf(fmaxf(y0 - pB.y, 0.f), pB.y - y1); dz = fmaxf(fmaxf(z0 - pB.z, 0.f), pB.z - z1);
        const float kB = dx * dx + dy * dy + dz * dz;
        const bool wantA = kA <= limA, wantB = kB <= limB;
        key[k] = wantA && wantB ? fminf(kA, kB) : (wantA ? kA : kB);
        cand[k] = (c != BVH4_EMPTY && (wantA || wantB)) ? c : BVH4_EMPTY;
      }
      if (!descend4(cand, key, stack_i, stack_k, sp, cur) && !pop()) { done = true; break; }
    }
    if (done) break;
    {
      if (visits) ++*visits;
      const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
      for (int i0 = 0; i0 < cnt; i0 += LG_PAIR_CHUNK) {
      float4 ta[LG_PAIR_CHUNK], tb[LG_PAIR_CHUNK], tc[LG_PAIR_CHUNK];      // (the faces of a leaf are fetched LG_PAIR_CHUNK at a time, as in closest_point)
#pragma unroll
      for (int u = 0; u < LG_PAIR_CHUNK; ++u) {
        const float4* T = M.tris + (size_t)(first + min(i0 + u, cnt - 1)) * 3;
        ta[u] = T[0]; tb[u] = T[1]; tc[u] = T[2];
      }
#pragma unroll
      for (int u = 0; u < LG_PAIR_CHUNK; ++u) {
        if (i0 + u >= cnt) continue;
        V3 a = v3(ta[u].x, ta[u].y, ta[u].z), b = v3(tb[u].x, tb[u].y, tb[u].z), cc = v3(tc[u].x, tc[u].y, tc[u].z);
        const bool nearA = onA && tri_box_dist2(pA, a, b, cc) <= bestA * (1.f + 1e-5f) + 1e-12f;
        const bool nearB = onB && tri_box_dist2(pB, a, b, cc) <= bestB * (1.f + 1e-5f) + 1e-12f;
        if (!(nearA || nearB)) continue;
        V3 fn = cross(b - a, cc - a); float fl = norm(fn);
        if (!(fl > 1e-10f)) continue;                      // zero-area faces are skipped (see closest_point)
        const V3 nh = (1.f / fl) * fn;
        if (nearA) {
          V3 q = closest_on_triangle(pA, a, b, cc);
          V3 dq = pA - q; float d2 = dot(dq, dq);
          if (d2 <= bestA * (1.f + 1e-5f) + 1e-12f) {
            const bool strictly = !fA || d2 < bestA * (1.f - 1e-5f) - 1e-12f;
            if (strictly) { absA = -1.f; nA = v3(0, 0, 1); }
            float sd = dot(dq, nh);
            float ab = fabsf(sd) * (sd > 0.f ? 1.001f : 1.f);
            if (ab > absA) { nA = nh; absA = ab; }
            if (!fA || d2 < bestA) { bestA = d2; cA = q; }
            fA = true;
          }
        }
        if (nearB) {
          V3 q = closest_on_triangle(pB, a, b, cc);
          V3 dq = pB - q; float d2 = dot(dq, dq);
          if (d2 <= bestB * (1.f + 1e-5f) + 1e-12f) {
            const bool strictly = !fB || d2 < bestB * (1.f - 1e-5f) - 1e-12f;
            if (strictly) { absB = -1.f; nB = v3(0, 0, 1); }
            float sd = dot(dq, nh);
            float ab = fabsf(sd) * (sd > 0.f ? 1.001f : 1.f);
            if (ab > absB) { nB = nh; absB = ab; }
            if (!fB || d2 < bestB) { bestB = d2; cB = q; }
            fB = true;
          }
        }
      }
      }
    }
    if (!pop()) done = true;
  }
  A.found = fA; A.cp = cA; A.fn = nA; B.found = fB; B.cp = cB; B.fn = nB;
}
LG_DEV void closest_point_pair(const MeshView& M, ClosestQuery& A, ClosestQuery& B, int* visits = nullptr) {
  closest_point_pair_t<false>(M, A, B, nullptr, nullptr, visits);
}

// one face of a grid / lattice mesh against the running closest point: `closest_point`'s arithmetic, tolerances and tie rule (order-free)
LG_DEV void closest_grid_triangle(V3 p, V3 a, V3 b, V3 cc, float& best2, bool& found, float& bestabs, V3& bestp, V3& bestn) {
  V3 fn = cross(b - a, cc - a); float fl = norm(fn);
  if (!(fl > 1e-10f)) return;                                 // zero-area faces of the slope correction
  V3 q = closest_on_triangle(p, a, b, cc);
  V3 dq = p - q; float d2 = dot(dq, dq);
  if (!(d2 <= best2 * (1.f + 1e-5f) + 1e-12f)) return;
  const bool strictly = !found || d2 < best2 * (1.f - 1e-5f) - 1e-12f;
  if (strictly) { bestabs = -1.f; bestn = v3(0, 0, 1); }
  V3 nh = (1.f / fl) * fn;
  float sd = dot(dq, nh);
  float ab = fabsf(sd) * (sd > 0.f ? 1.001f : 1.f);
  if (ab > bestabs) { bestn = nh; bestabs = ab; }
  if (!found || d2 < best2) { best2 = d2; bestp = q; }
  found = true;
}

// Closest point on a LATTICE mesh (lg_mesh.d_gcz / d_gcr: an OBJ mesh whose vertices sit on an evenly spaced lattice in x and y -- what the confined-space and
// heightfield converters write; any number of layers, ceilings, walls).  Every triangle is listed in each cell its xy bounding box overlaps, so the triangles
// that hold a point within R of p are listed in the cells the square [p - R, p + R] reaches: the cell under p first (its distance bounds the rest), then
// rounds of 4 x 4 cell records -- thirty-two independent loads, a box test per cell and height group from the record's z ranges -- and the faces of the groups
// that pass.  Per-face arithmetic, tolerances and tie rule are `closest_point`'s, which does not depend on the order the faces are met in (a face met twice
// changes nothing): the result is the tree walk's.  The walk pays ~25 DEPENDENT 128-byte node fetches per query.
LG_DEV void closest_point_lattice(const LatticeView& L, ClosestQuery& A, unsigned long long* v64 = nullptr) {
  if (!A.on) return;
  typedef unsigned u2v __attribute__((ext_vector_type(2))); typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const u2v __attribute__((address_space(1)))* gu2; typedef const f4v __attribute__((address_space(1)))* gf4;
  const gf4 CELL = (gf4)L.cell; const gu2 RUN = (gu2)L.run; const gf4 TRI = (gf4)L.tris;
  const V3 p = A.p; const float R = A.max_dist;
  const float ihx = frcp(L.hx), ihy = frcp(L.hy);
  float best2 = R * R, bestabs = -1.f; bool found = false;
  V3 bestp = p, bestn = v3(0, 0, 1);
  A.found = false; A.cp = p; A.fn = v3(0, 0, 1);
  const float fx = (p.x - L.x0) * ihx, fy = (p.y - L.y0) * ihy;
  // (cell indices from arithmetic boundaries: the true ones lie within LATTICE_TOL / 2 of a cell width of them, the window is taken that much wider)
  float grx = R * ihx + 2.f * LATTICE_TOL, gry = R * ihy + 2.f * LATTICE_TOL;
  int i0 = max((int)floorf(fx - grx), 0), i1 = min((int)floorf(fx + grx), L.nx - 1), j0 = max((int)floorf(fy - gry), 0), j1 = min((int)floorf(fy + gry), L.ny - 1);
  if (i0 > i1 || j0 > j1) return;
  // a run of faces: fetched four at a time, the box of each against the current best in front of the exact test
  auto exact = [&](int first, int cnt) {
#pragma unroll 1
    for (int t0 = 0; t0 < cnt; t0 += 4) {
      f4v ta[4], tb[4], tc[4];
      if (v64) *v64 += 1ull << 21;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const gf4 T = TRI + (size_t)(first + min(t0 + u, cnt - 1)) * 3;
        ta[u] = T[0]; tb[u] = T[1]; tc[u] = T[2];
      }
#pragma unroll 1
      for (int u = 0; u < 4; ++u) {
        if (t0 + u >= cnt) break;
        const V3 a = v3(ta[u].x, ta[u].y, ta[u].z), b = v3(tb[u].x, tb[u].y, tb[u].z), cc = v3(tc[u].x, tc[u].y, tc[u].z);
        if (!(tri_box_dist2(p, a, b, cc) <= best2 * (1.f + 1e-5f) + 1e-12f)) continue;
        if (v64) *v64 += 1ull << 42;
        closest_grid_triangle(p, a, b, cc, best2, found, bestabs, bestp, bestn);
      }
    }
  };
  // which of a cell's two groups (bit 0 lower, bit 1 upper) can hold a point within the current best distance
  auto cell_ok = [&](int i, int j, f4v z, u2v r) -> unsigned {
    const float xa = L.x0 + ((float)i - LATTICE_TOL) * L.hx, xb = L.x0 + ((float)(i + 1) + LATTICE_TOL) * L.hx;
    const float ya = L.y0 + ((float)j - LATTICE_TOL) * L.hy, yb = L.y0 + ((float)(j + 1) + LATTICE_TOL) * L.hy;
    const float dx = fmaxf(fmaxf(xa - p.x, 0.f), p.x - xb), dy = fmaxf(fmaxf(ya - p.y, 0.f), p.y - yb);
    const float dz0 = fmaxf(fmaxf(z.x - p.z, 0.f), p.z - z.y), dz1 = fmaxf(fmaxf(z.z - p.z, 0.f), p.z - z.w);     // (an empty group: 1e30 -> never passes)
    const float dxy = dx * dx + dy * dy, lim = best2 * (1.f + 1e-5f) + 1e-12f;
    return (((r.y & 0xffffu) != 0u && dxy + dz0 * dz0 <= lim) ? 1u : 0u) | (((r.y >> 16) != 0u && dxy + dz1 * dz1 <= lim) ? 2u : 0u);
  };
  auto visit = [&](int i, int j) {
    const size_t c = (size_t)j * L.nx + i;
    const f4v z = CELL[c]; const u2v r = RUN[c];
    if (v64) *v64 += 1ull;
    const unsigned ok = cell_ok(i, j, z, r);
    const int n0 = (int)(r.y & 0xffffu), n1 = (int)(r.y >> 16);
    // the nearer group first: what it finds may rule the other one out
    const bool upper_first = ok == 3u && fabsf(p.z - z.z) < fabsf(p.z - z.y);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const bool upper = (h == 1) != upper_first;
      if (!((h == 0 ? ok : cell_ok(i, j, z, r)) & (upper ? 2u : 1u))) continue;
      exact((int)r.x + (upper ? n0 : 0), upper ? n1 : n0);
    }
  };
  const int ci = max(i0, min((int)floorf(fx), i1)), cj = max(j0, min((int)floorf(fy), j1));
  visit(ci, cj);
  if (found) {
    const float r = sqrtf(best2) * (1.f + 1e-4f);
    grx = r * ihx + 2.f * LATTICE_TOL; gry = r * ihy + 2.f * LATTICE_TOL;
    i0 = max(i0, (int)floorf(fx - grx)); i1 = min(i1, (int)floorf(fx + grx)); j0 = max(j0, (int)floorf(fy - gry)); j1 = min(j1, (int)floorf(fy + gry));
  }
#pragma unroll 1
  for (int jb = j0; jb <= j1; jb += 4) {
#pragma unroll 1
    for (int ib = i0; ib <= i1; ib += 4) {
      unsigned pass = 0u;
      {
        f4v z[16]; u2v r[16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const size_t c = (size_t)min(jb + u, L.ny - 1) * L.nx + min(ib + t, L.nx - 1);
            z[4 * u + t] = CELL[c]; r[4 * u + t] = RUN[c];
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (jb + u <= j1 && ib + t <= i1 && !(ib + t == ci && jb + u == cj) && cell_ok(ib + t, jb + u, z[4 * u + t], r[4 * u + t])) pass |= 1u << (4 * u + t);
      }
      // (a cell that passed: its records again, cache hits -- indexing the register arrays by `bit` would put them in scratch; and the bound may have shrunk)
#pragma unroll 1
      while (pass) {
        const int bit = __ffs(pass) - 1; pass &= pass - 1u;
        visit(ib + (bit & 3), jb + (bit >> 2));
      }
    }
  }
  A.found = found; A.cp = bestp; A.fn = bestn;
}

// ---- the same lattice query answered by a GROUP of 16 lanes (one DPP row) for ONE point: the SDF of a robot's bodies (sdf_bodies_kernel) is a few thousand
// queries, each with a window of up to ~50 cells (a trunk 0.3 m above 0.1 m cells) -- lane by lane a wave waits for its trunk queries through ~40 dependent
// round trips (measured: slower than the tree walk).  Here the group shares the work of one query: the faces of the cell under the point are dealt over
// the lanes (their minimum bounds the window), then the window's cells are dealt over the lanes -- one round of independent record loads for up to 64
// cells, a box test each -- and every lane tests the faces of the cells it kept.  Each lane keeps `closest_point`'s running state for the faces IT saw; the
// group's answer is put together by that function's own tie rule, which does not depend on the order faces are met in: the point of the closest face, and
// among the lanes whose best is within the tolerance band of the group's minimum the normal of the face farthest from the point.
LG_DEV float row16_min(float x) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) x = fminf(x, __shfl_xor(x, m, 16));
  return x;
}
LG_DEV int row16_min(int x) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) x = min(x, __shfl_xor(x, m, 16));
  return x;
}
LG_DEV float row16_max(float x) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) x = fmaxf(x, __shfl_xor(x, m, 16));
  return x;
}
// p, R: the same on all 16 lanes of the group; k: this lane's index in the group.  Returns found (on every lane), *cp / *fn on every lane.
LG_DEV bool closest_point_lattice_row16(const LatticeView& L, V3 p, float R, int k, V3* cp_out, V3* fn_out) {
  typedef unsigned u2v __attribute__((ext_vector_type(2))); typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const u2v __attribute__((address_space(1)))* gu2; typedef const f4v __attribute__((address_space(1)))* gf4;
  const gf4 CELL = (gf4)L.cell; const gu2 RUN = (gu2)L.run; const gf4 TRI = (gf4)L.tris;
  const float ihx = frcp(L.hx), ihy = frcp(L.hy);
  float best2 = R * R, bestabs = -1.f; bool found = false;
  V3 bestp = p, bestn = v3(0, 0, 1);
  const float fx = (p.x - L.x0) * ihx, fy = (p.y - L.y0) * ihy;
  float grx = R * ihx + 2.f * LATTICE_TOL, gry = R * ihy + 2.f * LATTICE_TOL;
  int i0 = max((int)floorf(fx - grx), 0), i1 = min((int)floorf(fx + grx), L.nx - 1), j0 = max((int)floorf(fy - gry), 0), j1 = min((int)floorf(fy + gry), L.ny - 1);
  if (i0 > i1 || j0 > j1) return false;                                       // (group-uniform)
  auto face = [&](int f) {
    const gf4 T = TRI + (size_t)f * 3;
    const f4v ta = T[0], tb = T[1], tc = T[2];
    const V3 a = v3(ta.x, ta.y, ta.z), b = v3(tb.x, tb.y, tb.z), cc = v3(tc.x, tc.y, tc.z);
    if (!(tri_box_dist2(p, a, b, cc) <= best2 * (1.f + 1e-5f) + 1e-12f)) return;
    closest_grid_triangle(p, a, b, cc, best2, found, bestabs, bestp, bestn);
  };
  // a lane whose own best is not within the band of the group's minimum forgets it (its face cannot decide anything) and prunes with the group's bound
  auto share_bound = [&]() {
    const float gm = row16_min(found ? best2 : R * R);
    if (!(found && best2 <= gm * (1.f + 1e-5f) + 1e-12f)) { found = false; bestabs = -1.f; best2 = gm; }
  };
  // 1. the cell under the point: its faces dealt over the lanes
  const int ci = max(i0, min((int)floorf(fx), i1)), cj = max(j0, min((int)floorf(fy), j1));
  {
    const u2v r = RUN[(size_t)cj * L.nx + ci];
    const int cnt = (int)(r.y & 0xffffu) + (int)(r.y >> 16);
    for (int f = k; f < cnt; f += 16) face((int)r.x + f);
    share_bound();
    if (row16_min(found ? 1 : 2) == 1) {                                      // something found: the window shrinks to what can be closer
      const float rr = sqrtf(best2) * (1.f + 1e-4f);
      grx = rr * ihx + 2.f * LATTICE_TOL; gry = rr * ihy + 2.f * LATTICE_TOL;
      i0 = max(i0, (int)floorf(fx - grx)); i1 = min(i1, (int)floorf(fx + grx)); j0 = max(j0, (int)floorf(fy - gry)); j1 = min(j1, (int)floorf(fy + gry));
    }
  }
  // 2. the window's cells, 64 at a time: cell c of a round belongs to lane c % 16 (neighbours on different lanes)
  const int wx = i1 - i0 + 1, ncell = wx * (j1 - j0 + 1);
  for (int c0 = 0; c0 < ncell; c0 += 64) {
    f4v z[4]; u2v r[4]; int ii[4], jj[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = min(c0 + k + 16 * m, ncell - 1);
      jj[m] = j0 + c / wx; ii[m] = i0 + c - (c / wx) * wx;
      const size_t at = (size_t)jj[m] * L.nx + ii[m];
      z[m] = CELL[at]; r[m] = RUN[at];
    }
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
      if (c0 + k + 16 * m >= ncell || (ii[m] == ci && jj[m] == cj)) continue;
      const float xa = L.x0 + ((float)ii[m] - LATTICE_TOL) * L.hx, xb = L.x0 + ((float)(ii[m] + 1) + LATTICE_TOL) * L.hx;
      const float ya = L.y0 + ((float)jj[m] - LATTICE_TOL) * L.hy, yb = L.y0 + ((float)(jj[m] + 1) + LATTICE_TOL) * L.hy;
      const float dx = fmaxf(fmaxf(xa - p.x, 0.f), p.x - xb), dy = fmaxf(fmaxf(ya - p.y, 0.f), p.y - yb), dxy = dx * dx + dy * dy;
      const int n0 = (int)(r[m].y & 0xffffu), n1 = (int)(r[m].y >> 16);
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {                                           // the lower and the upper group of the cell's faces
        const float zlo = h == 0 ? z[m].x : z[m].z, zhi = h == 0 ? z[m].y : z[m].w;
        const float dz = fmaxf(fmaxf(zlo - p.z, 0.f), p.z - zhi);
        const int cnt = h == 0 ? n0 : n1, first = (int)r[m].x + (h == 0 ? 0 : n0);
        if (cnt == 0 || !(dxy + dz * dz <= best2 * (1.f + 1e-5f) + 1e-12f)) continue;
        for (int f = 0; f < cnt; ++f) face(first + f);
      }
    }
    if (c0 + 64 < ncell) share_bound();
  }
  // 3. the group's answer
  const float gm = row16_min(found ? best2 : 3.0e38f);
  if (!(gm < 3.0e37f)) return false;
  const int wp = row16_min((found && best2 == gm) ? k : 99);                  // the closest face's lane (lowest among equals)
  const bool band = found && best2 <= gm * (1.f + 1e-5f) + 1e-12f;
  const float ga = row16_max(band ? bestabs : -2.f);
  const int wn = row16_min((band && bestabs == ga) ? k : 99);
  *cp_out = v3(__shfl(bestp.x, wp, 16), __shfl(bestp.y, wp, 16), __shfl(bestp.z, wp, 16));
  *fn_out = v3(__shfl(bestn.x, wn, 16), __shfl(bestn.y, wn, 16), __shfl(bestn.z, wn, 16));
  return true;
}


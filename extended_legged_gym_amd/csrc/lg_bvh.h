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
        float4 a = T[0], b = T[1], cc = T[2];
        V3 v0 = v3(a.x, a.y, a.z), e1 = v3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = v3(cc.x - a.x, cc.y - a.y, cc.z - a.z);
        V3 p = cross(d, e2);
        float det = dot(e1, p);
        if (fabsf(det) < 1e-20f) continue;
        float idet = 1.f / det;
        V3 s = o - v0;
        float u = dot(s, p) * idet;
        if (u < -RAY_EDGE_EPS || u > 1.f + RAY_EDGE_EPS) continue;
        V3 q = cross(s, e1);
        float v = dot(d, q) * idet;
        if (v < -RAY_EDGE_EPS || u + v > 1.f + RAY_EDGE_EPS) continue;
        float t = dot(e2, q) * idet;
        if (t >= 0.f && t <= best) { best = t; hit = true; }
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
                 const float4* __restrict__ tris; };
LG_DEV int raygrid_locate(const float* b, int n, float x, int guess) {
  int i = guess < 0 ? 0 : (guess > n - 1 ? n - 1 : guess);
  while (i > 0 && x < b[i]) --i;
  while (i < n - 1 && x >= b[i + 1]) ++i;
  return i;
}
// tb: the boundary tables in LDS, x boundaries (nx + 1) followed by y boundaries (ny + 1): one dependent LDS read per cell crossed
LG_DEV float trace_ray_grid(const RayGrid& G, const float* tb, V3 o, V3 d, float max_dist) {
  const float* xb = tb; const float* yb = tb + G.nx + 1;
  const V3 inv = v3(1.f / (fabsf(d.x) > 1e-12f ? d.x : copysignf(1e-12f, d.x)), 1.f / (fabsf(d.y) > 1e-12f ? d.y : copysignf(1e-12f, d.y)),
                    1.f / (fabsf(d.z) > 1e-12f ? d.z : copysignf(1e-12f, d.z)));
  // a ray that does not move along an axis (|d| <= 1e-12: a vertical ray) never crosses that axis's lines: its boundary times are -inf / +inf while it
  // is between the outer lines (those included), not the 0 * 1e12 a start exactly ON a line gives
  const bool mvx = fabsf(d.x) > 1e-12f, mvy = fabsf(d.y) > 1e-12f;
  // the part of the ray over the lattice
  float t0 = 0.f, t1 = max_dist;
  {
    const bool inx = o.x >= xb[0] && o.x <= xb[G.nx], iny = o.y >= yb[0] && o.y <= yb[G.ny];
    const float ax = mvx ? (xb[0] - o.x) * inv.x : (inx ? -3.0e38f : 3.0e38f), bx = mvx ? (xb[G.nx] - o.x) * inv.x : (inx ? 3.0e38f : -3.0e38f);
    const float ay = mvy ? (yb[0] - o.y) * inv.y : (iny ? -3.0e38f : 3.0e38f), by = mvy ? (yb[G.ny] - o.y) * inv.y : (iny ? 3.0e38f : -3.0e38f);
    t0 = fmaxf(t0, fmaxf(fminf(ax, bx), fminf(ay, by)));
    t1 = fminf(t1, fminf(fmaxf(ax, bx), fmaxf(ay, by)));
  }
  if (!(t0 <= t1)) return -1.f;
  const float px = o.x + t0 * d.x, py = o.y + t0 * d.y;
  const float ux = (float)G.nx / (xb[G.nx] - xb[0]), uy = (float)G.ny / (yb[G.ny] - yb[0]);
  int ix = raygrid_locate(xb, G.nx, px, (int)((px - xb[0]) * ux)), iy = raygrid_locate(yb, G.ny, py, (int)((py - yb[0]) * uy));
  const int sx = inv.x >= 0.f ? 1 : -1, sy = inv.y >= 0.f ? 1 : -1;
  const int ox = sx > 0 ? 1 : 0, oy = (sy > 0 ? 1 : 0) + G.nx + 1;     // offsets into tb of the boundary AHEAD of cell ix / iy
  // (the boundary AHEAD of a cell on such an axis: +inf as well -- the cell was left before it was looked at: a miss where the tree walk hits)
  // ... and a ray without motion across the lines of an axis that starts exactly ON one of them runs down the border of two columns of cells: triangles
  // of the - side column reach it only with an edge (they are not listed in the + side cells), and where the slope correction folded the surface the
  // nearest hit may be one of theirs.  Such rays (a measure-zero set; tested wave-wide, so other waves pay three ballots) walk the - side column(s) too.
  const int ix0 = ix, iy0 = iy;
  const bool lx = !mvx && ix0 > 0 && px == xb[ix0], ly = !mvy && iy0 > 0 && py == yb[iy0];
  float best = max_dist; bool hit = false;
  for (int var = 0; var < 4; ++var) {
  const bool use = var == 0 || (((var & 1) == 0 || lx) && ((var & 2) == 0 || ly));
  if (var > 0 && __ballot(use) == 0ull) continue;
  ix = ix0 - (var & 1); iy = iy0 - (var >> 1);
  float tmx = mvx ? (tb[ix + ox] - o.x) * inv.x : 3.0e38f, tmy = mvy ? (tb[iy + oy] - o.y) * inv.y : 3.0e38f;
  float tcur = t0;
  // the ends of a cell's stretch of the ray are rounded: the z range is taken a little beyond either end
  const float padc = 2e-5f * fabsf(t1) + 1e-6f;
  bool done = !use;
  // one cell on, without branches: the axis whose boundary comes first, the boundary ahead of the new cell from LDS
  auto advance = [&]() {
    const bool stx = tmx <= tmy;
    const int inew = stx ? ix + sx : iy + sy;
    const bool out = inew < 0 || inew >= (stx ? G.nx : G.ny);
    ix = stx ? inew : ix; iy = stx ? iy : inew;
    const float bnd = tb[out ? 0 : inew + (stx ? ox : oy)];
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
        float4 a = T[0], b = T[1], cc = T[2];
        V3 v0 = v3(a.x, a.y, a.z), e1 = v3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = v3(cc.x - a.x, cc.y - a.y, cc.z - a.z);
        V3 p = cross(d, e2);
        float det = dot(e1, p);
        if (fabsf(det) < 1e-20f) continue;
        float idet = 1.f / det;
        V3 s = o - v0;
        float u = dot(s, p) * idet;
        if (u < -RAY_EDGE_EPS || u > 1.f + RAY_EDGE_EPS) continue;
        V3 q = cross(s, e1);
        float v = dot(d, q) * idet;
        if (v < -RAY_EDGE_EPS || u + v > 1.f + RAY_EDGE_EPS) continue;
        float t = dot(e2, q) * idet;
        if (t >= 0.f && t <= best) { best = t; hit = true; }
      }
      if ((hit && best <= tcur) || !(tnext < t1)) done = true; else advance();
      tcur = tnext;
    }
  }
  }
  return hit ? best : -1.f;
}

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


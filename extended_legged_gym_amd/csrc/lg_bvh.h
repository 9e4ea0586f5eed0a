// lg_bvh.h — BVH node / triangle layout and the two device traversals (closest ray hit, closest surface point) shared by
// the mesh-query kernels (lg_mesh.hip) and the triangle-mesh contact detection of the physics kernel (lg_step.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "lg_device.h"

struct BvhNode {          // 32 B
  float bmin[3]; int32_t left_first;   // inner: index of left child (right = left + 1); leaf: first triangle
  float bmax[3]; int32_t count;        // 0 = inner node, > 0 = number of triangles in the leaf
};

struct lg_mesh {
  int device = 0;
  int64_t n_tris = 0, n_nodes = 0;
  BvhNode* d_nodes = nullptr;
  float4* d_tris = nullptr;            // 3 float4 per triangle: v0, v1, v2 (w unused)
  std::string err;
};

// ------------------------------------------------------------------------------------------------ device: traversal
struct MeshView { const BvhNode* __restrict__ nodes; const float4* __restrict__ tris; };

LG_DEV bool slab(const BvhNode& n, V3 o, V3 inv, float tmax, float* tnear) {
  float tx1 = (n.bmin[0] - o.x) * inv.x, tx2 = (n.bmax[0] - o.x) * inv.x;
  float ty1 = (n.bmin[1] - o.y) * inv.y, ty2 = (n.bmax[1] - o.y) * inv.y;
  float tz1 = (n.bmin[2] - o.z) * inv.z, tz2 = (n.bmax[2] - o.z) * inv.z;
  float tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), 0.f));
  float tmx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), tmax));
  *tnear = tmin;
  return tmin <= tmx;
}

// closest two-sided hit with 0 <= t <= max_dist (Moller-Trumbore); returns t or -1
LG_DEV float trace_ray(const MeshView& M, V3 o, V3 d, float max_dist) {
  const V3 inv = v3(1.f / (fabsf(d.x) > 1e-12f ? d.x : copysignf(1e-12f, d.x)), 1.f / (fabsf(d.y) > 1e-12f ? d.y : copysignf(1e-12f, d.y)),
                    1.f / (fabsf(d.z) > 1e-12f ? d.z : copysignf(1e-12f, d.z)));
  float best = max_dist; bool hit = false;
  int stack[48]; int sp = 0;
  int cur = 0; float tn;
  if (!slab(M.nodes[0], o, inv, best, &tn)) return -1.f;
  while (true) {
    const BvhNode n = M.nodes[cur];
    if (n.count > 0) {
      for (int i = 0; i < n.count; ++i) {
        const float4* T = M.tris + (size_t)(n.left_first + i) * 3;
        float4 a = T[0], b = T[1], c = T[2];
        V3 v0 = v3(a.x, a.y, a.z), e1 = v3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = v3(c.x - a.x, c.y - a.y, c.z - a.z);
        V3 p = cross(d, e2);
        float det = dot(e1, p);
        if (fabsf(det) < 1e-20f) continue;
        float idet = 1.f / det;
        V3 s = o - v0;
        float u = dot(s, p) * idet;
        if (u < 0.f || u > 1.f) continue;
        V3 q = cross(s, e1);
        float v = dot(d, q) * idet;
        if (v < 0.f || u + v > 1.f) continue;
        float t = dot(e2, q) * idet;
        if (t >= 0.f && t <= best) { best = t; hit = true; }
      }
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    const int l = n.left_first, r = l + 1;
    float tl, tr;
    bool hl = slab(M.nodes[l], o, inv, best, &tl), hr = slab(M.nodes[r], o, inv, best, &tr);
    if (hl && hr) {
      int nearc = tl <= tr ? l : r, farc = tl <= tr ? r : l;
      if (sp < 48) stack[sp++] = farc;
      cur = nearc;
    } else if (hl) cur = l;
    else if (hr) cur = r;
    else { if (sp == 0) break; cur = stack[--sp]; }
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

LG_DEV float box_dist2(const BvhNode& n, V3 p) {
  float dx = fmaxf(fmaxf(n.bmin[0] - p.x, 0.f), p.x - n.bmax[0]);
  float dy = fmaxf(fmaxf(n.bmin[1] - p.y, 0.f), p.y - n.bmax[1]);
  float dz = fmaxf(fmaxf(n.bmin[2] - p.z, 0.f), p.z - n.bmax[2]);
  return dx * dx + dy * dy + dz * dz;
}

// closest point within max_dist; outputs the point and the unit normal of the face that decides the sign.  When several
// faces are equally close (the closest feature is a shared edge or vertex) the face whose plane is farthest from the
// query point decides: that rule is independent of traversal order, so the BVH and a brute-force scan agree.
LG_DEV bool closest_point(const MeshView& M, V3 p, float max_dist, V3* cp_out, V3* fn_out) {
  float best2 = max_dist * max_dist; bool found = false; float bestabs = -1.f;
  V3 bestp = p, bestn = v3(0, 0, 1);
  int stack[48]; int sp = 0; int cur = 0;
  if (box_dist2(M.nodes[0], p) > best2) return false;
  while (true) {
    const BvhNode n = M.nodes[cur];
    if (n.count > 0) {
      for (int i = 0; i < n.count; ++i) {
        const float4* T = M.tris + (size_t)(n.left_first + i) * 3;
        float4 a4 = T[0], b4 = T[1], c4 = T[2];
        V3 a = v3(a4.x, a4.y, a4.z), b = v3(b4.x, b4.y, b4.z), c = v3(c4.x, c4.y, c4.z);
        // zero-area faces (slope-corrected height-field meshes are full of them) are skipped: their points belong to
        // the edges of their neighbours, and the barycentric arithmetic below is 0/0 on them
        V3 fn = cross(b - a, c - a); float fl = norm(fn);
        if (!(fl > 1e-10f)) continue;
        V3 q = closest_on_triangle(p, a, b, c);
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
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    const int l = n.left_first, r = l + 1;
    float dl = box_dist2(M.nodes[l], p), dr = box_dist2(M.nodes[r], p);
    const float lim = best2 * (1.f + 1e-5f) + 1e-12f;
    bool hl = dl <= lim, hr = dr <= lim;
    if (hl && hr) { int nearc = dl <= dr ? l : r, farc = dl <= dr ? r : l; if (sp < 48) stack[sp++] = farc; cur = nearc; }
    else if (hl) cur = l;
    else if (hr) cur = r;
    else { if (sp == 0) break; cur = stack[--sp]; }
  }
  *cp_out = bestp; *fn_out = bestn;
  return found;
}

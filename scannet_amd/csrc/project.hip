// project.hip -- the per-frame body of the ProjectAnnotations tool on gfx950: the labelled mesh drawn from every frame's camera into
// an instance image (u8) and a label image (u16) at colour resolution, filtered against the sensor depth and by a 5x5 vote.
// Reference: AnnotationTools/ProjectAnnotations/Visualizer.cpp:57-193 (render), shaders/drawAnnotations.hlsl:9-33 (flat colour:
// `nointerpolation`, so a fragment carries the ids of its triangle's first vertex).
//
//   k_pa_vertex      position * worldViewProj for every vertex of every frame in the batch        drawAnnotations.hlsl:16-27
//   k_pa_raster      one wave per cluster of 64 triangles (culled as a whole against the view volume); set-up one lane per triangle
//                    (near-plane clip, 1/256-pixel snap, edge functions with the top-left rule, depth plane), traversal one triangle
//                    per wave step with the lanes on the pixels of a block; depth test = 64-bit atomicMin on {z bits, triangle
//                    index}; triangles spanning more than 64 pixels go to a queue
//   k_pa_raster_big  one workgroup per queued triangle, lanes strided over its bounding box
// (measured and dropped: binning the triangles to 64 x 64 screen tiles and rasterising each tile into an LDS depth buffer -- the
//  same-address atomics of the tile counters and the serial per-lane loops of one workgroup per tile cost 940 us per 8 frames
//  against 430 us for the global atomicMin below)
//   k_pa_resolve     colour/depth read-back (Visualizer.cpp:104-141) fused with the depth-consistency filter (:144-164)
//   k_pa_vote        5x5 neighbourhood vote (:166-186)
// Up to PA_MAX_BATCH frames per launch (blockIdx.y): the reference renders one frame per message-loop iteration and reads both
// buffers back through staging textures each time; here a frame is ~100 MB of HBM traffic (vertex transform + index stream), so
// batching only serves to hide launch latency and to make the D2H copies larger.
//
// The reference draws through Direct3D 11; the rules this software rasteriser follows (and which of mLib's conventions are not
// pinned by anything in the reference tree) are listed in the CPU checker's header; the two agree bit for bit.  The depth test
// resolves in any order (atomicMin of a total order), so the result does not depend on scheduling.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr int PA_MAX_BATCH = 8;
constexpr int PA_CLUSTER = 64;       // triangles per culling cluster = one wave of k_pa_raster

struct PaK {
  int w, h, dw, dh;
  float zn, zf, A, B, thresh;
  int filter_orig;
  uint32_t V, F;
};

struct PaBatch {
  float M[PA_MAX_BATCH][16];
  int valid[PA_MAX_BATCH];
  int has_depth;
};

struct ClipV { float x, y, z, w; };
struct ScrV { float px, py, z; };

struct TriSetup {
  long long x0, y0, x1, y1, x2, y2;
  long long ex0, ex1, ex2, ey0, ey1, ey2;
  float zc, dzdx, dzdy;   // depth plane: z(i, j) = (zc + (i - i0) dzdx) + (j - j0) dzdy
  int i0, i1, j0, j1;
  bool tl0, tl1, tl2;
};

__device__ inline long long snap(float p) {
  const float s = p * 256.0f;
  if (!(s > -1.0e9f && s < 1.0e9f)) return LLONG_MIN;
  return (long long)floorf(s + 0.5f);
}

__device__ inline bool tri_setup(TriSetup& S, int w, int h, const ScrV& a, const ScrV& b, const ScrV& c) {
  long long x0 = snap(a.px), y0 = snap(a.py), x1 = snap(b.px), y1 = snap(b.py), x2 = snap(c.px), y2 = snap(c.py);
  if (x0 == LLONG_MIN || y0 == LLONG_MIN || x1 == LLONG_MIN || y1 == LLONG_MIN || x2 == LLONG_MIN || y2 == LLONG_MIN) return false;
  float z0 = a.z, z1 = b.z, z2 = c.z;
  long long area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
  if (area == 0) return false;
  if (area < 0) {
    long long t;
    float tz;
    t = x1; x1 = x2; x2 = t;
    t = y1; y1 = y2; y2 = t;
    tz = z1; z1 = z2; z2 = tz;
    area = -area;
  }
  const long long minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2)), miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  long long i0 = (minx - 128 + 255) >> 8, i1 = (maxx - 128) >> 8, j0 = (miny - 128 + 255) >> 8, j1 = (maxy - 128) >> 8;
  if (i0 < 0) i0 = 0;
  if (j0 < 0) j0 = 0;
  if (i1 > w - 1) i1 = w - 1;
  if (j1 > h - 1) j1 = h - 1;
  if (i1 < i0 || j1 < j0) return false;
  S.x0 = x0; S.y0 = y0; S.x1 = x1; S.y1 = y1; S.x2 = x2; S.y2 = y2;
  S.ex0 = x1 - x0; S.ex1 = x2 - x1; S.ex2 = x0 - x2; S.ey0 = y1 - y0; S.ey1 = y2 - y1; S.ey2 = y0 - y2;
  S.tl0 = (S.ey0 == 0 && S.ex0 > 0) || (S.ey0 < 0);
  S.tl1 = (S.ey1 == 0 && S.ex1 > 0) || (S.ey1 < 0);
  S.tl2 = (S.ey2 == 0 && S.ex2 > 0) || (S.ey2 < 0);
  S.i0 = (int)i0; S.i1 = (int)i1; S.j0 = (int)j0; S.j1 = (int)j1;
  // the depth plane through the three snapped vertices, anchored at the first pixel of the bounding box: with the edge functions
  // e0 + e1 + e2 = area, z = z0 + (e2 (z1 - z0) + e0 (z2 - z0)) / area
  const long long px0 = 256 * i0 + 128, py0 = 256 * j0 + 128;
  const long long q0 = S.ex0 * (py0 - y0) - S.ey0 * (px0 - x0), q2 = S.ex2 * (py0 - y2) - S.ey2 * (px0 - x2);
  const float fa = (float)area, dz1 = z1 - z0, dz2 = z2 - z0;
  S.zc = z0 + ((float)q2 * dz1 + (float)q0 * dz2) / fa;
  S.dzdx = ((float)(-(S.ey2 * 256)) * dz1 + (float)(-(S.ey0 * 256)) * dz2) / fa;
  S.dzdy = ((float)(S.ex2 * 256) * dz1 + (float)(S.ex0 * 256) * dz2) / fa;
  return true;
}

__device__ inline void shade(const TriSetup& S, int i, int j, int w, unsigned long long* __restrict__ zbuf, uint32_t id) {
  const long long px = 256ll * i + 128, py = 256ll * j + 128;
  const long long e0 = S.ex0 * (py - S.y0) - S.ey0 * (px - S.x0);
  const long long e1 = S.ex1 * (py - S.y1) - S.ey1 * (px - S.x1);
  const long long e2 = S.ex2 * (py - S.y2) - S.ey2 * (px - S.x2);
  if (e0 < 0 || e1 < 0 || e2 < 0) return;
  if ((e0 == 0 && !S.tl0) || (e1 == 0 && !S.tl1) || (e2 == 0 && !S.tl2)) return;
  const float z = (S.zc + (float)(i - S.i0) * S.dzdx) + (float)(j - S.j0) * S.dzdy;
  if (!(z >= 0.0f && z <= 1.0f)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | id;   // z >= +0: float order = order of the bits
  atomicMin(&zbuf[(size_t)j * w + i], key);
}

__device__ inline ScrV to_screen(const ClipV& c, float W, float H) {
  ScrV s;
  s.px = (c.x / c.w + 1.0f) * 0.5f * W;
  s.py = (1.0f - c.y / c.w) * 0.5f * H;
  s.z = c.z / c.w;
  return s;
}

__device__ inline ClipV clip_point(const ClipV& in, const ClipV& out) {
  const float t = in.z / (in.z - out.z);
  ClipV q;
  q.x = in.x + t * (out.x - in.x);
  q.y = in.y + t * (out.y - in.y);
  q.z = 0.0f;
  q.w = in.w + t * (out.w - in.w);
  return q;
}

// triangle t of one frame -> up to two screen-space triangles (a fan over the near-clipped polygon); returns how many
__device__ inline int clip_and_project(const PaK& P, const uint32_t* __restrict__ tris, const float4* __restrict__ clip, uint32_t t, ScrV s[4]) {
  const uint32_t ia = tris[3 * (size_t)t], ib = tris[3 * (size_t)t + 1], ic = tris[3 * (size_t)t + 2];
  if (ia >= P.V || ib >= P.V || ic >= P.V) return 0;
  const float4 qa = clip[ia], qb = clip[ib], qc = clip[ic];
  const ClipV c[3] = {{qa.x, qa.y, qa.z, qa.w}, {qb.x, qb.y, qb.z, qb.w}, {qc.x, qc.y, qc.z, qc.w}};
  const bool in[3] = {c[0].z >= 0.0f, c[1].z >= 0.0f, c[2].z >= 0.0f};
  const int nin = (int)in[0] + (int)in[1] + (int)in[2];
  if (nin == 0) return 0;
  if (!(c[0].z == c[0].z && c[1].z == c[1].z && c[2].z == c[2].z)) return 0;
  // all three vertices in front of the near plane (w > 0) and beyond the same side of the view volume: the snapped bounding box
  // would miss every pixel centre (x/w <= -1 puts the vertex at or left of pixel 0's left edge, and so on), skip the divisions
  if (nin == 3 && ((c[0].x < -c[0].w && c[1].x < -c[1].w && c[2].x < -c[2].w) || (c[0].x > c[0].w && c[1].x > c[1].w && c[2].x > c[2].w) ||
                   (c[0].y < -c[0].w && c[1].y < -c[1].w && c[2].y < -c[2].w) || (c[0].y > c[0].w && c[1].y > c[1].w && c[2].y > c[2].w)))
    return 0;
  ClipV poly[4];
  int np = 0;
  if (nin == 3) { poly[0] = c[0]; poly[1] = c[1]; poly[2] = c[2]; np = 3; }
  else {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int k1 = (k + 1) % 3;
      if (in[k]) poly[np++] = c[k];
      if (in[k] != in[k1]) poly[np++] = in[k] ? clip_point(c[k], c[k1]) : clip_point(c[k1], c[k]);
    }
  }
  const float W = (float)P.w, H = (float)P.h;
  for (int k = 0; k < np; k++) s[k] = to_screen(poly[k], W, H);
  return np - 2;
}

__global__ __launch_bounds__(256) void k_pa_vertex(PaK P, PaBatch Bt, const float* __restrict__ xyz, float4* __restrict__ clip_all) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if (i >= P.V || !Bt.valid[q]) return;
  const float* M = Bt.M[q];
  const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  float4 c;
  c.x = M[0] * x + M[1] * y + M[2] * z + M[3];
  c.y = M[4] * x + M[5] * y + M[6] * z + M[7];
  c.z = M[8] * x + M[9] * y + M[10] * z + M[11];
  c.w = M[12] * x + M[13] * y + M[14] * z + M[15];
  clip_all[(size_t)q * P.V + i] = c;
}

// bounding box of every cluster of PA_CLUSTER consecutive triangles (once per mesh): {min x, y, z, -, max x, y, z, -}
__global__ __launch_bounds__(256) void k_pa_cluster_bounds(const float* __restrict__ xyz, const uint32_t* __restrict__ tris, uint32_t V, uint32_t F,
                                                           float* __restrict__ bounds) {
  const uint32_t cluster = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if ((size_t)cluster * PA_CLUSTER >= F) return;
  const uint32_t t = cluster * PA_CLUSTER + lane;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (t < F) {
    const uint32_t ia = tris[3 * (size_t)t], ib = tris[3 * (size_t)t + 1], ic = tris[3 * (size_t)t + 2];
    if (ia < V && ib < V && ic < V) {
      for (uint32_t v : {ia, ib, ic})
        for (int k = 0; k < 3; k++) { const float c = xyz[3 * (size_t)v + k]; lo[k] = fminf(lo[k], c); hi[k] = fmaxf(hi[k], c); }
    }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
  if (lane == 0) {
    float* o = bounds + 8 * (size_t)cluster;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
  }
}

// One wave per cluster.  A cluster whose box lies outside one plane of the view volume (with a margin far above the rounding of the
// vertex transform) is dropped by the whole wave before any triangle is read.  Set-up is one lane per triangle (clip, project, snap,
// edge functions, depth plane); the traversal is one triangle at a time for the whole wave: the owner's set-up is broadcast (readlane
// -> SGPRs) and the 64 lanes take the pixels of a 2^s x 2^(6-s) block that walks the bounding box, so the depth tests of one row of
// fragments fall into one or two cache lines instead of one line per fragment (a lane-per-triangle traversal was bound by exactly
// that: 388 us per 8 frames, the same with half the arithmetic).  Triangles spanning more than 64 pixels go to k_pa_raster_big.
// (Measured and dropped: set-up and traversal as two kernels with 64-byte records read back through scalar loads -- balanced, a
// single frame 36 us instead of 91, but 261 us instead of 200 for a batch of eight.)
// Depth test = 64-bit atomicMin on {z bits, triangle index}.
__global__ __launch_bounds__(256) void k_pa_raster(PaK P, PaBatch Bt, const uint32_t* __restrict__ tris, const float4* __restrict__ clip_all,
                                                   const float* __restrict__ bounds, unsigned long long* __restrict__ zbuf_all,
                                                   uint32_t* __restrict__ late_all, uint32_t* __restrict__ late_count) {
  const int q = blockIdx.y;
  if (!Bt.valid[q]) return;
  const uint32_t cluster = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if ((size_t)cluster * PA_CLUSTER >= P.F) return;
  {
    const float* b = bounds + 8 * (size_t)cluster;
    const float* M = Bt.M[q];
    const float x = b[(lane & 1) ? 4 : 0], y = b[(lane & 2) ? 5 : 1], z = b[(lane & 4) ? 6 : 2];
    const float cx = M[0] * x + M[1] * y + M[2] * z + M[3], cy = M[4] * x + M[5] * y + M[6] * z + M[7];
    const float cz = M[8] * x + M[9] * y + M[10] * z + M[11], cw = M[12] * x + M[13] * y + M[14] * z + M[15];
    const float m = 1.0e-3f * (1.0f + fabsf(cw) + fmaxf(fabsf(cx), fmaxf(fabsf(cy), fabsf(cz))));
    if (__all(cx + cw < -m) || __all(cw - cx < -m) || __all(cy + cw < -m) || __all(cw - cy < -m) || __all(cz < -m)) return;
  }
  // from here on the wave stays together: lanes without a triangle still rasterise the others' pixels
  const uint32_t t = cluster * PA_CLUSTER + lane;
  unsigned long long* zbuf = zbuf_all + (size_t)q * P.w * P.h;
  ScrV s[4];
  const int ntri = t < P.F ? clip_and_project(P, tris, clip_all + (size_t)q * P.V, t, s) : 0;
  uint32_t late_id[2];
  int nlate = 0;
  for (int sub = 0; sub < 2; sub++) {
    // this lane's triangle: 32-bit set-up when it spans at most 64 pixels each way (|edge| <= 2^14 subpixels, so every edge function
    // is below 2^30 inside the box), otherwise the queue
    int r0 = 0, r1 = 0, r2 = 0, sx0 = 0, sx1 = 0, sx2 = 0, sy0 = 0, sy1 = 0, sy2 = 0, origin = 0, extent = 0, tl = 0;
    float zc = 0.0f, dzdx = 0.0f, dzdy = 0.0f;
    bool small = false;
    if (sub < ntri) {
      TriSetup S;
      if (tri_setup(S, P.w, P.h, s[0], s[1 + sub], s[2 + sub])) {
        const long long span = max(max(llabs(S.ex0), llabs(S.ex1)), max(max(llabs(S.ex2), llabs(S.ey0)), max(llabs(S.ey1), llabs(S.ey2))));
        if (span <= 16384) {
          const long long px0 = 256ll * S.i0 + 128, py0 = 256ll * S.j0 + 128;
          r0 = (int)(S.ex0 * (py0 - S.y0) - S.ey0 * (px0 - S.x0));
          r1 = (int)(S.ex1 * (py0 - S.y1) - S.ey1 * (px0 - S.x1));
          r2 = (int)(S.ex2 * (py0 - S.y2) - S.ey2 * (px0 - S.x2));
          sx0 = (int)(-(S.ey0 * 256)); sx1 = (int)(-(S.ey1 * 256)); sx2 = (int)(-(S.ey2 * 256));
          sy0 = (int)(S.ex0 * 256); sy1 = (int)(S.ex1 * 256); sy2 = (int)(S.ex2 * 256);
          origin = S.i0 | (S.j0 << 16);
          extent = (S.i1 - S.i0 + 1) | ((S.j1 - S.j0 + 1) << 8);
          tl = (S.tl0 ? 1 : 0) | (S.tl1 ? 2 : 0) | (S.tl2 ? 4 : 0);
          zc = S.zc; dzdx = S.dzdx; dzdy = S.dzdy;
          small = true;
        } else {
          late_id[nlate++] = 2u * t + (uint32_t)sub;
        }
      }
    }
    unsigned long long todo = __ballot(small);
    while (todo) {
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
      todo &= todo - 1;
      const int R0 = __builtin_amdgcn_readlane(r0, src), R1 = __builtin_amdgcn_readlane(r1, src), R2 = __builtin_amdgcn_readlane(r2, src);
      const int SX0 = __builtin_amdgcn_readlane(sx0, src), SX1 = __builtin_amdgcn_readlane(sx1, src), SX2 = __builtin_amdgcn_readlane(sx2, src);
      const int SY0 = __builtin_amdgcn_readlane(sy0, src), SY1 = __builtin_amdgcn_readlane(sy1, src), SY2 = __builtin_amdgcn_readlane(sy2, src);
      const int org = __builtin_amdgcn_readlane(origin, src), ext = __builtin_amdgcn_readlane(extent, src), TL = __builtin_amdgcn_readlane(tl, src);
      const float ZC = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zc), src));
      const float DZX = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzdx), src));
      const float DZY = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzdy), src));
      const uint32_t id = cluster * PA_CLUSTER + (uint32_t)src;
      const int i0 = org & 0xffff, j0 = org >> 16, bw = ext & 0xff, bh = ext >> 8;
      const int sh = bw > 32 ? 6 : (bw > 16 ? 5 : (bw > 8 ? 4 : 3));   // block of 2^sh x 2^(6-sh) pixels
      const int lx = lane & ((1 << sh) - 1), ly = lane >> sh;
      const int step_x = 1 << sh, step_y = 64 >> sh;
      for (int by = 0; by < bh; by += step_y) {
        const int dy = by + ly;
        unsigned long long* row = zbuf + (size_t)(j0 + dy) * P.w + i0;
        const float zrow = (float)dy * DZY;
        for (int bx = 0; bx < bw; bx += step_x) {
          const int dx = bx + lx;
          if (dx < bw && dy < bh) {
            const int e0 = R0 + dx * SX0 + dy * SY0, e1 = R1 + dx * SX1 + dy * SY1, e2 = R2 + dx * SX2 + dy * SY2;
            if ((e0 | e1 | e2) >= 0 && !((e0 == 0 && !(TL & 1)) || (e1 == 0 && !(TL & 2)) || (e2 == 0 && !(TL & 4)))) {
              const float z = (ZC + (float)dx * DZX) + zrow;
              if (z >= 0.0f && z <= 1.0f) atomicMin(&row[dx], ((unsigned long long)__float_as_uint(z) << 32) | id);
            }
          }
        }
      }
    }
  }
  // queue appends, one atomic per wave and round
  for (int k = 0; k < 2; k++) {
    const bool has = k < nlate;
    const unsigned long long mask = __ballot(has);
    if (mask == 0) continue;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&late_count[q], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (has) late_all[(size_t)q * 2 * P.F + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = late_id[k];
  }
}

__global__ __launch_bounds__(256) void k_pa_raster_big(PaK P, PaBatch Bt, const uint32_t* __restrict__ tris, const float4* __restrict__ clip_all,
                                                       unsigned long long* __restrict__ zbuf_all, const uint32_t* __restrict__ queue_all,
                                                       const uint32_t* __restrict__ queue_count) {
  const int q = blockIdx.y;
  if (!Bt.valid[q]) return;
  const uint32_t n = queue_count[q];
  unsigned long long* zbuf = zbuf_all + (size_t)q * P.w * P.h;
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const uint32_t entry = queue_all[(size_t)q * 2 * P.F + e];
    const uint32_t t = entry >> 1;
    const int sub = (int)(entry & 1u);
    ScrV s[4];
    const int ntri = clip_and_project(P, tris, clip_all + (size_t)q * P.V, t, s);
    if (sub >= ntri) continue;
    TriSetup S;
    if (!tri_setup(S, P.w, P.h, s[0], s[1 + sub], s[2 + sub])) continue;
    const int bw = S.i1 - S.i0 + 1, bh = S.j1 - S.j0 + 1;
    const long long npx = (long long)bw * bh;
    for (long long p = threadIdx.x; p < npx; p += 256) shade(S, S.i0 + (int)(p % bw), S.j0 + (int)(p / bw), P.w, zbuf, t);
  }
}

__device__ inline float camera_z(float d, const PaK& P) {
  if (d == 0.0f || d == 1.0f) return 0.0f;
  const float z = P.B / (P.A - d);
  return (z >= P.zn && z <= P.zf) ? z : 0.0f;
}

__global__ __launch_bounds__(256) void k_pa_resolve(PaK P, PaBatch Bt, const uint32_t* __restrict__ tris, const uint8_t* __restrict__ vinst,
                                                    const uint16_t* __restrict__ vlabel, const unsigned long long* __restrict__ zbuf_all,
                                                    const uint16_t* __restrict__ depth_all, uint8_t* __restrict__ inst_all, uint16_t* __restrict__ label_all,
                                                    float* __restrict__ zcam_all) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  const int n = P.w * P.h;
  if (i >= n) return;
  const unsigned long long* zbuf = zbuf_all + (size_t)q * n;
  const unsigned long long key = zbuf[i];
  uint8_t inst = 0;
  uint16_t label = 0;
  float zc = 0.0f;
  if (key != ~0ull) {
    const uint32_t v0 = tris[3 * (size_t)(uint32_t)key];
    inst = vinst[v0];
    label = vlabel[v0];
    zc = camera_z(__uint_as_float((uint32_t)(key >> 32)), P);
  }
  if (zcam_all) zcam_all[(size_t)q * n + i] = zc;
  if (label != 0 && Bt.has_depth) {
    const int x = i % P.w, y = i / P.w;
    const float sw = (float)(P.dw - 1) / (float)(P.w - 1), sh = (float)(P.dh - 1) / (float)(P.h - 1);
    const float rw = (float)(P.w - 1) / (float)(P.dw - 1), rh = (float)(P.h - 1) / (float)(P.dh - 1);
    const unsigned dx = (unsigned)roundf(sw * (float)x), dy = (unsigned)roundf(sh * (float)y);
    int sx = (int)roundf((float)dx * rw), sy = (int)roundf((float)dy * rh);
    if (sx > P.w - 1) sx = P.w - 1;
    if (sy > P.h - 1) sy = P.h - 1;
    const unsigned long long k2 = zbuf[(size_t)sy * P.w + sx];
    const float z2 = k2 == ~0ull ? 0.0f : camera_z(__uint_as_float((uint32_t)(k2 >> 32)), P);
    const uint16_t drndr = (uint16_t)(z2 * 1000.0f);
    const uint16_t dorig = depth_all[(size_t)q * P.dw * P.dh + (size_t)dy * P.dw + dx];
    if ((P.filter_orig && dorig == 0) ||
        (drndr != 0 && dorig != 0 && fabsf((float)((int)drndr - (int)dorig) * 0.001f) > P.thresh + 0.01f * (float)dorig)) {
      label = 0;
      inst = 0;
    }
  }
  inst_all[(size_t)q * n + i] = inst;
  label_all[(size_t)q * n + i] = label;
}

// 64 x 8 pixels per workgroup, the labels of the tile and its two-pixel apron staged in LDS; a lane votes for two neighbouring pixels
// and reads their 6 x 5 window as 15 aligned 32-bit words (7.5 LDS reads per pixel instead of 25 -- the kernel is LDS-issue bound)
__global__ __launch_bounds__(256) void k_pa_vote(PaK P, const uint8_t* __restrict__ inst_in, const uint16_t* __restrict__ label_in,
                                                 uint8_t* __restrict__ inst_out, uint16_t* __restrict__ label_out) {
  __shared__ uint32_t tile32[12][36];   // 12 rows x 72 labels (68 used)
  uint16_t(*tile)[72] = reinterpret_cast<uint16_t(*)[72]>(tile32);
  const int tiles_x = (P.w + 63) / 64;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int q = blockIdx.y;
  const int n = P.w * P.h;
  const uint16_t* L = label_in + (size_t)q * n;
  const int x0 = tx * 64 - 2, y0 = ty * 8 - 2;
  for (int k = threadIdx.x; k < 12 * 68; k += 256) {
    const int r = k / 68, c = k % 68;
    const int x = x0 + c, y = y0 + r;
    tile[r][c] = (x >= 0 && x < P.w && y >= 0 && y < P.h) ? L[(size_t)y * P.w + x] : (uint16_t)0;
  }
  __syncthreads();
  const int lx = 2 * (threadIdx.x % 32), ly = threadIdx.x / 32;
  const int x = tx * 64 + lx, y = ty * 8 + ly;
  if (x >= P.w || y >= P.h) return;
  const uint32_t centre = tile32[ly + 2][lx / 2 + 1];
  const uint32_t va = centre & 0xffffu, vb = centre >> 16;   // labels of (x, y) and (x + 1, y)
  unsigned ca = 0, cb = 0;
  if (va != 0 || vb != 0) {
    // pixels outside the image hold 0 in the tile and a voting label is never 0, so they never count
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
      const uint32_t w0 = tile32[ly + dy][lx / 2], w1 = tile32[ly + dy][lx / 2 + 1], w2 = tile32[ly + dy][lx / 2 + 2];
      const uint32_t c0 = w0 & 0xffffu, c1 = w0 >> 16, c2 = w1 & 0xffffu, c3 = w1 >> 16, c4 = w2 & 0xffffu, c5 = w2 >> 16;
      ca += (c0 == va) + (c1 == va) + (c2 == va) + (c3 == va) + (c4 == va);
      cb += (c1 == vb) + (c2 == vb) + (c3 == vb) + (c4 == vb) + (c5 == vb);
    }
  }
  const int rows = min(y + 2, P.h - 1) - max(y - 2, 0) + 1;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int xx = x + k;
    if (xx >= P.w) break;
    const uint32_t v = k ? vb : va;
    const unsigned count = k ? cb : ca;
    const int i = y * P.w + xx;
    uint8_t inst = 0;
    uint16_t label = 0;
    if (v != 0) {
      const unsigned total = (unsigned)((min(xx + 2, P.w - 1) - max(xx - 2, 0) + 1) * rows);   // the in-image part of the window
      if (!((float)count / (float)total < 0.2f)) { label = (uint16_t)v; inst = inst_in[(size_t)q * n + i]; }
    }
    inst_out[(size_t)q * n + i] = inst;
    label_out[(size_t)q * n + i] = label;
  }
}

// worldViewProj (Visualizer.cpp:91-93): view = rigid inverse of camera-to-world from its normalised columns, projection from fx, fy with
// the principal point at the centre of the render target, image y down, Direct3D z
void world_view_proj(const float* c, float fx, float fy, uint32_t W, uint32_t H, float n, float f, float* M) {
  float ax[3][3];
  for (int k = 0; k < 3; k++) {
    const float x = c[k], y = c[4 + k], z = c[8 + k];
    const float len = sqrtf(x * x + y * y + z * z);
    ax[k][0] = x / len; ax[k][1] = y / len; ax[k][2] = z / len;
  }
  const float ex = c[3], ey = c[7], ez = c[11];
  float view[3][4];
  for (int k = 0; k < 3; k++) {
    view[k][0] = ax[k][0]; view[k][1] = ax[k][1]; view[k][2] = ax[k][2];
    view[k][3] = -(ax[k][0] * ex + ax[k][1] * ey + ax[k][2] * ez);
  }
  const float p00 = 2.0f * fx / (float)W, p11 = -(2.0f * fy / (float)H);
  const float A = f / (f - n), B = (n * f) / (f - n);
  for (int j = 0; j < 4; j++) {
    M[j] = p00 * view[0][j];
    M[4 + j] = p11 * view[1][j];
    M[8 + j] = A * view[2][j];
    M[12 + j] = view[2][j];
  }
  M[11] = M[11] - B;
}

}  // namespace

struct sf_projector {
  sf_project_params p;
  PaK k;
  int device = 0;
  hipStream_t stream = nullptr;
  float* xyz = nullptr;
  uint32_t* tris = nullptr;
  uint8_t* vinst = nullptr;
  uint16_t* vlabel = nullptr;
  float4* clip = nullptr;                 // PA_MAX_BATCH x V
  unsigned long long* zbuf = nullptr;     // PA_MAX_BATCH x w*h
  float* bounds = nullptr;                // clusters x 8
  uint32_t* late = nullptr;               // PA_MAX_BATCH x 2 F (triangle, half) ids with large bounding boxes
  uint32_t* late_count = nullptr;         // PA_MAX_BATCH
  uint16_t* depth = nullptr;              // PA_MAX_BATCH x dw*dh
  uint8_t *inst1 = nullptr, *inst2 = nullptr;
  uint16_t *label1 = nullptr, *label2 = nullptr;
  float* zcam = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

SF_API int sf_projector_max_batch(void) { return PA_MAX_BATCH; }

// page-locked host memory for the images handed to / returned by sf_projector_run: the copies then run at PCIe speed instead of
// being staged through the runtime's bounce buffers (6 GB/s measured from pageable memory)
SF_API int sf_host_alloc(uint64_t bytes, void** out) {
  if (!out || bytes == 0) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return SF_OK;
}
SF_API void sf_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

SF_API void sf_projector_destroy(sf_projector* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  for (void* d : {(void*)p->xyz, (void*)p->tris, (void*)p->vinst, (void*)p->vlabel, (void*)p->clip, (void*)p->zbuf, (void*)p->bounds, (void*)p->late, (void*)p->late_count,
                  (void*)p->depth, (void*)p->inst1, (void*)p->inst2, (void*)p->label1, (void*)p->label2, (void*)p->zcam})
    if (d) (void)hipFree(d);
  if (p->e0) (void)hipEventDestroy(p->e0);
  if (p->e1) (void)hipEventDestroy(p->e1);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

SF_API int sf_projector_create(const sf_project_params* params, int device, sf_projector** out) {
  if (!params || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (params->color_width < 2 || params->color_height < 2 || params->depth_width < 2 || params->depth_height < 2 || params->color_width > 16384 ||
      params->color_height > 16384)
    return sf::fail(SF_ERR_INVALID_ARG, "invalid image dimensions");
  if (!(params->fx > 0.0f) || !(params->fy > 0.0f) || !(params->depth_min > 0.0f) || !(params->depth_max > params->depth_min))
    return sf::fail(SF_ERR_INVALID_ARG, "invalid camera (fx %g fy %g depth range [%g, %g])", params->fx, params->fy, params->depth_min, params->depth_max);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device: libscanfuse has no CPU fallback, the annotation projector needs an MI355X");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  sf_projector* p = new sf_projector();
  p->p = *params;
  p->device = device;
  PaK& k = p->k;
  k.w = (int)params->color_width; k.h = (int)params->color_height; k.dw = (int)params->depth_width; k.dh = (int)params->depth_height;
  k.zn = params->depth_min; k.zf = params->depth_max;
  k.A = k.zf / (k.zf - k.zn);
  k.B = (k.zn * k.zf) / (k.zf - k.zn);
  k.thresh = params->depth_dist_thresh;
  k.filter_orig = params->filter_using_original_depth ? 1 : 0;
  k.V = k.F = 0;
  const size_t n = (size_t)k.w * k.h, dn = (size_t)k.dw * k.dh;
#define PA_ALLOC(ptr, bytes)                                                                                      \
  do {                                                                                                            \
    const hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                                     \
    if (e_ != hipSuccess) { sf_projector_destroy(p); return sf::fail(SF_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e_)); } \
  } while (0)
  PA_ALLOC(p->zbuf, PA_MAX_BATCH * n * 8);
  PA_ALLOC(p->late_count, PA_MAX_BATCH * 4);
  PA_ALLOC(p->depth, PA_MAX_BATCH * dn * 2);
  PA_ALLOC(p->inst1, PA_MAX_BATCH * n); PA_ALLOC(p->inst2, PA_MAX_BATCH * n);
  PA_ALLOC(p->label1, PA_MAX_BATCH * n * 2); PA_ALLOC(p->label2, PA_MAX_BATCH * n * 2);
  PA_ALLOC(p->zcam, PA_MAX_BATCH * n * 4);
#undef PA_ALLOC
  SF_HIP_CHECK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreate(&p->e0));
  SF_HIP_CHECK(hipEventCreate(&p->e1));
  *out = p;
  return SF_OK;
}

// MeshDataf with m_Colors = (r, g, instance, label) per vertex (Visualizer.cpp:259-295) -> device; replaces any previous mesh
SF_API int sf_projector_set_mesh(sf_projector* p, const float* xyz, uint64_t num_vertices, const uint32_t* triangles, uint64_t num_triangles,
                                 const uint8_t* vertex_instance, const uint16_t* vertex_label) {
  if (!p || !xyz || !triangles || !vertex_instance || !vertex_label) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (num_vertices == 0 || num_triangles == 0 || num_vertices > 0x7fffffffull || num_triangles > 0x7fffffffull)
    return sf::fail(SF_ERR_INVALID_ARG, "mesh size out of range (%llu vertices, %llu triangles)", (unsigned long long)num_vertices, (unsigned long long)num_triangles);
  SF_HIP_CHECK(hipSetDevice(p->device));
  SF_HIP_CHECK(hipStreamSynchronize(p->stream));
  for (void** d : {(void**)&p->xyz, (void**)&p->tris, (void**)&p->vinst, (void**)&p->vlabel, (void**)&p->clip, (void**)&p->bounds, (void**)&p->late}) {
    if (*d) (void)hipFree(*d);
    *d = nullptr;
  }
  p->k.V = p->k.F = 0;
  SF_HIP_CHECK(hipMalloc((void**)&p->xyz, num_vertices * 12));
  SF_HIP_CHECK(hipMalloc((void**)&p->tris, num_triangles * 12));
  SF_HIP_CHECK(hipMalloc((void**)&p->vinst, num_vertices));
  SF_HIP_CHECK(hipMalloc((void**)&p->vlabel, num_vertices * 2));
  SF_HIP_CHECK(hipMalloc((void**)&p->clip, (size_t)PA_MAX_BATCH * num_vertices * 16));
  SF_HIP_CHECK(hipMemcpy(p->xyz, xyz, num_vertices * 12, hipMemcpyHostToDevice));
  SF_HIP_CHECK(hipMemcpy(p->tris, triangles, num_triangles * 12, hipMemcpyHostToDevice));
  SF_HIP_CHECK(hipMemcpy(p->vinst, vertex_instance, num_vertices, hipMemcpyHostToDevice));
  SF_HIP_CHECK(hipMemcpy(p->vlabel, vertex_label, num_vertices * 2, hipMemcpyHostToDevice));
  const size_t clusters = (num_triangles + PA_CLUSTER - 1) / PA_CLUSTER;
  SF_HIP_CHECK(hipMalloc((void**)&p->bounds, clusters * 8 * 4));
  SF_HIP_CHECK(hipMalloc((void**)&p->late, (size_t)PA_MAX_BATCH * 2 * num_triangles * 4));
  hipLaunchKernelGGL(k_pa_cluster_bounds, dim3((unsigned)((clusters + 3) / 4)), dim3(256), 0, p->stream, p->xyz, p->tris, (uint32_t)num_vertices,
                     (uint32_t)num_triangles, p->bounds);
  SF_HIP_CHECK(hipGetLastError());
  SF_HIP_CHECK(hipStreamSynchronize(p->stream));
  p->k.V = (uint32_t)num_vertices;
  p->k.F = (uint32_t)num_triangles;
  return SF_OK;
}

// n <= sf_projector_max_batch() frames: cam2world n x 16 (row-major; first element -inf = no valid transform -> empty images,
// Visualizer.cpp:63,187-192), orig_depth n x depth_w*depth_h u16 millimetres (host; NULL = skip the depth-consistency filter),
// outputs n x color_w*color_h (host).  zcam_out (nullable): the rendered depth in metres at colour resolution.
SF_API int sf_projector_run(sf_projector* p, int n, const float* cam2world, const uint16_t* orig_depth, uint8_t* instance_out, uint16_t* label_out,
                            float* zcam_out, float* kernel_us) {
  if (!p || !cam2world || !instance_out || !label_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > PA_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (1..%d)", n, PA_MAX_BATCH);
  if (p->k.V == 0) return sf::fail(SF_ERR_INVALID_ARG, "no mesh set (sf_projector_set_mesh)");
  SF_HIP_CHECK(hipSetDevice(p->device));
  const PaK& k = p->k;
  const size_t np = (size_t)k.w * k.h, dn = (size_t)k.dw * k.dh;
  PaBatch b;
  std::memset(&b, 0, sizeof(b));
  for (int i = 0; i < n; i++) {
    const float* c = cam2world + 16 * (size_t)i;
    b.valid[i] = c[0] != -INFINITY;
    if (b.valid[i]) world_view_proj(c, p->p.fx, p->p.fy, p->p.color_width, p->p.color_height, k.zn, k.zf, b.M[i]);
  }
  b.has_depth = orig_depth != nullptr;
  SF_HIP_CHECK(hipMemsetAsync(p->zbuf, 0xFF, (size_t)n * np * 8, p->stream));
  SF_HIP_CHECK(hipMemsetAsync(p->late_count, 0, PA_MAX_BATCH * 4, p->stream));
  if (orig_depth) SF_HIP_CHECK(hipMemcpyAsync(p->depth, orig_depth, (size_t)n * dn * 2, hipMemcpyHostToDevice, p->stream));
  if (kernel_us) SF_HIP_CHECK(hipEventRecord(p->e0, p->stream));
  const unsigned cluster_blocks = (unsigned)(((k.F + PA_CLUSTER - 1) / PA_CLUSTER + 3) / 4);
  hipLaunchKernelGGL(k_pa_vertex, dim3((k.V + 255) / 256, n), dim3(256), 0, p->stream, k, b, p->xyz, p->clip);
  hipLaunchKernelGGL(k_pa_raster, dim3(cluster_blocks, n), dim3(256), 0, p->stream, k, b, p->tris, p->clip, p->bounds, p->zbuf, p->late, p->late_count);
  hipLaunchKernelGGL(k_pa_raster_big, dim3(512, n), dim3(256), 0, p->stream, k, b, p->tris, p->clip, p->zbuf, p->late, p->late_count);
  hipLaunchKernelGGL(k_pa_resolve, dim3((unsigned)((np + 255) / 256), n), dim3(256), 0, p->stream, k, b, p->tris, p->vinst, p->vlabel, p->zbuf, p->depth,
                     p->inst1, p->label1, zcam_out ? p->zcam : nullptr);
  hipLaunchKernelGGL(k_pa_vote, dim3((unsigned)(((k.w + 63) / 64) * ((k.h + 7) / 8)), n), dim3(256), 0, p->stream, k, p->inst1, p->label1, p->inst2, p->label2);
  SF_HIP_CHECK(hipGetLastError());
  if (kernel_us) SF_HIP_CHECK(hipEventRecord(p->e1, p->stream));
  SF_HIP_CHECK(hipMemcpyAsync(instance_out, p->inst2, (size_t)n * np, hipMemcpyDeviceToHost, p->stream));
  SF_HIP_CHECK(hipMemcpyAsync(label_out, p->label2, (size_t)n * np * 2, hipMemcpyDeviceToHost, p->stream));
  if (zcam_out) SF_HIP_CHECK(hipMemcpyAsync(zcam_out, p->zcam, (size_t)n * np * 4, hipMemcpyDeviceToHost, p->stream));
  SF_HIP_CHECK(hipStreamSynchronize(p->stream));
  if (kernel_us) {
    float ms = 0.0f;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, p->e0, p->e1));
    *kernel_us = 1000.0f * ms;
  }
  return SF_OK;
}

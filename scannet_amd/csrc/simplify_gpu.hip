// simplify_gpu.hip -- "Quadric Edge Collapse Decimation" on the GPU: the decimate stage's simplify.mlx (Server/scan_processor.py:144-145,
// Server/tools/meshclean/simplify.mlx:3-16) as ROUNDS OF INDEPENDENT COLLAPSES, an opt-in beside the sequential restatement in simplify.cpp.
//
// The sequential filter takes collapses one at a time from a global priority queue: 22 s of a scan's 24 s of host time, one thread, and
// with 16 usable CPUs per GPU the thing that sets scans per minute.  Same quadrics, same optimal placement, same priority
// and the same stop rule here (simplify_math.h is shared), but a different ORDER, so the triangles differ from the sequential result while
// the properties the filter guarantees hold (tests/test_simplify.py runs the same property tests on both):
//   per round  1. vertex -> face lists and the unique edges of the live faces (radix sorts; an edge with one face is a border edge)
//              2. per edge (v0 < v1): Q = Q0 + Q1, position x = minimiser of Q, priority = scale * Q(x) / min(QualityThr, worst quality of
//                 the faces around the pair after the move), floored at 1e-15 -- the host formulas; plus the link condition (the common
//                 neighbours of v0 and v1 are exactly the vertices opposite the edge): a global queue steers the sequential filter away
//                 from pinching the surface, nothing does here
//              3. candidates = the edges at or below the priority the sequential filter would reach for the collapses still needed
//                 (a quantile of this round's priorities), so no edge is taken that the global order would have left alone
//              4. every candidate writes its key (priority, scrambled edge id) over the closed 1-rings of both its end points (64-bit
//                 atomicMin into a lock word per vertex) and WINS when it still holds the minimum at its own end points: winners touch
//                 disjoint sets of faces and do not change what each other's priority, placement and link test read, so they are
//                 collapsed together; three such passes per round, the later ones among the candidates the earlier winners left
//                 untouched; the last round takes the winners in key order up to the face budget
//              5. collapse: the faces holding both vertices die, v0's other faces are re-pointed to v1, v1 moves to x and takes the
//                 summed quadric (and keeps its own colour, as VCG does)
//   then the shared end of the filter (simplify_finish: AutoClean, compaction).
// Deterministic: sorts, quantile, lock minima and the gather order of the initial quadrics (faces in index order, as the host adds
// them) do not depend on thread timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "hip_util.h"
#include "mesh.h"
#include "simplify_math.h"

sf_mesh* simplify_finish(const sf_mesh* in, const sf_simplify_params& P, std::vector<float>& pos, std::vector<uint32_t>& tri_in, std::vector<uint8_t>& fdel,
                         std::vector<uint8_t>& vdel, uint64_t nfaces, sf_simplify_stats& st);   // simplify.cpp

namespace {

using sf::DevBuf;
using sf::StreamGuard;

using sfq::Quadric;

constexpr int MAX_RING = 96;   // neighbours of one vertex held in registers / scratch for the link test; a longer ring rejects the collapse


struct Tri { uint32_t a, b, c; };


#define DEC_CHECK(call)                                                                                                \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess) return sf::fail(SF_ERR_DEVICE, "%s failed: %s (simplify_gpu.hip:%d)", #call, hipGetErrorString(e_), __LINE__); \
  } while (0)

struct Mesh {   // device view
  float* pos;            // 3 per vertex
  uint32_t* tri;         // 3 per face
  uint8_t* alive;        // per face
  uint8_t* vdel;         // per vertex
  Quadric* Q;            // per vertex
  const uint32_t* vbeg;  // per vertex: first entry of its corner list
  const uint32_t* vcnt;  // ... and how many
  const uint32_t* corner;  // sorted corner ids f * 3 + j
  uint32_t V, F;
};

// ---- 1. adjacency ----------------------------------------------------------------------------------------------------------------------
__global__ void k_emit(const uint32_t* __restrict__ tri, const uint8_t* __restrict__ alive, uint32_t F, uint32_t* ckey, uint32_t* cval, uint64_t* ekey) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const bool live = alive[f] != 0;
  uint32_t t[3] = {tri[3 * (size_t)f], tri[3 * (size_t)f + 1], tri[3 * (size_t)f + 2]};
  for (int j = 0; j < 3; j++) {
    ckey[3 * (size_t)f + j] = live ? t[j] : 0xFFFFFFFFu;
    cval[3 * (size_t)f + j] = 3 * f + (uint32_t)j;
    const uint32_t a = t[j], b = t[(j + 1) % 3];
    ekey[3 * (size_t)f + j] = live ? (((uint64_t)(a < b ? a : b) << 32) | (uint64_t)(a < b ? b : a)) : ~0ull;
  }
}
__global__ void k_vertex_ranges(const uint32_t* __restrict__ ckey, uint32_t n, uint32_t* vbeg, uint32_t* vcnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = ckey[i];
  if (k == 0xFFFFFFFFu) return;
  if (i == 0 || ckey[i - 1] != k) vbeg[k] = i;
  if (i + 1 == n || ckey[i + 1] != k) vcnt[k] = i + 1;   // end for now; turned into a count below
}
__global__ void k_fix_counts(uint32_t* vbeg, uint32_t* vcnt, uint32_t V) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) vcnt[v] = vcnt[v] > vbeg[v] ? vcnt[v] - vbeg[v] : 0u;
}

__device__ inline int edge_faces(const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ ucnt, uint32_t E, uint32_t a, uint32_t b) {
  const uint64_t key = ((uint64_t)(a < b ? a : b) << 32) | (uint64_t)(a < b ? b : a);
  uint32_t lo = 0, hi = E;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ukey[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < E && ukey[lo] == key) ? (int)ucnt[lo] : 0;
}

// ---- initial quadrics (simplify.cpp InitQuadric), gathered per vertex in the order the host adds them ---------------------------------------
__global__ void k_init_quadrics(Mesh M, const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ ucnt, uint32_t E, float boundary_weight, int planar) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= M.V) return;
  Quadric acc;
  acc.zero();
  for (uint32_t i = 0; i < M.vcnt[v]; i++) {
    const uint32_t c = M.corner[M.vbeg[v] + i], f = c / 3;
    const uint32_t* t = &M.tri[3 * (size_t)f];
    double p0[3], p1[3], p2[3];
    for (int k = 0; k < 3; k++) { p0[k] = M.pos[3 * (size_t)t[0] + k]; p1[k] = M.pos[3 * (size_t)t[1] + k]; p2[k] = M.pos[3 * (size_t)t[2] + k]; }
    const double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    Quadric q;
    q.by_plane(n, n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
    acc.add(q);
    for (int j = 0; j < 3; j++) {
      const uint32_t a = t[j], b = t[(j + 1) % 3];
      if (a != v && b != v) continue;
      const bool border = edge_faces(ukey, ucnt, E, a, b) == 1;
      if (!border && !planar) continue;
      double pa[3], pb[3], d[3];
      for (int k = 0; k < 3; k++) { pa[k] = M.pos[3 * (size_t)a + k]; pb[k] = M.pos[3 * (size_t)b + k]; d[k] = pb[k] - pa[k]; }
      const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      if (!(dl > 0.0)) continue;
      for (int k = 0; k < 3; k++) d[k] /= dl;
      const double wgt = border ? 0.5 * (double)boundary_weight : 0.5 * (double)boundary_weight / 100.0;
      const double bn[3] = {(n[1] * d[2] - n[2] * d[1]) * wgt, (n[2] * d[0] - n[0] * d[2]) * wgt, (n[0] * d[1] - n[1] * d[0]) * wgt};
      Quadric bq;
      bq.by_plane(bn, bn[0] * pa[0] + bn[1] * pa[1] + bn[2] * pa[2]);
      acc.add(bq);
    }
  }
  M.Q[v] = acc;
}

// ---- 2. priority, position, link condition per edge ------------------------------------------------------------------------------------------
__device__ inline void optimal_position(const Mesh& M, uint32_t v0, uint32_t v1, const Quadric& q, int optimal, float out[3]) {
  if (!optimal) { for (int k = 0; k < 3; k++) out[k] = M.pos[3 * (size_t)v1 + k]; return; }
  double p0[3], p1[3], mid[3], x[3];
  for (int k = 0; k < 3; k++) { p0[k] = M.pos[3 * (size_t)v0 + k]; p1[k] = M.pos[3 * (size_t)v1 + k]; mid[k] = 0.5 * (p0[k] + p1[k]); }
  sfq::minimise(q, mid, x);
  for (int k = 0; k < 3; k++) out[k] = (float)x[k];
  if (!(out[0] == out[0] && out[1] == out[1] && out[2] == out[2])) {   // NaN guard: best of the three candidates
    const double qm = q.apply(mid), q0 = q.apply(p0), q1 = q.apply(p1);
    const double* best = mid;
    if (q0 < qm) best = p0;
    if (q1 < qm && q1 < q0) best = p1;
    for (int k = 0; k < 3; k++) out[k] = (float)best[k];
  }
}

__global__ __launch_bounds__(128) void k_priority(Mesh M, const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ ucnt, uint32_t E, double scale,
                                                  float quality_thr, int optimal, float* pri, float* xout) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const uint32_t v0 = (uint32_t)(ukey[e] >> 32), v1 = (uint32_t)ukey[e];
  Quadric q = M.Q[v0];
  q.add(M.Q[v1]);
  float x[3];
  optimal_position(M, v0, v1, q, optimal, x);
  double min_qual = 1e300;
  uint32_t ring[MAX_RING];   // the neighbours of v0 (with repetitions: every fan face contributes two)
  int nring = 0;
  bool reject = false;
  int shared_faces = 0;
  for (int side = 0; side < 2; side++) {
    const uint32_t a = side ? v1 : v0, other = side ? v0 : v1;
    for (uint32_t i = 0; i < M.vcnt[a]; i++) {
      const uint32_t c = M.corner[M.vbeg[a] + i], f = c / 3;
      const uint32_t* t = &M.tri[3 * (size_t)f];
      const bool has_other = t[0] == other || t[1] == other || t[2] == other;
      if (has_other) { if (side == 0) shared_faces++; }
      else {
        const float* p[3];
        for (int k = 0; k < 3; k++) p[k] = t[k] == a ? x : &M.pos[3 * (size_t)t[k]];
        const double qt = sfq::quality(p[0], p[1], p[2]);
        if (qt < min_qual) min_qual = qt;
      }
      if (side == 0) {
        for (int k = 0; k < 3; k++)
          if (t[k] != a) { if (nring < MAX_RING) ring[nring++] = t[k]; else reject = true; }
      }
    }
  }
  // link condition: distinct common neighbours == faces on the edge
  int common = 0;
  if (!reject) {
    for (uint32_t i = 0; i < M.vcnt[v1] && !reject; i++) {
      const uint32_t c = M.corner[M.vbeg[v1] + i], f = c / 3, j = c % 3;
      // each neighbour of v1 once: as the vertex FOLLOWING v1 in a fan face, or -- at the open end of a border fan -- as the one preceding it in
      // a face whose edge (w, v1) has no second face
      const uint32_t* t = &M.tri[3 * (size_t)f];
      const uint32_t nxt = t[(j + 1) % 3], prv = t[(j + 2) % 3];
      uint32_t cand[2];
      int nc = 0;
      cand[nc++] = nxt;
      if (edge_faces(ukey, ucnt, E, prv, v1) == 1) cand[nc++] = prv;
      for (int q2 = 0; q2 < nc; q2++) {
        if (cand[q2] == v0) continue;
        bool in0 = false;
        for (int r = 0; r < nring; r++) in0 = in0 || ring[r] == cand[q2];
        common += in0 ? 1 : 0;
      }
    }
    if (common != shared_faces) reject = true;
  }
  const double xd[3] = {x[0], x[1], x[2]};
  double err = scale * q.apply(xd);
  if (min_qual > quality_thr) min_qual = quality_thr;
  if (err < 1e-15) err = 1e-15;   // QuadricEpsilon
  if (quality_thr > 0.0f) err = min_qual > 0.0 ? err / min_qual : 1e300;
  float pf = err > 3.0e38 ? 3.0e38f : (float)err;
  if (reject) pf = INFINITY;
  pri[e] = pf;
  xout[3 * (size_t)e] = x[0]; xout[3 * (size_t)e + 1] = x[1]; xout[3 * (size_t)e + 2] = x[2];
}

// ---- 4. independent winners -------------------------------------------------------------------------------------------------------------------
template <class Fn>
__device__ inline void for_closed_rings(const Mesh& M, uint32_t v0, uint32_t v1, Fn fn) {
  fn(v0);
  fn(v1);
  for (int side = 0; side < 2; side++) {
    const uint32_t a = side ? v1 : v0;
    for (uint32_t i = 0; i < M.vcnt[a]; i++) {
      const uint32_t* t = &M.tri[3 * (size_t)(M.corner[M.vbeg[a] + i] / 3)];
      for (int k = 0; k < 3; k++)
        if (t[k] != a) fn(t[k]);
    }
  }
}
// Lock key of an edge: its priority (positive, so the bit pattern orders it), ties broken by a BIJECTIVE scramble of the edge index salted
// with the round.  Not by the index itself: edge indices follow the vertex order, which on a marching-cubes mesh follows space, and the
// flat parts of a room carry the same floored priority on every edge -- the lowest index of a neighbourhood is then the lowest of a
// whole wall, one collapse per wall and round.  Scrambled, a constant fraction of the tied edges is a local minimum every round.
__device__ inline unsigned long long lock_key(float p, uint32_t e, uint32_t salt) {
  uint32_t h = e + salt;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   // invertible steps: distinct edges keep distinct keys
  return ((unsigned long long)__float_as_uint(p) << 32) | h;
}
// Two collapses e = (a, b) and w = (v0, v1) are independent iff neither has an end point inside the closed 1-rings of the other's end points
// (the relation is symmetric: a in ring(v0) <=> v0 in ring(a)): they then touch disjoint faces, and nothing either one's priority, placement
// or link test read is changed by the other.  Every candidate writes its key over its closed rings; it WINS iff it still holds the minimum
// at its own two end points -- any conflicting edge has one of them in its rings and a smaller key would show there.  `taken` marks the
// closed rings of the winners of earlier passes of the same round: an edge with a marked end point conflicts with one of them and sits
// the round out, the others are unaffected by what those winners will do and may compete again (more collapses per rebuild of the
// adjacency, which is what a round costs).
__global__ void k_lock(Mesh M, const uint64_t* __restrict__ ukey, uint32_t E, const float* __restrict__ pri, float tau, uint32_t salt, const uint8_t* __restrict__ taken,
                       unsigned long long* lock) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float p = pri[e];
  if (!(p <= tau)) return;
  const uint32_t v0 = (uint32_t)(ukey[e] >> 32), v1 = (uint32_t)ukey[e];
  if (taken[v0] | taken[v1]) return;
  const unsigned long long key = lock_key(p, e, salt);
  for_closed_rings(M, v0, v1, [&](uint32_t u) { atomicMin(&lock[u], key); });
}
// counters: [0] winners, [1] faces their collapses remove (2 per interior edge, 1 per border edge), [2] bit pattern of the largest priority
__global__ void k_winners(Mesh M, const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ ucnt, uint32_t E, const float* __restrict__ pri, float tau, uint32_t salt,
                          const unsigned long long* __restrict__ lock, uint8_t* taken_next, unsigned long long* win, uint32_t* counters) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const float p = e < E ? pri[e] : INFINITY;
  bool won = e < E && p <= tau;
  uint32_t v0 = 0, v1 = 0;
  if (won) {
    v0 = (uint32_t)(ukey[e] >> 32);
    v1 = (uint32_t)ukey[e];
    const unsigned long long key = lock_key(p, e, salt);
    won = lock[v0] == key && lock[v1] == key;   // only candidates that were still free wrote keys: no need to look at `taken` again
  }
  // the three counters are single words the whole launch shares: one atomic each per WAVE (a returning atomic per winner on one address is ~25 ns apiece, in a row)
  const uint64_t wm = __ballot(won);
  if (wm == 0ull) return;
  const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((unsigned long long)wm) - 1;
  uint32_t faces = won ? ucnt[e] : 0u, top = won ? __float_as_uint(p) : 0u;   // (winning priorities are finite and >= 0: their bit patterns order like the values)
  for (int o = 32; o > 0; o >>= 1) { faces += (uint32_t)__shfl_xor((int)faces, o); top = max(top, (uint32_t)__shfl_xor((int)top, o)); }
  uint32_t base = 0;
  if (lane == leader) {
    base = atomicAdd(&counters[0], (uint32_t)__popcll((unsigned long long)wm));
    atomicAdd(&counters[1], faces);
    atomicMax(&counters[2], top);
  }
  base = (uint32_t)__shfl((int)base, leader);
  if (!won) return;
  win[base + (uint32_t)__popcll((unsigned long long)(wm & ((1ull << lane) - 1ull)))] = ((unsigned long long)__float_as_uint(p) << 32) | e;
  for_closed_rings(M, v0, v1, [&](uint32_t u) { taken_next[u] = 1; });
}

// ---- 5. collapse ------------------------------------------------------------------------------------------------------------------------------
__global__ void k_collapse(Mesh M, const uint64_t* __restrict__ ukey, const float* __restrict__ xin, const unsigned long long* __restrict__ win, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t e = (uint32_t)win[i];
  const uint32_t v0 = (uint32_t)(ukey[e] >> 32), v1 = (uint32_t)ukey[e];
  for (uint32_t k = 0; k < M.vcnt[v0]; k++) {
    const uint32_t c = M.corner[M.vbeg[v0] + k], f = c / 3, j = c % 3;
    uint32_t* t = &M.tri[3 * (size_t)f];
    if (t[0] == v1 || t[1] == v1 || t[2] == v1) M.alive[f] = 0;
    else t[j] = v1;
  }
  Quadric q = M.Q[v0];
  q.add(M.Q[v1]);
  M.Q[v1] = q;
  for (int k = 0; k < 3; k++) M.pos[3 * (size_t)v1 + k] = xin[3 * (size_t)e + k];
  M.vdel[v0] = 1;
}

__global__ void k_iota64(unsigned long long* p, unsigned long long v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

SF_API int sf_mesh_simplify_gpu(const sf_mesh* in, const sf_simplify_params* p, int device, sf_mesh** out, sf_simplify_stats* stats) {
  if (!in || !p || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (p->preserve_boundary || p->preserve_normal || p->preserve_topology || p->quality_weight)
    return sf::fail(SF_ERR_UNSUPPORTED, "PreserveBoundary / PreserveNormal / PreserveTopology / QualityWeight are not implemented "
                                        "(simplify.mlx ships them all false)");
  if (!(p->target_perc >= 0.0f && p->target_perc <= 1.0f) || !(p->quality_thr >= 0.0f && p->quality_thr <= 1.0f) || !(p->boundary_weight > 0.0f))
    return sf::fail(SF_ERR_INVALID_ARG, "simplify parameters out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return sf::fail(SF_ERR_DEVICE, "no HIP device: sf_mesh_simplify_gpu needs an MI355X (sf_mesh_simplify is the sequential host filter)");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  DEC_CHECK(hipSetDevice(device));
  const uint32_t V = (uint32_t)(in->pos.size() / 3), F = (uint32_t)(in->tri.size() / 3);
  if (in->tri.size() / 3 > 0x2FFFFFFFull) return sf::fail(SF_ERR_CAPACITY, "mesh too large for 32-bit corner ids");
  sf_simplify_stats st;
  std::memset(&st, 0, sizeof(st));
  st.vertices_in = V;
  st.faces_in = F;
  std::vector<float> pos(in->pos.begin(), in->pos.end());
  std::vector<uint32_t> tri(in->tri.begin(), in->tri.end());
  std::vector<uint8_t> alive_h(F, 1), vdel_h(V, 0);
  uint64_t nalive = 0;
  for (uint32_t v : tri)
    if (v >= V) return sf::fail(SF_ERR_FORMAT, "face references vertex %u of %u", v, V);
  for (uint32_t f = 0; f < F; f++) {   // faces with a repeated vertex cannot take part: dropped up front, as the host filter does
    const uint32_t* t = &tri[3 * (size_t)f];
    if (t[0] == t[1] || t[1] == t[2] || t[0] == t[2]) alive_h[f] = 0;
    else nalive++;
  }
  uint64_t target = p->target_faces;
  if (p->target_perc != 0.0f) target = (uint64_t)((double)F * (double)p->target_perc);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t v = 0; v < V; v++)
    for (int k = 0; k < 3; k++) {
      const double c = pos[3 * (size_t)v + k];
      if (!(c == c) || c > 1e30 || c < -1e30) return sf::fail(SF_ERR_FORMAT, "vertex %u has a non-finite coordinate", v);
      lo[k] = std::min(lo[k], c);
      hi[k] = std::max(hi[k], c);
    }
  double scale = 1.0;
  if (V) {
    const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    scale = diag > 0.0 ? 1e8 * std::pow(1.0 / diag, 6.0) : 1.0;
  }
  if (F > 0 && V > 0 && nalive > target) {
    StreamGuard sg;   // its own stream: host threads finishing several meshes, and the fuser of the next scan, share the device
    DEC_CHECK(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
    hipStream_t s = sg.s;
    const size_t NC0 = 3 * (size_t)F;
    DevBuf d_pos, d_tri, d_alive, d_vdel, d_Q, d_vbeg, d_vcnt, d_ckey, d_cval, d_ckey2, d_cval2, d_ekey, d_ekey2, d_ukey, d_ucnt, d_nruns, d_pri, d_pri2, d_pri3, d_x, d_lock,
        d_win, d_win2, d_nwin, d_taken, d_tmp;
    DEC_CHECK(d_pos.alloc((size_t)V * 12)); DEC_CHECK(d_tri.alloc(NC0 * 4)); DEC_CHECK(d_alive.alloc(F)); DEC_CHECK(d_vdel.alloc(V));
    DEC_CHECK(d_Q.alloc((size_t)V * sizeof(Quadric))); DEC_CHECK(d_vbeg.alloc((size_t)V * 4)); DEC_CHECK(d_vcnt.alloc((size_t)V * 4));
    DEC_CHECK(d_ckey.alloc(NC0 * 4)); DEC_CHECK(d_cval.alloc(NC0 * 4)); DEC_CHECK(d_ckey2.alloc(NC0 * 4)); DEC_CHECK(d_cval2.alloc(NC0 * 4));
    DEC_CHECK(d_ekey.alloc(NC0 * 8)); DEC_CHECK(d_ekey2.alloc(NC0 * 8)); DEC_CHECK(d_ukey.alloc(NC0 * 8)); DEC_CHECK(d_ucnt.alloc(NC0 * 4)); DEC_CHECK(d_nruns.alloc(8));
    DEC_CHECK(d_pri.alloc(NC0 * 4)); DEC_CHECK(d_pri2.alloc(NC0 * 4)); DEC_CHECK(d_pri3.alloc(NC0 * 4)); DEC_CHECK(d_x.alloc(NC0 * 12)); DEC_CHECK(d_lock.alloc((size_t)V * 8));
    DEC_CHECK(d_win.alloc(NC0 * 8)); DEC_CHECK(d_win2.alloc(NC0 * 8)); DEC_CHECK(d_nwin.alloc(16)); DEC_CHECK(d_taken.alloc(V));
    DEC_CHECK(hipMemcpyAsync(d_pos.p, pos.data(), (size_t)V * 12, hipMemcpyHostToDevice, s));
    DEC_CHECK(hipMemcpyAsync(d_tri.p, tri.data(), NC0 * 4, hipMemcpyHostToDevice, s));
    DEC_CHECK(hipMemcpyAsync(d_alive.p, alive_h.data(), F, hipMemcpyHostToDevice, s));
    DEC_CHECK(hipMemsetAsync(d_vdel.p, 0, V, s));
    // scratch for the rocprim calls: sized once for the largest request
    size_t tmp_bytes = 0, need = 0;
    DEC_CHECK(rocprim::radix_sort_pairs(nullptr, need, d_ckey.as<uint32_t>(), d_ckey2.as<uint32_t>(), d_cval.as<uint32_t>(), d_cval2.as<uint32_t>(), NC0, 0, 32, s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(rocprim::radix_sort_keys(nullptr, need, d_ekey.as<uint64_t>(), d_ekey2.as<uint64_t>(), NC0, 0, 64, s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(rocprim::run_length_encode(nullptr, need, d_ekey2.as<uint64_t>(), (unsigned int)NC0, d_ukey.as<uint64_t>(), d_ucnt.as<uint32_t>(), d_nruns.as<uint32_t>(), s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(rocprim::radix_sort_keys(nullptr, need, d_pri.as<uint32_t>(), d_pri2.as<uint32_t>(), NC0, 0, 32, s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(rocprim::radix_sort_keys(nullptr, need, d_win.as<uint64_t>(), d_win2.as<uint64_t>(), NC0, 0, 64, s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(rocprim::select(nullptr, need, d_tri.as<Tri>(), d_alive.as<uint8_t>(), d_ckey.as<Tri>(), d_nruns.as<uint32_t>(), (size_t)F, s));
    tmp_bytes = std::max(tmp_bytes, need);
    DEC_CHECK(d_tmp.alloc(tmp_bytes));
    Mesh M{d_pos.as<float>(), d_tri.as<uint32_t>(), d_alive.as<uint8_t>(), d_vdel.as<uint8_t>(), d_Q.as<Quadric>(), d_vbeg.as<uint32_t>(), d_vcnt.as<uint32_t>(),
           d_cval2.as<uint32_t>(), V, F};
    const unsigned gV = (V + 255) / 256;
    uint32_t Fc = F;   // faces in the device arrays: dead ones are squeezed out (in order) when a quarter has gone, the sorts shrink with the mesh
    bool first = true;
    int stalled = 0;
    constexpr int PASSES = 3;
    std::vector<unsigned long long> win_h;
    std::vector<uint32_t> ucnt_h;
    while (nalive > target) {
      size_t tb = tmp_bytes;
      if (nalive * 4 < (uint64_t)Fc * 3) {
        DEC_CHECK(rocprim::select(d_tmp.p, tb, d_tri.as<Tri>(), d_alive.as<uint8_t>(), d_ckey.as<Tri>(), d_nruns.as<uint32_t>(), (size_t)Fc, s));
        uint32_t kept = 0;
        DEC_CHECK(hipMemcpyAsync(&kept, d_nruns.p, 4, hipMemcpyDeviceToHost, s));
        DEC_CHECK(hipStreamSynchronize(s));
        if (kept != nalive) return sf::fail(SF_ERR_DEVICE, "simplify_gpu: %u faces alive on the device, %llu counted", kept, (unsigned long long)nalive);
        DEC_CHECK(hipMemcpyAsync(d_tri.p, d_ckey.p, (size_t)kept * 12, hipMemcpyDeviceToDevice, s));
        DEC_CHECK(hipMemsetAsync(d_alive.p, 1, kept, s));
        Fc = kept;
        M.F = Fc;
      }
      const size_t NC = 3 * (size_t)Fc;
      const unsigned gF = (Fc + 255) / 256, gC = (unsigned)((NC + 255) / 256);
      // 1. adjacency of the live faces
      hipLaunchKernelGGL(k_emit, dim3(gF), dim3(256), 0, s, M.tri, M.alive, Fc, d_ckey.as<uint32_t>(), d_cval.as<uint32_t>(), d_ekey.as<uint64_t>());
      tb = tmp_bytes;
      DEC_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_ckey.as<uint32_t>(), d_ckey2.as<uint32_t>(), d_cval.as<uint32_t>(), d_cval2.as<uint32_t>(), NC, 0, 32, s));
      DEC_CHECK(hipMemsetAsync(d_vbeg.p, 0, (size_t)V * 4, s));
      DEC_CHECK(hipMemsetAsync(d_vcnt.p, 0, (size_t)V * 4, s));
      hipLaunchKernelGGL(k_vertex_ranges, dim3(gC), dim3(256), 0, s, d_ckey2.as<uint32_t>(), (uint32_t)NC, d_vbeg.as<uint32_t>(), d_vcnt.as<uint32_t>());
      hipLaunchKernelGGL(k_fix_counts, dim3(gV), dim3(256), 0, s, d_vbeg.as<uint32_t>(), d_vcnt.as<uint32_t>(), V);
      tb = tmp_bytes;
      DEC_CHECK(rocprim::radix_sort_keys(d_tmp.p, tb, d_ekey.as<uint64_t>(), d_ekey2.as<uint64_t>(), NC, 0, 64, s));
      tb = tmp_bytes;
      DEC_CHECK(rocprim::run_length_encode(d_tmp.p, tb, d_ekey2.as<uint64_t>(), (unsigned int)NC, d_ukey.as<uint64_t>(), d_ucnt.as<uint32_t>(), d_nruns.as<uint32_t>(), s));
      uint32_t E = 0;
      DEC_CHECK(hipMemcpyAsync(&E, d_nruns.p, 4, hipMemcpyDeviceToHost, s));
      DEC_CHECK(hipStreamSynchronize(s));
      if (E > 0) {   // the run of dead faces' keys (~0) sorts last
        uint64_t lastkey = 0;
        DEC_CHECK(hipMemcpyAsync(&lastkey, d_ukey.as<uint64_t>() + (E - 1), 8, hipMemcpyDeviceToHost, s));
        DEC_CHECK(hipStreamSynchronize(s));
        if (lastkey == ~0ull) E--;
      }
      if (E == 0) break;
      const unsigned gE = (E + 255) / 256;
      if (first) {
        hipLaunchKernelGGL(k_init_quadrics, dim3(gV), dim3(256), 0, s, M, d_ukey.as<uint64_t>(), d_ucnt.as<uint32_t>(), E, p->boundary_weight, p->planar_quadric);
        first = false;
      }
      // 2. priorities
      hipLaunchKernelGGL(k_priority, dim3((E + 127) / 128), dim3(128), 0, s, M, d_ukey.as<uint64_t>(), d_ucnt.as<uint32_t>(), E, scale, p->quality_thr,
                         p->optimal_placement, d_pri.as<float>(), d_x.as<float>());
      // 3. the priority the collapses still needed would reach: a quantile of this round's priorities (half as many again, the rounds overlap)
      const uint64_t needed = (nalive - target + 1) / 2;
      float tau = INFINITY;
      if (stalled == 0) {
        DEC_CHECK(hipMemcpyAsync(d_pri3.p, d_pri.p, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
        tb = tmp_bytes;
        DEC_CHECK(rocprim::radix_sort_keys(d_tmp.p, tb, d_pri3.as<uint32_t>(), d_pri2.as<uint32_t>(), (size_t)E, 0, 32, s));
        const uint64_t kth = std::min<uint64_t>((uint64_t)E - 1, needed + needed / 2 + 64);
        DEC_CHECK(hipMemcpyAsync(&tau, d_pri2.as<float>() + kth, 4, hipMemcpyDeviceToHost, s));
        DEC_CHECK(hipStreamSynchronize(s));
      }
      if (!(tau < 3.0e38f)) tau = 3.0e38f;   // never an edge the link test rejected (infinite priority)
      // 4. winners: up to PASSES independent sets per round, each among the candidates the earlier ones left untouched
      DEC_CHECK(hipMemsetAsync(d_nwin.p, 0, 16, s));
      DEC_CHECK(hipMemsetAsync(d_taken.p, 0, V, s));
      for (int pass = 0; pass < PASSES; pass++) {
        const uint32_t salt = ((uint32_t)st.rounds * PASSES + (uint32_t)pass) * 0x9E3779B9u;
        hipLaunchKernelGGL(k_iota64, dim3(gV), dim3(256), 0, s, d_lock.as<unsigned long long>(), ~0ull, V);
        hipLaunchKernelGGL(k_lock, dim3(gE), dim3(256), 0, s, M, d_ukey.as<uint64_t>(), E, d_pri.as<float>(), tau, salt, d_taken.as<uint8_t>(), d_lock.as<unsigned long long>());
        hipLaunchKernelGGL(k_winners, dim3(gE), dim3(256), 0, s, M, d_ukey.as<uint64_t>(), d_ucnt.as<uint32_t>(), E, d_pri.as<float>(), tau, salt,
                           d_lock.as<unsigned long long>(), d_taken.as<uint8_t>(), d_win.as<unsigned long long>(), d_nwin.as<uint32_t>());
      }
      uint32_t cnt[4] = {0, 0, 0, 0};
      DEC_CHECK(hipMemcpyAsync(cnt, d_nwin.p, 16, hipMemcpyDeviceToHost, s));
      DEC_CHECK(hipStreamSynchronize(s));
      const uint32_t nwin = cnt[0];
      st.rounds++;
      if (nwin == 0) {
        if (stalled++ >= 1) break;   // nothing collapsible even without the threshold
        continue;
      }
      stalled = 0;
      uint32_t take = nwin;
      const unsigned long long* d_take = d_win.as<unsigned long long>();
      if (nalive - cnt[1] >= target) {   // every winner fits under the budget: winners are independent, their order does not matter
        float mp;
        std::memcpy(&mp, &cnt[2], 4);
        if (mp > st.max_priority) st.max_priority = mp;
        nalive -= cnt[1];
      } else {   // the last round: winners in (priority, edge) order until the face count reaches the target
        tb = tmp_bytes;
        DEC_CHECK(rocprim::radix_sort_keys(d_tmp.p, tb, d_win.as<uint64_t>(), d_win2.as<uint64_t>(), (size_t)nwin, 0, 64, s));
        d_take = d_win2.as<unsigned long long>();
        win_h.resize(nwin);
        DEC_CHECK(hipMemcpyAsync(win_h.data(), d_win2.p, (size_t)nwin * 8, hipMemcpyDeviceToHost, s));
        ucnt_h.resize(E);
        DEC_CHECK(hipMemcpyAsync(ucnt_h.data(), d_ucnt.p, (size_t)E * 4, hipMemcpyDeviceToHost, s));
        DEC_CHECK(hipStreamSynchronize(s));
        uint64_t removed = 0;
        take = 0;
        while (take < nwin && nalive - removed > target) {
          removed += ucnt_h[(uint32_t)win_h[take]];
          take++;
        }
        float mp;
        const uint32_t top = (uint32_t)(win_h[take - 1] >> 32);
        std::memcpy(&mp, &top, 4);
        if (mp > st.max_priority) st.max_priority = mp;
        nalive -= removed;
      }
      st.collapses += take;
      // 5. collapse
      hipLaunchKernelGGL(k_collapse, dim3((take + 255) / 256), dim3(256), 0, s, M, d_ukey.as<uint64_t>(), d_x.as<float>(), d_take, take);
      DEC_CHECK(hipGetLastError());
    }
    DEC_CHECK(hipStreamSynchronize(s));
    DEC_CHECK(hipMemcpyAsync(pos.data(), d_pos.p, (size_t)V * 12, hipMemcpyDeviceToHost, s));
    tri.resize(3 * (size_t)Fc);
    alive_h.resize(Fc);
    DEC_CHECK(hipMemcpyAsync(tri.data(), d_tri.p, (size_t)Fc * 12, hipMemcpyDeviceToHost, s));
    DEC_CHECK(hipMemcpyAsync(alive_h.data(), d_alive.p, Fc, hipMemcpyDeviceToHost, s));
    DEC_CHECK(hipMemcpyAsync(vdel_h.data(), d_vdel.p, V, hipMemcpyDeviceToHost, s));
    DEC_CHECK(hipStreamSynchronize(s));
  }
  const size_t Fh = alive_h.size();   // the device squeezes dead faces out of its arrays as it goes
  std::vector<uint8_t> fdel(Fh);
  uint64_t nf = 0;
  for (size_t f = 0; f < Fh; f++) { fdel[f] = alive_h[f] ? 0 : 1; nf += alive_h[f] ? 1 : 0; }
  sf_mesh* m = simplify_finish(in, *p, pos, tri, fdel, vdel_h, nf, st);
  st.target_faces = target;
  if (stats) *stats = st;
  *out = m;
  return SF_OK;
}

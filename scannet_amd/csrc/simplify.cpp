// simplify.cpp -- "Quadric Edge Collapse Decimation", the first filter of the `decimate` stage's simplify.mlx.
//
// Replaces the two `meshlabserver -i X_vh_clean.ply -o X_vh_clean_1.ply -m vc -s simplify.mlx` calls of
// Server/scan_processor.py:144-145 (each keeps 20 % of the faces: Server/tools/meshclean/simplify.mlx:3-16; the four
// cleaning filters that follow in the script, :17-24, are clean.cpp).  Parameters as shipped: TargetFaceNum 0,
// TargetPerc 0.2, QualityThr 0.3, PreserveBoundary false, BoundaryWeight 1, PreserveNormal false, PreserveTopology
// false, OptimalPlacement true, PlanarQuadric false, QualityWeight false, AutoClean true, Selected false.
//
// MeshLab / VCG are not in the reference tree and no version is pinned (Server/config.py:19), so this restates the
// published VCG algorithm (vcg/complex/algorithms/local_optimization/tri_edge_collapse_quadric.h, quadric.h,
// edge_collapse.h, local_optimization.h) -- PARITY UNPINNED, checked by properties (tests/test_simplify.py):
//   * quadrics in double, one per vertex: sum over incident faces of the plane quadric with the UN-normalised normal
//     (UseArea: weight (2 area)^2); every border edge adds the quadric of the plane through the edge orthogonal to its
//     face, scaled by BoundaryQuadricWeight (0.5 x BoundaryWeight); PlanarQuadric adds the same for every edge at 1/100;
//   * a collapse (v0 -> v1) has priority  ScaleFactor * Q(x) / min(QualityThr, worst quality of the faces around the
//     pair after the move),  Q = Q0 + Q1, x = the minimiser of Q (OptimalPlacement) or v1's position, face quality =
//     2 area / (longest edge)^2, ScaleFactor = 1e8 / diag^6 (ScaleIndependent), priority floor 1e-15 (QuadricEpsilon);
//   * a min-heap (4-ary here) of all edges once (lower index = v0, symmetric because placement is optimal); an entry
//     is stale when either vertex died or was touched after the entry was made (time stamps);
//     executing a collapse deletes the faces holding both vertices, re-points v0's other faces to v1 (prepended to
//     v1's face list as VCG does), moves v1 to x, gives v1 the summed quadric and v1's own colour, and pushes fresh
//     entries (v1, w) for every neighbour w; the heap is rebuilt from its live entries when it outgrows 3 x faces;
//   * until the face count reaches the target; then AutoClean: zero-area faces, duplicate vertices, unreferenced
//     vertices removed, compaction in index order.
// One deliberate difference: where Q's 3x3 system is rank deficient (flat areas), VCG's full-pivot LU returns a point
// with the free coordinates at ZERO -- the "bad spikes in very flat areas" its own tooltip warns of (simplify.mlx:12).
// Here the minimiser closest to the edge midpoint is taken (pseudo-inverse on the eigen-decomposition, eigenvalues
// below 1e-9 of the largest treated as zero): identical wherever the system is well conditioned, no spikes on walls.
// Sequential by nature (a global priority order); ~0.1 M collapses per second on one host core (2 M faces -> 0.4 M in ~10 s).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mesh.h"
#include "simplify_math.h"

namespace {

using sfq::Quadric;
using sfq::minimise;

struct HeapElem {
  float pri;
  uint32_t v0, v1;
  uint32_t mark;
};
// The order collapses are taken in: by priority, ties by (v0, v1, mark) -- a total order, so the sequence of collapses is a property
// of the mesh and the parameters, not of the queue's internals.
inline bool elem_less(const HeapElem& a, const HeapElem& b) {
  if (a.pri != b.pri) return a.pri < b.pri;
  if (a.v0 != b.v0) return a.v0 < b.v0;
  if (a.v1 != b.v1) return a.v1 < b.v1;
  return a.mark < b.mark;
}
// 4-ary min-heap (four 16-byte children share one cache line: about half the misses of a binary heap)
struct Heap4 {
  std::vector<HeapElem> h;
  bool empty() const { return h.empty(); }
  size_t size() const { return h.size(); }
  void push(const HeapElem& e) {
    size_t i = h.size();
    h.push_back(e);
    while (i > 0) {
      const size_t p = (i - 1) >> 2;
      if (!elem_less(e, h[p])) break;
      h[i] = h[p];
      i = p;
    }
    h[i] = e;
  }
  void sift_down(size_t i, const HeapElem& e) {
    const size_t n = h.size();
    for (;;) {
      const size_t c0 = 4 * i + 1;
      if (c0 >= n) break;
      size_t m = c0;
      const size_t ce = std::min(c0 + 4, n);
      for (size_t c = c0 + 1; c < ce; c++)
        if (elem_less(h[c], h[m])) m = c;
      if (!elem_less(h[m], e)) break;
      h[i] = h[m];
      i = m;
    }
    h[i] = e;
  }
  HeapElem pop() {
    const HeapElem top = h[0], last = h.back();
    h.pop_back();
    if (!h.empty()) sift_down(0, last);
    return top;
  }
  void heapify() {  // Floyd
    const size_t n = h.size();
    if (n < 2) return;
    for (size_t s = (n - 2) / 4 + 1; s-- > 0;) { const HeapElem e = h[s]; sift_down(s, e); }
  }
};

// The queue of a scan-sized mesh holds tens of millions of entries, and a heap that large misses the cache on every level of every
// pop -- microseconds per collapse.  Only the entries that can be popped soon need a heap: `hot` holds everything with priority <=
// threshold (a few MiB, cache-resident), `cold` is an unsorted append-only array of the rest.  When `hot` runs dry the stale entries of
// `cold` are dropped, a new threshold is taken from a sample of its priorities so that HOT_TARGET entries or a sixth of `cold` qualify,
// and those move over.  Every entry <= threshold is in the heap at all times, so the pop order is exactly that of one big heap under elem_less.
struct LadderQueue {
  static constexpr size_t HOT_TARGET = 1u << 19;
  Heap4 hot;
  std::vector<HeapElem> cold;
  float threshold = -INFINITY;
  size_t size() const { return hot.size() + cold.size(); }
  void push(const HeapElem& e) {
    if (e.pri <= threshold) hot.push(e);
    else cold.push_back(e);
  }
  // drop what `keep` rejects everywhere (the stale entries); rebuild the heap
  template <class Keep>
  void compact(Keep keep) {
    size_t w = 0;
    for (size_t i = 0; i < hot.h.size(); i++) if (keep(hot.h[i])) hot.h[w++] = hot.h[i];
    hot.h.resize(w);
    hot.heapify();
    w = 0;
    for (size_t i = 0; i < cold.size(); i++) if (keep(cold[i])) cold[w++] = cold[i];
    cold.resize(w);
  }
  // make `hot` non-empty if anything is left; false when the queue is exhausted
  template <class Keep>
  bool refill(Keep keep) {
    while (hot.empty()) {
      size_t w = 0;
      for (size_t i = 0; i < cold.size(); i++) if (keep(cold[i])) cold[w++] = cold[i];
      cold.resize(w);
      if (cold.empty()) return false;
      // how much moves over: at least HOT_TARGET, and a sixth of what is left -- every refill reads all of `cold`, so the number of
      // refills has to stay logarithmic in its size (a fixed HOT_TARGET meant ~85 passes over 6 M entries on a 2 M-face mesh)
      const size_t want = std::max(HOT_TARGET, cold.size() / 6);
      if (cold.size() <= 2 * want) threshold = INFINITY;
      else {   // the want / size quantile of an evenly spaced sample of 4096 priorities
        std::vector<float> sample(4096);
        for (size_t k = 0; k < sample.size(); k++) sample[k] = cold[(size_t)((double)k * (double)cold.size() / (double)sample.size())].pri;
        const size_t q = std::min(sample.size() - 1, (size_t)((double)sample.size() * (double)want / (double)cold.size()));
        std::nth_element(sample.begin(), sample.begin() + (long)q, sample.end());
        threshold = sample[q];
      }
      w = 0;
      for (size_t i = 0; i < cold.size(); i++) {
        if (cold[i].pri <= threshold) hot.h.push_back(cold[i]);
        else cold[w++] = cold[i];
      }
      cold.resize(w);
      hot.heapify();
    }
    return true;
  }
};

struct Simplifier {
  const sf_simplify_params& P;
  std::vector<float> pos;
  std::vector<uint32_t> tri;
  std::vector<uint8_t> fdel, vdel, visited;
  std::vector<Quadric> Q;
  std::vector<uint32_t> imark;
  // VF adjacency as VCG keeps it: an intrusive singly linked list per vertex through the (face, corner) slots
  std::vector<int32_t> vf_face;   // head per vertex: face or -1
  std::vector<uint8_t> vf_idx;    // head per vertex: corner
  std::vector<int32_t> nx_face;   // per (face, corner): next face or -1
  std::vector<uint8_t> nx_idx;
  LadderQueue heap;
  std::vector<std::pair<int32_t, int>> both, only0;   // scratch of collapse()
  uint32_t global_mark = 0;
  uint64_t nfaces = 0;
  double scale = 1.0;
  sf_simplify_stats st;

  explicit Simplifier(const sf_simplify_params& p) : P(p) { std::memset(&st, 0, sizeof(st)); }

  size_t nv() const { return pos.size() / 3; }
  void pd(uint32_t v, double p[3]) const { p[0] = pos[3 * v]; p[1] = pos[3 * v + 1]; p[2] = pos[3 * v + 2]; }

  static float quality(const float* p0, const float* p1, const float* p2) { return sfq::quality(p0, p1, p2); }

  void vf_prepend(uint32_t v, int32_t f, int j) {
    nx_face[3 * (size_t)f + j] = vf_face[v];
    nx_idx[3 * (size_t)f + j] = vf_idx[v];
    vf_face[v] = f;
    vf_idx[v] = (uint8_t)j;
  }
  void vf_detach(int32_t f, int j) {  // remove (f, j) from the list of its vertex
    const uint32_t v = tri[3 * (size_t)f + j];
    if (vf_face[v] == f && vf_idx[v] == j) {
      vf_face[v] = nx_face[3 * (size_t)f + j];
      vf_idx[v] = nx_idx[3 * (size_t)f + j];
      return;
    }
    int32_t cf = vf_face[v];
    int cj = vf_idx[v];
    while (cf >= 0) {
      const int32_t nf = nx_face[3 * (size_t)cf + cj];
      const int nj = nx_idx[3 * (size_t)cf + cj];
      if (nf == f && nj == j) {
        nx_face[3 * (size_t)cf + cj] = nx_face[3 * (size_t)f + j];
        nx_idx[3 * (size_t)cf + cj] = nx_idx[3 * (size_t)f + j];
        return;
      }
      cf = nf;
      cj = nj;
    }
  }

  void optimal(uint32_t v0, uint32_t v1, const Quadric& q, float out[3]) const {
    if (!P.optimal_placement) { out[0] = pos[3 * v1]; out[1] = pos[3 * v1 + 1]; out[2] = pos[3 * v1 + 2]; return; }
    double p0[3], p1[3], mid[3], x[3];
    pd(v0, p0); pd(v1, p1);
    for (int i = 0; i < 3; i++) mid[i] = 0.5 * (p0[i] + p1[i]);
    minimise(q, mid, x);
    for (int i = 0; i < 3; i++) out[i] = (float)x[i];
    if (!(out[0] == out[0] && out[1] == out[1] && out[2] == out[2])) {  // NaN guard: best of the three candidates
      const double qm = q.apply(mid), q0 = q.apply(p0), q1 = q.apply(p1);
      const double* best = mid;
      if (q0 < qm) best = p0;
      if (q1 < qm && q1 < q0) best = p1;
      for (int i = 0; i < 3; i++) out[i] = (float)best[i];
    }
  }

  float priority(uint32_t v0, uint32_t v1) const {
    Quadric q = Q[v0];
    q.add(Q[v1]);
    float x[3];
    optimal(v0, v1, q, x);
    double min_qual = 1e300;
    // the faces around v0 that do not hold v1, with v0 at x; then the same around v1
    for (int side = 0; side < 2; side++) {
      const uint32_t a = side ? v1 : v0, other = side ? v0 : v1;
      int32_t f = vf_face[a];
      int j = vf_idx[a];
      while (f >= 0) {
        const uint32_t* t = &tri[3 * (size_t)f];
        if (t[0] != other && t[1] != other && t[2] != other) {
          const float* p[3];
          for (int k = 0; k < 3; k++) p[k] = t[k] == a ? x : &pos[3 * (size_t)t[k]];
          const double qt = quality(p[0], p[1], p[2]);
          if (qt < min_qual) min_qual = qt;
        }
        const int32_t nf = nx_face[3 * (size_t)f + j];
        j = nx_idx[3 * (size_t)f + j];
        f = nf;
      }
    }
    const double xd[3] = {x[0], x[1], x[2]};
    double err = scale * q.apply(xd);
    if (min_qual > P.quality_thr) min_qual = P.quality_thr;
    if (err < 1e-15) err = 1e-15;  // QuadricEpsilon
    if (P.quality_thr > 0.0f) err = min_qual > 0.0 ? err / min_qual : 1e300;
    return err > 3.0e38 ? 3.0e38f : (float)err;
  }

  void push(uint32_t v0, uint32_t v1) {
    heap.push(HeapElem{priority(v0, v1), v0, v1, global_mark});
  }

  int init(const sf_mesh* in) {
    pos.assign(in->pos.begin(), in->pos.end());
    tri.assign(in->tri.begin(), in->tri.end());
    const size_t n_v = nv(), n_f = tri.size() / 3;
    for (uint32_t v : tri)
      if (v >= n_v) return sf::fail(SF_ERR_FORMAT, "face references vertex %u of %zu", v, n_v);
    fdel.assign(n_f, 0);
    vdel.assign(n_v, 0);
    visited.assign(n_v, 0);
    imark.assign(n_v, 0);
    Q.resize(n_v);
    for (Quadric& q : Q) q.zero();
    vf_face.assign(n_v, -1);
    vf_idx.assign(n_v, 0);
    nx_face.assign(3 * n_f, -1);
    nx_idx.assign(3 * n_f, 0);
    // faces with a repeated vertex cannot take part (VCG would assert): dropped up front
    for (size_t f = 0; f < n_f; f++) {
      const uint32_t* t = &tri[3 * f];
      if (t[0] == t[1] || t[1] == t[2] || t[0] == t[2]) { fdel[f] = 1; continue; }
      nfaces++;
      for (int j = 0; j < 3; j++) vf_prepend(t[j], (int32_t)f, j);  // UpdateTopology::VertexFace: each face prepends itself
    }
    // border edges: exactly one incident face (counted through the VF list of the edge's first vertex)
    auto edge_faces = [&](uint32_t a, uint32_t b) {
      int count = 0;
      int32_t f = vf_face[a];
      int j = vf_idx[a];
      while (f >= 0) {
        const uint32_t* t = &tri[3 * (size_t)f];
        count += t[0] == b || t[1] == b || t[2] == b;
        const int32_t nf = nx_face[3 * (size_t)f + j];
        j = nx_idx[3 * (size_t)f + j];
        f = nf;
      }
      return count;
    };
    // InitQuadric
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    std::vector<uint8_t> referenced(n_v, 0);
    for (size_t f = 0; f < n_f; f++) {
      if (fdel[f]) continue;
      const uint32_t* t = &tri[3 * f];
      double p0[3], p1[3], p2[3];
      pd(t[0], p0); pd(t[1], p1); pd(t[2], p2);
      const double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      Quadric q;
      q.by_plane(n, n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
      for (int j = 0; j < 3; j++) { Q[t[j]].add(q); referenced[t[j]] = 1; }
      for (int j = 0; j < 3; j++) {
        const bool border = edge_faces(t[j], t[(j + 1) % 3]) == 1;
        if (!border && !P.planar_quadric) continue;
        double pa[3], pb[3];
        pd(t[j], pa); pd(t[(j + 1) % 3], pb);
        double d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(dl > 0.0)) continue;
        for (int k = 0; k < 3; k++) d[k] /= dl;
        const double wgt = border ? 0.5 * (double)P.boundary_weight : 0.5 * (double)P.boundary_weight / 100.0;
        const double bn[3] = {(n[1] * d[2] - n[2] * d[1]) * wgt, (n[2] * d[0] - n[0] * d[2]) * wgt, (n[0] * d[1] - n[1] * d[0]) * wgt};
        Quadric bq;
        bq.by_plane(bn, bn[0] * pa[0] + bn[1] * pa[1] + bn[2] * pa[2]);
        Q[t[j]].add(bq);
        Q[t[(j + 1) % 3]].add(bq);
      }
    }
    for (size_t v = 0; v < n_v; v++) {  // UpdateBounding::Box: every vertex of the container
      for (int k = 0; k < 3; k++) {
        const double c = pos[3 * v + k];
        if (!(c == c) || c > 1e30 || c < -1e30) return sf::fail(SF_ERR_FORMAT, "vertex %zu has a non-finite coordinate", v);
        lo[k] = std::min(lo[k], c);
        hi[k] = std::max(hi[k], c);
      }
    }
    if (n_v) {
      const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
      scale = diag > 0.0 ? 1e8 * std::pow(1.0 / diag, 6.0) : 1.0;
    }
    // the heap: every edge once, (lower index, higher index); neighbours enumerated through the VF list as VCG does.  The mesh
    // is read-only here, so the ~1.5 priorities per face of the start (a third of all priority evaluations of a run) are
    // computed by all host threads over contiguous vertex ranges and concatenated in vertex order: same heap content as the
    // sequential loop
    {
      int nt = sf::usable_cpus();   // the cgroup quota, not the logical CPUs the container shows
      nt = std::max(1, std::min(nt, 64));
      if (n_v < 20000) nt = 1;
      std::vector<std::vector<HeapElem>> part((size_t)nt);
      auto work = [&](int t) {
        const uint32_t lo_v = (uint32_t)((uint64_t)n_v * (uint64_t)t / (uint64_t)nt), hi_v = (uint32_t)((uint64_t)n_v * (uint64_t)(t + 1) / (uint64_t)nt);
        std::vector<HeapElem>& out = part[(size_t)t];
        out.reserve((size_t)(hi_v - lo_v) * 3 + 16);
        std::vector<uint32_t> seen;  // neighbours of v already pushed (valence is small: linear search)
        for (uint32_t v = lo_v; v < hi_v; v++) {
          if (vf_face[v] < 0) continue;
          seen.clear();
          int32_t f = vf_face[v];
          int j = vf_idx[v];
          while (f >= 0) {
            const uint32_t w[2] = {tri[3 * (size_t)f + (j + 1) % 3], tri[3 * (size_t)f + (j + 2) % 3]};
            for (int q = 0; q < 2; q++) {
              if (!(v < w[q])) continue;
              bool dup = false;
              for (uint32_t u : seen) dup = dup || u == w[q];
              if (dup) continue;
              seen.push_back(w[q]);
              out.push_back(HeapElem{priority(v, w[q]), v, w[q], 0});
            }
            const int32_t nf = nx_face[3 * (size_t)f + j];
            j = nx_idx[3 * (size_t)f + j];
            f = nf;
          }
        }
      };
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
      work(0);
      for (std::thread& th : pool) th.join();
      size_t total = 0;
      for (const auto& v : part) total += v.size();
      heap.cold.reserve(std::max(total + 16, 3 * (size_t)nfaces + 4096 + 16));
      for (const auto& v : part) heap.cold.insert(heap.cold.end(), v.begin(), v.end());
    }
    return SF_OK;
  }

  bool up_to_date(const HeapElem& h) const { return !vdel[h.v0] && !vdel[h.v1] && h.mark >= imark[h.v0] && h.mark >= imark[h.v1]; }

  void collapse(uint32_t v0, uint32_t v1) {
    Quadric q = Q[v0];
    q.add(Q[v1]);
    float x[3];
    optimal(v0, v1, q, x);
    Q[v1] = q;
    // FindSets over VF(v0): faces with both vertices die, the others are re-pointed
    both.clear();
    only0.clear();
    {
      int32_t f = vf_face[v0];
      int j = vf_idx[v0];
      while (f >= 0) {
        const uint32_t* t = &tri[3 * (size_t)f];
        if (t[0] == v1 || t[1] == v1 || t[2] == v1) both.emplace_back(f, j);
        else only0.emplace_back(f, j);
        const int32_t nf = nx_face[3 * (size_t)f + j];
        j = nx_idx[3 * (size_t)f + j];
        f = nf;
      }
    }
    for (auto& fj : both) {
      vf_detach(fj.first, (fj.second + 1) % 3);
      vf_detach(fj.first, (fj.second + 2) % 3);
      fdel[fj.first] = 1;
      nfaces--;
    }
    for (auto& fj : only0) {
      tri[3 * (size_t)fj.first + fj.second] = v1;
      vf_prepend(v1, fj.first, fj.second);
    }
    vf_face[v0] = -1;
    vdel[v0] = 1;
    pos[3 * v1] = x[0]; pos[3 * v1 + 1] = x[1]; pos[3 * v1 + 2] = x[2];
    st.collapses++;
    // UpdateHeap
    global_mark++;
    imark[v1] = global_mark;
    for (int pass = 0; pass < 2; pass++) {
      int32_t f = vf_face[v1];
      int j = vf_idx[v1];
      while (f >= 0) {
        const uint32_t w1 = tri[3 * (size_t)f + (j + 1) % 3], w2 = tri[3 * (size_t)f + (j + 2) % 3];
        if (pass == 0) {
          visited[w1] = 0; visited[w2] = 0;
          // the second pass evaluates one collapse per neighbour: start fetching what it will read (quadric, fan head, first fan face)
          for (uint32_t w : {w1, w2}) {
            __builtin_prefetch(&Q[w]);
            __builtin_prefetch(reinterpret_cast<const char*>(&Q[w]) + 64);
            const int32_t wf = vf_face[w];
            if (wf >= 0) { __builtin_prefetch(&tri[3 * (size_t)wf]); __builtin_prefetch(&nx_face[3 * (size_t)wf]); }
          }
        }
        else {
          if (!visited[w1]) { visited[w1] = 1; push(v1, w1); }
          if (!visited[w2]) { visited[w2] = 1; push(w2, v1); }
        }
        const int32_t nf = nx_face[3 * (size_t)f + j];
        j = nx_idx[3 * (size_t)f + j];
        f = nf;
      }
    }
  }

  void run(uint64_t target) {
    const auto keep = [this](const HeapElem& e) { return up_to_date(e); };
    while (nfaces > target) {
      if (heap.size() > 3 * (size_t)nfaces + 4096) heap.compact(keep);  // ClearHeap: drop the stale entries
      if (!heap.refill(keep)) break;
      const HeapElem h = heap.hot.pop();
      // whichever entry comes next is among the first few of the array: start fetching what its validity test and its collapse read
      for (size_t c = 0; c < 5 && c < heap.hot.h.size(); c++) {
        const HeapElem& nx = heap.hot.h[c];
        __builtin_prefetch(&imark[nx.v0]); __builtin_prefetch(&imark[nx.v1]);
        __builtin_prefetch(&vdel[nx.v0]); __builtin_prefetch(&vdel[nx.v1]);
        if (c < 2) { __builtin_prefetch(&Q[nx.v0]); __builtin_prefetch(&Q[nx.v1]); __builtin_prefetch(&vf_face[nx.v0]); __builtin_prefetch(&vf_face[nx.v1]); }
      }
      if (!up_to_date(h)) { st.stale_popped++; continue; }
      if (h.pri > st.max_priority) st.max_priority = h.pri;
      collapse(h.v0, h.v1);
    }
  }
};

}  // namespace

// The end of the filter, shared by the sequential and the GPU variant: AutoClean (zero-area faces, duplicate vertices -- bit-identical
// positions -> lowest index --, unreferenced vertices), compaction in index order, colours travel with the surviving vertex.
sf_mesh* simplify_finish(const sf_mesh* in, const sf_simplify_params& P, std::vector<float>& pos, std::vector<uint32_t>& tri_in, std::vector<uint8_t>& fdel,
                         std::vector<uint8_t>& vdel, uint64_t nfaces, sf_simplify_stats& st) {
  // AutoClean: zero-area faces, duplicate vertices (bit-identical positions -> lowest index), unreferenced vertices
  const size_t n_v = (pos.size() / 3), n_f = tri_in.size() / 3;
  std::vector<uint32_t> target_of(n_v);
  for (size_t v = 0; v < n_v; v++) target_of[v] = (uint32_t)v;
  if (P.auto_clean) {
    for (size_t f = 0; f < n_f; f++) {
      if (fdel[f]) continue;
      const uint32_t* t = &tri_in[3 * f];
      const float *a = &pos[3 * (size_t)t[0]], *b = &pos[3 * (size_t)t[1]], *c = &pos[3 * (size_t)t[2]];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      if (!(std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 0.0f)) { fdel[f] = 1; nfaces--; st.faces_zero_area++; }
    }
    // duplicate vertices: bit-identical positions (-0.0 == +0.0) go to the lowest index.  One pass in index order over an open-addressing
    // table keyed by the canonical 12 bytes (a sort by position did the same in O(n log n): 0.15 s of a scan's decimation on the GPU path)
    size_t live = 0;
    for (size_t v = 0; v < n_v; v++) live += vdel[v] ? 0 : 1;
    size_t cap = 16;
    while (cap < 2 * live + 2) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0xFFFFFFFFu);
    auto canon = [&](uint32_t v, uint32_t* q) {
      for (int c = 0; c < 3; c++) { const float f = pos[3 * (size_t)v + c] + 0.0f; std::memcpy(&q[c], &f, 4); }
    };
    for (size_t v = 0; v < n_v; v++) {
      if (vdel[v]) continue;
      uint32_t a[3], b[3];
      canon((uint32_t)v, a);
      uint64_t h = ((uint64_t)a[0] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)a[1] * 0xC2B2AE3D27D4EB4Full) ^ ((uint64_t)a[2] * 0x165667B19E3779F9ull);
      h ^= h >> 29;
      size_t i = (size_t)(h * 0xBF58476D1CE4E5B9ull >> 20) & (cap - 1);
      for (;;) {
        if (slot[i] == 0xFFFFFFFFu) { slot[i] = (uint32_t)v; break; }   // first (lowest) index with this position
        canon(slot[i], b);
        if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) { target_of[v] = slot[i]; st.vertices_duplicate++; break; }
        i = (i + 1) & (cap - 1);
      }
    }
  }
  std::vector<uint8_t> used(n_v, 0);
  std::vector<uint32_t> tri;
  tri.reserve(3 * (size_t)nfaces);
  for (size_t f = 0; f < n_f; f++) {
    if (fdel[f]) continue;
    const uint32_t a = target_of[tri_in[3 * f]], b = target_of[tri_in[3 * f + 1]], c = target_of[tri_in[3 * f + 2]];
    if (P.auto_clean && (a == b || b == c || a == c)) { st.faces_zero_area++; continue; }
    tri.push_back(a); tri.push_back(b); tri.push_back(c);
    used[a] = used[b] = used[c] = 1;
  }
  std::vector<uint32_t> remap(n_v, 0xFFFFFFFFu);
  uint32_t w = 0;
  for (size_t v = 0; v < n_v; v++) {
    const bool keep = P.auto_clean ? used[v] != 0 : (!vdel[v]);
    if (keep) remap[v] = w++;
  }
  sf_mesh* m = new sf_mesh();
  m->pos.resize(3 * (size_t)w);
  if (!in->col.empty()) m->col.resize(4 * (size_t)w);
  for (size_t v = 0; v < n_v; v++) {
    if (remap[v] == 0xFFFFFFFFu) continue;
    std::memcpy(&m->pos[3 * (size_t)remap[v]], &pos[3 * v], 12);
    if (!in->col.empty()) std::memcpy(&m->col[4 * (size_t)remap[v]], &in->col[4 * v], 4);
  }
  m->tri.resize(tri.size());
  for (size_t i = 0; i < tri.size(); i++) m->tri[i] = remap[tri[i]];
  st.vertices_out = w;
  st.faces_out = tri.size() / 3;
  return m;
}

SF_API void sf_simplify_default_params(sf_simplify_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->target_faces = 0;
  p->target_perc = 0.2f;        // simplify.mlx:5
  p->quality_thr = 0.3f;        // :6
  p->preserve_boundary = 0;     // :7
  p->boundary_weight = 1.0f;    // :8
  p->preserve_normal = 0;       // :9
  p->preserve_topology = 0;     // :10
  p->optimal_placement = 1;     // :11
  p->planar_quadric = 0;        // :12
  p->quality_weight = 0;        // :13
  p->auto_clean = 1;            // :14
}

// scanfuse_internal.h: the quadric, the optimal position and the priority the filter assigns to the collapse v0 -> v1 of `in` BEFORE any
// collapse has happened -- what tests/test_simplify_known_answers.py checks against values derived by hand.
SF_API int sf_simplify_probe_edge(const sf_mesh* in, const sf_simplify_params* p, uint32_t v0, uint32_t v1, double quadric10[10], float position[3],
                                  float* priority, double* scale_factor) {
  if (!in || !p) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  Simplifier S(*p);
  const int rc = S.init(in);
  if (rc != SF_OK) return rc;
  if (v0 >= S.nv() || v1 >= S.nv() || v0 == v1) return sf::fail(SF_ERR_INVALID_ARG, "sf_simplify_probe_edge: vertices %u, %u of %zu", v0, v1, S.nv());
  Quadric q = S.Q[v0];
  q.add(S.Q[v1]);
  if (quadric10) {
    for (int i = 0; i < 6; i++) quadric10[i] = q.a[i];
    for (int i = 0; i < 3; i++) quadric10[6 + i] = q.b[i];
    quadric10[9] = q.c;
  }
  if (position) S.optimal(v0, v1, q, position);
  if (priority) *priority = S.priority(v0, v1);
  if (scale_factor) *scale_factor = S.scale;
  return SF_OK;
}

SF_API int sf_mesh_simplify(const sf_mesh* in, const sf_simplify_params* p, sf_mesh** out, sf_simplify_stats* stats) {
  if (!in || !p || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (p->preserve_boundary || p->preserve_normal || p->preserve_topology || p->quality_weight)
    return sf::fail(SF_ERR_UNSUPPORTED, "PreserveBoundary / PreserveNormal / PreserveTopology / QualityWeight are not implemented "
                                        "(simplify.mlx ships them all false)");
  if (!(p->target_perc >= 0.0f && p->target_perc <= 1.0f) || !(p->quality_thr >= 0.0f && p->quality_thr <= 1.0f) || !(p->boundary_weight > 0.0f))
    return sf::fail(SF_ERR_INVALID_ARG, "simplify parameters out of range");
  Simplifier S(*p);
  const int rc = S.init(in);
  if (rc != SF_OK) return rc;
  S.st.vertices_in = S.nv();
  S.st.faces_in = in->tri.size() / 3;
  // MeshLab: TargetFaceNum, or TargetPerc x fn when that is non-zero
  uint64_t target = p->target_faces;
  if (p->target_perc != 0.0f) target = (uint64_t)((double)(in->tri.size() / 3) * (double)p->target_perc);
  S.run(target);
  sf_mesh* m = simplify_finish(in, *p, S.pos, S.tri, S.fdel, S.vdel, S.nfaces, S.st);
  S.st.target_faces = target;
  if (stats) *stats = S.st;
  *out = m;
  return SF_OK;
}

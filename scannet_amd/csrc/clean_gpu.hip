// clean_gpu.hip -- the four cleaning filters of clean.mlx / cleanLoRes.mlx (Server/tools/meshclean/clean.mlx:3-10) on the GPU, an opt-in beside
// clean.cpp with IDENTICAL output (same arrays, same statistics): on a scan-sized mesh (7.9 M faces) the host filters are 1.7 s of one
// thread -- the largest host stage of a scan once the decimation runs on the GPU.
//
//   1. Merge Close Vertices.  VCG clusters greedily in index order: vertex i, if no earlier centre has absorbed it, becomes a centre and
//      absorbs every not yet absorbed vertex closer than the threshold (original positions on both sides; float distance as clean.cpp
//      computes it).  Hence: j is absorbed by the LOWEST-index centre among its lower-index neighbours L(j) = {i < j, |p_i - p_j| < r}, and
//      is itself a centre iff L(j) holds no centre.  That recursion only looks down the index order, so it is resolved in rounds: a vertex
//      settles once all of L(j) has (round 0 settles everybody with empty L(j): on a marching-cubes mesh almost all vertices).  Neighbours
//      come from a uniform grid of cell 2r (vertices sorted by cell key, the eight cells on the point's side of its cell).  r == 0:
//      bit-identical positions merge into the lowest index (two stable sorts).
//   2. Remove Duplicate Faces: same vertex set, lowest face index survives -- two stable radix sorts of the sorted triples, first of a run.
//   3. Remove Isolated pieces: faces sharing an edge are connected (every face of a non-manifold edge too): edges sorted by key, faces of
//      a run linked in a lock-free union-find (larger root hooked under the smaller by CAS, path halving), component sizes by atomics.
//   4. Remove Unreferenced Vertex + compaction in index order: flags, exclusive scan, gather.
// Every compaction is a stable select, so vertex and face order are those of the host filter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "hip_util.h"
#include "mesh.h"
#include "scanfuse.h"

namespace {

using sf::DevBuf;
using sf::StreamGuard;

struct Tri { uint32_t a, b, c; };

#define CL_CHECK(call)                                                                                                                    \
  do {                                                                                                                                    \
    hipError_t e_ = (call);                                                                                                               \
    if (e_ != hipSuccess) return sf::fail(SF_ERR_DEVICE, "%s failed: %s (clean_gpu.hip:%d)", #call, hipGetErrorString(e_), __LINE__);    \
  } while (0)

constexpr uint32_t UNSETTLED = 0xFFFFFFFFu;

struct Grid {
  double inv;        // cells per metre
  int64_t base[3];   // cell index of the lowest corner, one cell of slack
};
__host__ __device__ inline void cell_of(const Grid& G, const float* q, int64_t* c3, int64_t* side) {
  for (int c = 0; c < 3; c++) {
    const double u = (double)q[c] * G.inv, fl = floor(u);
    c3[c] = (int64_t)fl - G.base[c];
    side[c] = (u - fl) < 0.5 ? -1 : 1;
  }
}
__host__ __device__ inline uint64_t pack_cell(int64_t x, int64_t y, int64_t z) { return ((uint64_t)z << 42) | ((uint64_t)y << 21) | (uint64_t)x; }

__global__ void k_cell_keys(const float* __restrict__ pos, uint32_t nv, Grid G, uint64_t* keys, uint32_t* idx) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  int64_t c3[3], side[3];
  cell_of(G, pos + 3 * (size_t)v, c3, side);
  keys[v] = pack_cell(c3[0], c3[1], c3[2]);
  idx[v] = v;
}
__global__ void k_iota(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// lower_bound of `key` in the sorted unique cell keys; returns the cell's position or -1
__device__ inline int find_cell(const uint64_t* __restrict__ ukeys, uint32_t ncells, uint64_t key) {
  uint32_t lo = 0, hi = ncells;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < ncells && ukeys[lo] == key) ? (int)lo : -1;
}

// One pass over L(j) (file header).  target[j] == UNSETTLED until j settles; a settled vertex is a centre iff target[j] == j.
// Reads of target[] race with this very kernel's writes: harmless, a value only ever goes from UNSETTLED to its final one.
// round0: every vertex; later rounds: the `pending` list of the round before.
__global__ void k_settle(const float* __restrict__ pos, uint32_t nv, Grid G, float radius, const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ cstart,
                         uint32_t ncells, const uint32_t* __restrict__ members, const uint32_t* __restrict__ pending_in, uint32_t n_in, uint32_t* target,
                         uint32_t* pending_out, uint32_t* n_out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_in) return;
  const uint32_t j = pending_in ? pending_in[t] : t;
  const float px = pos[3 * (size_t)j], py = pos[3 * (size_t)j + 1], pz = pos[3 * (size_t)j + 2];
  int64_t c3[3], side[3];
  cell_of(G, pos + 3 * (size_t)j, c3, side);
  bool waiting = false;
  uint32_t best = UNSETTLED;   // lowest-index centre in L(j)
  for (int oz = 0; oz < 2; oz++)
    for (int oy = 0; oy < 2; oy++)
      for (int ox = 0; ox < 2; ox++) {
        const int c = find_cell(ukeys, ncells, pack_cell(c3[0] + ox * side[0], c3[1] + oy * side[1], c3[2] + oz * side[2]));
        if (c < 0) continue;
        for (uint32_t k = cstart[c]; k < cstart[c + 1]; k++) {
          const uint32_t i = members[k];
          if (i >= j) break;   // members of a cell are in index order
          // the centre's coordinates minus the candidate's, as clean.cpp's sweep computes it (squares make the order immaterial)
          const float ex = pos[3 * (size_t)i] - px, ey = pos[3 * (size_t)i + 1] - py, ez = pos[3 * (size_t)i + 2] - pz;
          const float dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
          if (!(dist < radius)) continue;
          const uint32_t ti = __atomic_load_n(&target[i], __ATOMIC_RELAXED);
          if (ti == UNSETTLED) waiting = true;
          else if (ti == i && i < best) best = i;
        }
      }
  if (waiting) pending_out[atomicAdd(n_out, 1u)] = j;
  else __atomic_store_n(&target[j], best == UNSETTLED ? j : best, __ATOMIC_RELAXED);
}

// r == 0: canonical bit patterns (-0.0 -> +0.0) as sort keys
__global__ void k_pos_keys(const float* __restrict__ pos, uint32_t nv, uint32_t* kz, uint64_t* kxy) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const float x = pos[3 * (size_t)v] + 0.0f, y = pos[3 * (size_t)v + 1] + 0.0f, z = pos[3 * (size_t)v + 2] + 0.0f;
  kz[v] = __float_as_uint(z);
  kxy[v] = ((uint64_t)__float_as_uint(x) << 32) | __float_as_uint(y);
}
__global__ void k_gather64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm, uint32_t n, uint64_t* dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}
// sorted by (kxy, kz) with ties in index order: the head of a run is the lowest index
__global__ void k_run_heads(const uint64_t* __restrict__ kxy_sorted, const uint32_t* __restrict__ kz, const uint32_t* __restrict__ perm, uint32_t n, uint32_t* head_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head_flag[i] = (i == 0 || kxy_sorted[i] != kxy_sorted[i - 1] || kz[perm[i]] != kz[perm[i - 1]]) ? i : 0u;
}
__global__ void k_targets_from_heads(const uint32_t* __restrict__ head_pos, const uint32_t* __restrict__ perm, uint32_t n, uint32_t* target) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) target[perm[i]] = perm[head_pos[i]];
}

// ---- faces ------------------------------------------------------------------------------------------------------------------------------
__global__ void k_remap_faces(const uint32_t* __restrict__ tri_in, const uint32_t* __restrict__ target, uint32_t nf, Tri* tri, uint8_t* keep) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const uint32_t a = target[tri_in[3 * (size_t)f]], b = target[tri_in[3 * (size_t)f + 1]], c = target[tri_in[3 * (size_t)f + 2]];
  tri[f] = Tri{a, b, c};
  keep[f] = (a == b || b == c || a == c) ? 0 : 1;
}
__device__ inline void sort3(uint32_t& a, uint32_t& b, uint32_t& c) {
  uint32_t t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
}
__global__ void k_face_keys(const Tri* __restrict__ tri, uint32_t n, uint32_t* k2, uint64_t* k01) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  uint32_t a = tri[f].a, b = tri[f].b, c = tri[f].c;
  sort3(a, b, c);
  k2[f] = c;
  k01[f] = ((uint64_t)a << 32) | b;
}
// faces sorted by their vertex set, ties in face order: all but the first of a run are duplicates
__global__ void k_dup_flags(const uint64_t* __restrict__ k01_sorted, const uint32_t* __restrict__ k2, const uint32_t* __restrict__ perm, uint32_t n, uint8_t* keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keep[perm[i]] = (i > 0 && k01_sorted[i] == k01_sorted[i - 1] && k2[perm[i]] == k2[perm[i - 1]]) ? 0 : 1;
}
__global__ void k_edge_keys(const Tri* __restrict__ tri, uint32_t n, uint64_t* ekey, uint32_t* eface) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const uint32_t v[3] = {tri[f].a, tri[f].b, tri[f].c};
  for (int e = 0; e < 3; e++) {
    const uint32_t a = v[e], b = v[(e + 1) % 3];
    ekey[3 * (size_t)f + e] = ((uint64_t)(a < b ? a : b) << 32) | (a < b ? b : a);
    eface[3 * (size_t)f + e] = f;
  }
}
__device__ inline uint32_t uf_root(uint32_t* parent, uint32_t x) {
  for (;;) {
    const uint32_t p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    if (p == x) return x;
    const uint32_t g = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
    if (g != p) __atomic_store_n(&parent[x], g, __ATOMIC_RELAXED);   // path halving: an ancestor replaces the parent
    x = p;
  }
}
__global__ void k_link(const uint64_t* __restrict__ ekey, const uint32_t* __restrict__ eface, size_t n, uint32_t* parent) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i == 0 || i >= n || ekey[i] != ekey[i - 1]) return;
  uint32_t a = eface[i - 1], b = eface[i];
  for (;;) {
    a = uf_root(parent, a);
    b = uf_root(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return;   // the larger root goes under the smaller: a component's root is its lowest face
  }
}
__global__ void k_roots(uint32_t* parent, uint32_t n, uint32_t* root, uint32_t* size) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const uint32_t r = uf_root(parent, f);
  root[f] = r;
  atomicAdd(&size[r], 1u);
}
// counters: [0] components, [1] components below the minimum, [2] faces in them
__global__ void k_component_flags(const uint32_t* __restrict__ root, const uint32_t* __restrict__ size, uint32_t n, uint32_t min_faces, uint8_t* keep, uint32_t* counters) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const bool small = size[root[f]] < min_faces;
  keep[f] = small ? 0 : 1;
  if (small) atomicAdd(&counters[2], 1u);
  if (root[f] == f) {
    atomicAdd(&counters[0], 1u);
    if (small) atomicAdd(&counters[1], 1u);
  }
}

// ---- vertices ---------------------------------------------------------------------------------------------------------------------------
__global__ void k_mark_used(const Tri* __restrict__ tri, uint32_t n, uint32_t* used) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  used[tri[f].a] = 1u; used[tri[f].b] = 1u; used[tri[f].c] = 1u;
}
__global__ void k_count_merged(const uint32_t* __restrict__ target, uint32_t nv, uint32_t* counter) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool merged = v < nv && target[v] != v;
  const unsigned long long b = __ballot(merged);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(counter, (uint32_t)__popcll(b));
}
__global__ void k_gather_vertices(const float* __restrict__ pos, const uint8_t* __restrict__ col, const uint32_t* __restrict__ used, const uint32_t* __restrict__ remap, uint32_t nv,
                                  float* out_pos, uint8_t* out_col) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv || !used[v]) return;
  const uint32_t w = remap[v];
  for (int c = 0; c < 3; c++) out_pos[3 * (size_t)w + c] = pos[3 * (size_t)v + c];   // a surviving vertex is a centre: it never moved
  if (col) *reinterpret_cast<uint32_t*>(out_col + 4 * (size_t)w) = *reinterpret_cast<const uint32_t*>(col + 4 * (size_t)v);
}
__global__ void k_remap_tris(const Tri* __restrict__ tri, const uint32_t* __restrict__ remap, uint32_t n, uint32_t* out) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  out[3 * (size_t)f] = remap[tri[f].a];
  out[3 * (size_t)f + 1] = remap[tri[f].b];
  out[3 * (size_t)f + 2] = remap[tri[f].c];
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

SF_API int sf_mesh_clean_gpu(const sf_mesh* in, float merge_distance, uint32_t min_component_faces, int device, sf_mesh** out, sf_clean_stats* stats) {
  if (!in || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!(merge_distance >= 0.0f)) return sf::fail(SF_ERR_INVALID_ARG, "merge distance must be >= 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return sf::fail(SF_ERR_DEVICE, "no HIP device: sf_mesh_clean_gpu needs an MI355X (sf_mesh_clean is the host filter)");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  CL_CHECK(hipSetDevice(device));
  const size_t nv = in->pos.size() / 3, nf = in->tri.size() / 3;
  if (nv >= 0xFFFFFFF0ull || nf >= 0x55555550ull) return sf::fail(SF_ERR_CAPACITY, "mesh too large for 32-bit indices");
  sf_clean_stats st;
  std::memset(&st, 0, sizeof(st));
  st.vertices_in = nv;
  st.faces_in = nf;
  for (uint32_t v : in->tri)
    if (v >= nv) return sf::fail(SF_ERR_FORMAT, "face references vertex %u of %zu", v, nv);
  sf_mesh* m = new sf_mesh();
  if (nv == 0) {
    if (stats) *stats = st;
    *out = m;
    return SF_OK;
  }
  struct Bail { sf_mesh* m; ~Bail() { delete m; } } bail{m};   // released on success
  Grid G{0.0, {0, 0, 0}};
  if (merge_distance > 0.0f) {
    // the host filter's grid: cells 1e-5 larger than 2 r (the float distance test may accept a point a few 1e-7 relative beyond r)
    G.inv = 0.5 / ((double)merge_distance * (1.0 + 1e-5));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < nv; i++)
      for (int c = 0; c < 3; c++) {
        const double q = (double)in->pos[3 * i + c];
        if (!(q == q) || q > 1e30 || q < -1e30) return sf::fail(SF_ERR_FORMAT, "vertex %zu has a non-finite coordinate", i);
        lo[c] = std::min(lo[c], q); hi[c] = std::max(hi[c], q);
      }
    for (int c = 0; c < 3; c++) {
      G.base[c] = (int64_t)std::floor(lo[c] * G.inv) - 1;
      if ((int64_t)std::floor(hi[c] * G.inv) + 1 - G.base[c] >= (1ll << 21))
        return sf::fail(SF_ERR_UNSUPPORTED, "mesh extent / merge distance exceeds 2^21 cells per axis");
    }
  }
  StreamGuard sg;
  CL_CHECK(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
  hipStream_t s = sg.s;
  const uint32_t V = (uint32_t)nv, F = (uint32_t)nf;
  const size_t N = std::max<size_t>(nv, 3 * nf);   // the longest array any sort sees
  DevBuf d_pos, d_col, d_tri_in, d_target, d_k64a, d_k64b, d_k32a, d_k32b, d_p32a, d_p32b, d_cstart, d_ncells, d_pending_a, d_pending_b, d_count, d_tri, d_tri2, d_keep,
      d_parent, d_root, d_size, d_used, d_remap, d_out_pos, d_out_col, d_out_tri, d_tmp;
  const size_t nfa = std::max<size_t>(nf, 1);   // a mesh without faces still has its vertices clustered (and then dropped as unreferenced)
  CL_CHECK(d_pos.alloc(nv * 12)); CL_CHECK(d_tri_in.alloc(nfa * 12)); CL_CHECK(d_target.alloc(nv * 4));
  if (!in->col.empty()) CL_CHECK(d_col.alloc(nv * 4));
  CL_CHECK(d_k64a.alloc(N * 8)); CL_CHECK(d_k64b.alloc(N * 8)); CL_CHECK(d_k32a.alloc((N + 1) * 4)); CL_CHECK(d_k32b.alloc(N * 4));
  CL_CHECK(d_p32a.alloc(N * 4)); CL_CHECK(d_p32b.alloc(N * 4)); CL_CHECK(d_cstart.alloc((nv + 1) * 4)); CL_CHECK(d_ncells.alloc(16));
  CL_CHECK(d_pending_a.alloc(nv * 4)); CL_CHECK(d_pending_b.alloc(nv * 4)); CL_CHECK(d_count.alloc(64));
  CL_CHECK(d_tri.alloc(nfa * 12)); CL_CHECK(d_tri2.alloc(nfa * 12)); CL_CHECK(d_keep.alloc(nfa));
  CL_CHECK(d_parent.alloc(nfa * 4)); CL_CHECK(d_root.alloc(nfa * 4)); CL_CHECK(d_size.alloc(nfa * 4));
  CL_CHECK(d_used.alloc(nv * 4)); CL_CHECK(d_remap.alloc(nv * 4));
  size_t tmp_bytes = 0, need = 0;
  CL_CHECK(rocprim::radix_sort_pairs(nullptr, need, d_k64a.as<uint64_t>(), d_k64b.as<uint64_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), N, 0, 64, s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(rocprim::radix_sort_pairs(nullptr, need, d_k32a.as<uint32_t>(), d_k32b.as<uint32_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), N, 0, 32, s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(rocprim::run_length_encode(nullptr, need, d_k64b.as<uint64_t>(), (unsigned int)nv, d_k64a.as<uint64_t>(), d_k32a.as<uint32_t>(), d_ncells.as<uint32_t>(), s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(rocprim::exclusive_scan(nullptr, need, d_k32a.as<uint32_t>(), d_cstart.as<uint32_t>(), 0u, nv + 1, rocprim::plus<uint32_t>(), s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(rocprim::inclusive_scan(nullptr, need, d_k32a.as<uint32_t>(), d_k32b.as<uint32_t>(), nv, rocprim::maximum<uint32_t>(), s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(rocprim::select(nullptr, need, d_tri.as<Tri>(), d_keep.as<uint8_t>(), d_tri2.as<Tri>(), d_ncells.as<uint32_t>(), nfa, s));
  tmp_bytes = std::max(tmp_bytes, need);
  CL_CHECK(d_tmp.alloc(tmp_bytes));
  CL_CHECK(hipMemcpyAsync(d_pos.p, in->pos.data(), nv * 12, hipMemcpyHostToDevice, s));
  if (nf > 0) CL_CHECK(hipMemcpyAsync(d_tri_in.p, in->tri.data(), nf * 12, hipMemcpyHostToDevice, s));
  if (!in->col.empty()) CL_CHECK(hipMemcpyAsync(d_col.p, in->col.data(), nv * 4, hipMemcpyHostToDevice, s));
  size_t tb;
  uint32_t* target = d_target.as<uint32_t>();

  // ---- 1. merge close vertices ---------------------------------------------------------------------------------------------------------
  if (merge_distance > 0.0f) {
    hipLaunchKernelGGL(k_cell_keys, dim3(grid_for(nv)), dim3(256), 0, s, d_pos.as<float>(), V, G, d_k64a.as<uint64_t>(), d_p32a.as<uint32_t>());
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k64a.as<uint64_t>(), d_k64b.as<uint64_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), nv, 0, 63, s));
    // unique cells -> d_k64a, their sizes -> d_k32a, their first member -> d_cstart; members (vertex ids, index order inside a cell) = d_p32b
    tb = tmp_bytes;
    CL_CHECK(rocprim::run_length_encode(d_tmp.p, tb, d_k64b.as<uint64_t>(), (unsigned int)nv, d_k64a.as<uint64_t>(), d_k32a.as<uint32_t>(), d_ncells.as<uint32_t>(), s));
    uint32_t ncells = 0;
    CL_CHECK(hipMemcpyAsync(&ncells, d_ncells.p, 4, hipMemcpyDeviceToHost, s));
    CL_CHECK(hipStreamSynchronize(s));
    CL_CHECK(hipMemsetAsync(d_k32a.as<uint32_t>() + ncells, 0, 4, s));   // the scan below reads one element past the counts
    tb = tmp_bytes;
    CL_CHECK(rocprim::exclusive_scan(d_tmp.p, tb, d_k32a.as<uint32_t>(), d_cstart.as<uint32_t>(), 0u, (size_t)ncells + 1, rocprim::plus<uint32_t>(), s));
    CL_CHECK(hipMemsetAsync(target, 0xFF, nv * 4, s));
    uint32_t n_in = V;
    const uint32_t* pending_in = nullptr;
    uint32_t* pend[2] = {d_pending_a.as<uint32_t>(), d_pending_b.as<uint32_t>()};
    for (int round = 0;; round++) {
      CL_CHECK(hipMemsetAsync(d_count.p, 0, 4, s));
      hipLaunchKernelGGL(k_settle, dim3(grid_for(n_in)), dim3(256), 0, s, d_pos.as<float>(), V, G, merge_distance, d_k64a.as<uint64_t>(), d_cstart.as<uint32_t>(), ncells,
                         d_p32b.as<uint32_t>(), pending_in, n_in, target, pend[round & 1], d_count.as<uint32_t>());
      uint32_t left = 0;
      CL_CHECK(hipMemcpyAsync(&left, d_count.p, 4, hipMemcpyDeviceToHost, s));
      CL_CHECK(hipStreamSynchronize(s));
      if (left == 0) break;
      if (left == n_in && round > 0) return sf::fail(SF_ERR_DEVICE, "sf_mesh_clean_gpu: clustering made no progress (%u vertices)", left);
      pending_in = pend[round & 1];
      n_in = left;
    }
  } else {
    // bit-identical positions -> lowest index: stable sort by z, then by (x, y); run heads by a running maximum
    hipLaunchKernelGGL(k_pos_keys, dim3(grid_for(nv)), dim3(256), 0, s, d_pos.as<float>(), V, d_k32a.as<uint32_t>(), d_k64a.as<uint64_t>());
    hipLaunchKernelGGL(k_iota, dim3(grid_for(nv)), dim3(256), 0, s, d_p32a.as<uint32_t>(), V);
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k32a.as<uint32_t>(), d_k32b.as<uint32_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), nv, 0, 32, s));
    hipLaunchKernelGGL(k_gather64, dim3(grid_for(nv)), dim3(256), 0, s, d_k64a.as<uint64_t>(), d_p32b.as<uint32_t>(), V, d_k64b.as<uint64_t>());
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k64b.as<uint64_t>(), d_k64a.as<uint64_t>(), d_p32b.as<uint32_t>(), d_p32a.as<uint32_t>(), nv, 0, 64, s));
    // sorted (x, y) keys in d_k64a, permutation in d_p32a, z keys by vertex in d_k32a
    hipLaunchKernelGGL(k_run_heads, dim3(grid_for(nv)), dim3(256), 0, s, d_k64a.as<uint64_t>(), d_k32a.as<uint32_t>(), d_p32a.as<uint32_t>(), V, d_k32b.as<uint32_t>());
    tb = tmp_bytes;
    CL_CHECK(rocprim::inclusive_scan(d_tmp.p, tb, d_k32b.as<uint32_t>(), d_p32b.as<uint32_t>(), nv, rocprim::maximum<uint32_t>(), s));
    hipLaunchKernelGGL(k_targets_from_heads, dim3(grid_for(nv)), dim3(256), 0, s, d_p32b.as<uint32_t>(), d_p32a.as<uint32_t>(), V, target);
  }
  CL_CHECK(hipMemsetAsync(d_count.p, 0, 64, s));
  hipLaunchKernelGGL(k_count_merged, dim3(grid_for(nv)), dim3(256), 0, s, target, V, d_count.as<uint32_t>() + 4);

  // ---- faces through the merge; degenerate ones out ---------------------------------------------------------------------------------------
  if (F > 0) hipLaunchKernelGGL(k_remap_faces, dim3(grid_for(nf)), dim3(256), 0, s, d_tri_in.as<uint32_t>(), target, F, d_tri.as<Tri>(), d_keep.as<uint8_t>());
  auto compact_faces = [&](uint32_t n, uint32_t* kept) -> int {   // d_tri -> (stable select by d_keep) -> d_tri
    size_t t2 = tmp_bytes;
    CL_CHECK(rocprim::select(d_tmp.p, t2, d_tri.as<Tri>(), d_keep.as<uint8_t>(), d_tri2.as<Tri>(), d_ncells.as<uint32_t>(), (size_t)n, s));
    CL_CHECK(hipMemcpyAsync(kept, d_ncells.p, 4, hipMemcpyDeviceToHost, s));
    CL_CHECK(hipStreamSynchronize(s));
    std::swap(d_tri.p, d_tri2.p);
    return SF_OK;
  };
  uint32_t n1 = 0;
  if (F > 0) { const int rc = compact_faces(F, &n1); if (rc != SF_OK) return rc; }
  st.faces_degenerate = F - n1;

  // ---- 2. duplicate faces -----------------------------------------------------------------------------------------------------------------
  uint32_t n2 = n1;
  if (n1 > 0) {
    hipLaunchKernelGGL(k_face_keys, dim3(grid_for(n1)), dim3(256), 0, s, d_tri.as<Tri>(), n1, d_k32a.as<uint32_t>(), d_k64a.as<uint64_t>());
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n1)), dim3(256), 0, s, d_p32a.as<uint32_t>(), n1);
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k32a.as<uint32_t>(), d_k32b.as<uint32_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), (size_t)n1, 0, 32, s));
    hipLaunchKernelGGL(k_gather64, dim3(grid_for(n1)), dim3(256), 0, s, d_k64a.as<uint64_t>(), d_p32b.as<uint32_t>(), n1, d_k64b.as<uint64_t>());
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k64b.as<uint64_t>(), d_k64a.as<uint64_t>(), d_p32b.as<uint32_t>(), d_p32a.as<uint32_t>(), (size_t)n1, 0, 64, s));
    hipLaunchKernelGGL(k_dup_flags, dim3(grid_for(n1)), dim3(256), 0, s, d_k64a.as<uint64_t>(), d_k32a.as<uint32_t>(), d_p32a.as<uint32_t>(), n1, d_keep.as<uint8_t>());
    const int rc = compact_faces(n1, &n2);
    if (rc != SF_OK) return rc;
  }
  st.faces_duplicate = n1 - n2;

  // ---- 3. small connected components ------------------------------------------------------------------------------------------------------
  uint32_t n3 = n2;
  if (n2 > 0) {
    const size_t ne = 3 * (size_t)n2;
    hipLaunchKernelGGL(k_edge_keys, dim3(grid_for(n2)), dim3(256), 0, s, d_tri.as<Tri>(), n2, d_k64a.as<uint64_t>(), d_p32a.as<uint32_t>());
    tb = tmp_bytes;
    CL_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tb, d_k64a.as<uint64_t>(), d_k64b.as<uint64_t>(), d_p32a.as<uint32_t>(), d_p32b.as<uint32_t>(), ne, 0, 64, s));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n2)), dim3(256), 0, s, d_parent.as<uint32_t>(), n2);
    hipLaunchKernelGGL(k_link, dim3(grid_for(ne)), dim3(256), 0, s, d_k64b.as<uint64_t>(), d_p32b.as<uint32_t>(), ne, d_parent.as<uint32_t>());
    CL_CHECK(hipMemsetAsync(d_size.p, 0, (size_t)n2 * 4, s));
    hipLaunchKernelGGL(k_roots, dim3(grid_for(n2)), dim3(256), 0, s, d_parent.as<uint32_t>(), n2, d_root.as<uint32_t>(), d_size.as<uint32_t>());
    hipLaunchKernelGGL(k_component_flags, dim3(grid_for(n2)), dim3(256), 0, s, d_root.as<uint32_t>(), d_size.as<uint32_t>(), n2, min_component_faces, d_keep.as<uint8_t>(),
                       d_count.as<uint32_t>());
    const int rc = compact_faces(n2, &n3);
    if (rc != SF_OK) return rc;
  }

  // ---- 4. unreferenced vertices, compaction ------------------------------------------------------------------------------------------------
  CL_CHECK(hipMemsetAsync(d_used.p, 0, nv * 4, s));
  if (n3 > 0) hipLaunchKernelGGL(k_mark_used, dim3(grid_for(n3)), dim3(256), 0, s, d_tri.as<Tri>(), n3, d_used.as<uint32_t>());
  tb = tmp_bytes;
  CL_CHECK(rocprim::exclusive_scan(d_tmp.p, tb, d_used.as<uint32_t>(), d_remap.as<uint32_t>(), 0u, nv, rocprim::plus<uint32_t>(), s));
  uint32_t last_used = 0, last_remap = 0, counters[16];
  CL_CHECK(hipMemcpyAsync(&last_used, d_used.as<uint32_t>() + (nv - 1), 4, hipMemcpyDeviceToHost, s));
  CL_CHECK(hipMemcpyAsync(&last_remap, d_remap.as<uint32_t>() + (nv - 1), 4, hipMemcpyDeviceToHost, s));
  CL_CHECK(hipMemcpyAsync(counters, d_count.p, 64, hipMemcpyDeviceToHost, s));
  CL_CHECK(hipStreamSynchronize(s));
  const uint32_t vout = last_remap + last_used;
  st.vertices_merged = counters[4];
  st.components_in = counters[0];
  st.components_removed = counters[1];
  st.faces_small_component = counters[2];
  st.vertices_out = vout;
  st.faces_out = n3;
  st.vertices_unreferenced = nv - st.vertices_merged - st.vertices_out;
  m->pos.resize((size_t)vout * 3);
  if (!in->col.empty()) m->col.resize((size_t)vout * 4);
  m->tri.resize((size_t)n3 * 3);
  if (vout > 0) {
    CL_CHECK(d_out_pos.alloc((size_t)vout * 12));
    if (!in->col.empty()) CL_CHECK(d_out_col.alloc((size_t)vout * 4));
    hipLaunchKernelGGL(k_gather_vertices, dim3(grid_for(nv)), dim3(256), 0, s, d_pos.as<float>(), in->col.empty() ? nullptr : d_col.as<uint8_t>(), d_used.as<uint32_t>(),
                       d_remap.as<uint32_t>(), V, d_out_pos.as<float>(), in->col.empty() ? nullptr : d_out_col.as<uint8_t>());
    CL_CHECK(hipMemcpyAsync(m->pos.data(), d_out_pos.p, (size_t)vout * 12, hipMemcpyDeviceToHost, s));
    if (!in->col.empty()) CL_CHECK(hipMemcpyAsync(m->col.data(), d_out_col.p, (size_t)vout * 4, hipMemcpyDeviceToHost, s));
  }
  if (n3 > 0) {
    CL_CHECK(d_out_tri.alloc((size_t)n3 * 12));
    hipLaunchKernelGGL(k_remap_tris, dim3(grid_for(n3)), dim3(256), 0, s, d_tri.as<Tri>(), d_remap.as<uint32_t>(), n3, d_out_tri.as<uint32_t>());
    CL_CHECK(hipMemcpyAsync(m->tri.data(), d_out_tri.p, (size_t)n3 * 12, hipMemcpyDeviceToHost, s));
  }
  CL_CHECK(hipStreamSynchronize(s));
  CL_CHECK(hipGetLastError());
  bail.m = nullptr;
  if (stats) *stats = st;
  *out = m;
  return SF_OK;
}

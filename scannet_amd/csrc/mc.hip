// mc.hip -- iso-surface extraction from the voxel-hash TSDF: marching cubes with an exact integer weld.
//
// Replaces the mesh extraction of the external DepthSensing.exe ("improve" stage, Server/scan_processor.py:
// 137-138 -> <id>_vh.ply); in-tree anchors are s_SDFMarchingCubeThreshFactor / s_marchingCubesMaxNumTriangles
// (Server/tools/recons/zParametersScanNet.txt:48,106) and the PLY surface (README.md:45-46).  Spec: DESIGN.md 3.7.
//
// One 512-thread workgroup per live block.  The block's 8^3 tile plus the +1 halo from its 7 neighbour
// blocks (hash look-ups) is staged in LDS as a 9^3 tile of {sdf, rgbw}; each thread owns one cube, reads
// its 8 corners from LDS, classifies it with the generated case table (also in LDS) and
//   pass 1 counts triangles per block,  [device exclusive scan over blocks]
//   pass 2 writes the triangle soup at the block's offset: cube key + 3 x (edge key, position, colour).
// Every vertex is identified by the integer key of the grid edge it lies on, so welding is exact:
// radix sort the 3T edge keys, flag run heads, scan, scatter.  Triangles are sorted by (cube key, table
// order); vertices by edge key -- the canonical mesh of the CPU checker, byte for byte.
// rocPRIM supplies the device radix sort / scan (plumbing); the kernels are hand-written.
#include <hip/hip_runtime.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <sys/mman.h>

#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "fuser_internal.h"
#include "mc_tables.h"
#include "mesh.h"

namespace {

constexpr int KEY_BIAS = 1 << 19;

__device__ inline uint64_t edge_key(int gx, int gy, int gz, int axis) {
  return ((uint64_t)(uint32_t)(gx + KEY_BIAS) << 42) | ((uint64_t)(uint32_t)(gy + KEY_BIAS) << 22) |
         ((uint64_t)(uint32_t)(gz + KEY_BIAS) << 2) | (uint64_t)axis;
}

struct McTables {
  signed char corner_off[8][4];
  signed char edge_lo[12], edge_hi[12], edge_axis[12];
  unsigned char num_tris[256];
  signed char tris[256][MC_MAX_TRIS * 3];
};

struct Tile {
  float sdf[729];
  uint32_t cw[729];
};

// stage the 9^3 tile; returns false for the threads' view when a corner is unusable later (handled per cube)
__device__ inline void load_tile(Tile& t, int* s_nb, const uint4* __restrict__ voxels, const HashEntry* __restrict__ table,
                                 const uint64_t* __restrict__ block_keys, const ParamsK& P, int slot, int& bx, int& by, int& bz) {
  unpack_key(block_keys[slot], bx, by, bz);
  if (threadIdx.x < 8) {
    const int i = threadIdx.x;
    s_nb[i] = i == 0 ? slot : hash_lookup(table, P, bx + (i & 1), by + ((i >> 1) & 1), bz + (i >> 2));
  }
  __syncthreads();
  const uint2* vox = reinterpret_cast<const uint2*>(voxels);
  for (int i = threadIdx.x; i < 729; i += 512) {
    const int tx = i % 9, ty = (i / 9) % 9, tz = i / 81;
    const int nb = (tx >> 3) | ((ty >> 3) << 1) | ((tz >> 3) << 2);
    const int s = s_nb[nb];
    uint2 v = make_uint2(0u, 0u);  // weight 0 => cube skipped
    if (s >= 0) v = vox[(size_t)s * 512 + (tz & 7) * 64 + (ty & 7) * 8 + (tx & 7)];
    t.sdf[i] = __uint_as_float(v.x);
    t.cw[i] = v.y;
  }
  __syncthreads();
}

__device__ inline int classify(const Tile& t, const McTables& T, int lx, int ly, int lz, float thresh, float* d, uint32_t* c) {
  int cs = 0;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int idx = (lz + T.corner_off[i][2]) * 81 + (ly + T.corner_off[i][1]) * 9 + lx + T.corner_off[i][0];
    d[i] = t.sdf[idx];
    c[i] = t.cw[idx];
    ok = ok && ((c[i] >> 24) != 0u) && (fabsf(d[i]) <= thresh);
    if (d[i] < 0.0f) cs |= 1 << i;
  }
  return ok ? cs : -1;
}

__global__ __launch_bounds__(512) void k_mc(const uint4* __restrict__ voxels, const HashEntry* __restrict__ table,
                                            const uint64_t* __restrict__ block_keys, const int32_t* __restrict__ live, int n_live,
                                            const McTables* __restrict__ tables, ParamsK P, float thresh, int emit,
                                            uint32_t* __restrict__ tri_count, const uint32_t* __restrict__ tri_off,
                                            uint64_t* __restrict__ tri_key, uint64_t* __restrict__ vkey, float* __restrict__ vpos,
                                            uint32_t* __restrict__ vcol) {
  __shared__ Tile tile;
  __shared__ McTables T;
  __shared__ int s_nb[8];
  __shared__ uint32_t s_wsum[8];
  for (int i = threadIdx.x; i < (int)(sizeof(McTables) / 4); i += 512) reinterpret_cast<uint32_t*>(&T)[i] = reinterpret_cast<const uint32_t*>(tables)[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
  for (int bi = blockIdx.x; bi < n_live; bi += gridDim.x) {
    int bx, by, bz;
    load_tile(tile, s_nb, voxels, table, block_keys, P, live[bi], bx, by, bz);
    float d[8];
    uint32_t c[8];
    const int cs = classify(tile, T, lx, ly, lz, thresh, d, c);
    const uint32_t nt = cs < 0 ? 0u : (uint32_t)T.num_tris[cs];
    // block-wide exclusive prefix of nt in cube-index order: wave scan + LDS
    uint32_t incl = nt;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < 8; w++) {
      if (w < wave) base += s_wsum[w];
      total += s_wsum[w];
    }
    if (!emit) {
      if (threadIdx.x == 0) tri_count[bi] = total;
    } else if (nt) {
      const int gx = 8 * bx + lx, gy = 8 * by + ly, gz = 8 * bz + lz;
      const uint64_t cube = (edge_key(gx, gy, gz, 0) >> 2) << 3;
      const uint32_t first = tri_off[bi] + base + (incl - nt);
      for (uint32_t t = 0; t < nt; t++) {
        const size_t ti = (size_t)first + t;
        tri_key[ti] = cube | (uint64_t)t;
        for (int k = 0; k < 3; k++) {
          const int e = T.tris[cs][3 * t + k];
          const int a = T.edge_lo[e], b = T.edge_hi[e], axis = T.edge_axis[e];
          const int glx = gx + T.corner_off[a][0], gly = gy + T.corner_off[a][1], glz = gz + T.corner_off[a][2];
          const float dl = d[a], dh = d[b];
          const float mu = dl / (dl - dh);
          float px = (float)glx * P.voxel, py = (float)gly * P.voxel, pz = (float)glz * P.voxel;
          if (axis == 0) px = ((float)glx + mu) * P.voxel;
          else if (axis == 1) py = ((float)gly + mu) * P.voxel;
          else pz = ((float)glz + mu) * P.voxel;
          uint32_t col = 0xFF000000u;
          for (int q = 0; q < 3; q++) {
            const float cl = (float)((c[a] >> (8 * q)) & 0xFFu), ch = (float)((c[b] >> (8 * q)) & 0xFFu);
            col |= ((uint32_t)(fmaf(mu, ch - cl, cl) + 0.5f) & 0xFFu) << (8 * q);
          }
          const size_t vi = 3 * ti + k;
          vkey[vi] = edge_key(glx, gly, glz, axis);
          vpos[3 * vi] = px; vpos[3 * vi + 1] = py; vpos[3 * vi + 2] = pz;
          vcol[vi] = col;
        }
      }
    }
    __syncthreads();
  }
}

__global__ void k_iota(uint32_t* a, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void k_flag_heads(const uint64_t* __restrict__ sorted_keys, uint32_t* flag, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || sorted_keys[i] != sorted_keys[i - 1]) ? 1u : 0u;
}
// sorted position j (soup index src[j]) -> vertex id = (inclusive head count) - 1
__global__ void k_weld(const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ src, const uint32_t* __restrict__ flag,
                       const uint32_t* __restrict__ excl, uint32_t n, const float* __restrict__ vpos, const uint32_t* __restrict__ vcol,
                       uint32_t* __restrict__ soup_vid, float* out_pos, uint32_t* out_col, uint64_t* out_key) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t vid = excl[j] + flag[j] - 1u;
  const uint32_t s = src[j];
  soup_vid[s] = vid;
  if (flag[j]) {
    out_pos[3 * (size_t)vid] = vpos[3 * (size_t)s]; out_pos[3 * (size_t)vid + 1] = vpos[3 * (size_t)s + 1]; out_pos[3 * (size_t)vid + 2] = vpos[3 * (size_t)s + 2];
    out_col[vid] = vcol[s];
    out_key[vid] = sorted_keys[j];
  }
}
__global__ void k_gather_tris(const uint32_t* __restrict__ tri_src, const uint32_t* __restrict__ soup_vid, uint32_t n_tris, uint32_t* out_idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tris) return;
  const uint32_t t = tri_src[i];
  out_idx[3 * (size_t)i] = soup_vid[3 * (size_t)t]; out_idx[3 * (size_t)i + 1] = soup_vid[3 * (size_t)t + 1]; out_idx[3 * (size_t)i + 2] = soup_vid[3 * (size_t)t + 2];
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T* as() { return (T*)p; }
};

McTables host_tables() {
  McTables t;
  std::memset(&t, 0, sizeof(t));
  for (int i = 0; i < 8; i++) for (int q = 0; q < 3; q++) t.corner_off[i][q] = MC_CORNER_OFF[i][q];
  for (int e = 0; e < 12; e++) {
    int a = MC_EDGE_CORNERS[e][0], b = MC_EDGE_CORNERS[e][1], axis = 0;
    for (int q = 0; q < 3; q++) if (MC_CORNER_OFF[a][q] != MC_CORNER_OFF[b][q]) axis = q;
    if (MC_CORNER_OFF[a][axis] > MC_CORNER_OFF[b][axis]) { const int tmp = a; a = b; b = tmp; }
    t.edge_lo[e] = (signed char)a; t.edge_hi[e] = (signed char)b; t.edge_axis[e] = (signed char)axis;
  }
  for (int c = 0; c < 256; c++) {
    t.num_tris[c] = MC_NUM_TRIS[c];
    for (int k = 0; k < MC_MAX_TRIS * 3; k++) t.tris[c][k] = MC_TRIS[c][k];
  }
  return t;
}

#define MC_CHECK(call)                                                                                        \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return sf::fail(SF_ERR_DEVICE, "%s failed: %s (mc.hip:%d)", #call, hipGetErrorString(e_), __LINE__); \
  } while (0)


// Device -> host for the mesh arrays (a scan-sized mesh is ~250 MB going into freshly allocated, never touched memory): the runtime's own path for
// pageable destinations stages through a small pinned buffer on ONE thread -- copy and first-touch page faults serialised, ~6 GB/s.  Here: two
// 16 MiB page-locked bounce buffers owned by the fuser; chunk i + 1 travels over the link while a small team of threads copies chunk i out,
// each thread faulting its own pages (up to 8: the 4 of the first version left the link waiting, 13 GB/s).
struct DlSeg { void* dst; const void* src; size_t bytes; };
constexpr size_t DL_CHUNK = 16u << 20;

void team_copy(uint8_t* dst, const uint8_t* src, size_t n) {
  const int nt = n < (2u << 20) ? 1 : std::max(1, std::min(8, sf::usable_cpus()));
  if (nt == 1) { std::memcpy(dst, src, n); return; }
  std::vector<std::thread> team;
  const size_t per = ((n / (size_t)nt) + 4095) & ~(size_t)4095;
  for (int t = 1; t < nt; t++) {
    const size_t a = std::min(n, per * (size_t)t), b = std::min(n, per * (size_t)(t + 1));
    if (b > a) team.emplace_back([=] { std::memcpy(dst + a, src + a, b - a); });
  }
  std::memcpy(dst, src, std::min(n, per));
  for (std::thread& t : team) t.join();
}

int download_segments(sf_fuser* f, hipStream_t s, const DlSeg* segs, int nseg) {
  // the destinations are fresh anonymous memory: with 4 KiB pages the copy team spends its time in page faults (8 threads: 10 GB/s, less than half
  // of what the link delivers); ask for transparent huge pages on the 2 MiB-aligned inside of every large array -- 512x fewer faults
  for (int i = 0; i < nseg; i++) {
    const uintptr_t a = ((uintptr_t)segs[i].dst + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), b = ((uintptr_t)segs[i].dst + segs[i].bytes) & ~(uintptr_t)((2u << 20) - 1);
    if (b > a) (void)madvise((void*)a, (size_t)(b - a), MADV_HUGEPAGE);   // advisory: a kernel without THP just says no
  }
  for (int q = 0; q < 2; q++) {
    if (!f->mc_bounce[q]) MC_CHECK(hipHostMalloc(&f->mc_bounce[q], DL_CHUNK, hipHostMallocDefault));
    if (!f->mc_bounce_ev[q]) MC_CHECK(hipEventCreateWithFlags(&f->mc_bounce_ev[q], hipEventDisableTiming));
  }
  struct Pending { uint8_t* dst; size_t n; bool live; } pend[2] = {{nullptr, 0, false}, {nullptr, 0, false}};
  int slot = 0;
  auto drain = [&](int q) -> int {
    if (!pend[q].live) return SF_OK;
    MC_CHECK(hipEventSynchronize(f->mc_bounce_ev[q]));
    team_copy(pend[q].dst, (const uint8_t*)f->mc_bounce[q], pend[q].n);
    pend[q].live = false;
    return SF_OK;
  };
  for (int i = 0; i < nseg; i++) {
    for (size_t off = 0; off < segs[i].bytes; off += DL_CHUNK) {
      const size_t n = std::min(DL_CHUNK, segs[i].bytes - off);
      int rc = drain(slot);   // the bounce buffer is free once its previous chunk has been copied out
      if (rc != SF_OK) return rc;
      MC_CHECK(hipMemcpyAsync(f->mc_bounce[slot], (const uint8_t*)segs[i].src + off, n, hipMemcpyDeviceToHost, s));
      MC_CHECK(hipEventRecord(f->mc_bounce_ev[slot], s));
      pend[slot] = {(uint8_t*)segs[i].dst + off, n, true};
      slot ^= 1;
      rc = drain(slot);       // while that one travels, copy the other out
      if (rc != SF_OK) return rc;
    }
  }
  int rc = drain(0);
  if (rc != SF_OK) return rc;
  return drain(1);
}

}  // namespace

// phase clock of the most recent extraction (scanfuse_internal.h sf_fuser_mc_timing): host wall clock between the points where the host
// waits for the stream anyway, HIP events for the phases in between
namespace {
struct McClock {
  hipEvent_t ev[8] = {nullptr};
  int n = 0;
  hipStream_t s;
  explicit McClock(hipStream_t st) : s(st) {}
  ~McClock() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
  void mark() { if (n < 8 && hipEventCreate(&ev[n]) == hipSuccess) { (void)hipEventRecord(ev[n], s); n++; } }
  double ms(int a, int b) const { float t = 0; return (a < n && b < n && hipEventElapsedTime(&t, ev[a], ev[b]) == hipSuccess) ? (double)t : -1.0; }
};
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

SF_API int sf_fuser_mc_timing(const sf_fuser* f, double* out, int n) {
  if (!f || !out || n < 0) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  for (int i = 0; i < n; i++) out[i] = i < 12 ? f->mc_timing[i] : 0.0;
  return SF_OK;
}

SF_API int sf_fuser_extract_mesh(sf_fuser* f, sf_mesh** out) {
  if (!f || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  MC_CHECK(hipSetDevice(f->device));
  const double t_begin = wall_ms();
  for (double& v : f->mc_timing) v = 0.0;
  int32_t n_live = 0;
  int rc = sf_compact_live(f, &n_live, 0);  // ghost copies of a neighbour slab's blocks are read as neighbours, never meshed
  if (rc != SF_OK) return rc;
  std::unique_ptr<sf_mesh> m(new sf_mesh());
  if (n_live == 0) { *out = m.release(); return SF_OK; }
  hipStream_t s = f->stream;
  McClock clk(s);
  const double t_compact = wall_ms();
  const float thresh = f->p.mc_thresh_factor * f->p.voxel_size;
  const McTables ht = host_tables();
  DevBuf d_tab, d_cnt, d_off, d_tmp;
  MC_CHECK(d_tab.alloc(sizeof(McTables)));
  MC_CHECK(hipMemcpyAsync(d_tab.p, &ht, sizeof(McTables), hipMemcpyHostToDevice, s));
  MC_CHECK(d_cnt.alloc((size_t)(n_live + 1) * 4));
  MC_CHECK(d_off.alloc((size_t)(n_live + 1) * 4));
  MC_CHECK(hipMemsetAsync(d_cnt.p, 0, (size_t)(n_live + 1) * 4, s));
  const int grid = n_live < f->num_cus * 8 ? n_live : f->num_cus * 8;
  clk.mark();   // 0
  hipLaunchKernelGGL(k_mc, dim3(grid), dim3(512), 0, s, f->voxels, f->table, f->block_keys, f->compact, n_live, d_tab.as<McTables>(), f->pk,
                     thresh, 0, d_cnt.as<uint32_t>(), (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr, (float*)nullptr, (uint32_t*)nullptr);
  clk.mark();   // 1: count pass done
  size_t tmp_bytes = 0;
  MC_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt.as<uint32_t>(), d_off.as<uint32_t>(), 0u, (size_t)(n_live + 1), rocprim::plus<uint32_t>(), s));
  MC_CHECK(d_tmp.alloc(tmp_bytes));
  MC_CHECK(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, d_cnt.as<uint32_t>(), d_off.as<uint32_t>(), 0u, (size_t)(n_live + 1), rocprim::plus<uint32_t>(), s));
  uint32_t T = 0;
  MC_CHECK(hipMemcpyAsync(&T, d_off.as<uint32_t>() + n_live, 4, hipMemcpyDeviceToHost, s));
  MC_CHECK(hipStreamSynchronize(s));
  if (T == 0) { *out = m.release(); return SF_OK; }
  if ((uint64_t)T * 3 > 0xFFFFFFF0ull) { return sf::fail(SF_ERR_CAPACITY, "mesh too large: %u triangles", T); }
  const uint32_t NV = 3 * T;
  const double t_counted = wall_ms();
  DevBuf d_tkey, d_vkey, d_vpos, d_vcol;
  MC_CHECK(d_tkey.alloc((size_t)T * 8));
  MC_CHECK(d_vkey.alloc((size_t)NV * 8));
  MC_CHECK(d_vpos.alloc((size_t)NV * 12));
  MC_CHECK(d_vcol.alloc((size_t)NV * 4));
  hipLaunchKernelGGL(k_mc, dim3(grid), dim3(512), 0, s, f->voxels, f->table, f->block_keys, f->compact, n_live, d_tab.as<McTables>(), f->pk,
                     thresh, 1, d_cnt.as<uint32_t>(), d_off.as<uint32_t>(), d_tkey.as<uint64_t>(), d_vkey.as<uint64_t>(), d_vpos.as<float>(),
                     d_vcol.as<uint32_t>());
  clk.mark();   // 2: emit pass queued behind this
  // ---- weld: sort edge keys, flag heads, scan, scatter
  DevBuf d_vkey_s, d_src, d_src_s, d_flag, d_excl, d_tmp2;
  MC_CHECK(d_vkey_s.alloc((size_t)NV * 8));
  MC_CHECK(d_src.alloc((size_t)NV * 4));
  MC_CHECK(d_src_s.alloc((size_t)NV * 4));
  hipLaunchKernelGGL(k_iota, dim3((NV + 255) / 256), dim3(256), 0, s, d_src.as<uint32_t>(), NV);
  size_t sort_bytes = 0;
  MC_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_vkey.as<uint64_t>(), d_vkey_s.as<uint64_t>(), d_src.as<uint32_t>(), d_src_s.as<uint32_t>(), (size_t)NV, 0, 64, s));
  MC_CHECK(d_tmp2.alloc(sort_bytes));
  MC_CHECK(rocprim::radix_sort_pairs(d_tmp2.p, sort_bytes, d_vkey.as<uint64_t>(), d_vkey_s.as<uint64_t>(), d_src.as<uint32_t>(), d_src_s.as<uint32_t>(), (size_t)NV, 0, 64, s));
  clk.mark();   // 3: vertex sort
  MC_CHECK(d_flag.alloc((size_t)(NV + 1) * 4));
  MC_CHECK(d_excl.alloc((size_t)(NV + 1) * 4));
  MC_CHECK(hipMemsetAsync(d_flag.p, 0, (size_t)(NV + 1) * 4, s));
  hipLaunchKernelGGL(k_flag_heads, dim3((NV + 255) / 256), dim3(256), 0, s, d_vkey_s.as<uint64_t>(), d_flag.as<uint32_t>(), NV);
  size_t scan_bytes = 0;
  DevBuf d_tmp3;
  MC_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag.as<uint32_t>(), d_excl.as<uint32_t>(), 0u, (size_t)(NV + 1), rocprim::plus<uint32_t>(), s));
  MC_CHECK(d_tmp3.alloc(scan_bytes));
  MC_CHECK(rocprim::exclusive_scan(d_tmp3.p, scan_bytes, d_flag.as<uint32_t>(), d_excl.as<uint32_t>(), 0u, (size_t)(NV + 1), rocprim::plus<uint32_t>(), s));
  uint32_t NU = 0;
  MC_CHECK(hipMemcpyAsync(&NU, d_excl.as<uint32_t>() + NV, 4, hipMemcpyDeviceToHost, s));
  MC_CHECK(hipStreamSynchronize(s));
  DevBuf d_vid, d_opos, d_ocol, d_okey;
  MC_CHECK(d_vid.alloc((size_t)NV * 4));
  MC_CHECK(d_opos.alloc((size_t)NU * 12));
  MC_CHECK(d_ocol.alloc((size_t)NU * 4));
  MC_CHECK(d_okey.alloc((size_t)NU * 8));
  hipLaunchKernelGGL(k_weld, dim3((NV + 255) / 256), dim3(256), 0, s, d_vkey_s.as<uint64_t>(), d_src_s.as<uint32_t>(), d_flag.as<uint32_t>(),
                     d_excl.as<uint32_t>(), NV, d_vpos.as<float>(), d_vcol.as<uint32_t>(), d_vid.as<uint32_t>(), d_opos.as<float>(),
                     d_ocol.as<uint32_t>(), d_okey.as<uint64_t>());
  clk.mark();   // 4: heads, scan, weld
  // ---- canonical triangle order: sort by (cube key, table order)
  DevBuf d_tkey_s, d_tsrc, d_tsrc_s, d_tmp4, d_oidx;
  MC_CHECK(d_tkey_s.alloc((size_t)T * 8));
  MC_CHECK(d_tsrc.alloc((size_t)T * 4));
  MC_CHECK(d_tsrc_s.alloc((size_t)T * 4));
  hipLaunchKernelGGL(k_iota, dim3((T + 255) / 256), dim3(256), 0, s, d_tsrc.as<uint32_t>(), T);
  size_t tsort_bytes = 0;
  MC_CHECK(rocprim::radix_sort_pairs(nullptr, tsort_bytes, d_tkey.as<uint64_t>(), d_tkey_s.as<uint64_t>(), d_tsrc.as<uint32_t>(), d_tsrc_s.as<uint32_t>(), (size_t)T, 0, 64, s));
  MC_CHECK(d_tmp4.alloc(tsort_bytes));
  MC_CHECK(rocprim::radix_sort_pairs(d_tmp4.p, tsort_bytes, d_tkey.as<uint64_t>(), d_tkey_s.as<uint64_t>(), d_tsrc.as<uint32_t>(), d_tsrc_s.as<uint32_t>(), (size_t)T, 0, 64, s));
  MC_CHECK(d_oidx.alloc((size_t)T * 12));
  hipLaunchKernelGGL(k_gather_tris, dim3((T + 255) / 256), dim3(256), 0, s, d_tsrc_s.as<uint32_t>(), d_vid.as<uint32_t>(), T, d_oidx.as<uint32_t>());
  clk.mark();   // 5: triangle sort + gather
  // ---- download
  const double t_queued = wall_ms();
  m->pos.resize((size_t)NU * 3);
  m->col.resize((size_t)NU * 4);
  m->keys.resize(NU);
  m->tri.resize((size_t)T * 3);
  m->tkeys.resize(T);
  {
    const DlSeg segs[5] = {{m->tkeys.data(), d_tkey_s.p, (size_t)T * 8}, {m->pos.data(), d_opos.p, (size_t)NU * 12}, {m->col.data(), d_ocol.p, (size_t)NU * 4},
                           {m->keys.data(), d_okey.p, (size_t)NU * 8}, {m->tri.data(), d_oidx.p, (size_t)T * 12}};
    rc = download_segments(f, s, segs, 5);
    if (rc != SF_OK) return rc;
  }
  clk.mark();   // 6: downloads
  MC_CHECK(hipStreamSynchronize(s));
  MC_CHECK(hipGetLastError());
  const double t_end = wall_ms();
  // [0] whole call  [1] live-block list  [2] count pass  [3] emit pass  [4] vertex sort  [5] heads + scan + weld  [6] triangle sort + gather
  // [7] downloads (device time)  [8] host: output arrays allocated and zero-filled + waiting for the downloads  [9] live blocks  [10] triangles  [11] vertices
  double* tm = f->mc_timing;
  tm[0] = t_end - t_begin; tm[1] = t_compact - t_begin; tm[2] = clk.ms(0, 1); tm[3] = clk.ms(1, 2); tm[4] = clk.ms(2, 3); tm[5] = clk.ms(3, 4);
  tm[6] = clk.ms(4, 5); tm[7] = clk.ms(5, 6); tm[8] = t_end - t_queued; tm[9] = (double)n_live; tm[10] = (double)T; tm[11] = (double)NU;
  (void)t_counted;
  *out = m.release();
  return SF_OK;
}

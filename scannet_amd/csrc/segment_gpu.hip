// segment_gpu.hip -- the two data-parallel stages of the Segmentator on gfx950: vertex normals and edge weights.
//
// Replaces, for sf_segment_mesh_gpu, the loops of Segmentator/segmentator.cpp:185-229 that csrc/segment.cpp runs on the host.  What is sequential by
// definition is made parallel by ordering, not by reassociating:
//   * a vertex normal is the RUNNING MEAN of the unit normals of the vertex's faces in face order (n <- t fn + (1 - t) n, t = 1 / (corners of earlier
//     faces + 1): segmentator.cpp:113-116, :205-207) -- its value depends on the order, so the host walks the faces one after the other.  Here the 3F face
//     corners are sorted by vertex (stable radix sort: within a vertex they stay in corner = face order), and ONE LANE PER VERTEX walks its own corners
//     in that order: the same chain of separately rounded fp32 operations per vertex, F / 64 of them side by side;
//   * the unit face normals (cross product, sqrtf, three divisions) and the edge weights (1 - n_u . n_v, squared on convex edges) are independent per
//     face / per edge: one lane each.
// Every operation is the host's, un-contracted (-ffp-contract=off) IEEE fp32 with HIP's correctly rounded division and square root: the weight keys are
// the host's bit for bit (NaN normals of zero-area faces included), and everything behind them -- libstdc++'s std::sort, the two sweeps -- stays on the host
// (segment.cpp: the permutation of tied and NaN weights must be that sort's).  tests/test_segmentator_gpu.py holds the labels against the host path's and
// the reference binary's.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>   // rocprim's texture iterator calls memset without including it

#include <rocprim/rocprim.hpp>

#include <vector>

#include "common.h"

namespace {

struct WeightKeyD {
  float w;
  uint32_t edge;
};

__global__ __launch_bounds__(256) void k_seg_corner_keys(const uint32_t* __restrict__ tri, uint32_t n, uint32_t* __restrict__ key, uint32_t* __restrict__ corner) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) { key[i] = tri[i]; corner[i] = i; }
}

// unit normal of every face: (B - A) x (C - A) divided by its length (NaN for a zero-area face: segmentator.cpp:107-112)
__global__ __launch_bounds__(256) void k_seg_face_normals(const float* __restrict__ xyz, const uint32_t* __restrict__ tri, uint32_t nf, float* __restrict__ fn) {
  const uint32_t f = blockIdx.x * 256u + threadIdx.x;
  if (f >= nf) return;
  const uint32_t* t = tri + 3 * (size_t)f;
  const float* A = xyz + 3 * (size_t)t[0];
  const float* B = xyz + 3 * (size_t)t[1];
  const float* C = xyz + 3 * (size_t)t[2];
  const float ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2];
  const float vx = C[0] - A[0], vy = C[1] - A[1], vz = C[2] - A[2];
  float fx = uy * vz - uz * vy, fy = uz * vx - ux * vz, fz = ux * vy - uy * vx;
  const float flen = sqrtf(fx * fx + fy * fy + fz * fz);
  fx /= flen; fy /= flen; fz /= flen;
  fn[3 * (size_t)f] = fx; fn[3 * (size_t)f + 1] = fy; fn[3 * (size_t)f + 2] = fz;
}

// one lane per vertex: its corners (sorted by vertex, in face order inside) blended into the running mean exactly as the host's face loop does for it
__global__ __launch_bounds__(256) void k_seg_vertex_normals(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ scorner, uint32_t nc, const float* __restrict__ fn,
                                                            uint32_t nv, float4* __restrict__ vn) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= nv) return;
  uint32_t lo = 0, hi = nc;   // first corner whose vertex is >= v
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (skey[mid] < v) lo = mid + 1; else hi = mid;
  }
  float nx = 0.0f, ny = 0.0f, nz = 0.0f;
  uint32_t seen = 0, done = 0, cur_face = 0xFFFFFFFFu;   // seen: corners of EARLIER faces (the count the host holds while it blends a face: it moves after all three corners)
  for (uint32_t i = lo; i < nc && skey[i] == v; i++) {
    const uint32_t f = scorner[i] / 3u;
    if (f != cur_face) { seen = done; cur_face = f; }
    const float wnew = 1.0f / ((float)seen + 1.0f), wold = 1.0f - wnew;
    const float fx = fn[3 * (size_t)f], fy = fn[3 * (size_t)f + 1], fz = fn[3 * (size_t)f + 2];
    nx = wnew * fx + wold * nx;
    ny = wnew * fy + wold * ny;
    nz = wnew * fz + wold * nz;
    done++;
  }
  vn[v] = make_float4(nx, ny, nz, 0.0f);
}

// edge 3f + c of face (i, j, k): c = 0: i-j, c = 1: i-k, c = 2: k-j (segmentator.cpp:199-204); weight :211-229
__global__ __launch_bounds__(256) void k_seg_edge_weights(const float* __restrict__ xyz, const uint32_t* __restrict__ tri, const float4* __restrict__ vn, uint32_t ne,
                                                          WeightKeyD* __restrict__ keys) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= ne) return;
  const uint32_t* t = tri + 3 * (size_t)(e / 3u);
  const uint32_t c = e % 3u;
  const uint32_t u = c == 2u ? t[2] : t[0], w_ = c == 1u ? t[2] : t[1];
  const float* P = xyz + 3 * (size_t)u;
  const float* Q = xyz + 3 * (size_t)w_;
  const float4 nu = vn[u], nw = vn[w_];
  float ex = Q[0] - P[0], ey = Q[1] - P[1], ez = Q[2] - P[2];
  const float elen = sqrtf(ex * ex + ey * ey + ez * ez);
  ex /= elen; ey /= elen; ez /= elen;
  const float across = nu.x * nw.x + nu.y * nw.y + nu.z * nw.z;
  const float along = nw.x * ex + nw.y * ey + nw.z * ez;
  float w = 1.0f - across;
  if (along > 0) w = w * w;
  keys[e] = WeightKeyD{w, e};
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

// keys_out: 3 * nf records {weight, edge number} (host memory), what segment.cpp's own loops produce.  Indices are validated by the caller.
int segment_weight_keys_gpu(const float* xyz, size_t nv, const uint32_t* tri, size_t nf, int device, void* keys_out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device: sf_segment_mesh_gpu needs one (sf_segment_mesh is the host path)");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  const size_t nc = nf * 3;
  if (nc == 0) return SF_OK;
  DevBuf d_xyz, d_tri, d_fn, d_vn, d_k0, d_k1, d_c0, d_c1, d_keys, d_tmp;
  SF_HIP_CHECK(d_xyz.alloc(nv * 12)); SF_HIP_CHECK(d_tri.alloc(nc * 4)); SF_HIP_CHECK(d_fn.alloc(nf * 12)); SF_HIP_CHECK(d_vn.alloc(nv * 16));
  SF_HIP_CHECK(d_k0.alloc(nc * 4)); SF_HIP_CHECK(d_k1.alloc(nc * 4)); SF_HIP_CHECK(d_c0.alloc(nc * 4)); SF_HIP_CHECK(d_c1.alloc(nc * 4)); SF_HIP_CHECK(d_keys.alloc(nc * 8));
  hipStream_t s = nullptr;
  SF_HIP_CHECK(hipMemcpyAsync(d_xyz.p, xyz, nv * 12, hipMemcpyHostToDevice, s));
  SF_HIP_CHECK(hipMemcpyAsync(d_tri.p, tri, nc * 4, hipMemcpyHostToDevice, s));
  const unsigned gc = (unsigned)((nc + 255) / 256), gf = (unsigned)((nf + 255) / 256), gv = (unsigned)((nv + 255) / 256);
  hipLaunchKernelGGL(k_seg_corner_keys, dim3(gc), dim3(256), 0, s, d_tri.as<uint32_t>(), (uint32_t)nc, d_k0.as<uint32_t>(), d_c0.as<uint32_t>());
  hipLaunchKernelGGL(k_seg_face_normals, dim3(gf), dim3(256), 0, s, d_xyz.as<float>(), d_tri.as<uint32_t>(), (uint32_t)nf, d_fn.as<float>());
  size_t need = 0;
  int bits = 1;
  while (bits < 32 && (nv >> bits) != 0) bits++;   // the keys are vertex numbers below nv
  SF_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, d_k0.as<uint32_t>(), d_k1.as<uint32_t>(), d_c0.as<uint32_t>(), d_c1.as<uint32_t>(), nc, 0, (unsigned)bits, s));
  SF_HIP_CHECK(d_tmp.alloc(need));
  SF_HIP_CHECK(rocprim::radix_sort_pairs(d_tmp.p, need, d_k0.as<uint32_t>(), d_k1.as<uint32_t>(), d_c0.as<uint32_t>(), d_c1.as<uint32_t>(), nc, 0, (unsigned)bits, s));
  if (nv) hipLaunchKernelGGL(k_seg_vertex_normals, dim3(gv), dim3(256), 0, s, d_k1.as<uint32_t>(), d_c1.as<uint32_t>(), (uint32_t)nc, d_fn.as<float>(), (uint32_t)nv, d_vn.as<float4>());
  hipLaunchKernelGGL(k_seg_edge_weights, dim3(gc), dim3(256), 0, s, d_xyz.as<float>(), d_tri.as<uint32_t>(), d_vn.as<float4>(), (uint32_t)nc, d_keys.as<WeightKeyD>());
  SF_HIP_CHECK(hipGetLastError());
  SF_HIP_CHECK(hipMemcpyAsync(keys_out, d_keys.p, nc * 8, hipMemcpyDeviceToHost, s));
  SF_HIP_CHECK(hipStreamSynchronize(s));
  return SF_OK;
}

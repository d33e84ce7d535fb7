// h2d_bw.hip -- what the host -> HBM link gives on this box, to size sf_fuse_run's copy pipeline: hipMemcpyAsync from pinned memory on
// 1, 2, 4, 8 streams (each stream = one SDMA engine at a time) and a kernel reading the pinned buffer directly (zero-copy).
//   hipcc -O2 --offload-arch=gfx950 tools/gpu/h2d_bw.hip -o /tmp/h2d_bw && /tmp/h2d_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t total = 1ull << 30, chunk = 10ull << 20;
  void *h = nullptr, *d = nullptr;
  CK(hipHostMalloc(&h, total, hipHostMallocDefault));
  CK(hipMalloc(&d, total));
  std::memset(h, 1, total);
  printf("{");
  for (int ns : {1, 2, 3, 4, 8}) {
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      size_t off = 0; int k = 0;
      while (off < total) { const size_t n = std::min(chunk, total - off); CK(hipMemcpyAsync((char*)d + off, (char*)h + off, n, hipMemcpyHostToDevice, st[k++ % ns])); off += n; }
      CK(hipDeviceSynchronize());
      if (rep == 1) printf("\"memcpy_%d_streams_GBs\": %.1f, ", ns, total / (now() - t0) / 1e9);
    }
    for (auto& s : st) CK(hipStreamDestroy(s));
  }
  for (int wgs : {64, 256, 1024, 4096}) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      hipLaunchKernelGGL(k_pull, dim3(wgs), dim3(256), 0, 0, (const uint4*)h, (uint4*)d, total / 16);
      CK(hipDeviceSynchronize());
      if (rep == 1) printf("\"kernel_pull_%d_wgs_GBs\": %.1f, ", wgs, total / (now() - t0) / 1e9);
    }
  }
  {  // a pull kernel beside two copy streams
    hipStream_t a, b, c;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    const double t0 = now();
    const size_t half = total / 2;
    hipLaunchKernelGGL(k_pull, dim3(256), dim3(256), 0, c, (const uint4*)h, (uint4*)d, half / 16);
    size_t off = half; int k = 0;
    while (off < total) { const size_t n = std::min(chunk, total - off); CK(hipMemcpyAsync((char*)d + off, (char*)h + off, n, hipMemcpyHostToDevice, (k++ & 1) ? a : b)); off += n; }
    CK(hipDeviceSynchronize());
    printf("\"pull_plus_2_copy_streams_GBs\": %.1f", total / (now() - t0) / 1e9);
  }
  printf("}\n");
  return 0;
}

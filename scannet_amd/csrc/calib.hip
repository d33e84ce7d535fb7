// calib.hip -- PMC calibration helper: a plain 16 B-per-lane read-modify-write stream over a known byte
// count, the access pattern of k_integrate.  Used only to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md section HBM: FETCH_SIZE under-reports wide coalesced reads) -- see
// tools/pmc_calibrate.py and DESIGN.md section 5.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {
__global__ __launch_bounds__(256) void k_calib_rmw(uint4* __restrict__ buf, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    uint4 v = buf[i];
    v.x += 1u; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
    buf[i] = v;
  }
}
__global__ __launch_bounds__(256) void k_calib_read(const uint4* __restrict__ buf, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = buf[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) *sink = acc;
}
}  // namespace

// Runs `iters` launches of a read+write stream and `iters` of a read-only stream over `bytes` bytes.
SF_API int sf_calib_stream(int device, uint64_t bytes, int iters) {
  SF_HIP_CHECK(hipSetDevice(device));
  uint4* buf = nullptr;
  uint32_t* sink = nullptr;
  SF_HIP_CHECK(hipMalloc((void**)&buf, bytes));
  SF_HIP_CHECK(hipMalloc((void**)&sink, 4));
  SF_HIP_CHECK(hipMemset(buf, 1, bytes));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_calib_rmw, dim3(8192), dim3(256), 0, 0, buf, (size_t)(bytes / 16));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_calib_read, dim3(8192), dim3(256), 0, 0, buf, (size_t)(bytes / 16), sink);
  SF_HIP_CHECK(hipDeviceSynchronize());
  (void)hipFree(buf);
  (void)hipFree(sink);
  return SF_OK;
}

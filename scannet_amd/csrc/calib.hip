// calib.hip -- PMC calibration helper: a plain 16 B-per-lane read-modify-write stream over a known byte
// count, the access pattern of k_integrate.  Used only to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md section HBM: FETCH_SIZE under-reports wide coalesced reads) -- see
// tools/pmc_calibrate.py and DESIGN.md section 5.
#include <hip/hip_runtime.h>

#include "common.h"
#include "fuser_internal.h"

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

// ---------------------------------------------------------------------------------------------------------------
// Device self-test of the hand-expanded divisions (fuser_internal.h) against the hardware's IEEE division:
//   recip: every one of the 2^23 mantissas at 9 exponents spanning 2^-20 .. 2^20 (camera-space depths in metres),
//   quot : every integer divisor 1..511 against 2^20 numerators each (counter-based bit patterns with exponents in the
//          TSDF range, plus exact multiples and their neighbours: the near-halfway quotients).
//   div  : k_alloc's general quotient div_rn(a, b, recip_rn(b)): 2^28 operand pairs -- counter-based bit patterns with both signs and
//          exponents 2^-24 .. 2^16 for the divisor (ray directions, the voxel size) and 2^-30 .. 2^12 for the dividend, plus for every pair
//          the dividends that sit within 3 ulp of an exact multiple and of a half-way multiple of the divisor (the hard cases of rounding).
// Returns the number of lanes whose result differs in any bit (expected: 0, 0 and 0).
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_selftest_recip(unsigned long long* bad) {
  const uint32_t mant = blockIdx.x * 256 + threadIdx.x;  // 2^23 threads
  const uint32_t expo[9] = {107, 117, 122, 126, 127, 128, 132, 137, 147};
  unsigned int n = 0;
  for (int e = 0; e < 9; e += 2) {
    const float b0 = __uint_as_float((expo[e] << 23) | mant), b1 = __uint_as_float((expo[(e + 1) % 9] << 23) | mant);
    const v2f r = recip_rn((v2f){b0, b1});
    n += __float_as_uint(r.x) != __float_as_uint(1.0f / b0);
    n += __float_as_uint(r.y) != __float_as_uint(1.0f / b1);
  }
  if (n) atomicAdd(bad, (unsigned long long)n);
}
__global__ __launch_bounds__(256) void k_selftest_quot(unsigned long long* bad) {
  const uint32_t m_int = blockIdx.y + 1;  // 1..511
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;  // 2^20 numerators per divisor
  const float m = (float)m_int, r = 1.0f / m;
  uint64_t x = ((uint64_t)m_int << 32 | i) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  float n0, n1;
  {
    const uint32_t e = 100u + (uint32_t)(x >> 58);  // 2^-27 .. 2^36
    n0 = __uint_as_float(((uint32_t)x & 0x807FFFFFu) | (e << 23));
    const float qf = __uint_as_float(0x3F800000u | ((uint32_t)(x >> 24) & 0x7FFFFFu));  // quotient near 1..2
    n1 = __uint_as_float(__float_as_uint(qf * m) + ((uint32_t)(x >> 50) & 7u) - 3u);    // a multiple of m, +-3 ulp
  }
  const v2f q = quot_rn((v2f){n0, n1}, splat(m), splat(r));
  unsigned int n = (__float_as_uint(q.x) != __float_as_uint(n0 / m)) + (__float_as_uint(q.y) != __float_as_uint(n1 / m));
  if (n) atomicAdd(bad, (unsigned long long)n);
}
__global__ __launch_bounds__(256) void k_selftest_div(unsigned long long* bad) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;   // 2^26 threads, 4 quotients each
  uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  uint64_t y = (x ^ 0xD6E8FEB86659FD93ull) * 0x94D049BB133111EBull;
  y ^= y >> 31;
  const uint32_t eb = 103u + (uint32_t)(x >> 58) % 41u;          // divisor 2^-24 .. 2^16
  const uint32_t ea = 97u + (uint32_t)(y >> 58) % 43u;           // dividend 2^-30 .. 2^12
  const float b = __uint_as_float(((uint32_t)x & 0x807FFFFFu) | (eb << 23));
  const float a0 = __uint_as_float(((uint32_t)y & 0x807FFFFFu) | (ea << 23));
  const float qf = __uint_as_float(0x3F800000u | ((uint32_t)(y >> 24) & 0x7FFFFFu));                         // a quotient in [1, 2)
  const float a1 = __uint_as_float(__float_as_uint(qf * b) + ((uint32_t)(x >> 50) & 7u) - 3u);               // near an exact multiple of b
  const float qh = __uint_as_float(__float_as_uint(qf) & 0xFFFFFFFEu);                                       // even significand: qh + ulp/2 is half-way
  const float a2 = __uint_as_float(__float_as_uint(fmaf(qh, b, 0x1p-24f * b)) + ((uint32_t)(y >> 50) & 7u) - 3u);   // near a half-way quotient
  const float a3 = __uint_as_float(__float_as_uint(a0) ^ 0x00400000u);
  const float rb = recip_rn(b);
  unsigned int n = 0;
  n += __float_as_uint(div_rn(a0, b, rb)) != __float_as_uint(a0 / b);
  n += __float_as_uint(div_rn(a1, b, rb)) != __float_as_uint(a1 / b);
  n += __float_as_uint(div_rn(a2, b, rb)) != __float_as_uint(a2 / b);
  n += __float_as_uint(div_rn(a3, b, rb)) != __float_as_uint(a3 / b);
  n += __float_as_uint(rb) != __float_as_uint(1.0f / b);
  if (n) atomicAdd(bad, (unsigned long long)n);
}
}  // namespace

SF_API int sf_selftest_division(int device, uint64_t* recip_mismatches, uint64_t* quot_mismatches, uint64_t* div_mismatches) {
  if (!recip_mismatches || !quot_mismatches || !div_mismatches) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(device));
  unsigned long long* d = nullptr;
  SF_HIP_CHECK(hipMalloc((void**)&d, 24));
  SF_HIP_CHECK(hipMemset(d, 0, 24));
  hipLaunchKernelGGL(k_selftest_recip, dim3((1u << 23) / 256), dim3(256), 0, nullptr, d);
  hipLaunchKernelGGL(k_selftest_quot, dim3((1u << 20) / 256, 511), dim3(256), 0, nullptr, d + 1);
  hipLaunchKernelGGL(k_selftest_div, dim3((1u << 26) / 256), dim3(256), 0, nullptr, d + 2);
  unsigned long long h[3] = {0, 0, 0};
  const hipError_t e = hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "self-test failed: %s", hipGetErrorString(e));
  *recip_mismatches = h[0];
  *quot_mismatches = h[1];
  *div_mismatches = h[2];
  return SF_OK;
}

// jpeg_gpu.hip -- the data-parallel half of baseline JPEG decoding on gfx950.  The frame pipeline's host threads only entropy-decode
// (jpeg.cpp: jpeg_decode_coef -- Huffman decoding is serial per frame); the non-zero quantised coefficients go over PCIe in place of the
// RGB image (about a third of its bytes) and two kernels reconstruct the picture where the fuser wants it anyway, in HBM:
//   k_jpeg_idct   one lane per 8 x 8 block: its non-zero coefficients (fetched eight at a time) scattered into a zeroed 16-bit LDS column, dequantised,
//                 integer 1-D passes down the columns and along the rows, clamp, eight 8-byte stores into the plane
//   k_jpeg_rgb    four pixels per lane: chroma upsampling, fixed-point YCbCr -> RGB, three dword stores
// Every arithmetic step is the integer function of jpeg_idct.h the host decoder is built from, so the bytes are the ones
// sf_sens_decode_color produces -- and the ones the reference's decoder produces (tests/test_gpu_pipeline.py, tests/test_sens.py).
// Replaces, for this path, the SSE2 IDCT + resampling + colour conversion of stb_image as RGBDFrame::decompressColorAlloc_stb
// calls it (SensReader/c++/src/sensorData.h:609-616, stb_image.h:2028-2207).
#include <hip/hip_runtime.h>

#include <atomic>
#include <vector>

#include "common.h"
#include "jpeg_idct.h"

namespace {

constexpr int JPEG_MAX_BATCH = 16;

struct JpegBatch {
  const uint8_t* payload[JPEG_MAX_BATCH];   // SfJpegLayout + coefficients, device
  uint8_t* rgb[JPEG_MAX_BATCH];             // W x H x 3 out, device; nullptr: slot unused
  uint8_t* planes[JPEG_MAX_BATCH];          // scratch for the component planes of this frame
};

// LDS: one 16-bit column of 64 dequantised coefficients per lane (32 KiB per workgroup: five workgroups per CU; as 32-bit words it was 64 KiB, two per CU, and the
// kernel -- two waves per SIMD, each waiting for its block's entries one load at a time -- took 280-410 us per 16 pictures beside the fusion).
__global__ __launch_bounds__(256) void k_jpeg_idct(JpegBatch B) {
  __shared__ uint4 s_blk4[64 * 256 * 2 / 16];   // [64][256] int16: coefficient z of lane t at z * 256 + t
  int16_t* const s_blk = reinterpret_cast<int16_t*>(s_blk4);
  const int f = blockIdx.y;
  if (B.rgb[f] == nullptr) return;
  const SfJpegLayout* __restrict__ L = reinterpret_cast<const SfJpegLayout*>(B.payload[f]);
  const uint32_t* __restrict__ table = reinterpret_cast<const uint32_t*>(B.payload[f] + sizeof(SfJpegLayout));
  const uint32_t* __restrict__ entries = table + L->nblocks;
  __shared__ int s_q[3][64];
  for (int t = threadIdx.x; t < 64 * L->ncomp; t += 256) s_q[t >> 6][t & 63] = L->q[t >> 6][t & 63];
#pragma unroll
  for (int k = 0; k < 8; k++) s_blk4[k * 256 + threadIdx.x] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t block = blockIdx.x * 256 + threadIdx.x;   // block index over all components
  if (block >= L->nblocks) return;
  int c = L->ncomp - 1;
  while (c > 0 && block < L->block_off[c]) c--;
  size_t plane_off = 0;
  for (int i = 0; i < c; i++) plane_off += (size_t)L->bw[i] * L->bh[i];
  const uint32_t b = block - L->block_off[c];
  const int blocks_w = L->bw[c] / 8;
  // the block's non-zero coefficients scattered into the lane's (zeroed) LDS column, dequantised on the way; eight entries are asked for at a time
  int16_t* col = s_blk + threadIdx.x;
  const uint32_t te = table[block];
  const uint32_t* e = entries + (te >> 7);
  const uint32_t cnt = te & 127u;
  for (uint32_t k0 = 0; k0 < cnt; k0 += 8) {
    uint32_t w[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) w[k] = e[min(k0 + k, cnt - 1u)];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++)
      if (k0 + k < cnt) {
        const int z = (int)((w[k] >> 16) & 63u);
        col[z * 256] = (int16_t)sf_jpeg_dequant16((int)(int16_t)(w[k] & 0xffffu), s_q[c][z]);
      }
  }
  int blk[64];
#pragma unroll
  for (int z = 0; z < 64; z++) blk[z] = col[z * 256];
  sf_idct_block_int(blk);
  uint8_t* out = B.planes[f] + plane_off + (size_t)(b / blocks_w) * 8 * L->bw[c] + (size_t)(b % blocks_w) * 8;
#pragma unroll
  for (int y = 0; y < 8; y++) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
      lo |= (uint32_t)blk[8 * y + x] << (8 * x);
      hi |= (uint32_t)blk[8 * y + 4 + x] << (8 * x);
    }
    *reinterpret_cast<uint2*>(out + (size_t)y * L->bw[c]) = make_uint2(lo, hi);
  }
}

// Four pixels of one row per lane, a workgroup = 4 rows x 256 pixels, blockIdx.z = frame: no division by the (run-time) image width anywhere
// (round-2 profile: the linear-index version spent 235 us per 16 frames of 1296x968, most of it in 64-bit i / W and i % W).
__global__ __launch_bounds__(256) void k_jpeg_rgb(JpegBatch B) {
  const int f = blockIdx.z;
  if (B.rgb[f] == nullptr) return;
  const SfJpegLayout* __restrict__ L = reinterpret_cast<const SfJpegLayout*>(B.payload[f]);
  const int W = L->width, H = L->height;
  const int x0 = ((int)blockIdx.x * 64 + (int)(threadIdx.x & 63)) * 4, y = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
  if (x0 >= W || y >= H) return;
  const uint8_t* p0 = B.planes[f];
  const int ncomp = L->ncomp;
  // per component: sampling ratio (1 or 2 on this path: jpeg_decode_coef leaves anything else to the host decoder), valid samples, plane
  int sx[3], sy[3], cw[3], ch[3], bw[3];
  const uint8_t* plane[3];
  {
    const uint8_t* q = p0;
    for (int c = 0; c < 3; c++) {
      const int cc = c < ncomp ? c : 0;
      sx[c] = L->hmax > L->h[cc] ? 2 : 1; sy[c] = L->vmax > L->v[cc] ? 2 : 1;
      cw[c] = (W + sx[c] - 1) >> (sx[c] - 1); ch[c] = (H * L->v[cc] + L->vmax - 1) >> (L->vmax - 1);
      bw[c] = L->bw[cc];
      plane[c] = q;
      if (c < ncomp) q += (size_t)L->bw[cc] * L->bh[cc];
    }
  }
  uint8_t px[12];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = x0 + k;
    uint8_t* o = px + 3 * k;
    o[0] = o[1] = o[2] = 0;
    if (x >= W) continue;
    if (ncomp == 1) { o[0] = o[1] = o[2] = p0[(size_t)y * bw[0] + x]; continue; }
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      __builtin_assume(sx[c] >= 1 && sx[c] <= 2 && sy[c] >= 1 && sy[c] <= 2);
      v[c] = sf_jpeg_upsample(plane[c], bw[c], cw[c], ch[c], sx[c], sy[c], x, y);
    }
    sf_jpeg_ycc_to_rgb(v[0], v[1], v[2], o);
  }
  uint8_t* dst = B.rgb[f] + 3 * ((size_t)y * W + x0);
  if (x0 + 4 <= W && ((uintptr_t)dst & 3) == 0) {
    uint32_t* o = reinterpret_cast<uint32_t*>(dst);
    o[0] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)px[3] << 24);
    o[1] = (uint32_t)px[4] | ((uint32_t)px[5] << 8) | ((uint32_t)px[6] << 16) | ((uint32_t)px[7] << 24);
    o[2] = (uint32_t)px[8] | ((uint32_t)px[9] << 8) | ((uint32_t)px[10] << 16) | ((uint32_t)px[11] << 24);
  } else {
    for (int k = 0; k < 4 && x0 + k < W; k++) { dst[3 * k] = px[3 * k]; dst[3 * k + 1] = px[3 * k + 1]; dst[3 * k + 2] = px[3 * k + 2]; }
  }
}

}  // namespace

// Reconstruct up to 16 entropy-decoded frames on `stream`.  d_payload[i]: SfJpegLayout + coefficients (16-byte aligned), d_rgb[i]: the
// RGB image out (nullptr: skip the slot), d_planes[i]: scratch of at least the summed plane sizes.  max_blocks / max_width x max_height bound the
// grid (the layouts live on the device).
// The code object of this file is loaded by the runtime when one of its kernels is first used (milliseconds, inside a scan's first sf_fuse_run unless somebody asks
// earlier): the preparation thread of the frame pipeline asks (pipeline.hip, sf_run_resources_prepare_ex).
void jpeg_gpu_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_jpeg_idct));
  (void)hipGetLastError();
}

int jpeg_gpu_reconstruct(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_rgb, uint8_t* const* d_planes, uint32_t max_blocks,
                         uint32_t max_width, uint32_t max_height) {
  if (n < 1 || n > JPEG_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_gpu_reconstruct: %d frames", n);
  JpegBatch b;
  for (int i = 0; i < JPEG_MAX_BATCH; i++) {
    b.payload[i] = i < n ? d_payload[i] : nullptr;
    b.rgb[i] = i < n ? d_rgb[i] : nullptr;
    b.planes[i] = i < n ? d_planes[i] : nullptr;
  }
  hipLaunchKernelGGL(k_jpeg_idct, dim3((max_blocks + 255) / 256, n), dim3(256), 0, stream, b);
  hipLaunchKernelGGL(k_jpeg_rgb, dim3((max_width + 255) / 256, (max_height + 3) / 4, n), dim3(256), 0, stream, b);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
}

// The component planes only (k_jpeg_idct): for a consumer that converts the pixels it needs itself -- the fuser's pre-pass looks up ONE colour pixel per depth
// pixel (640x480 of a 1296x968 picture: a quarter of them), so the frame pipeline no longer has k_jpeg_rgb write 3.8 MB of RGB per picture for it to pick from.
int jpeg_gpu_planes(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_planes, uint32_t max_blocks) {
  if (n < 1 || n > JPEG_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_gpu_planes: %d frames", n);
  JpegBatch b;
  for (int i = 0; i < JPEG_MAX_BATCH; i++) {
    b.payload[i] = i < n ? d_payload[i] : nullptr;
    b.rgb[i] = i < n ? d_planes[i] : nullptr;   // k_jpeg_idct only asks whether the slot is used
    b.planes[i] = i < n ? d_planes[i] : nullptr;
  }
  hipLaunchKernelGGL(k_jpeg_idct, dim3((max_blocks + 255) / 256, n), dim3(256), 0, stream, b);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
}

int jpeg_decode_rgb(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h);                                          // jpeg.cpp
int jpeg_decode_coef(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);  // jpeg.cpp


// The same picture through the GPU path of the frame pipeline: entropy decoding here, reconstruction on `device`, result copied back.
// One frame, synchronous -- an entry point for callers that want the pixels in HBM anyway and for the parity test; SF_ERR_UNSUPPORTED
// for layouts the GPU path leaves to the host decoder (sampling factors above 2).
SF_API int sf_jpeg_decode_gpu(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, int device, uint8_t* dst_rgb) {
  if (!data || !dst_rgb || width == 0 || height == 0) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device: libscanfuse has no CPU fallback for this entry point (sf_jpeg_decode is the host decoder)");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  const uint64_t padded = (uint64_t)((width + 15) & ~15u) * ((height + 15) & ~15u);
  std::vector<uint32_t> host((sizeof(SfJpegLayout) + padded * 3 / 64 * 4 + padded * 3 * 4) / 4 + 64);   // every coefficient of a 4:4:4 frame non-zero
  const int rc = jpeg_decode_coef(data, bytes, width, height, reinterpret_cast<uint8_t*>(host.data()), host.size() * 4);
  if (rc != SF_OK) return rc;
  const SfJpegLayout* L = reinterpret_cast<const SfJpegLayout*>(host.data());
  uint8_t *d_pay = nullptr, *d_rgb = nullptr, *d_planes = nullptr;
  const size_t pay_b = sf_jpeg_payload_bytes(*L), rgb_b = (size_t)width * height * 3;
  auto release = [&]() { if (d_pay) (void)hipFree(d_pay); if (d_rgb) (void)hipFree(d_rgb); if (d_planes) (void)hipFree(d_planes); };
  hipError_t e = hipMalloc((void**)&d_pay, pay_b);
  if (e == hipSuccess) e = hipMalloc((void**)&d_rgb, rgb_b);
  if (e == hipSuccess) e = hipMalloc((void**)&d_planes, sf_jpeg_plane_bytes(*L));
  if (e == hipSuccess) e = hipMemcpy(d_pay, host.data(), pay_b, hipMemcpyHostToDevice);
  int out = SF_OK;
  if (e == hipSuccess) {
    const uint8_t* pp = d_pay;
    out = jpeg_gpu_reconstruct(nullptr, 1, &pp, &d_rgb, &d_planes, L->nblocks, width, height);
    if (out == SF_OK) e = hipMemcpy(dst_rgb, d_rgb, rgb_b, hipMemcpyDeviceToHost);
  }
  release();
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "sf_jpeg_decode_gpu: %s", hipGetErrorString(e));
  return out;
}

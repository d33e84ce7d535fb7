// filter2d.hip -- the 2-D annotation filter on gfx950: the CUDA kernels of AnnotationTools/Filter2dAnnotations/filter.cu that
// Filter2dAnnotations.cpp calls, and the per-frame sequence in which it calls them (Filter2dAnnotations.cpp:326-397).
//   k_f2d_prepare         convertToFloat / convertToGrayscale (host loops in the reference)   Filter2dAnnotations.cpp:232-256
//   k_f2d_bilateral       bilateralFilterFloatMapDevice                                       filter.cu:210-247
//   k_f2d_resample_float  resampleFloatMapDevice + bilinearInterpolationFloat                 filter.cu:514-560
//   k_f2d_resample_uchar  resampleUCharMapDevice                                              filter.cu:647-665
//   k_f2d_vote            filterAnnotations_Kernel                                            filter.cu:1020-1059
//   k_f2d_to_label        convertInstanceToLabel_Kernel                                       filter.cu:1082-1091
// What is MI355X-specific: the reference keeps the per-pixel vote histogram (80 floats) in GLOBAL memory -- a 401 MB scratch
// buffer at 1296x968, memset before every launch and hit with a read-modify-write per window tap (filter.cu:1045,1069).  Here the
// 80 x 256 histogram of a workgroup lives in LDS (80 KiB of the CU's 160 KiB; bin-major, so a wave's 64 lanes always hit 64
// different banks whatever bins they vote for): no scratch buffer, no memset, no HBM traffic for votes.  The spatial Gaussian of
// a launch is tabulated once per workgroup in LDS ((2r+1)^2 floats); the range Gaussians are evaluated per tap in binary64 as
// the source's 2.0 literal demands (gaussR, filter.cu:190-193) -- this is fp64-ALU-bound work, which gfx950 has.
// Arithmetic: statement by statement as the CUDA source, no contraction (-ffp-contract=off), exp() = sf_exp64_t (exp64.h, its table in LDS).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "common.h"
#include "exp64.h"

namespace {

constexpr int F2D_LABELS = 80;  // MAX_NUM_LABELS_PER_SCENE, GlobalDefines.h:12
#define F2D_MINF (-INFINITY)

// exp(-(double)(dist * dist) / (2.0 * sigma * sigma)) (filter.cu:190-193) with the divisor c = 2 sigma^2 and rc = RN(1 / c) fixed per
// launch: q = a * rc, r = fma(-c, q, a) (exact), q + r * rc is the correctly rounded quotient a / c (Markstein; rc is the correctly
// rounded reciprocal and q is within an ulp) -- three operations instead of the dozen of a binary64 division, same bits (the CPU
// checker divides; 1.7e8 random (dist, sigma) pairs compared in the making).
struct GaussR { double c, rc; };
__device__ inline GaussR gauss_r_setup(float sigma) {
  GaussR g;
  g.c = 2.0 * (double)sigma * (double)sigma;
  g.rc = 1.0 / g.c;
  return g;
}
__device__ inline float gauss_r(const GaussR& g, float dist, const double (*T)[2]) {
  const double a = -(double)(dist * dist);
  const double q = a * g.rc;
  const double r = fma(-g.c, q, a);
  return (float)sf_exp64_t(fma(r, g.rc, q), T);
}
__device__ inline float gauss_d2(float sigma, int x, int y, const double (*T)[2]) {  // filter.cu:200-203
  return (float)sf_exp64_t((double)(-((float)(x * x + y * y) / (2.0f * sigma * sigma))), T);
}
// the 2^(j/32) table of exp64.h into LDS (all 256 lanes call it; followed by the caller's barrier)
__device__ inline void load_exp_table(double (*T)[2]) {
  static const double table[32][2] = SF_EXP64_TABLE;
  if (threadIdx.x < 64) T[threadIdx.x >> 1][threadIdx.x & 1] = table[threadIdx.x >> 1][threadIdx.x & 1];
}

__global__ __launch_bounds__(256) void k_f2d_prepare(const uint16_t* __restrict__ depth16, float* __restrict__ depth, int dn,
                                                     const uint8_t* __restrict__ rgb, float* __restrict__ intensity, int cn) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < dn) depth[i] = depth16[i] == 0 ? F2D_MINF : (float)depth16[i] * 0.001f;
  if (i < cn) {
    const float inv = 1.0f / 255.0f;
    intensity[i] = (0.299f * (float)rgb[3 * i] + 0.587f * (float)rgb[3 * i + 1] + 0.114f * (float)rgb[3 * i + 2]) * inv;
  }
}

// 16 x 16 pixel tiles, one lane per pixel; the spatial weights of the launch in LDS
__global__ __launch_bounds__(256) void k_f2d_bilateral(float* __restrict__ out, const float* __restrict__ in, float sigma_d, float sigma_r, int w, int h,
                                                       int radius) {
  extern __shared__ float s_gd[];  // (2r+1)^2
  __shared__ double s_exp[32][2];
  load_exp_table(s_exp);
  __syncthreads();
  const int side = 2 * radius + 1;
  for (int t = threadIdx.x; t < side * side; t += 256) s_gd[t] = gauss_d2(sigma_d, t / side - radius, t % side - radius, s_exp);  // [dx + r][dy + r]
  __syncthreads();
  const GaussR gr = gauss_r_setup(sigma_r);
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= w || y >= h) return;
  float result = F2D_MINF;
  float sum = 0.0f, sum_weight = 0.0f;
  const float center = in[(size_t)y * w + x];
  if (center != F2D_MINF) {
    for (int m = x - radius; m <= x + radius; m++)
      for (int n = y - radius; n <= y + radius; n++) {
        if (!(m >= 0 && n >= 0 && m < w && n < h)) continue;
        const float cur = in[(size_t)n * w + m];
        if (cur == F2D_MINF) continue;
        const float weight = s_gd[(m - x + radius) * side + (n - y + radius)] * gauss_r(gr, cur - center, s_exp);
        sum_weight += weight;
        sum += weight * cur;
      }
    if (sum_weight > 0.0f) result = sum / sum_weight;
  }
  out[(size_t)y * w + x] = result;
}

// Validity-aware bilinear sample (what filter.cu:514-541 computes): a horizontal blend per source row over the taps that exist and are
// valid, renormalised by the weight that took part, then the same vertically over the rows that produced something.  The order of
// the floating-point operations (left tap before right tap, upper row before lower row, division after accumulation) is part of the
// result and is kept; the checker (oracle/filter2d_oracle.c) pins it bit for bit.
struct RowBlend { float sum, weight; };
__device__ inline RowBlend blend_row(const float* __restrict__ img, unsigned iw, unsigned ih, int x0, int yr, float ax) {
  RowBlend r = {0.0f, 0.0f};
  if ((unsigned)yr >= ih) return r;
  const float* row = img + (size_t)yr * iw;
  const float wl = 1.0f - ax;
#pragma unroll
  for (int tap = 0; tap < 2; tap++) {
    const int xt = x0 + tap;
    if ((unsigned)xt >= iw) continue;
    const float v = row[xt];
    if (v == F2D_MINF) continue;
    const float wt = tap ? ax : wl;
    r.sum += wt * v;
    r.weight += wt;
  }
  return r;
}
__device__ inline float bilinear(float x, float y, const float* __restrict__ in, unsigned iw, unsigned ih) {
  const int x0 = (int)floorf(x), y0 = (int)floorf(y);
  const float ax = x - (float)x0, ay = y - (float)y0;
  const RowBlend up = blend_row(in, iw, ih, x0, y0, ax), lo = blend_row(in, iw, ih, x0, y0 + 1, ax);
  float acc = 0.0f, wsum = 0.0f;
  if (up.weight > 0.0f) { acc += (1.0f - ay) * (up.sum / up.weight); wsum += (1.0f - ay); }
  if (lo.weight > 0.0f) { acc += ay * (lo.sum / lo.weight); wsum += ay; }
  return wsum > 0.0f ? acc / wsum : F2D_MINF;
}

// Output pixel (x, y) of a resample looks at source position (x * (iw - 1) / (ow - 1), y * (ih - 1) / (oh - 1)); pixels whose rounded
// source position falls outside keep what the output held (filter.cu:543-573, 647-665).  One lane per output pixel, 16 x 16 tiles.
struct ResampleAt { float fx, fy; unsigned nx, ny; bool inside; };
__device__ inline ResampleAt resample_at(int x, int y, int ow, int oh, int iw, int ih) {
  ResampleAt r;
  r.fx = (float)x * ((float)(iw - 1) / (float)(ow - 1));
  r.fy = (float)y * ((float)(ih - 1) / (float)(oh - 1));
  r.nx = (unsigned)(r.fx + 0.5f);
  r.ny = (unsigned)(r.fy + 0.5f);
  r.inside = r.nx < (unsigned)iw && r.ny < (unsigned)ih;
  return r;
}

__global__ __launch_bounds__(256) void k_f2d_resample_float(float* __restrict__ out, int ow, int oh, const float* __restrict__ in, int iw, int ih) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= ow || y >= oh) return;
  const ResampleAt at = resample_at(x, y, ow, oh, iw, ih);
  if (at.inside) out[(size_t)y * ow + x] = bilinear(at.fx, at.fy, in, (unsigned)iw, (unsigned)ih);
}

__global__ __launch_bounds__(256) void k_f2d_resample_uchar(uint8_t* __restrict__ out, int ow, int oh, const uint8_t* __restrict__ in, int iw, int ih) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= ow || y >= oh) return;
  const ResampleAt at = resample_at(x, y, ow, oh, iw, ih);
  if (at.inside) out[(size_t)y * ow + x] = in[(size_t)at.ny * iw + at.nx];
}

// the per-pixel histogram in LDS: vote[bin][lane]; + the spatial table behind it
__global__ __launch_bounds__(256) void k_f2d_vote(uint8_t* __restrict__ out, const uint8_t* __restrict__ in, const float* __restrict__ depth,
                                                  const float* __restrict__ intensity, const uint8_t* __restrict__ instance_to_idx,
                                                  const uint8_t* __restrict__ idx_to_instance, int radius, int w, int h, float sigma_d, float sigma_r,
                                                  float intensity_scale) {
  extern __shared__ float s_mem[];
  float* vote = s_mem;                       // F2D_LABELS x 256
  float* s_gd = s_mem + F2D_LABELS * 256;    // (2r+1)^2: [i + r][j + r]  (i = dy, j = dx)
  __shared__ uint8_t s_to_idx[256];
  __shared__ double s_exp[32][2];
  load_exp_table(s_exp);
  __syncthreads();
  const GaussR gr = gauss_r_setup(sigma_r);
  const int side = 2 * radius + 1;
  for (int t = threadIdx.x; t < side * side; t += 256) s_gd[t] = gauss_d2(sigma_d, t % side - radius, t / side - radius, s_exp);
  s_to_idx[threadIdx.x] = instance_to_idx[threadIdx.x];
#pragma unroll 4
  for (int b = 0; b < F2D_LABELS; b++) vote[b * 256 + threadIdx.x] = 0.0f;
  __syncthreads();
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= w || y >= h) return;
  const float dc = depth[(size_t)y * w + x], ic = intensity[(size_t)y * w + x];
  for (int i = -radius; i <= radius; i++) {
    if (y + i < 0 || y + i >= h) continue;
    for (int j = -radius; j <= radius; j++) {
      if (x + j < 0 || x + j >= w) continue;
      const size_t q = (size_t)(y + i) * w + (x + j);
      const float d = depth[q], in_ = intensity[q];
      const float io = fabsf(ic - in_) * intensity_scale;
      float doff = 0.0f;
      if (dc != F2D_MINF && d != F2D_MINF) doff = fabsf(dc - d);
      const float weight = s_gd[(i + radius) * side + (j + radius)] * gauss_r(gr, doff, s_exp) * gauss_r(gr, io, s_exp);
      const uint8_t idx = s_to_idx[in[q]];
      if (idx < F2D_LABELS) vote[idx * 256 + threadIdx.x] += weight;
    }
  }
  float best = 0.0f;
  uint8_t best_val = 0;
  for (int b = 0; b < F2D_LABELS; b++) {
    const float v = vote[b * 256 + threadIdx.x];
    if (v > best) { best = v; best_val = idx_to_instance[b]; }
  }
  out[(size_t)y * w + x] = best_val;
}

__global__ __launch_bounds__(256) void k_f2d_to_label(uint16_t* __restrict__ out, const uint8_t* __restrict__ instance, const uint16_t* __restrict__ lut, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = lut[instance[i]];
}

}  // namespace

struct sf_filter2d {
  int device = 0;
  int dw = 0, dh = 0, cw = 0, ch = 0;
  size_t big = 0;
  hipStream_t stream = nullptr;
  uint16_t* d_depth16 = nullptr;
  uint8_t* d_rgb = nullptr;
  float *depth = nullptr, *depth_h = nullptr, *depth_orig = nullptr, *inten = nullptr, *inten_h = nullptr, *inten_orig = nullptr;
  uint8_t *inst = nullptr, *inst_h = nullptr, *to_idx = nullptr, *to_inst = nullptr;
  uint16_t *label = nullptr, *to_label = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

SF_API void sf_filter2d_destroy(sf_filter2d* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  for (void* p : {(void*)f->d_depth16, (void*)f->d_rgb, (void*)f->depth, (void*)f->depth_h, (void*)f->depth_orig, (void*)f->inten, (void*)f->inten_h,
                  (void*)f->inten_orig, (void*)f->inst, (void*)f->inst_h, (void*)f->to_idx, (void*)f->to_inst, (void*)f->label, (void*)f->to_label})
    if (p) (void)hipFree(p);
  if (f->e0) (void)hipEventDestroy(f->e0);
  if (f->e1) (void)hipEventDestroy(f->e1);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
}

SF_API int sf_filter2d_create(int depth_width, int depth_height, int color_width, int color_height, int device, sf_filter2d** out) {
  if (!out || depth_width < 2 || depth_height < 2 || color_width < 2 || color_height < 2) return sf::fail(SF_ERR_INVALID_ARG, "invalid image dimensions");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device: libscanfuse has no CPU fallback, the annotation filter needs an MI355X");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  sf_filter2d* f = new sf_filter2d();
  f->device = device;
  f->dw = depth_width; f->dh = depth_height; f->cw = color_width; f->ch = color_height;
  const size_t dn = (size_t)depth_width * depth_height, cn = (size_t)color_width * color_height;
  f->big = std::max(std::max(dn, cn), (size_t)320 * 240);
#define F2_ALLOC(ptr, bytes)                                                                                     \
  do {                                                                                                           \
    const hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                                    \
    if (e_ != hipSuccess) { sf_filter2d_destroy(f); return sf::fail(SF_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e_)); } \
    (void)hipMemset((ptr), 0, (bytes));                                                                          \
  } while (0)
  F2_ALLOC(f->d_depth16, dn * 2);
  F2_ALLOC(f->d_rgb, cn * 3);
  F2_ALLOC(f->depth, f->big * 4); F2_ALLOC(f->depth_h, f->big * 4); F2_ALLOC(f->depth_orig, dn * 4);
  F2_ALLOC(f->inten, f->big * 4); F2_ALLOC(f->inten_h, f->big * 4); F2_ALLOC(f->inten_orig, cn * 4);
  F2_ALLOC(f->inst, f->big); F2_ALLOC(f->inst_h, f->big);
  F2_ALLOC(f->to_idx, 256); F2_ALLOC(f->to_inst, 256); F2_ALLOC(f->to_label, 512);
  F2_ALLOC(f->label, f->big * 2);
#undef F2_ALLOC
  SF_HIP_CHECK(hipMemset(f->to_idx, 0xFF, 256));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreate(&f->e0));
  SF_HIP_CHECK(hipEventCreate(&f->e1));
  SF_HIP_CHECK(hipFuncSetAttribute((const void*)k_f2d_vote, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  *out = f;
  return SF_OK;
}

// FilterData::init (Filter2dAnnotations.cpp:52-61): instance -> histogram bin, bin -> instance, instance -> label.  256 / 80 / 256
// entries (the reference passes 80-entry vectors for all three and reads out of bounds for instance values >= 80).
SF_API int sf_filter2d_set_tables(sf_filter2d* f, const uint8_t instance_to_idx[256], const uint8_t idx_to_instance[80], const uint16_t instance_to_label[256]) {
  if (!f || !instance_to_idx || !idx_to_instance || !instance_to_label) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(hipMemcpy(f->to_idx, instance_to_idx, 256, hipMemcpyHostToDevice));
  SF_HIP_CHECK(hipMemcpy(f->to_inst, idx_to_instance, 80, hipMemcpyHostToDevice));
  SF_HIP_CHECK(hipMemcpy(f->to_label, instance_to_label, 512, hipMemcpyHostToDevice));
  return SF_OK;
}

namespace {
inline dim3 tiles(int w, int h) { return dim3((unsigned)((w + 15) / 16), (unsigned)((h + 15) / 16)); }
}

// One frame: Filter2dAnnotations.cpp:326-397.  Host buffers: depth dw*dh u16 (mm), rgb cw*ch*3, instance_in cw*ch u8 (the rendered
// annotation); instance_out cw*ch u8, label_out cw*ch u16.  kernel_us (nullable): duration of the frame's kernels.
SF_API int sf_filter2d_frame(sf_filter2d* f, const uint16_t* depth, const uint8_t* rgb, const uint8_t* instance_in, uint8_t* instance_out,
                             uint16_t* label_out, float* kernel_us) {
  if (!f || !depth || !rgb || !instance_in || !instance_out || !label_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  hipStream_t s = f->stream;
  const int dw = f->dw, dh = f->dh, cw = f->cw, ch = f->ch;
  const size_t dn = (size_t)dw * dh, cn = (size_t)cw * ch;
  SF_HIP_CHECK(hipMemcpyAsync(f->d_depth16, depth, dn * 2, hipMemcpyHostToDevice, s));
  SF_HIP_CHECK(hipMemcpyAsync(f->d_rgb, rgb, cn * 3, hipMemcpyHostToDevice, s));
  SF_HIP_CHECK(hipMemcpyAsync(f->inst_h, instance_in, cn, hipMemcpyHostToDevice, s));
  SF_HIP_CHECK(hipEventRecord(f->e0, s));
  float *dep = f->depth, *dep_h = f->depth_h, *inten = f->inten, *inten_h = f->inten_h;
  uint8_t *inst = f->inst, *inst_h = f->inst_h;
  hipLaunchKernelGGL(k_f2d_prepare, dim3((unsigned)((std::max(dn, cn) + 255) / 256)), dim3(256), 0, s, f->d_depth16, dep, (int)dn, f->d_rgb, inten, (int)cn);
  SF_HIP_CHECK(hipMemcpyAsync(f->depth_orig, dep, dn * 4, hipMemcpyDeviceToDevice, s));
  SF_HIP_CHECK(hipMemcpyAsync(f->inten_orig, inten, cn * 4, hipMemcpyDeviceToDevice, s));
  auto bilateral = [&](float* out, const float* in, float sd, float sr, int w, int h) {
    const int radius = (int)std::ceil(2.0 * (double)sd);
    const size_t lds = (size_t)(2 * radius + 1) * (2 * radius + 1) * 4;
    hipLaunchKernelGGL(k_f2d_bilateral, tiles(w, h), dim3(256), lds, s, out, in, sd, sr, w, h, radius);
  };
  bilateral(inten_h, inten, 6.0f, 0.1f, cw, ch);   // :334
  bilateral(dep_h, dep, 2.0f, 0.1f, dw, dh);       // :335
  const int fw[2] = {320, cw}, fh[2] = {240, ch}, radii[2] = {12, 10};
  const float iscale[2] = {10.0f, 4.0f};
  int cur_dw = dw, cur_cw = cw;
  if (fw[0] != cw) hipLaunchKernelGGL(k_f2d_resample_uchar, tiles(fw[0], fh[0]), dim3(256), 0, s, inst, fw[0], fh[0], inst_h, cw, ch);
  else SF_HIP_CHECK(hipMemcpyAsync(inst, inst_h, cn, hipMemcpyDeviceToDevice, s));
  for (int iter = 0; iter < 2; iter++) {
    if (cur_dw != fw[iter]) {
      if (fw[iter] == dw) {
        if (iter + 1 == 2) std::swap(dep, dep_h);
        else SF_HIP_CHECK(hipMemcpyAsync(dep, f->depth_orig, dn * 4, hipMemcpyDeviceToDevice, s));
      } else hipLaunchKernelGGL(k_f2d_resample_float, tiles(fw[iter], fh[iter]), dim3(256), 0, s, dep, fw[iter], fh[iter], dep_h, dw, dh);
      cur_dw = fw[iter];
    }
    if (cur_cw != fw[iter]) {
      if (fw[iter] == cw) {
        if (iter + 1 == 2) std::swap(inten, inten_h);
        else SF_HIP_CHECK(hipMemcpyAsync(inten, f->inten_orig, cn * 4, hipMemcpyDeviceToDevice, s));
      } else hipLaunchKernelGGL(k_f2d_resample_float, tiles(fw[iter], fh[iter]), dim3(256), 0, s, inten, fw[iter], fh[iter], inten_h, cw, ch);
      cur_cw = fw[iter];
    }
    const int r = radii[iter];
    const size_t lds = (size_t)F2D_LABELS * 256 * 4 + (size_t)(2 * r + 1) * (2 * r + 1) * 4;
    hipLaunchKernelGGL(k_f2d_vote, tiles(fw[iter], fh[iter]), dim3(256), lds, s, inst_h, inst, dep, inten, f->to_idx, f->to_inst, r, fw[iter], fh[iter], 5.0f,
                       0.1f, iscale[iter]);
    if (iter + 1 == 2) std::swap(inst_h, inst);
    else hipLaunchKernelGGL(k_f2d_resample_uchar, tiles(fw[iter + 1], fh[iter + 1]), dim3(256), 0, s, inst, fw[iter + 1], fh[iter + 1], inst_h, fw[iter], fh[iter]);
  }
  hipLaunchKernelGGL(k_f2d_to_label, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, s, f->label, inst, f->to_label, (int)cn);
  SF_HIP_CHECK(hipGetLastError());
  SF_HIP_CHECK(hipEventRecord(f->e1, s));
  SF_HIP_CHECK(hipMemcpyAsync(instance_out, inst, cn, hipMemcpyDeviceToHost, s));
  SF_HIP_CHECK(hipMemcpyAsync(label_out, f->label, cn * 2, hipMemcpyDeviceToHost, s));
  SF_HIP_CHECK(hipStreamSynchronize(s));
  if (kernel_us) {
    float ms = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, f->e0, f->e1));
    *kernel_us = ms * 1e3f;
  }
  return SF_OK;
}

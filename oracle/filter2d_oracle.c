/* oracle/filter2d_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the CUDA kernels AnnotationTools/Filter2dAnnotations/filter.cu that Filter2dAnnotations.cpp calls, and of
 * the per-frame sequence in which it calls them:
 *   or_f2d_bilateral          bilateralFilterFloatMapDevice      filter.cu:210-247 (gaussD :200-203, gaussR :190-193)
 *   or_f2d_resample_float     resampleFloatMapDevice             filter.cu:543-560 (bilinearInterpolationFloat :514-541)
 *   or_f2d_resample_uchar     resampleUCharMapDevice             filter.cu:647-665
 *   or_f2d_vote               filterAnnotations_Kernel           filter.cu:1020-1059 (+ the memset of the vote buffer :1069)
 *   or_f2d_to_label           convertInstanceToLabel_Kernel      filter.cu:1082-1091
 *   or_f2d_frame              the frame body of process()        Filter2dAnnotations.cpp:326-397 (convertToFloat :245-256,
 *                             convertToGrayscale :232-243, filter sizes / radii / intensity scales :287-290)
 * PARITY UNPINNED: the reference needs CUDA + mLib + FreeImage and holds no test or fixture for this tool; nvcc's default
 * fused-multiply-add contraction and libdevice's exp are not reproducible off an NVIDIA toolchain.  What is restated: every
 * statement in source order, binary32 / binary64 exactly where C++'s promotion rules put them (gaussR is evaluated in double
 * because of its 2.0 literal, gaussD in float), no contraction, exp() = the fixed IEEE sequence or_exp64 below (within 1 ulp of
 * glibc: tests/test_filter2d.py).  Instance values index 256-entry tables here (the reference allocates 80 entries and reads
 * out of bounds for larger values); table entries >= 80 cast no vote. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_NUM_LABELS_PER_SCENE 80 /* GlobalDefines.h:12 */

/* the same sequence as scannet_amd/csrc/exp64.h: x = k ln2/32 + r, exp(x) = 2^(k >> 5) * T[k & 31] * (1 + p(r)), T as a double-double */
static const double EXP_T[32][2] = {
  {0x1p+0, 0x0p+0},
  {0x1.059b0d3158574p+0, 0x1.d7p-55},
  {0x1.0b5586cf9890fp+0, 0x1.8a8p-54},
  {0x1.11301d0125b51p+0, -0x1.6c8p-54},
  {0x1.172b83c7d517bp+0, -0x1.19p-55},
  {0x1.1d4873168b9aap+0, 0x1.ep-54},
  {0x1.2387a6e756238p+0, 0x1.9bp-54},
  {0x1.29e9df51fdee1p+0, 0x1.61p-55},
  {0x1.306fe0a31b715p+0, 0x1.6fp-55},
  {0x1.371a7373aa9cbp+0, -0x1.638p-54},
  {0x1.3dea64c123422p+0, 0x1.aep-55},
  {0x1.44e086061892dp+0, 0x1.8p-59},
  {0x1.4bfdad5362a27p+0, 0x1.d4p-56},
  {0x1.5342b569d4f82p+0, -0x1.08p-55},
  {0x1.5ab07dd485429p+0, 0x1.63p-54},
  {0x1.6247eb03a5585p+0, -0x1.38p-54},
  {0x1.6a09e667f3bcdp+0, -0x1.bep-54},
  {0x1.71f75e8ec5f74p+0, -0x1.17p-55},
  {0x1.7a11473eb0187p+0, -0x1.42p-55},
  {0x1.82589994cce13p+0, -0x1.d5p-54},
  {0x1.8ace5422aa0dbp+0, 0x1.6e8p-54},
  {0x1.93737b0cdc5e5p+0, -0x1.78p-57},
  {0x1.9c49182a3f09p+0, 0x1.c8p-56},
  {0x1.a5503b23e255dp+0, -0x1.d3p-54},
  {0x1.ae89f995ad3adp+0, 0x1.7ap-54},
  {0x1.b7f76f2fb5e47p+0, -0x1.56p-56},
  {0x1.c199bdd85529cp+0, 0x1.11p-55},
  {0x1.cb720dcef9069p+0, 0x1.5p-56},
  {0x1.d5818dcfba487p+0, 0x1.2fp-55},
  {0x1.dfc97337b9b5fp+0, -0x1.1a8p-54},
  {0x1.ea4afa2a490dap+0, -0x1.eap-54},
  {0x1.f50765b6e454p+0, 0x1.9dp-54},
};
double or_exp64(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return (double)INFINITY;
  if (x < -745.2) return 0.0;
  const double inv = 0x1.71547652b82fep+5, l_hi = 0x1.62e42feep-6, l_lo = 0x1.a39ef358p-38;
  const double kf = floor(x * inv + 0.5);
  const double r = fma(-kf, l_lo, fma(-kf, l_hi, x));
  double q = 1.0 / 720;
  q = fma(q, r, 1.0 / 120); /* explicit fused multiply-add: one rounding, the same on both sides */
  q = fma(q, r, 1.0 / 24);
  q = fma(q, r, 1.0 / 6);
  q = fma(q, r, 0.5);
  const double p = fma(r * r, q, r);
  const int k = (int)kf, j = k & 31, m = k >> 5;
  const double y = fma(EXP_T[j][0], p, EXP_T[j][1]) + EXP_T[j][0];
  return ldexp(y, m);
}

static float gauss_r(float sigma, float dist) { /* filter.cu:190-193: (dist*dist) in float, the rest in double */
  return (float)or_exp64(-(double)(dist * dist) / (2.0 * (double)sigma * (double)sigma));
}
static float gauss_d2(float sigma, int x, int y) { /* filter.cu:200-203: all float; exp on a float argument */
  return (float)or_exp64((double)(-((float)(x * x + y * y) / (2.0f * sigma * sigma))));
}

void or_f2d_bilateral(float* out, const float* in, float sigma_d, float sigma_r, int w, int h) {
  const int radius = (int)ceil(2.0 * (double)sigma_d);
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      out[(size_t)y * w + x] = -INFINITY;
      float sum = 0.0f, sum_weight = 0.0f;
      const float center = in[(size_t)y * w + x];
      if (center == -INFINITY) continue;
      for (int m = x - radius; m <= x + radius; m++)
        for (int n = y - radius; n <= y + radius; n++) {
          if (!(m >= 0 && n >= 0 && m < w && n < h)) continue;
          const float cur = in[(size_t)n * w + m];
          if (cur == -INFINITY) continue;
          const float weight = gauss_d2(sigma_d, m - x, n - y) * gauss_r(sigma_r, cur - center);
          sum_weight += weight;
          sum += weight * cur;
        }
      if (sum_weight > 0.0f) out[(size_t)y * w + x] = sum / sum_weight;
    }
}

static float bilinear(float x, float y, const float* in, unsigned iw, unsigned ih) { /* filter.cu:514-541 */
  const int p00x = (int)floorf(x), p00y = (int)floorf(y);
  const int p01x = p00x, p01y = p00y + 1, p10x = p00x + 1, p10y = p00y, p11x = p00x + 1, p11y = p00y + 1;
  const float alpha = x - (float)p00x, beta = y - (float)p00y;
  float s0 = 0.0f, w0 = 0.0f;
  if ((unsigned)p00x < iw && (unsigned)p00y < ih) { const float v = in[(size_t)p00y * iw + p00x]; if (v != -INFINITY) { s0 += (1.0f - alpha) * v; w0 += (1.0f - alpha); } }
  if ((unsigned)p10x < iw && (unsigned)p10y < ih) { const float v = in[(size_t)p10y * iw + p10x]; if (v != -INFINITY) { s0 += alpha * v; w0 += alpha; } }
  float s1 = 0.0f, w1 = 0.0f;
  if ((unsigned)p01x < iw && (unsigned)p01y < ih) { const float v = in[(size_t)p01y * iw + p01x]; if (v != -INFINITY) { s1 += (1.0f - alpha) * v; w1 += (1.0f - alpha); } }
  if ((unsigned)p11x < iw && (unsigned)p11y < ih) { const float v = in[(size_t)p11y * iw + p11x]; if (v != -INFINITY) { s1 += alpha * v; w1 += alpha; } }
  const float p0 = s0 / w0, p1 = s1 / w1;
  float ss = 0.0f, ww = 0.0f;
  if (w0 > 0.0f) { ss += (1.0f - beta) * p0; ww += (1.0f - beta); }
  if (w1 > 0.0f) { ss += beta * p1; ww += beta; }
  return ww > 0.0f ? ss / ww : -INFINITY;
}

/* pixels the kernel does not write keep what `out` held (the reference reuses its buffers) */
void or_f2d_resample_float(float* out, int ow, int oh, const float* in, int iw, int ih) {
  const float sw = (float)(iw - 1) / (float)(ow - 1), sh = (float)(ih - 1) / (float)(oh - 1);
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) {
      const unsigned xi = (unsigned)((float)x * sw + 0.5f), yi = (unsigned)((float)y * sh + 0.5f);
      if (xi < (unsigned)iw && yi < (unsigned)ih) out[(size_t)y * ow + x] = bilinear((float)x * sw, (float)y * sh, in, (unsigned)iw, (unsigned)ih);
    }
}

void or_f2d_resample_uchar(uint8_t* out, int ow, int oh, const uint8_t* in, int iw, int ih) {
  const float sw = (float)(iw - 1) / (float)(ow - 1), sh = (float)(ih - 1) / (float)(oh - 1);
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) {
      const unsigned xi = (unsigned)((float)x * sw + 0.5f), yi = (unsigned)((float)y * sh + 0.5f);
      if (xi < (unsigned)iw && yi < (unsigned)ih) out[(size_t)y * ow + x] = in[(size_t)yi * iw + xi];
    }
}

void or_f2d_vote(uint8_t* out, const uint8_t* in, const float* depth, const float* intensity, const uint8_t* instance_to_idx /*256*/,
                 const uint8_t* idx_to_instance /*80*/, int radius, int w, int h, float sigma_d, float sigma_r, float intensity_scale) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float vote[MAX_NUM_LABELS_PER_SCENE];
      memset(vote, 0, sizeof(vote));
      const float dc = depth[(size_t)y * w + x], ic = intensity[(size_t)y * w + x];
      for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
          if (!(x + j >= 0 && x + j < w && y + i >= 0 && y + i < h)) continue;
          const float d = depth[(size_t)(y + i) * w + (x + j)], in_ = intensity[(size_t)(y + i) * w + (x + j)];
          const float io = fabsf(ic - in_) * intensity_scale;
          float doff = 0.0f;
          if (dc != -INFINITY && d != -INFINITY) doff = fabsf(dc - d);
          const float weight = gauss_d2(sigma_d, j, i) * gauss_r(sigma_r, doff) * gauss_r(sigma_r, io);
          const uint8_t idx = instance_to_idx[in[(size_t)(y + i) * w + (x + j)]];
          if (idx < MAX_NUM_LABELS_PER_SCENE) vote[idx] += weight;
        }
      float best = 0.0f;
      uint8_t best_val = 0;
      for (int i = 0; i < MAX_NUM_LABELS_PER_SCENE; i++)
        if (vote[i] > best) { best = vote[i]; best_val = idx_to_instance[i]; }
      out[(size_t)y * w + x] = best_val;
    }
}

void or_f2d_to_label(uint16_t* out, const uint8_t* instance, const uint16_t* instance_to_label /*256*/, int w, int h) {
  for (size_t i = 0; i < (size_t)w * h; i++) out[i] = instance_to_label[instance[i]];
}

/* Filter2dAnnotations.cpp:326-397 for one frame.  depth dw x dh u16 (mm), rgb cw x ch x 3, instance cw x ch u8. */
void or_f2d_frame(const uint16_t* depth16, int dw, int dh, const uint8_t* rgb, int cw, int ch, const uint8_t* instance_in, const uint8_t* instance_to_idx,
                  const uint8_t* idx_to_instance, const uint16_t* instance_to_label, uint8_t* instance_out, uint16_t* label_out) {
  const int fw[2] = {320, cw}, fh[2] = {240, ch}, radii[2] = {12, 10};     /* :287-289 */
  const float iscale[2] = {10.0f, 4.0f};                                    /* :290 */
  const size_t cn = (size_t)cw * ch, dn = (size_t)dw * dh;
  size_t big = cn > dn ? cn : dn;
  if (big < (size_t)320 * 240) big = (size_t)320 * 240;  /* the intermediate resolution of the first pass */
  float* depth = (float*)malloc(big * 4);
  float* depth_h = (float*)malloc(big * 4);
  float* inten = (float*)malloc(big * 4);
  float* inten_h = (float*)malloc(big * 4);
  uint8_t* inst = (uint8_t*)calloc(big, 1);
  uint8_t* inst_h = (uint8_t*)calloc(big, 1);
  for (size_t i = 0; i < big; i++) depth[i] = depth_h[i] = inten[i] = inten_h[i] = 0.0f;
  for (size_t i = 0; i < dn; i++) depth[i] = depth16[i] == 0 ? -INFINITY : (float)depth16[i] * 0.001f;                 /* :245-256 */
  const float inv = 1.0f / 255.0f;
  for (size_t i = 0; i < cn; i++) inten[i] = (0.299f * (float)rgb[3 * i] + 0.587f * (float)rgb[3 * i + 1] + 0.114f * (float)rgb[3 * i + 2]) * inv; /* :232-243 */
  or_f2d_bilateral(inten_h, inten, 6.0f, 0.1f, cw, ch);                                                               /* :334 */
  or_f2d_bilateral(depth_h, depth, 2.0f, 0.1f, dw, dh);                                                               /* :335 */
  memcpy(inst_h, instance_in, cn);                                                                                    /* :342 */
  int cur_dw = dw, cur_cw = cw;
  if (fw[0] != cw) or_f2d_resample_uchar(inst, fw[0], fh[0], inst_h, cw, ch);                                          /* :355-356 */
  else memcpy(inst, inst_h, cn);
  float* depth_orig = (float*)malloc(dn * 4);
  for (size_t i = 0; i < dn; i++) depth_orig[i] = depth16[i] == 0 ? -INFINITY : (float)depth16[i] * 0.001f;
  float* inten_orig = (float*)malloc(cn * 4);
  memcpy(inten_orig, inten, cn * 4);
  for (int iter = 0; iter < 2; iter++) {
    if (cur_dw != fw[iter]) {                                                                                         /* :359-372 */
      if (fw[iter] == dw) {
        if (iter + 1 == 2) { float* t = depth; depth = depth_h; depth_h = t; }
        else memcpy(depth, depth_orig, dn * 4);
      } else or_f2d_resample_float(depth, fw[iter], fh[iter], depth_h, dw, dh);
      cur_dw = fw[iter];
    }
    if (cur_cw != fw[iter]) {                                                                                         /* :373-385 */
      if (fw[iter] == cw) {
        if (iter + 1 == 2) { float* t = inten; inten = inten_h; inten_h = t; }
        else memcpy(inten, inten_orig, cn * 4);
      } else or_f2d_resample_float(inten, fw[iter], fh[iter], inten_h, cw, ch);
      cur_cw = fw[iter];
    }
    or_f2d_vote(inst_h, inst, depth, inten, instance_to_idx, idx_to_instance, radii[iter], fw[iter], fh[iter], 5.0f, 0.1f, iscale[iter]); /* :386-388 */
    if (iter + 1 == 2) { uint8_t* t = inst_h; inst_h = inst; inst = t; }                                               /* :390-391 */
    else or_f2d_resample_uchar(inst, fw[iter + 1], fh[iter + 1], inst_h, fw[iter], fh[iter]);                          /* :393 */
  }
  or_f2d_to_label(label_out, inst, instance_to_label, fw[1], fh[1]);                                                   /* :395 */
  memcpy(instance_out, inst, (size_t)fw[1] * fh[1]);
  free(depth); free(depth_h); free(inten); free(inten_h); free(inst); free(inst_h); free(depth_orig); free(inten_orig);
}

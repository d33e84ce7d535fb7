// exp64.h -- exp(x) in binary64, one fixed sequence of IEEE operations (no library call; the fused multiply-adds are explicit): the annotation
// filter's weights are defined through exp() in double (AnnotationTools/Filter2dAnnotations/filter.cu:190-208), and the GPU
// path and its CPU checker must agree bit for bit -- which two different libm / ocml implementations do not promise.
// x = k ln2/32 + r with |r| <= ln2/64 (Cody-Waite, two fma), exp(x) = 2^(k >> 5) * T[k & 31] * (1 + p(r)): T = 2^(j/32) as a
// double-double from the table below, p = Taylor polynomial of degree 6 in Horner form with fma (remainder < 2^-57), the power of two
// applied by ldexp (one rounding, also into the subnormal range).  Measured against glibc on 2e7 arguments in [-800, 0]: within 1 ulp,
// identical after rounding to float.  The kernels keep the table in LDS (sf_exp64_table -> shared memory) and call sf_exp64_t.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__
#else
#define SF_HD
#endif

#define SF_EXP64_TABLE                                     \
  {                                                        \
    {0x1p+0, 0x0p+0}, \
    {0x1.059b0d3158574p+0, 0x1.d7p-55}, \
    {0x1.0b5586cf9890fp+0, 0x1.8a8p-54}, \
    {0x1.11301d0125b51p+0, -0x1.6c8p-54}, \
    {0x1.172b83c7d517bp+0, -0x1.19p-55}, \
    {0x1.1d4873168b9aap+0, 0x1.ep-54}, \
    {0x1.2387a6e756238p+0, 0x1.9bp-54}, \
    {0x1.29e9df51fdee1p+0, 0x1.61p-55}, \
    {0x1.306fe0a31b715p+0, 0x1.6fp-55}, \
    {0x1.371a7373aa9cbp+0, -0x1.638p-54}, \
    {0x1.3dea64c123422p+0, 0x1.aep-55}, \
    {0x1.44e086061892dp+0, 0x1.8p-59}, \
    {0x1.4bfdad5362a27p+0, 0x1.d4p-56}, \
    {0x1.5342b569d4f82p+0, -0x1.08p-55}, \
    {0x1.5ab07dd485429p+0, 0x1.63p-54}, \
    {0x1.6247eb03a5585p+0, -0x1.38p-54}, \
    {0x1.6a09e667f3bcdp+0, -0x1.bep-54}, \
    {0x1.71f75e8ec5f74p+0, -0x1.17p-55}, \
    {0x1.7a11473eb0187p+0, -0x1.42p-55}, \
    {0x1.82589994cce13p+0, -0x1.d5p-54}, \
    {0x1.8ace5422aa0dbp+0, 0x1.6e8p-54}, \
    {0x1.93737b0cdc5e5p+0, -0x1.78p-57}, \
    {0x1.9c49182a3f09p+0, 0x1.c8p-56}, \
    {0x1.a5503b23e255dp+0, -0x1.d3p-54}, \
    {0x1.ae89f995ad3adp+0, 0x1.7ap-54}, \
    {0x1.b7f76f2fb5e47p+0, -0x1.56p-56}, \
    {0x1.c199bdd85529cp+0, 0x1.11p-55}, \
    {0x1.cb720dcef9069p+0, 0x1.5p-56}, \
    {0x1.d5818dcfba487p+0, 0x1.2fp-55}, \
    {0x1.dfc97337b9b5fp+0, -0x1.1a8p-54}, \
    {0x1.ea4afa2a490dap+0, -0x1.eap-54}, \
    {0x1.f50765b6e454p+0, 0x1.9dp-54}, \
  }

SF_HD inline double sf_exp64_t(double x, const double (*T)[2]) {
  if (x != x) return x;
  if (x > 709.782712893384) return (double)INFINITY;
  if (x < -745.2) return 0.0;
  const double inv = 0x1.71547652b82fep+5, l_hi = 0x1.62e42feep-6, l_lo = 0x1.a39ef358p-38;   // 32/ln2, ln2/32 = hi + lo (hi: 32 bits)
  const double kf = floor(x * inv + 0.5);
  const double r = fma(-kf, l_lo, fma(-kf, l_hi, x));
  double q = 1.0 / 720;
  q = fma(q, r, 1.0 / 120);  /* explicit fused multiply-add: one rounding, the same on both sides */
  q = fma(q, r, 1.0 / 24);
  q = fma(q, r, 1.0 / 6);
  q = fma(q, r, 0.5);
  const double p = fma(r * r, q, r);
  const int k = (int)kf, j = k & 31, m = k >> 5;
  const double y = fma(T[j][0], p, T[j][1]) + T[j][0];
  return ldexp(y, m);
}

inline double sf_exp64(double x) {   // host: the table as a constant
  static const double T[32][2] = SF_EXP64_TABLE;
  return sf_exp64_t(x, T);
}

// simplify_math.h -- the arithmetic of the quadric edge collapse shared by the sequential host implementation (simplify.cpp) and the
// parallel GPU one (simplify_gpu.hip): plane quadrics in double, the optimal position (adjugate inverse where the 3x3 system is well
// conditioned, pseudo-inverse around the edge midpoint where it is not: simplify.cpp header), VCG's face quality.  One source, so the
// two pick the same position and the same priority for the same collapse.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SF_SHD __host__ __device__
#else
#define SF_SHD
#endif

namespace sfq {

struct Quadric {
  double a[6], b[3], c;
  SF_SHD void zero() { for (int i = 0; i < 6; i++) a[i] = 0.0; for (int i = 0; i < 3; i++) b[i] = 0.0; c = 0.0; }
  SF_SHD void by_plane(const double n[3], double off) {
    a[0] = n[0] * n[0]; a[1] = n[0] * n[1]; a[2] = n[0] * n[2];
    a[3] = n[1] * n[1]; a[4] = n[1] * n[2]; a[5] = n[2] * n[2];
    b[0] = -2.0 * off * n[0]; b[1] = -2.0 * off * n[1]; b[2] = -2.0 * off * n[2];
    c = off * off;
  }
  SF_SHD void add(const Quadric& q) {
    for (int i = 0; i < 6; i++) a[i] += q.a[i];
    for (int i = 0; i < 3; i++) b[i] += q.b[i];
    c += q.c;
  }
  SF_SHD double apply(const double p[3]) const {
    return p[0] * p[0] * a[0] + 2 * p[0] * p[1] * a[1] + 2 * p[0] * p[2] * a[2] + p[0] * b[0] + p[1] * p[1] * a[3] + 2 * p[1] * p[2] * a[4] +
           p[1] * b[1] + p[2] * p[2] * a[5] + p[2] * b[2] + c;
  }
};

// symmetric 3x3 eigen-decomposition (cyclic Jacobi): A = V diag(w) V^T
SF_SHD inline void eigen_sym3(const double a[6], double w[3], double V[3][3]) {
  double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-18 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) {  // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq;
          A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk;
          A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cs * vkp - sn * vkq;
          V[k][q] = sn * vkp + cs * vkq;
        }
      }
  }
  for (int i = 0; i < 3; i++) w[i] = A[i][i];
}

// minimiser of q closest to `mid`: x = mid + pinv(A) (-b/2 - A mid)
SF_SHD inline void minimise(const Quadric& q, const double mid[3], double x[3]) {
  // Well conditioned (the usual case off the flat areas): plain inverse through the adjugate.  For a positive
  // semi-definite A, lambda_min >= det / trace^2 and lambda_max <= trace, so det > 1e-6 trace^3 guarantees a condition
  // number below 1e6 -- the pseudo-inverse below would use all three eigenvalues and return the same point.
  {
    const double* a = q.a;
    const double c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, tr = a[0] + a[3] + a[5];
    if (det > 1e-6 * tr * tr * tr && tr > 0.0) {
      const double c11 = a[0] * a[5] - a[2] * a[2], c12 = a[1] * a[2] - a[0] * a[4], c22 = a[0] * a[3] - a[1] * a[1];
      const double r0 = -0.5 * q.b[0], r1 = -0.5 * q.b[1], r2 = -0.5 * q.b[2], inv = 1.0 / det;
      x[0] = (c00 * r0 + c01 * r1 + c02 * r2) * inv;
      x[1] = (c01 * r0 + c11 * r1 + c12 * r2) * inv;
      x[2] = (c02 * r0 + c12 * r1 + c22 * r2) * inv;
      return;
    }
  }
  double w[3], V[3][3];
  eigen_sym3(q.a, w, V);
  const double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
  const double Am[3] = {q.a[0] * mid[0] + q.a[1] * mid[1] + q.a[2] * mid[2], q.a[1] * mid[0] + q.a[3] * mid[1] + q.a[4] * mid[2],
                        q.a[2] * mid[0] + q.a[4] * mid[1] + q.a[5] * mid[2]};
  const double r[3] = {-0.5 * q.b[0] - Am[0], -0.5 * q.b[1] - Am[1], -0.5 * q.b[2] - Am[2]};
  x[0] = mid[0]; x[1] = mid[1]; x[2] = mid[2];
  if (!(wmax > 0.0)) return;
  for (int k = 0; k < 3; k++) {
    if (!(fabs(w[k]) > 1e-9 * wmax)) continue;
    const double proj = (V[0][k] * r[0] + V[1][k] * r[1] + V[2][k] * r[2]) / w[k];
    for (int i = 0; i < 3; i++) x[i] += V[i][k] * proj;
  }
}

// vcg::Quality(p0, p1, p2) = 2 area / longest edge squared, in float as CMeshO does
SF_SHD inline float quality(const float* p0, const float* p1, const float* p2) {
  const float d10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const float d20[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const float d12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const float x[3] = {d10[1] * d20[2] - d10[2] * d20[1], d10[2] * d20[0] - d10[0] * d20[2], d10[0] * d20[1] - d10[1] * d20[0]};
  const float a = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (a == 0) return 0;
  float b = d10[0] * d10[0] + d10[1] * d10[1] + d10[2] * d10[2];
  if (b == 0) return 0;
  float t = d20[0] * d20[0] + d20[1] * d20[1] + d20[2] * d20[2];
  if (b < t) b = t;
  t = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
  if (b < t) b = t;
  return a / b;
}


}  // namespace sfq

/* tools/check_division.c -- CPU brute-force check of the hand-expanded divisions of k_integrate and k_alloc (fuser.hip, fuser_internal.h).
 *
 *   gcc -O2 -mfma -ffp-contract=off -o /tmp/check_division tools/check_division.c -lm && /tmp/check_division
 *
 * (1) n / m with m an integer in [1, 511]: r = RN(1/m) from a table, q0 = RN(n*r), q1 = fma(fma(-m, q0, n), r, q0).
 *     Claim: q1 == RN(n/m) (Markstein: one correction suffices when the reciprocal is correctly rounded).
 *     Checked on ~1.4e9 numerators per run: random bit patterns, near-multiples of m (near-halfway quotients), scaled ints.
 * (2) 1 / b: seed r0 within 2 ulp of 1/b (v_rcp_f32 is specified to 1 ulp), two Newton steps
 *     r1 = fma(fma(-b, r0, 1), r0, r0), r2 = fma(fma(-b, r1, 1), r1, r1).
 *     Claim: r2 is bit-identical to the result of the compiler's full fdiv expansion (one more correction step) for
 *     every mantissa and every such seed -- so dropping the third step and the div_scale/div_fixup range handling does
 *     not change a bit for normal-range b.  Checked exhaustively over all 2^23 mantissas x 3 exponents x 5 seed errors.
 * (3) a / b for general normal-range operands (k_alloc: world -> block through / voxel, the DDA's tMax / tDelta through / direction):
 *     y = RN(1/b), q0 = RN(a*y), q1 = fma(fma(-b, q0, a), y, q0), q = fma(fma(-b, q1, a), y, q1) -- the quotient part of the compiler's
 *     own expansion without div_scale / div_fixup.  Claim: q == RN(a/b).  Checked on 2^30 operand pairs: random bit patterns in the
 *     ranges k_alloc sees, dividends within 3 ulp of exact multiples and of half-way multiples of the divisor.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(void) {
  long bad1 = 0, tot = 0;
  for (int m = 1; m < 512; m++) {
    const float fm = (float)m, r = 1.0f / fm;
    for (long it = 0; it < 3000000; it++) {
      float n;
      const uint64_t x = rnd();
      if ((it & 3) == 0) { n = asf((uint32_t)x); if (!isfinite(n) || fabsf(n) < 0x1p-100f || fabsf(n) > 1e30f) continue; }
      else if ((it & 3) == 1) { const float qf = asf(0x3f800000u | ((uint32_t)(x >> 40) & 0x7fffff)); n = qf * fm; n = asf(asu(n) + (int)((x >> 8) & 7) - 3); }
      else { n = (float)((int64_t)(x >> 20) - (1ll << 43)) * asf(0x2f800000u + (uint32_t)((x & 15) << 23)); if (n == 0) continue; }
      const float ref = n / fm, q0 = n * r, q1 = fmaf(fmaf(-fm, q0, n), r, q0);
      tot++;
      if (asu(q1) != asu(ref)) bad1++;
    }
  }
  printf("(1) table reciprocal + one correction: %ld cases, %ld mismatches vs n/m\n", tot, bad1);
  long differ = 0, wrong_exact_seed = 0;
  for (uint32_t mant = 0; mant < (1u << 23); mant++)
    for (int ex = 0; ex < 3; ex++) {
      const float b = asf((ex == 0 ? 0x3f800000u : ex == 1 ? 0x3c000000u : 0x43000000u) | mant), ref = 1.0f / b;
      for (int d = -2; d <= 2; d++) {
        const float r0 = asf(asu(ref) + d);
        const float r1 = fmaf(fmaf(-b, r0, 1.0f), r0, r0);
        const float r2 = fmaf(fmaf(-b, r1, 1.0f), r1, r1);
        const float r3 = fmaf(fmaf(-b, r2, 1.0f), r1, r2); /* the compiler's third step */
        if (asu(r2) != asu(r3)) differ++;
        if (d == 0 && asu(r2) != asu(ref)) wrong_exact_seed++;
      }
    }
  printf("(2) two Newton steps vs the compiler's three: %ld differences; exact seed -> wrong result %ld times\n", differ, wrong_exact_seed);
  long bad3 = 0, tot3 = 0;
  for (long it = 0; it < (1l << 28); it++) {
    const uint64_t x = rnd(), z = rnd();
    const float b = asf(((uint32_t)x & 0x807fffffu) | ((103u + (uint32_t)(x >> 58) % 41u) << 23));
    const float a0 = asf(((uint32_t)z & 0x807fffffu) | ((97u + (uint32_t)(z >> 58) % 43u) << 23));
    const float qf = asf(0x3f800000u | ((uint32_t)(z >> 24) & 0x7fffffu));
    const float a1 = asf(asu(qf * b) + ((uint32_t)(x >> 50) & 7u) - 3u);
    const float qh = asf(asu(qf) & 0xfffffffeu);
    const float a2 = asf(asu(fmaf(qh, b, 0x1p-24f * b)) + ((uint32_t)(z >> 50) & 7u) - 3u);
    const float a3 = asf(asu(a0) ^ 0x00400000u);
    const float y = 1.0f / b;
    const float as[4] = {a0, a1, a2, a3};
    for (int k = 0; k < 4; k++) {
      const float a = as[k], q0 = a * y, q1 = fmaf(fmaf(-b, q0, a), y, q0), q = fmaf(fmaf(-b, q1, a), y, q1);
      tot3++;
      if (asu(q) != asu(a / b)) bad3++;
    }
  }
  printf("(3) reciprocal + two residual corrections: %ld cases, %ld mismatches vs a/b\n", tot3, bad3);
  return (bad1 || differ || wrong_exact_seed || bad3) ? 1 : 0;
}

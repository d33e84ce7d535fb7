// jpeg_idct.h -- the sample reconstruction of the baseline JPEG decoder as functions both the host decoder (jpeg.cpp) and the GPU
// path of the frame pipeline (jpeg_gpu.hip) are built from, so that the two produce the same bytes -- and the bytes of the
// reference: RGBDFrame::decompressColorAlloc_stb (SensReader/c++/src/sensorData.h:609-616) decodes with stb_image v2.08, whose
// reconstruction is pure integer arithmetic, so identity (not a tolerance) is the bar (tests/test_sens.py).  Restated here:
//   * dequantisation to 16 bits: (int16)(coefficient * step)                                        stb_image.h:1726,1751,1764
//   * 8 x 8 inverse DCT, the "slow integer" factorisation with 12-bit constants: columns keep 2 extra bits ((x + 512) >> 10),
//     rows remove 17 with the level shift folded into the rounding term ((x + 65536 + (128 << 17)) >> 17), clamp   stb_image.h:1928-2026
//   * chroma upsampling chosen per component from (hs, vs) = (hmax / h, vmax / v): (1,1) copy, (1,2) / (2,1) / (2,2) triangle filter
//     (3/4 nearer + 1/4 farther sample), anything else nearest; rows beyond the last VALID chroma row replicate it   stb_image.h:2871-2933,3052-3063,3339-3383
//     (2,1) keeps stb's own last-column rule: pixel 2 (w - 1) = (3 c[w - 2] + c[w - 1] + 2) >> 2, pixel 2 w - 1 = c[w - 1]   stb_image.h:2899-2900
//   * YCbCr -> RGB in 12.20 fixed point, the blue-difference term of green truncated to its upper 16 bits                stb_image.h:3093-3119
// All arithmetic is 32-bit two's complement (wrapping), shifts of negative values are arithmetic.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SF_JHD __host__ __device__
#else
#define SF_JHD
#endif

SF_JHD inline uint8_t sf_jpeg_clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// (x >> shift) clamped to 0..255.  On the device the shifted value passes through an empty asm statement: hipcc (ROCm 7.2) otherwise fuses
// "arithmetic shift + clamp" of two neighbouring samples into gfx950's v_ashr_pk_u8_i32 and then ORs further bytes into the upper half of
// its result as if it were zero -- on the MI355X that half holds other bits (measured: samples 2 and 3 of a packed group came back as
// value | 0x80 / 0xff; tools/gpu/debug_jpeg.py).  The barrier keeps the two operations apart; the result is the plain integer one.
SF_JHD inline uint8_t sf_jpeg_shift_clamp8(int32_t x, int shift) {
  int32_t t = x >> shift;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(t));
#endif
  return sf_jpeg_clamp8(t);
}

// One 8-point pass.  in: the eight inputs; out k = (value_k + bias) >> shift; `emit(k, value_k + bias, shift)` stores it.
// 32-bit wrapping arithmetic: unsigned for the sums and products, the final shift on the signed reinterpretation.
#if defined(__HIP_DEVICE_COMPILE__)
extern "C" __device__ __attribute__((const)) int __ockl_mul24_i32(int, int);
#endif
// a * c in 32-bit wrapping arithmetic where `a` is known to fit 24 signed bits (below: dequantised 16-bit coefficients, sums of two of them,
// and -- in the row pass -- column results, which are 32-bit values shifted right by 10, and sums of up to four of those: < 2^23 in magnitude)
// and `c` a 14-bit constant: on the device that is the full-rate v_mul_i32_i24 instead of the quarter-rate 32-bit multiply; same low 32 bits.
SF_JHD inline uint32_t sf_mul24c(uint32_t a, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__ockl_mul24_i32((int)a, c);   // what HIP's __mul24 is (this header is also included where <hip/hip_runtime.h> is not)
#else
  return a * (uint32_t)c;
#endif
}

template <class Emit>
SF_JHD inline void sf_idct8_int(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, uint32_t bias, int shift, Emit emit) {
  typedef uint32_t u;
  // even half: rotation of (s2, s6) by 3 pi / 8, butterfly of (s0, s4) scaled by 2^12
  const u rot = sf_mul24c((u)s2 + (u)s6, 2217);
  const u ev2 = rot + sf_mul24c((u)s6, -7567);
  const u ev3 = rot + sf_mul24c((u)s2, 3135);
  const u ev0 = ((u)s0 + (u)s4) * 4096u, ev1 = ((u)s0 - (u)s4) * 4096u;
  const u a0 = ev0 + ev3 + bias, a3 = ev0 - ev3 + bias, a1 = ev1 + ev2 + bias, a2 = ev1 - ev2 + bias;
  // odd half
  const u q3 = (u)s7 + (u)s3, q4 = (u)s5 + (u)s1, q1 = (u)s7 + (u)s1, q2 = (u)s5 + (u)s3;
  const u q5 = sf_mul24c(q3 + q4, 4816);
  const u r1 = q5 + sf_mul24c(q1, -3685), r2 = q5 + sf_mul24c(q2, -10497), r3 = sf_mul24c(q3, -8034), r4 = sf_mul24c(q4, -1597);
  const u od3 = sf_mul24c((u)s1, 6149) + r1 + r4;
  const u od2 = sf_mul24c((u)s3, 12586) + r2 + r3;
  const u od1 = sf_mul24c((u)s5, 8410) + r2 + r4;
  const u od0 = sf_mul24c((u)s7, 1223) + r1 + r3;
  // emit receives the UNSHIFTED sum and the shift (the row pass clamps behind the shift: sf_jpeg_shift_clamp8)
  emit(0, (int32_t)(a0 + od3), shift); emit(7, (int32_t)(a0 - od3), shift);
  emit(1, (int32_t)(a1 + od2), shift); emit(6, (int32_t)(a1 - od2), shift);
  emit(2, (int32_t)(a2 + od1), shift); emit(5, (int32_t)(a2 - od1), shift);
  emit(3, (int32_t)(a3 + od0), shift); emit(4, (int32_t)(a3 - od0), shift);
}

// dequantised 16-bit coefficients (natural order, blk[8 v + u]) -> 64 levels in place
SF_JHD inline void sf_idct_block_int(int* blk) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int c = 0; c < 8; c++) {
    int* col = blk + c;
    sf_idct8_int(col[0], col[8], col[16], col[24], col[32], col[40], col[48], col[56], 512u, 10, [&](int k, int32_t v, int sh) { col[8 * k] = v >> sh; });
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 8; r++) {
    int* row = blk + 8 * r;
    sf_idct8_int(row[0], row[1], row[2], row[3], row[4], row[5], row[6], row[7], 65536u + (128u << 17), 17, [&](int k, int32_t v, int sh) { row[k] = (int)sf_jpeg_shift_clamp8(v, sh); });
  }
}

// the 16-bit dequantised coefficient stb keeps: (short)(coefficient * step)
SF_JHD inline int sf_jpeg_dequant16(int coef, int step) { return (int)(int16_t)(uint16_t)((uint32_t)coef * (uint32_t)step); }

SF_JHD inline void sf_jpeg_ycc_to_rgb(int Y, int cb, int cr, uint8_t* o) {
  cb -= 128;
  cr -= 128;
  const uint32_t yf = ((uint32_t)Y << 20) + (1u << 19);
  const int32_t r = (int32_t)(yf + (uint32_t)cr * 1470208u);
  const int32_t g = (int32_t)(yf + (uint32_t)cr * (uint32_t)-748800 + (((uint32_t)cb * (uint32_t)-360960) & 0xffff0000u));
  const int32_t b = (int32_t)(yf + (uint32_t)cb * 1858048u);
  o[0] = sf_jpeg_shift_clamp8(r, 20);
  o[1] = sf_jpeg_shift_clamp8(g, 20);
  o[2] = sf_jpeg_shift_clamp8(b, 20);
}

// ---- entropy-decoded frame handed to the GPU (jpeg_gpu.hip): this header, a block table, the non-zero coefficients ----------------
// Component c holds (bw[c] / 8) x (bh[c] / 8) blocks in raster order; block_off[c] is the index of its first block in the table.
// table[b] = (first entry << 7) | number of entries of block b; an entry = (natural-order position v * 8 + u) << 16 | the quantised
// coefficient as 16 bits.  A 4:2:0 frame at quality 90 holds ~10 non-zero coefficients per block: a third of the bytes of the pixels.
struct SfJpegLayout {
  uint16_t width, height;
  uint8_t ncomp, hmax, vmax, reserved0;
  uint8_t h[3], v[3];
  uint16_t bw[3], bh[3];     // plane sizes in samples, padded to whole MCUs
  uint16_t reserved1;
  uint32_t block_off[3];
  uint32_t nblocks;          // of all components; the table (nblocks x uint32) starts 512 bytes into the payload
  uint32_t nentries;         // the entries (uint32 each) follow the table
  uint16_t q[3][64];         // quantiser steps per component, natural order
  uint8_t pad[80];
};
static_assert(sizeof(SfJpegLayout) == 512, "SfJpegLayout is the 512-byte header of a coefficient payload");
inline size_t sf_jpeg_payload_bytes(const SfJpegLayout& L) { return sizeof(SfJpegLayout) + 4 * ((size_t)L.nblocks + L.nentries); }
inline size_t sf_jpeg_plane_bytes(const SfJpegLayout& L) { size_t n = 0; for (int c = 0; c < L.ncomp; c++) n += (size_t)L.bw[c] * L.bh[c]; return n; }

// full-resolution sample (x, y) of a component stored at 1/sx x 1/sy, cw x ch valid samples -- the per-pixel form of jpeg.cpp's row loops
// (same integers); the dispatch on (sx, sy) and the last-column rule of (2, 1) are stb's (header comment)
SF_JHD inline int sf_jpeg_upsample(const uint8_t* plane, int bw, int cw, int ch, int sx, int sy, int x, int y) {
  if (sx == 1 && sy == 1) return plane[(size_t)y * bw + x];
  const bool tri_v = sy == 2 && sx <= 2, tri_h = sx == 2 && sy <= 2;
  if (!tri_v && !tri_h) {   // nearest
    const int cy = y / sy < ch ? y / sy : ch - 1;
    return plane[(size_t)cy * bw + x / sx];
  }
  int wn = 4, wf = 0, yn = y < ch ? y : ch - 1, yf = yn;   // vertical weights (of 4) and rows of the nearer / farther sample
  if (tri_v) {
    const int cy = y >> 1;
    yn = cy;
    yf = (y & 1) ? (cy + 1 < ch ? cy + 1 : cy) : (cy > 0 ? cy - 1 : cy);
    wn = 3; wf = 1;
  }
  const uint8_t* rn = plane + (size_t)yn * bw;
  const uint8_t* rf = plane + (size_t)yf * bw;
  if (!tri_h) return (wn * rn[x] + wf * rf[x] + 2) >> 2;
  const int cx = x >> 1;
  if (!tri_v) {   // (2, 1)
    if (cw == 1) return rn[0];
    if (x == 0 || x == 2 * cw - 1) return rn[cx];
    if (x == 2 * cw - 2) return (3 * rn[cw - 2] + rn[cw - 1] + 2) >> 2;
    const int cn = (x & 1) ? cx + 1 : cx - 1;
    return (3 * rn[cx] + rn[cn] + 2) >> 2;
  }
  const int cn = (x & 1) ? (cx + 1 < cw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : cx);
  const int a = 3 * rn[cx] + rf[cx], b = 3 * rn[cn] + rf[cn];
  return (3 * a + b + 8) >> 4;
}

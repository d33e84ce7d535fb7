// jpeg_idct.h -- the sample reconstruction of the baseline JPEG decoder as functions both the host decoder (jpeg.cpp) and the GPU
// path of the frame pipeline (jpeg_gpu.hip) are built from, so that the two produce the same bytes: dequantisation with the
// AAN scale factors folded in, a separable 8-point inverse DCT after Arai / Agui / Nakajima (5 multiplications per 1-D pass, plain
// IEEE binary32, no contraction), rounding to the nearest level, triangle-filter 2x chroma upsampling and BT.601 full-range
// YCbCr -> RGB in 16.16 fixed point.  T.81 does not define bit-exact decoding; against the reference's stb decoder (integer IDCT)
// the result stays within the tolerance tests/test_sens.py states.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SF_JHD __host__ __device__
#else
#define SF_JHD
#endif

// s[0] = 1, s[k] = sqrt(2) cos(k pi / 16): coefficient (v, u) is multiplied by s[u] s[v] / 8 before the butterflies
SF_JHD inline float sf_jpeg_aan(int k) {
  const float s[8] = {1.0f, 1.387039845f, 1.306562965f, 1.175875602f, 1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
  return s[k];
}
// multiplier of the quantised coefficient at natural-order position z of a table entry q
SF_JHD inline float sf_jpeg_dequant(uint16_t q, int z) { return (float)q * (sf_jpeg_aan(z & 7) * sf_jpeg_aan(z >> 3) * 0.125f); }

// one 8-point pass in place over v[0], v[s], ..., v[7 s]
SF_JHD inline void sf_idct8(float* v, int s) {
  const float t0 = v[0], t1 = v[2 * s], t2 = v[4 * s], t3 = v[6 * s];
  const float a10 = t0 + t2, a11 = t0 - t2, a13 = t1 + t3, a12 = (t1 - t3) * 1.414213562f - a13;
  const float e0 = a10 + a13, e3 = a10 - a13, e1 = a11 + a12, e2 = a11 - a12;
  const float o4 = v[s], o5 = v[3 * s], o6 = v[5 * s], o7 = v[7 * s];
  const float z13 = o6 + o5, z10 = o6 - o5, z11 = o4 + o7, z12 = o4 - o7;
  const float b7 = z11 + z13, b11 = (z11 - z13) * 1.414213562f;
  const float z5 = (z10 + z12) * 1.847759065f;
  const float b10 = z5 - z12 * 1.082392200f;
  const float b12 = z5 - z10 * 2.613125930f;
  const float b6 = b12 - b7, b5 = b11 - b6, b4 = b10 - b5;
  v[0] = e0 + b7; v[7 * s] = e0 - b7;
  v[s] = e1 + b6; v[6 * s] = e1 - b6;
  v[2 * s] = e2 + b5; v[5 * s] = e2 - b5;
  v[3 * s] = e3 + b4; v[4 * s] = e3 - b4;
}

SF_JHD inline uint8_t sf_jpeg_level(float v) {   // sample = nearest integer to v + 128, clamped
  const float r = rintf(v + 128.0f);
  return (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
}

SF_JHD inline uint8_t sf_jpeg_clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// BT.601 full range, 16.16 fixed point
SF_JHD inline void sf_jpeg_ycc_to_rgb(int Y, int cb, int cr, uint8_t* o) {
  cb -= 128;
  cr -= 128;
  o[0] = sf_jpeg_clamp8((Y * 65536 + 91881 * cr + 32768) >> 16);
  o[1] = sf_jpeg_clamp8((Y * 65536 - 22554 * cb - 46802 * cr + 32768) >> 16);
  o[2] = sf_jpeg_clamp8((Y * 65536 + 116130 * cb + 32768) >> 16);
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

// full-resolution sample (x, y) of a component stored at 1/sx x 1/sy: triangle filter for a factor of 2 (3/4 nearer + 1/4 farther sample,
// vertically then horizontally, borders replicated), nearest otherwise -- the per-pixel form of jpeg.cpp's row loops (same integers)
SF_JHD inline int sf_jpeg_upsample(const uint8_t* plane, int bw, int cw, int ch, int sx, int sy, int x, int y) {
  if (sx == 1 && sy == 1) return plane[(size_t)y * bw + x];
  int y0, y1, wy0, wy1;
  if (sy == 2) { const int cy = y >> 1; y0 = cy; y1 = (y & 1) ? (cy + 1 < ch ? cy + 1 : cy) : (cy > 0 ? cy - 1 : cy); wy0 = 3; wy1 = 1; }
  else { y0 = y1 = (y / sy < ch ? y / sy : ch - 1); wy0 = 4; wy1 = 0; }
  const uint8_t* r0 = plane + (size_t)y0 * bw;
  const uint8_t* r1 = plane + (size_t)y1 * bw;
  if (sx == 2) {
    const int cx = x >> 1;
    const int cn = (x & 1) ? (cx + 1 < cw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : cx);
    const int a = wy0 * r0[cx] + wy1 * r1[cx], b = wy0 * r0[cn] + wy1 * r1[cn];   // each scaled by 4
    return (3 * a + b + 8) >> 4;
  }
  const int cx = x / sx < cw ? x / sx : cw - 1;
  return (wy0 * r0[cx] + wy1 * r1[cx] + 2) >> 2;
}

// jpeg_enc.cpp -- baseline JPEG encoder (ITU-T T.81 sequential DCT, Huffman, 8 bit, YCbCr 4:2:0 or 4:4:4) for the colour
// frames the `calibrate` stage writes back into a .sens.
//
// Replaces RGBDFrame::compressColor with TYPE_JPEG (SensReader/c++/src/sensorData.h:565-596), which the reference delegates to
// the Occipital uplink encoder behind _USE_UPLINK_COMPRESSION (Windows builds only; on Linux it throws "need UPLINK_COMPRESSION",
// :590) -- called by replaceColor (:505-508) from Calibration::calibrateScan (Calibrate/src/calibration.h:268).  The byte
// stream of that encoder is not reproducible (closed library, unknown tables) and need not be: any baseline JPEG of the
// same pixels is a valid TYPE_JPEG blob for every reader of the format (stb_image in the reference, jpeg.cpp here).
// Quantisation tables: T.81 Annex K.1 scaled the IJG way (quality 1..100, default 90); Huffman tables: Annex K.3; the
// forward DCT is a separable double-precision 8x8 DCT-II (an encoder runs once per frame on a decode-bound stage).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// Annex K.1 (natural order)
const uint8_t kQLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                            18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// Annex K.3: BITS (codes per length 1..16) and HUFFVAL
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffEnc {
  uint16_t code[256];
  uint8_t len[256];
  HuffEnc(const uint8_t* bits, const uint8_t* vals) {
    std::memset(code, 0, sizeof(code));
    std::memset(len, 0, sizeof(len));
    uint32_t c = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {  // Annex C: canonical codes in order of increasing length
      for (int i = 0; i < bits[l - 1]; i++) { code[vals[k]] = (uint16_t)c++; len[vals[k]] = (uint8_t)l; k++; }
      c <<= 1;
    }
  }
};

struct Out {
  std::vector<uint8_t>& v;
  uint32_t acc = 0;
  int n = 0;
  explicit Out(std::vector<uint8_t>& o) : v(o) {}
  void byte(uint8_t b) { v.push_back(b); }
  void word(uint16_t w) { v.push_back((uint8_t)(w >> 8)); v.push_back((uint8_t)w); }
  void bits(uint32_t code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1u));
    n += len;
    while (n >= 8) {
      const uint8_t b = (uint8_t)(acc >> (n - 8));
      v.push_back(b);
      if (b == 0xFF) v.push_back(0);  // byte stuffing (B.1.1.5)
      n -= 8;
    }
  }
  void flush() {
    if (n > 0) bits(0x7F, 8 - n);  // pad with ones
  }
};

// One 8-point forward pass after Arai / Agui / Nakajima, in place over v[0], v[s], ..., v[7 s]; after both passes coefficient (v, u)
// carries the factor 8 * a[u] * a[v] (a[0] = 1, a[k] = sqrt(2) cos(k pi / 16)), which the quantisation multipliers absorb
inline void fdct8(float* v, int s) {
  const float t0 = v[0] + v[7 * s], t7 = v[0] - v[7 * s], t1 = v[s] + v[6 * s], t6 = v[s] - v[6 * s];
  const float t2 = v[2 * s] + v[5 * s], t5 = v[2 * s] - v[5 * s], t3 = v[3 * s] + v[4 * s], t4 = v[3 * s] - v[4 * s];
  const float e10 = t0 + t3, e13 = t0 - t3, e11 = t1 + t2, e12 = t1 - t2;
  v[0] = e10 + e11; v[4 * s] = e10 - e11;
  const float z1 = (e12 + e13) * 0.707106781f;
  v[2 * s] = e13 + z1; v[6 * s] = e13 - z1;
  const float o10 = t4 + t5, o11 = t5 + t6, o12 = t6 + t7;
  const float z5 = (o10 - o12) * 0.382683433f, z2 = 0.541196100f * o10 + z5, z4 = 1.306562965f * o12 + z5, z3 = o11 * 0.707106781f;
  const float z11 = t7 + z3, z13 = t7 - z3;
  v[5 * s] = z13 + z2; v[3 * s] = z13 - z2; v[s] = z11 + z4; v[7 * s] = z11 - z4;
}

// quantiser steps -> multipliers 1 / (q * 8 a[u] a[v]) in natural order
struct QuantMul {
  float m[64];
  explicit QuantMul(const uint8_t* q) {
    const double a[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};
    for (int v = 0; v < 8; v++)
      for (int u = 0; u < 8; u++) m[8 * v + u] = (float)(1.0 / ((double)q[8 * v + u] * 8.0 * a[u] * a[v]));
  }
};

void encode_block(Out& o, float* px, const QuantMul& qm, int& dc_pred, const HuffEnc& dc, const HuffEnc& ac) {
  // columns first (eight independent 1-D passes: the compiler vectorises across them), transpose, columns again (= the rows)
  for (int c = 0; c < 8; c++) fdct8(px + c, 8);
  float t[64];
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) t[x * 8 + y] = px[y * 8 + x];
  for (int c = 0; c < 8; c++) fdct8(t + c, 8);   // t[u * 8 + v] = coefficient (v, u)
  int nat[64];
  for (int v = 0; v < 8; v++)
    for (int u = 0; u < 8; u++) nat[8 * v + u] = (int)std::lrintf(t[u * 8 + v] * qm.m[8 * v + u]);
  int zz[64];
  for (int i = 0; i < 64; i++) zz[i] = nat[kZigzag[i]];
  auto magnitude = [](int v, int& nbits, uint32_t& bitsv) {
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    nbits = a ? 32 - __builtin_clz(a) : 0;
    bitsv = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << nbits) - 1u);
  };
  int nb;
  uint32_t bv;
  const int diff = zz[0] - dc_pred;
  dc_pred = zz[0];
  magnitude(diff, nb, bv);
  o.bits(dc.code[nb], dc.len[nb]);
  if (nb) o.bits(bv, nb);
  int run = 0, last = 63;
  while (last > 0 && zz[last] == 0) last--;
  for (int i = 1; i <= last; i++) {
    if (zz[i] == 0) { run++; continue; }
    while (run > 15) { o.bits(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
    magnitude(zz[i], nb, bv);
    const int sym = (run << 4) | nb;
    o.bits(ac.code[sym], ac.len[sym]);
    o.bits(bv, nb);
    run = 0;
  }
  if (last < 63) o.bits(ac.code[0x00], ac.len[0x00]);  // EOB
}

}  // namespace

// rgb: width*height*3 bytes.  quality 1..100 (0: 90).  subsample != 0: 4:2:0 chroma.  The blob is appended to `out`.
int jpeg_encode_rgb(const uint8_t* rgb, uint32_t width, uint32_t height, int quality, int subsample, std::vector<uint8_t>& out) {
  if (!rgb || width == 0 || height == 0 || width > 65535 || height > 65535) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_encode_rgb: bad image");
  if (quality <= 0) quality = 90;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;  // IJG jpeg_quality_scaling
  uint8_t ql[64], qc[64];
  for (int i = 0; i < 64; i++) {
    int a = (kQLuma[i] * scale + 50) / 100, b = (kQChroma[i] * scale + 50) / 100;
    ql[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
    qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
  }
  const QuantMul qml(ql), qmc(qc);
  static const HuffEnc dcl(kDcLumaBits, kDcVals), dcc(kDcChromaBits, kDcVals), acl(kAcLumaBits, kAcLumaVals), acc(kAcChromaBits, kAcChromaVals);
  out.reserve(out.size() + (size_t)width * height / 2 + 1024);
  Out o(out);
  o.word(0xFFD8);
  o.word(0xFFE0); o.word(16); o.byte('J'); o.byte('F'); o.byte('I'); o.byte('F'); o.byte(0); o.word(0x0101); o.byte(0); o.word(1); o.word(1); o.byte(0); o.byte(0);
  o.word(0xFFDB); o.word(2 + 2 * 65);
  o.byte(0); for (int i = 0; i < 64; i++) o.byte(ql[kZigzag[i]]);
  o.byte(1); for (int i = 0; i < 64; i++) o.byte(qc[kZigzag[i]]);
  o.word(0xFFC0); o.word(17); o.byte(8); o.word((uint16_t)height); o.word((uint16_t)width); o.byte(3);
  o.byte(1); o.byte(subsample ? 0x22 : 0x11); o.byte(0);
  o.byte(2); o.byte(0x11); o.byte(1);
  o.byte(3); o.byte(0x11); o.byte(1);
  auto dht = [&](int cls_id, const uint8_t* bits, const uint8_t* vals, int nvals) {
    o.word(0xFFC4); o.word((uint16_t)(2 + 1 + 16 + nvals)); o.byte((uint8_t)cls_id);
    for (int i = 0; i < 16; i++) o.byte(bits[i]);
    for (int i = 0; i < nvals; i++) o.byte(vals[i]);
  };
  dht(0x00, kDcLumaBits, kDcVals, 12);
  dht(0x10, kAcLumaBits, kAcLumaVals, 162);
  dht(0x01, kDcChromaBits, kDcVals, 12);
  dht(0x11, kAcChromaBits, kAcChromaVals, 162);
  o.word(0xFFDA); o.word(12); o.byte(3); o.byte(1); o.byte(0x00); o.byte(2); o.byte(0x11); o.byte(3); o.byte(0x11); o.byte(0); o.byte(63); o.byte(0);
  const int mcu = subsample ? 16 : 8;
  const uint32_t mx = (width + mcu - 1) / mcu, my = (height + mcu - 1) / mcu;
  int pred[3] = {0, 0, 0};
  std::vector<float> Y((size_t)mcu * mcu), Cb((size_t)mcu * mcu), Cr((size_t)mcu * mcu);
  for (uint32_t by = 0; by < my; by++)
    for (uint32_t bx = 0; bx < mx; bx++) {
      const bool inside = (bx + 1) * mcu <= width && (by + 1) * mcu <= height;
      for (int y = 0; y < mcu; y++) {
        const uint32_t sy = std::min(by * mcu + y, height - 1);  // edge replication
        const uint8_t* row = rgb + 3 * ((size_t)sy * width);
        float* yo = &Y[(size_t)y * mcu];
        float* bo = &Cb[(size_t)y * mcu];
        float* ro = &Cr[(size_t)y * mcu];
        for (int x = 0; x < mcu; x++) {
          const uint8_t* p = row + 3 * (size_t)(inside ? bx * mcu + x : std::min(bx * mcu + x, width - 1));
          const float r = p[0], g = p[1], b = p[2];
          yo[x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
          bo[x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
          ro[x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
      }
      float blk[64];
      if (subsample) {
        for (int q = 0; q < 4; q++) {
          const int oy = (q >> 1) * 8, ox = (q & 1) * 8;
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) blk[8 * y + x] = Y[(size_t)(oy + y) * 16 + ox + x];
          encode_block(o, blk, qml, pred[0], dcl, acl);
        }
        for (int c = 0; c < 2; c++) {
          const std::vector<float>& S = c ? Cr : Cb;
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++)
              blk[8 * y + x] = 0.25f * (S[(size_t)(2 * y) * 16 + 2 * x] + S[(size_t)(2 * y) * 16 + 2 * x + 1] + S[(size_t)(2 * y + 1) * 16 + 2 * x] + S[(size_t)(2 * y + 1) * 16 + 2 * x + 1]);
          encode_block(o, blk, qmc, pred[1 + c], dcc, acc);
        }
      } else {
        encode_block(o, Y.data(), qml, pred[0], dcl, acl);
        encode_block(o, Cb.data(), qmc, pred[1], dcc, acc);
        encode_block(o, Cr.data(), qmc, pred[2], dcc, acc);
      }
    }
  o.flush();
  o.word(0xFFD9);
  return SF_OK;
}

SF_API int sf_jpeg_encode(const uint8_t* rgb, uint32_t width, uint32_t height, int quality, int subsample, uint8_t* dst, uint64_t dst_capacity, uint64_t* out_bytes) {
  if (!out_bytes) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::vector<uint8_t> v;
  v.reserve((size_t)width * height / 2 + 1024);
  const int rc = jpeg_encode_rgb(rgb, width, height, quality, subsample, v);
  if (rc != SF_OK) return rc;
  *out_bytes = v.size();
  if (!dst || dst_capacity < v.size()) return sf::fail(SF_ERR_BOUNDS, "sf_jpeg_encode: %zu bytes needed", v.size());
  std::memcpy(dst, v.data(), v.size());
  return SF_OK;
}

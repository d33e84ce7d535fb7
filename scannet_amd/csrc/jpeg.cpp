// jpeg.cpp -- JPEG decoder (8-bit Huffman: baseline SOF0 / SOF1, and progressive SOF2 on the host) for .sens colour frames.
//
// Replaces stb::stbi_load_from_memory as called by RGBDFrame::decompressColorAlloc_stb
// (SensReader/c++/src/sensorData.h:609-616 -> sensorData/stb_image.h:1067,3411).  ScanNet colour frames are
// baseline YCbCr 4:2:0 / 4:2:2 JPEGs written by the capture app -- the case the device paths take; progressive
// pictures (which the reference decodes too, stb_image.h:1771-1900) are decoded here on the host.  Entropy decoding is written from ITU-T T.81 (and is the serial part); the reconstruction --
// integer IDCT, chroma upsampling, YCbCr -> RGB (jpeg_idct.h, shared with the GPU path of the frame pipeline) -- reproduces the
// integer arithmetic of the reference's decoder, so the pixels are IDENTICAL to the reference's (tests/test_sens.py), including
// its handling of the last columns of a 4:2:2 picture.  Like the reference: 8-bit quantisation tables only, over-subscribed
// Huffman tables rejected, DRI segments must be 4 bytes long (stb_image.h:2619-2650,1521-1556).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"
#include "jpeg_huff.h"
#include "jpeg_idct.h"

namespace {

struct HuffDC_AC {
  // canonical decode tables (T.81 annex F.2.2.3) + 9-bit lookahead
  uint8_t bits[17];
  uint8_t vals[256];
  int32_t mincode[17], maxcode[18], valptr[17];
  uint16_t look[512];  // (len << 8) | symbol, 0 = slow path
  // AC tables: for a 12-bit window that holds a whole short code AND the magnitude bits behind it,
  // (value << 8) | (run << 4) | total bits; 0 = decode symbol and magnitude separately
  int16_t fast_ac[4096];
  bool present = false;
  // false: the code lengths over-subscribe the code space (more codes of some length than the prefix tree has leaves left) --
  // such a table would index past `look`; stb_image.h:1543 rejects it ("bad code lengths")
  bool build(bool with_fast_ac) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      if (bits[l] && code - 1 >= (1 << l)) return false;
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    std::memset(look, 0, sizeof(look));
    code = 0; k = 0;
    for (int l = 1; l <= 9; l++) {
      for (int i = 0; i < bits[l]; i++, k++) {
        const int first = code << (9 - l);
        for (int f = 0; f < (1 << (9 - l)); f++) look[first + f] = (uint16_t)((l << 8) | vals[k]);
        code++;
      }
      code <<= 1;
    }
    for (int w = 0; w < 4096 && with_fast_ac; w++) {
      fast_ac[w] = 0;
      const uint16_t e = look[w >> 3];
      if (!e) continue;
      const int len = e >> 8, run = (e >> 4) & 15, size = e & 15;
      if (size == 0 || len + size > 12) continue;
      int v = (w >> (12 - len - size)) & ((1 << size) - 1);
      if (v < (1 << (size - 1))) v += 1 - (1 << size);   // T.81 F.2.2.1 EXTEND
      if (v >= -128 && v <= 127) fast_ac[w] = (int16_t)(v * 256 + run * 16 + len + size);
    }
    present = true;
    return true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int pred = 0;
  int bw = 0, bh = 0;  // plane size in samples (padded to whole MCUs)
  uint8_t* plane = nullptr;  // points into per-thread scratch (a frame is several MB: no allocation / page faults per frame)
};

struct BitSrc {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;   // the next bits of the entropy-coded segment, left-aligned
  int cnt = 0;        // how many of them are valid
  bool hit_marker = false;
  // top up to more than 32 valid bits: four bytes at once while none of them is 0xFF (stuffing / markers go through the byte loop)
  inline void fill() {
    if (!hit_marker && p + 4 <= end) {
      uint32_t w;
      std::memcpy(&w, p, 4);
      const uint32_t n = ~w;
      if (!((n - 0x01010101u) & ~n & 0x80808080u)) {   // no byte of w is 0xFF
        buf |= (uint64_t)__builtin_bswap32(w) << (32 - cnt);
        cnt += 32;
        p += 4;
        return;
      }
    }
    while (cnt <= 56) {
      uint64_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          const uint8_t nx = (p + 1 < end) ? p[1] : 0xD9;
          if (nx == 0) p += 2;
          else { hit_marker = true; b = 0; }
        } else p++;
      }
      buf |= b << (56 - cnt);
      cnt += 8;
    }
  }
  inline int peek(int n) { return (int)(buf >> (64 - n)); }
  inline void drop(int n) { buf <<= n; cnt -= n; }
  inline int get(int n) {
    if (n == 0) return 0;
    if (cnt < n) fill();
    const int v = peek(n);
    drop(n);
    return v;
  }
  void reset() { buf = 0; cnt = 0; hit_marker = false; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

inline int decode_huff(BitSrc& bs, const HuffDC_AC& h) {
  if (bs.cnt < 32) bs.fill();
  const uint16_t e = h.look[bs.peek(9)];
  if (e) { bs.drop(e >> 8); return e & 0xFF; }
  int code = bs.peek(9);
  int l = 9;
  const uint64_t all = bs.buf;
  while (l < 17 && code > h.maxcode[l]) {
    l++;
    code = (int)(all >> (64 - l));
  }
  if (l > 16) return -1;
  bs.drop(l);
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}

// One AC symbol with a single refill check: a code (<= 16 bits) and its magnitude bits (<= 11) always fit the 32 bits that are valid
// after fill().  Returns 0 = a coefficient (`run` zeros, then `value`), 1 = end of block, 2 = sixteen zeros, -1 = invalid code.
inline int decode_ac(BitSrc& bs, const HuffDC_AC& h, int& run, int& value) {
  if (bs.cnt < 32) bs.fill();
  const uint32_t w = (uint32_t)(bs.buf >> 32);
  const int fa = h.fast_ac[w >> 20];
  if (fa) {   // run, size and magnitude from one look-up
    run = (fa >> 4) & 15;
    value = fa >> 8;
    bs.drop(fa & 15);
    return 0;
  }
  int len, rs;
  const uint16_t e = h.look[w >> 23];
  if (e) { len = e >> 8; rs = e & 0xFF; }
  else {
    len = 9;
    int code = (int)(w >> 23);
    while (len < 17 && code > h.maxcode[len]) {
      len++;
      code = (int)(w >> (32 - len));
    }
    if (len > 16) return -1;
    rs = h.vals[h.valptr[len] + code - h.mincode[len]];
  }
  const int sz = rs & 15;
  run = rs >> 4;
  if (sz == 0) { bs.drop(len); return run == 15 ? 2 : 1; }
  value = extend((int)((w << len) >> (32 - sz)), sz);
  bs.drop(len + sz);
  return 0;
}

const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// dequantised 16-bit coefficients (natural order) -> 8 x 8 samples
void idct_block(int* blk, bool dc_only, uint8_t* out, int stride) {
  if (dc_only) {   // every other input of both passes is zero: all 64 samples equal ((dc << 14) + rounding + level shift) >> 17
    const uint8_t v = sf_jpeg_shift_clamp8((int32_t)((uint32_t)blk[0] * 16384u + 65536u + (128u << 17)), 17);
    for (int y = 0; y < 8; y++) std::memset(out + (size_t)y * stride, v, 8);
    return;
  }
  sf_idct_block_int(blk);
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) out[(size_t)y * stride + x] = (uint8_t)blk[y * 8 + x];
}

// ---- progressive JPEG (SOF2), host only.  The reference's decoder takes these (stb_image.h:1771-1900 block decoders, :2520-2556 scan loops, :2582-2598
// dequantise + IDCT at the end) and RGBDFrame::decompressColorAlloc_stb hands them through like any other picture; ScanNet's own colour frames are baseline,
// so this path is for .sens files that passed through other tools.  Coefficients are 16-bit as the reference keeps them (every store wraps to short, the
// dequantisation multiplies in short): T.81 G.1.2 spectral selection + successive approximation, with the reference's checks.
struct ProgScan {
  int ns = 0, order[3] = {0, 0, 0};
  int ss = 0, se = 0, ah = 0, al = 0;
};

// first DC scan / DC refinement of one block
inline int prog_dc(BitSrc& bs, int16_t* blk, const HuffDC_AC& h, Component& c, const ProgScan& sc) {
  if (sc.ah == 0) {
    std::memset(blk, 0, 64 * sizeof(int16_t));
    const int t = decode_huff(bs, h);
    if (t < 0 || t > 15) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DC code");
    const int diff = t ? extend(bs.get(t), t) : 0;
    c.pred += diff;
    blk[0] = (int16_t)(uint16_t)((uint32_t)c.pred << sc.al);
  } else if (bs.get(1)) {
    blk[0] = (int16_t)(blk[0] + (int16_t)(1 << sc.al));
  }
  return SF_OK;
}

// first AC scan / AC refinement of one block (band [ss, se] of the zig-zag order); eob_run carries over blocks
inline int prog_ac(BitSrc& bs, int16_t* blk, const HuffDC_AC& h, const ProgScan& sc, int& eob_run) {
  if (sc.ah == 0) {
    if (eob_run) { --eob_run; return SF_OK; }
    int k = sc.ss;
    do {
      const int rs = decode_huff(bs, h);
      if (rs < 0) return sf::fail(SF_ERR_FORMAT, "jpeg: bad AC code");
      const int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r < 15) {
          eob_run = 1 << r;
          if (r) eob_run += bs.get(r);
          --eob_run;
          break;
        }
        k += 16;
      } else {
        k += r;
        if (k > 63) return sf::fail(SF_ERR_FORMAT, "jpeg: AC run past the end of the block");
        blk[ZIGZAG[k++]] = (int16_t)(uint16_t)((uint32_t)extend(bs.get(s), s) << sc.al);
      }
    } while (k <= sc.se);
    return SF_OK;
  }
  const int16_t bit = (int16_t)(1 << sc.al);
  auto refine = [&](int16_t& v) {   // a correction bit for a coefficient that is non-zero already
    if (bs.get(1) && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
  };
  if (eob_run) {
    --eob_run;
    for (int k = sc.ss; k <= sc.se; ++k) {
      int16_t& v = blk[ZIGZAG[k]];
      if (v != 0) refine(v);
    }
    return SF_OK;
  }
  int k = sc.ss;
  do {
    const int rs = decode_huff(bs, h);
    if (rs < 0) return sf::fail(SF_ERR_FORMAT, "jpeg: bad AC code");
    int s = rs & 15, r = rs >> 4;
    if (s == 0) {
      if (r < 15) {
        eob_run = (1 << r) - 1;
        if (r) eob_run += bs.get(r);
        r = 64;   // the rest of the band only receives correction bits
      }           // r == 15: sixteen zero-history coefficients are skipped, nothing is written
    } else {
      if (s != 1) return sf::fail(SF_ERR_FORMAT, "jpeg: bad AC refinement code");
      s = bs.get(1) ? bit : -bit;
    }
    while (k <= sc.se) {
      int16_t& v = blk[ZIGZAG[k++]];
      if (v != 0) refine(v);
      else {
        if (r == 0) { v = (int16_t)s; break; }
        --r;
      }
    }
  } while (k <= sc.se);
  return SF_OK;
}

}  // namespace

// dst != nullptr: decode to RGB.  dst == nullptr: entropy-decode only -- the layout goes to *L, block table and non-zero quantised
// coefficients behind it (payload: SfJpegLayout, table, entries; the GPU reconstructs: jpeg_gpu.hip); SF_ERR_UNSUPPORTED when the layout is
// one the GPU path does not take or the payload does not fit payload_capacity bytes.
static int decode_impl(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity,
                       bool prepare_only = false) {
  SfJpegLayout* L = reinterpret_cast<SfJpegLayout*>(payload);
  uint32_t* table = nullptr;
  uint32_t* entries = nullptr;
  uint64_t max_entries = 0, nent = 0;
  if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return sf::fail(SF_ERR_FORMAT, "jpeg: missing SOI");
  uint16_t qt[4][64];   // natural order
  bool qt_ok[4] = {false, false, false, false};
  HuffDC_AC hdc[4], hac[4];
  Component comp[3];
  int ncomp = 0, width = 0, height = 0, restart = 0, hmax = 1, vmax = 1;
  uint64_t pos = 2;
  bool have_sof = false;
  auto u16 = [&](uint64_t at) { return (int)((data[at] << 8) | data[at + 1]); };
  bool progressive = false;
  auto parse_dqt = [&](uint64_t seg, uint64_t seg_end) -> int {
    uint64_t q = seg;
    while (q < seg_end) {
      const int pq = data[q] >> 4, tq = data[q] & 15;
      q++;
      if (pq != 0) return sf::fail(SF_ERR_FORMAT, "jpeg: 16-bit quantisation table (the reference decoder takes 8-bit tables only, stb_image.h:2625)");
      if (tq > 3 || q + 64 > seg_end) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DQT");
      for (int i = 0; i < 64; i++) qt[tq][ZIGZAG[i]] = data[q + i];
      q += 64;
      qt_ok[tq] = true;
    }
    return SF_OK;
  };
  auto parse_dht = [&](uint64_t seg, uint64_t seg_end) -> int {
    uint64_t q = seg;
    while (q < seg_end) {
      const int tc = data[q] >> 4, th = data[q] & 15;
      q++;
      if (tc > 1 || th > 3 || q + 16 > seg_end) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DHT");
      HuffDC_AC& h = tc ? hac[th] : hdc[th];
      int total = 0;
      h.bits[0] = 0;
      for (int i = 1; i <= 16; i++) { h.bits[i] = data[q + i - 1]; total += h.bits[i]; }
      q += 16;
      if (total > 256 || q + total > seg_end) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DHT counts");
      std::memcpy(h.vals, data + q, (size_t)total);
      q += total;
      if (!h.build(tc != 0 && !progressive)) return sf::fail(SF_ERR_FORMAT, "jpeg: bad Huffman code lengths");   // the progressive block decoders take symbols one at a time
    }
    return SF_OK;
  };
  uint64_t first_sos = 0;   // progressive: where the first scan header starts (its length field)
  while (true) {
    if (pos + 4 > n) return sf::fail(SF_ERR_FORMAT, "jpeg: truncated before SOS");
    if (data[pos] != 0xFF) return sf::fail(SF_ERR_FORMAT, "jpeg: expected a marker");
    while (pos < n && data[pos] == 0xFF) pos++;
    if (pos >= n) return sf::fail(SF_ERR_FORMAT, "jpeg: truncated inside a marker");
    const int m = data[pos++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (pos + 2 > n) return sf::fail(SF_ERR_FORMAT, "jpeg: truncated segment");
    const int len = u16(pos);
    if (len < 2 || pos + len > n) return sf::fail(SF_ERR_FORMAT, "jpeg: bad segment length");
    const uint64_t seg = pos + 2, seg_end = pos + len;
    if (m == 0xDB) {
      const int rc = parse_dqt(seg, seg_end);
      if (rc != SF_OK) return rc;
    } else if (m == 0xC4) {
      const int rc = parse_dht(seg, seg_end);
      if (rc != SF_OK) return rc;
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      progressive = m == 0xC2;
      if (len < 8 || data[seg] != 8) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: only 8-bit precision is supported");
      height = u16(seg + 1); width = u16(seg + 3); ncomp = data[seg + 5];
      if (ncomp != 1 && ncomp != 3) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: %d components not supported", ncomp);
      if (len < 8 + 3 * ncomp || width == 0 || height == 0) return sf::fail(SF_ERR_FORMAT, "jpeg: bad SOF");
      for (int i = 0; i < ncomp; i++) {
        comp[i].id = data[seg + 6 + 3 * i];
        comp[i].h = data[seg + 7 + 3 * i] >> 4; comp[i].v = data[seg + 7 + 3 * i] & 15;
        comp[i].tq = data[seg + 8 + 3 * i];
        if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return sf::fail(SF_ERR_FORMAT, "jpeg: bad sampling factors");
        hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax;
      }
      have_sof = true;
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: lossless / hierarchical / arithmetic JPEG (SOF%d) is not supported", m - 0xC0);
    } else if (m == 0xDD) {
      if (len != 4) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DRI length");
      restart = u16(seg);
    } else if (m == 0xDA) {
      if (!have_sof) return sf::fail(SF_ERR_FORMAT, "jpeg: SOS before SOF");
      if (progressive) { first_sos = pos; break; }   // scan headers of a progressive picture are read by its scan loop below
      const int ns = data[seg];
      if (ns != ncomp || len < 6 + 2 * ns) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: non-interleaved scans are not supported");
      for (int i = 0; i < ns; i++) {
        const int cid = data[seg + 1 + 2 * i];
        int ci = -1;
        for (int k = 0; k < ncomp; k++) if (comp[k].id == cid) ci = k;
        if (ci < 0) return sf::fail(SF_ERR_FORMAT, "jpeg: scan refers to an unknown component");
        comp[ci].td = data[seg + 2 + 2 * i] >> 4; comp[ci].ta = data[seg + 2 + 2 * i] & 15;
        if (comp[ci].td > 3 || comp[ci].ta > 3) return sf::fail(SF_ERR_FORMAT, "jpeg: bad table selector");
      }
      pos = seg_end;
      break;
    } else if (m == 0xD9) {
      return sf::fail(SF_ERR_FORMAT, "jpeg: EOI before any scan");
    }
    pos = seg_end;
  }
  if ((uint32_t)width != expect_w || (uint32_t)height != expect_h)
    return sf::fail(SF_ERR_FORMAT, "jpeg: image is %dx%d, header says %ux%u", width, height, expect_w, expect_h);
  for (int i = 0; i < ncomp && !progressive; i++)
    if (!qt_ok[comp[i].tq] || !hdc[comp[i].td].present || !hac[comp[i].ta].present) return sf::fail(SF_ERR_FORMAT, "jpeg: missing quantisation / Huffman table");
  if (progressive && dst == nullptr)
    return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: a progressive picture takes the host decoder (the device paths reconstruct one scan)");
  const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
  const int mcux = (width + mcu_w - 1) / mcu_w, mcuy = (height + mcu_h - 1) / mcu_h;
  const bool to_coef = dst == nullptr;
  if (to_coef) {
    if (hmax > 2 || vmax > 2) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: sampling factors above 2 take the host decoder");
    if (mcux * mcu_w > 65535 || mcuy * mcu_h > 65535) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: picture too large for the coefficient layout");
    if (payload_capacity < sizeof(SfJpegLayout)) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: payload too small");
    std::memset(L, 0, sizeof(*L));
    L->width = (uint16_t)width; L->height = (uint16_t)height;
    L->ncomp = (uint8_t)ncomp; L->hmax = (uint8_t)hmax; L->vmax = (uint8_t)vmax;
    uint32_t nb = 0;
    for (int i = 0; i < ncomp; i++) {
      if ((hmax % comp[i].h) || (vmax % comp[i].v)) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: fractional sampling ratios are not supported");
      L->h[i] = (uint8_t)comp[i].h; L->v[i] = (uint8_t)comp[i].v;
      L->bw[i] = (uint16_t)(mcux * comp[i].h * 8); L->bh[i] = (uint16_t)(mcuy * comp[i].v * 8);
      L->block_off[i] = nb;
      nb += (uint32_t)(L->bw[i] / 8) * (uint32_t)(L->bh[i] / 8);
      for (int z = 0; z < 64; z++) L->q[i][z] = qt[comp[i].tq][z];
    }
    L->nblocks = nb;
    if (prepare_only) {
      // jpeg_prepare_huff: nothing is decoded here -- the layout, the Huffman tables per component, the shape of an MCU and the entropy-coded
      // segment with its byte stuffing removed go to the device, which decodes it (jpeg_huff.h / jpeg_huff_gpu.hip)
      if (restart) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: restart intervals take the host entropy decoder");
      if (sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) + 16 > payload_capacity) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: payload too small");
      SfJpegHuffDesc* D = reinterpret_cast<SfJpegHuffDesc*>(payload + sizeof(SfJpegLayout));
      std::memset(D, 0, sizeof(*D));
      D->total_blocks = nb;
      D->mcux = (uint32_t)mcux;
      int bpm = 0;
      for (int i = 0; i < ncomp; i++)
        for (int by = 0; by < comp[i].v; by++)
          for (int bx = 0; bx < comp[i].h; bx++) {
            if (bpm >= JH_MAX_MCU_BLOCKS) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: more than 10 blocks per MCU");
            D->comp_of[bpm] = (uint8_t)i; D->bx_of[bpm] = (uint8_t)bx; D->by_of[bpm] = (uint8_t)by;
            bpm++;
          }
      D->blocks_per_mcu = (uint32_t)bpm;
      if ((uint32_t)(mcux * mcuy * bpm) != nb) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: block count does not match the MCU grid");
      auto put = [](SfJpegHuffTable& t, const HuffDC_AC& h) {
        std::memcpy(t.look, h.look, sizeof(t.look));
        for (int l = 0; l < 18; l++) t.maxcode[l] = h.maxcode[l];
        for (int l = 0; l < 17; l++) { t.mincode[l] = h.mincode[l]; t.valptr[l] = h.valptr[l]; }
        std::memcpy(t.vals, h.vals, 256);
      };
      for (int i = 0; i < ncomp; i++) { put(D->dc[i], hdc[comp[i].td]); put(D->ac[i], hac[comp[i].ta]); }
      uint8_t* ecs = payload + sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc);
      const uint64_t cap = payload_capacity - sizeof(SfJpegLayout) - sizeof(SfJpegHuffDesc);
      // runs between 0xFF bytes are copied whole (memchr + memcpy: a 200 KB segment holds ~800 of them; byte by byte this loop was 40 % of a host
      // thread's time per frame); 0xFF 0x00 is a stuffed 0xFF, any other 0xFF ends the segment (EOI; RSTn cannot occur: no restart interval)
      uint64_t w = 0, i = pos;
      while (i < n) {
        const uint8_t* ff = static_cast<const uint8_t*>(std::memchr(data + i, 0xFF, (size_t)(n - i)));
        const uint64_t run = ff ? (uint64_t)(ff - (data + i)) : n - i;
        if (w + run + 25 > cap) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: the entropy-coded segment does not fit the payload");
        std::memcpy(ecs + w, data + i, (size_t)run);
        w += run;
        i += run;
        if (!ff) break;
        if (i + 1 < n && data[i + 1] == 0x00) { ecs[w++] = 0xFF; i += 2; }
        else break;
      }
      D->ecs_bytes = (uint32_t)w;
      while (w & 3) ecs[w++] = 0;
      for (int i = 0; i < 16; i++) ecs[w++] = 0;   // the lanes fetch four words at a time (jpeg_huff.h jh_fill): the last fill stays inside
      D->ecs_words = (uint32_t)(w / 4);
      return SF_OK;
    }
    if (sizeof(SfJpegLayout) + 4ull * nb > payload_capacity) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: the block table does not fit the payload");
    table = reinterpret_cast<uint32_t*>(payload + sizeof(SfJpegLayout));
    entries = table + nb;
    max_entries = (payload_capacity - sizeof(SfJpegLayout) - 4ull * nb) / 4;
    if (max_entries > (1u << 25) - 64) max_entries = (1u << 25) - 64;   // a table word keeps 25 bits of entry index
  }
  for (int i = 0; i < ncomp; i++) {
    comp[i].bw = mcux * comp[i].h * 8; comp[i].bh = mcuy * comp[i].v * 8;
    if (to_coef) { comp[i].pred = 0; continue; }
    static thread_local std::vector<uint8_t> scratch[3];
    if (scratch[i].size() < (size_t)comp[i].bw * comp[i].bh) scratch[i].resize((size_t)comp[i].bw * comp[i].bh);  // every sample is written by the IDCT
    comp[i].plane = scratch[i].data();
    comp[i].pred = 0;
  }
  if (progressive) {
    static thread_local std::vector<int16_t> coef[3];
    int cbw[3] = {0, 0, 0};   // blocks per row of the component's (MCU-padded) plane
    for (int i = 0; i < ncomp; i++) {
      cbw[i] = comp[i].bw / 8;
      coef[i].assign((size_t)cbw[i] * (size_t)(comp[i].bh / 8) * 64, 0);
    }
    // how many blocks of a component a NON-interleaved scan holds: its real samples, whatever the MCU grid pads (stb_image.h:2521-2527)
    auto blocks_w = [&](int i) { return ((width * comp[i].h + hmax - 1) / hmax + 7) >> 3; };
    auto blocks_h = [&](int i) { return ((height * comp[i].v + vmax - 1) / vmax + 7) >> 3; };
    uint64_t at = first_sos;   // at a scan header's length field
    bool more = true;
    while (more) {
      // ---- scan header (stbi__process_scan_header, stb_image.h:2663-2698)
      if (at + 2 > n) return sf::fail(SF_ERR_FORMAT, "jpeg: truncated scan header");
      const int Ls = u16(at);
      if (Ls < 6 || at + Ls > n) return sf::fail(SF_ERR_FORMAT, "jpeg: bad SOS length");
      ProgScan sc;
      sc.ns = data[at + 2];
      if (sc.ns < 1 || sc.ns > ncomp || Ls != 6 + 2 * sc.ns) return sf::fail(SF_ERR_FORMAT, "jpeg: bad SOS component count");
      for (int i = 0; i < sc.ns; i++) {
        const int cid = data[at + 3 + 2 * i], tt = data[at + 4 + 2 * i];
        int ci = -1;
        for (int k = 0; k < ncomp; k++) if (comp[k].id == cid) { ci = k; break; }
        if (ci < 0) return sf::fail(SF_ERR_FORMAT, "jpeg: scan refers to an unknown component");
        comp[ci].td = tt >> 4; comp[ci].ta = tt & 15;
        if (comp[ci].td > 3 || comp[ci].ta > 3) return sf::fail(SF_ERR_FORMAT, "jpeg: bad table selector");
        sc.order[i] = ci;
      }
      sc.ss = data[at + 3 + 2 * sc.ns]; sc.se = data[at + 4 + 2 * sc.ns];
      sc.ah = data[at + 5 + 2 * sc.ns] >> 4; sc.al = data[at + 5 + 2 * sc.ns] & 15;
      if (sc.ss > 63 || sc.se > 63 || sc.ss > sc.se || sc.ah > 13 || sc.al > 13) return sf::fail(SF_ERR_FORMAT, "jpeg: bad SOS");
      if (sc.ss == 0 ? sc.se != 0 : sc.ns != 1) return sf::fail(SF_ERR_FORMAT, "jpeg: a scan cannot merge DC and AC coefficients");
      for (int i = 0; i < sc.ns; i++) {
        const Component& c = comp[sc.order[i]];
        if (sc.ss == 0 ? (sc.ah == 0 && !hdc[c.td].present) : !hac[c.ta].present) return sf::fail(SF_ERR_FORMAT, "jpeg: missing Huffman table");
      }
      // ---- entropy-coded data of the scan
      BitSrc bs{data + at + Ls, data + n};
      int todo = restart ? restart : 0x7FFFFFFF, eob_run = 0;
      for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
      auto restart_point = [&]() {   // as the baseline loop below: on to the RSTn marker, predictors and the end-of-band run start over
        bs.reset();
        // ... if that is what comes next: behind the last interval of a scan stands the next scan's header, not a restart marker
        while (bs.p + 1 < bs.end && !(bs.p[0] == 0xFF && bs.p[1] != 0x00 && bs.p[1] != 0xFF)) bs.p++;
        if (bs.p + 1 < bs.end && bs.p[1] >= 0xD0 && bs.p[1] <= 0xD7) bs.p += 2;
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eob_run = 0;
        todo = restart;
      };
      if (sc.ns == 1) {
        const int ci = sc.order[0];
        Component& c = comp[ci];
        const int bwn = blocks_w(ci), bhn = blocks_h(ci);
        for (int by = 0; by < bhn; by++)
          for (int bx = 0; bx < bwn; bx++) {
            int16_t* blk = coef[ci].data() + 64 * ((size_t)bx + (size_t)by * cbw[ci]);
            const int rc = sc.ss == 0 ? prog_dc(bs, blk, hdc[c.td], c, sc) : prog_ac(bs, blk, hac[c.ta], sc, eob_run);
            if (rc != SF_OK) return rc;
            if (--todo <= 0) restart_point();
          }
      } else {
        for (int my = 0; my < mcuy; my++)
          for (int mx = 0; mx < mcux; mx++) {
            for (int k = 0; k < sc.ns; k++) {
              const int ci = sc.order[k];
              Component& c = comp[ci];
              for (int by = 0; by < c.v; by++)
                for (int bx = 0; bx < c.h; bx++) {
                  int16_t* blk = coef[ci].data() + 64 * ((size_t)(mx * c.h + bx) + (size_t)(my * c.v + by) * cbw[ci]);
                  const int rc = prog_dc(bs, blk, hdc[c.td], c, sc);
                  if (rc != SF_OK) return rc;
                }
            }
            if (--todo <= 0) restart_point();
          }
      }
      // ---- on to the next marker; tables may be redefined between scans
      uint64_t q = (uint64_t)(bs.p - data);
      more = false;
      while (q + 1 < n) {
        if (data[q] != 0xFF || data[q + 1] == 0x00 || data[q + 1] == 0xFF || (data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7)) { q++; continue; }
        const int m = data[q + 1];
        q += 2;
        if (m == 0xD9) break;   // EOI
        if (q + 2 > n) return sf::fail(SF_ERR_FORMAT, "jpeg: truncated segment");
        const int len = u16(q);
        if (len < 2 || q + len > n) return sf::fail(SF_ERR_FORMAT, "jpeg: bad segment length");
        if (m == 0xDA) { at = q; more = true; break; }
        int rc = SF_OK;
        if (m == 0xDB) rc = parse_dqt(q + 2, q + len);
        else if (m == 0xC4) rc = parse_dht(q + 2, q + len);
        else if (m == 0xDD) { if (len != 4) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DRI length"); restart = u16(q + 2); }
        if (rc != SF_OK) return rc;
        q += len;
      }
    }
    // ---- every coefficient is in: dequantise in 16 bits and reconstruct the blocks that hold real samples (stbi__jpeg_finish, stb_image.h:2582-2598)
    int blk[64];
    for (int ci = 0; ci < ncomp; ci++) {
      Component& c = comp[ci];
      if (!qt_ok[c.tq]) return sf::fail(SF_ERR_FORMAT, "jpeg: missing quantisation table");
      const uint16_t* q = qt[c.tq];
      const int bwn = blocks_w(ci), bhn = blocks_h(ci);
      for (int by = 0; by < bhn; by++)
        for (int bx = 0; bx < bwn; bx++) {
          const int16_t* src = coef[ci].data() + 64 * ((size_t)bx + (size_t)by * cbw[ci]);
          for (int z = 0; z < 64; z++) blk[z] = sf_jpeg_dequant16(src[z], q[z]);
          idct_block(blk, false, c.plane + (size_t)(by * 8) * c.bw + bx * 8, c.bw);
        }
    }
  }
  BitSrc bs{data + pos, data + n};
  int todo = restart ? restart : 0x7FFFFFFF;
  int blk[64];
  if (!progressive)
  for (int my = 0; my < mcuy; my++)
    for (int mx = 0; mx < mcux; mx++) {
      for (int ci = 0; ci < ncomp; ci++) {
        Component& c = comp[ci];
        const uint16_t* q = qt[c.tq];
        for (int by = 0; by < c.v; by++)
          for (int bx = 0; bx < c.h; bx++) {
            if (to_coef) {   // the same walk, the non-zero coefficients appended instead of reconstructed
              if (nent + 64 > max_entries) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: more coefficients than the payload holds");
              const uint32_t block = L->block_off[ci] + (uint32_t)(my * c.v + by) * (uint32_t)(c.bw / 8) + (uint32_t)(mx * c.h + bx);
              const uint64_t first_entry = nent;
              const int t = decode_huff(bs, hdc[c.td]);
              if (t < 0 || t > 11) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DC code");
              c.pred += t ? extend(bs.get(t), t) : 0;
              if (c.pred != 0) entries[nent++] = (uint32_t)(uint16_t)(int16_t)c.pred;
              const HuffDC_AC& ac = hac[c.ta];
              for (int k = 1; k < 64;) {
                int run, value;
                const int what = decode_ac(bs, ac, run, value);
                if (what == 0) {
                  k += run;
                  if (k > 63) return sf::fail(SF_ERR_FORMAT, "jpeg: AC run past the end of the block");
                  entries[nent++] = ((uint32_t)ZIGZAG[k] << 16) | (uint32_t)(uint16_t)(int16_t)value;
                  k++;
                } else if (what == 2) k += 16;
                else if (what == 1) break;
                else return sf::fail(SF_ERR_FORMAT, "jpeg: bad AC code");
              }
              table[block] = ((uint32_t)first_entry << 7) | (uint32_t)(nent - first_entry);
              continue;
            }
            std::memset(blk, 0, sizeof(blk));
            const int t = decode_huff(bs, hdc[c.td]);
            if (t < 0 || t > 11) return sf::fail(SF_ERR_FORMAT, "jpeg: bad DC code");
            const int diff = t ? extend(bs.get(t), t) : 0;
            c.pred += diff;
            blk[0] = sf_jpeg_dequant16(c.pred, q[0]);
            bool dc_only = true;
            const HuffDC_AC& ac = hac[c.ta];
            for (int k = 1; k < 64;) {
              int run, value;
              const int what = decode_ac(bs, ac, run, value);
              if (what == 0) {
                k += run;
                if (k > 63) return sf::fail(SF_ERR_FORMAT, "jpeg: AC run past the end of the block");
                const int z = ZIGZAG[k];
                blk[z] = sf_jpeg_dequant16(value, q[z]);
                dc_only = false;
                k++;
              } else if (what == 2) k += 16;
              else if (what == 1) break;
              else return sf::fail(SF_ERR_FORMAT, "jpeg: bad AC code");
            }
            idct_block(blk, dc_only, c.plane + (size_t)((my * c.v + by) * 8) * c.bw + (mx * c.h + bx) * 8, c.bw);
          }
      }
      if (--todo <= 0) {
        // restart interval: skip to the RSTn marker, reset predictors
        bs.reset();
        while (bs.p + 1 < bs.end && !(bs.p[0] == 0xFF && bs.p[1] >= 0xD0 && bs.p[1] <= 0xD7)) bs.p++;
        if (bs.p + 1 < bs.end) bs.p += 2;
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        todo = restart;
      }
    }
  if (to_coef) { L->nentries = (uint32_t)nent; return SF_OK; }
  // upsample + colour convert
  if (ncomp == 1) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) {
        const uint8_t g = comp[0].plane[(size_t)y * comp[0].bw + x];
        uint8_t* o = dst + 3 * ((size_t)y * width + x);
        o[0] = o[1] = o[2] = g;
      }
    return SF_OK;
  }
  // components to full resolution and YCbCr -> RGB, one output row at a time.  Per component, by (sx, sy) = (hmax / h, vmax / v):
  // (1,1) as is; (1,2), (2,1), (2,2) triangle filter (3/4 of the nearer sample, 1/4 of the farther one; rows beyond the last valid
  // one replicate it); everything else nearest.  Same integers as sf_jpeg_upsample (jpeg_idct.h), which the GPU path evaluates per pixel.
  static thread_local std::vector<int16_t> vb[3];    // 3 x nearer + farther chroma row
  static thread_local std::vector<uint8_t> row[3];   // full-resolution row of each component
  const uint8_t* full[3];
  for (int ci = 0; ci < 3; ci++) {
    Component& c = comp[ci];
    if ((hmax % c.h) || (vmax % c.v)) return sf::fail(SF_ERR_UNSUPPORTED, "jpeg: fractional sampling ratios are not supported");
    if (vb[ci].size() < (size_t)c.bw + 2) vb[ci].resize((size_t)c.bw + 2);
    if (row[ci].size() < (size_t)c.bw * (size_t)(hmax / c.h) + 8) row[ci].resize((size_t)c.bw * (size_t)(hmax / c.h) + 8);
  }
  for (int y = 0; y < height; y++) {
    for (int ci = 0; ci < 3; ci++) {
      Component& c = comp[ci];
      const int sx = hmax / c.h, sy = vmax / c.v;
      if (sx == 1 && sy == 1) { full[ci] = c.plane + (size_t)y * c.bw; continue; }
      const int cw = (width + sx - 1) / sx, ch = (height * c.v + vmax - 1) / vmax;  // valid samples of the component
      uint8_t* o = row[ci].data();
      full[ci] = o;
      const bool tri_v = sy == 2 && sx <= 2, tri_h = sx == 2 && sy <= 2;
      if (!tri_v && !tri_h) {
        const uint8_t* r = c.plane + (size_t)(y / sy < ch ? y / sy : ch - 1) * c.bw;
        for (int x = 0; x < width; x++) o[x] = r[x / sx];
        continue;
      }
      int yn = y < ch ? y : ch - 1, yf = yn;
      if (tri_v) { const int cy = y >> 1; yn = cy; yf = (y & 1) ? (cy + 1 < ch ? cy + 1 : cy) : (cy > 0 ? cy - 1 : cy); }
      const uint8_t* rn = c.plane + (size_t)yn * c.bw;
      const uint8_t* rf = c.plane + (size_t)yf * c.bw;
      if (!tri_h) {
        for (int x = 0; x < width; x++) o[x] = (uint8_t)((3 * rn[x] + rf[x] + 2) >> 2);
      } else if (!tri_v) {   // horizontal only; the last chroma column follows the reference (jpeg_idct.h)
        if (cw == 1) { o[0] = o[1] = rn[0]; continue; }
        o[0] = rn[0];
        o[1] = (uint8_t)((3 * rn[0] + rn[1] + 2) >> 2);
        for (int cx = 1; cx < cw - 1; cx++) {
          const int m = 3 * rn[cx] + 2;
          o[2 * cx] = (uint8_t)((m + rn[cx - 1]) >> 2);
          o[2 * cx + 1] = (uint8_t)((m + rn[cx + 1]) >> 2);
        }
        o[2 * cw - 2] = (uint8_t)((3 * rn[cw - 2] + rn[cw - 1] + 2) >> 2);
        o[2 * cw - 1] = rn[cw - 1];
      } else {
        int16_t* t = vb[ci].data();
        for (int x = 0; x < cw; x++) t[x] = (int16_t)(3 * rn[x] + rf[x]);
        o[0] = (uint8_t)((t[0] + 2) >> 2);
        for (int cx = 1; cx < cw; cx++) {
          o[2 * cx - 1] = (uint8_t)((3 * t[cx - 1] + t[cx] + 8) >> 4);
          o[2 * cx] = (uint8_t)((3 * t[cx] + t[cx - 1] + 8) >> 4);
        }
        o[2 * cw - 1] = (uint8_t)((t[cw - 1] + 2) >> 2);
      }
    }
    const uint8_t* __restrict__ py = full[0];
    const uint8_t* __restrict__ pb = full[1];
    const uint8_t* __restrict__ pr = full[2];
    uint8_t* __restrict__ o = dst + 3 * (size_t)y * width;
    for (int x = 0; x < width; x++) sf_jpeg_ycc_to_rgb(py[x], pb[x], pr[x], o + 3 * x);
  }
  return SF_OK;
}

int jpeg_decode_rgb(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h) {
  if (!dst) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_decode_rgb: NULL destination");
  return decode_impl(data, n, dst, expect_w, expect_h, nullptr, 0);
}

// entropy decoding only: payload = SfJpegLayout + block table + non-zero coefficients (4-byte aligned, capacity in bytes)
int jpeg_decode_coef(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity) {
  if (!payload) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_decode_coef: NULL argument");
  return decode_impl(data, n, nullptr, expect_w, expect_h, payload, payload_capacity);
}

// nothing decoded: SfJpegLayout + SfJpegHuffDesc + the unstuffed entropy-coded segment, for the device's entropy decoder (jpeg_huff_gpu.hip).
// SF_ERR_UNSUPPORTED for what that decoder leaves to the host (restart intervals, sampling factors above 2, a segment larger than the payload).
int jpeg_prepare_huff(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity) {
  if (!payload) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_prepare_huff: NULL argument");
  return decode_impl(data, n, nullptr, expect_w, expect_h, payload, payload_capacity, true);
}

// Baseline JPEG -> RGB on the host (what sf_sens_decode_color does for a TYPE_JPEG frame)
SF_API int sf_jpeg_decode(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, uint8_t* dst_rgb) {
  if (!data || !dst_rgb) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  return jpeg_decode_rgb(data, bytes, dst_rgb, width, height);
}

// jpeg_huff.h -- the entropy decoder of the baseline-JPEG colour path as LANE PROGRAMS: compiled for the device (jpeg_huff_gpu.hip: one lane per
// chunk of the entropy-coded segment, one 1024-lane workgroup per picture) and for the host (tools/jpeg_parallelism: lock-step emulation).
//
// Replaces, for the frame pipeline, the serial Huffman decoding of stb_image as RGBDFrame::decompressColorAlloc_stb calls it
// (SensReader/c++/src/sensorData.h:600-616; stbi__jpeg_decode_block / stbi__jpeg_huff_decode, stb_image.h:1521-1700, 1703-1760): T.81 F.2.2 with
// the reference's limits (8-bit precision, Huffman tables of class 0 / 1, DC category <= 11, AC run / size nibbles).
//
// Why it can be parallel: a decoder that starts at an arbitrary bit with a guessed state falls in step with the true decoder once bit position,
// coefficient index AND the block's place in the MCU agree (luma and chroma blocks use different tables, so a wrong place decodes garbage at the
// next chroma block and the decoder re-synchronises somewhere else -- until it lands on the truth, which it never leaves again).  The segment is
// cut into C <= 1024 chunks:
//   stage A  lane c decodes chunk c from the END state of lane c - 1 in the previous round (round 0: a guess) and records its own end state and
//            what it saw: blocks completed, entries produced, the sum of the DC differences per component; lanes whose start did not change
//            do nothing.  Lane 0 starts from the truth, so the fixed point is the sequential decode (measured on 1296x968 pictures: 3-6 rounds,
//            2.0-2.3 scans per chunk: tools/jpeg_parallelism/README.md).
//   stage B  exclusive prefix sums over the lanes: block ordinal, entry index, DC predictor per component at the start of every chunk.
//   stage C  every lane decodes its chunk once more and writes: one entry per DC value (absolute) and per non-zero AC coefficient, and the table
//            word of every block it completes -- the payload k_jpeg_idct reads (jpeg_idct.h SfJpegLayout: table[b] = first entry << 7 | count).
#pragma once
#include <stdint.h>

#include "jpeg_idct.h"

#if defined(__HIPCC__)
#define JH_HD __host__ __device__ inline
#else
#define JH_HD inline
#endif

constexpr int JH_MAX_CHUNKS = 1024;
constexpr uint32_t JH_MIN_CHUNK_BITS = 512;
constexpr int JH_MAX_MCU_BLOCKS = 10;   // T.81 B.2.3

struct SfJpegHuffTable {
  uint16_t look[512];   // the next 9 bits -> (code length << 8) | value; 0: the code is longer than 9 bits (or invalid)
  int32_t maxcode[18];  // per length: largest code of that length, -1 if none ([17] = sentinel)
  int32_t mincode[17];
  int32_t valptr[17];
  uint8_t vals[256];
};
// the prepared picture the host hands to the device: SfJpegLayout (nentries 0), this descriptor, then ecs_words * 4 bytes of entropy-coded data
// with the byte stuffing removed (0xFF00 -> 0xFF), big-endian bit order as in the file, zero padded
struct SfJpegHuffGeom {
  uint32_t ecs_bytes, ecs_words;
  uint32_t total_blocks, mcux;
  uint32_t blocks_per_mcu;
  uint32_t reserved;
  uint8_t comp_of[12], bx_of[12], by_of[12];   // per block of an MCU: component, block column / row inside the MCU's share of that component
};
struct SfJpegHuffDesc : SfJpegHuffGeom {
  SfJpegHuffTable dc[3], ac[3];     // per component
};
static_assert(sizeof(SfJpegHuffGeom) == 60 && sizeof(SfJpegHuffTable) % 4 == 0 && sizeof(SfJpegHuffDesc) % 4 == 0, "the prepared payload is read as 32-bit words");

struct JHState {
  uint32_t p;      // bit position
  uint16_t bi, k;  // block within the MCU; next coefficient index (0 = the DC symbol comes next)
  uint32_t open;   // entries produced so far for the block that is open
};
JH_HD bool jh_same(const JHState& a, const JHState& b) { return a.p == b.p && a.bi == b.bi && a.k == b.k && a.open == b.open; }

struct JHCounts {
  uint32_t blocks, entries;
  int32_t dc_sum[3];
  uint32_t bad;   // an invalid code was met (counts only on the true path)
};

// 32 bits of the segment starting at bit p (words are the file's bytes in order: big-endian bit numbering), out of a lane's buffer of FOUR words.
// The buffer is refilled by every lane at once, every third symbol, wherever the lane stands: a symbol takes at most 16 + 15 bits, so from a bit
// offset below 32 three symbols look no further than bit 93 + 32 of the 128 (the segment is padded with 16 zero bytes for the last fill:
// jpeg_prepare_huff).  Round 5 kept two words per lane and reloaded when the position crossed into another word -- one lane in six per symbol, so nearly
// every symbol step of a WAVE ran the reload path and waited for its load; now one 16-byte load per lane and three symbols, the same for all lanes.
JH_HD uint32_t jh_be32(uint32_t w) { return (w >> 24) | ((w >> 8) & 0xFF00u) | ((w << 8) & 0xFF0000u) | (w << 24); }
// four words of the segment from word w on, byte-swapped
JH_HD void jh_fill(const uint32_t* words, uint32_t w, uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t jh_u32x4 __attribute__((ext_vector_type(4), aligned(4)));   // ONE global_load_dwordx4 at a 4-byte aligned address
  const jh_u32x4 v = *reinterpret_cast<const jh_u32x4*>(words + w);
  d0 = jh_be32(v.x); d1 = jh_be32(v.y); d2 = jh_be32(v.z); d3 = jh_be32(v.w);
#else
  d0 = jh_be32(words[w]); d1 = jh_be32(words[w + 1]); d2 = jh_be32(words[w + 2]); d3 = jh_be32(words[w + 3]);
#endif
}
// one Huffman symbol from the 32 bits `w`: its value, code length in len; -1 = no such code
JH_HD int jh_symbol(const SfJpegHuffTable& t, uint32_t w, int& len) {
  const uint16_t e = t.look[w >> 23];
  if (e) { len = e >> 8; return e & 0xFF; }
  int l = 9, code = (int)(w >> 23);
  while (l < 17 && code > t.maxcode[l]) {
    l++;
    code = (int)(w >> (32 - l));
  }
  if (l > 16) { len = 1; return -1; }
  len = l;
  return t.vals[(t.valptr[l] + code - t.mincode[l]) & 255];
}
// The same through a 12-bit first-level table (what the device keeps in LDS, built there from the canonical ranges): with 9 bits one symbol in twenty of a
// photo-like picture took the ladder above, so nearly every symbol step of a WAVE did, up to seven rounds of it; with 12 bits it is one in hundreds.
constexpr int JH_LOOK12 = 4096;
JH_HD uint16_t jh_look12_entry(const SfJpegHuffTable& t, uint32_t idx) {   // idx = the next 12 bits
  int l = 1, code = (int)(idx >> 11);
  while (l < 13 && code > t.maxcode[l]) {
    l++;
    code = (int)(idx >> (12 - l));
  }
  if (l > 12) return 0;
  return (uint16_t)((l << 8) | t.vals[(t.valptr[l] + code - t.mincode[l]) & 255]);
}
JH_HD int jh_symbol12(const SfJpegHuffTable& t, const uint16_t* look12, uint32_t w, int& len) {
  const uint16_t e = look12[w >> 20];
  if (e) { len = e >> 8; return e & 0xFF; }
  int l = 12, code = (int)(w >> 20);
  while (l < 17 && code > t.maxcode[l]) {
    l++;
    code = (int)(w >> (32 - l));
  }
  if (l > 16) { len = 1; return -1; }
  len = l;
  return t.vals[(t.valptr[l] + code - t.mincode[l]) & 255];
}
JH_HD int jh_extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// natural-order position of zig-zag index k (T.81 figure A.6)
JH_HD int jh_zigzag(int k) {
  const uint8_t Z[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return Z[k & 63];
}

// Decode from state s while symbols START before `limit` (and inside the segment).  E receives what stage C writes:
//   E.entry(is_dc, component, zig-zag index, value) -> at every DC symbol (value = the DC difference) and every non-zero AC coefficient;
//   E.block_done(component, bi, entries of the block) returns false to stop (the picture is complete).  Stage A passes an E that does nothing.
// A DC and an AC symbol go down ONE path (the table is a select, the rest arithmetic on the same registers): the lanes of a wave sit at different
// coefficient indices, and with a branch per kind every symbol step of the wave ran both.
template <class Emit>
JH_HD JHState jh_run(const SfJpegHuffGeom& D, const SfJpegHuffTable* dc, const SfJpegHuffTable* ac, const uint16_t* dc12, const uint16_t* ac12, const uint32_t* words, JHState s,
                     uint32_t limit, JHCounts& n, Emit& E) {   // dc12 / ac12: [3][JH_LOOK12] first-level tables (jh_look12_entry) or nullptr: the tables' own 9-bit ones
  const uint32_t nbits = D.ecs_bytes * 8u;
  n.blocks = n.entries = 0;
  n.dc_sum[0] = n.dc_sum[1] = n.dc_sum[2] = 0;
  n.bad = 0;
  uint32_t wb = 0u, d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;   // the lane's window: words wb .. wb + 3 (plain registers: a struct handed about by reference ended up in LDS)
  uint32_t until_fill = 0u;
  while (s.p < limit && s.p < nbits) {
    if (until_fill == 0u) { wb = s.p >> 5; jh_fill(words, wb, d0, d1, d2, d3); until_fill = 3u; }
    until_fill--;
    const int ci = D.comp_of[s.bi];
    uint32_t w;
    {
      const uint32_t i = (s.p >> 5) - wb, sh = s.p & 31u;   // i <= 2 on this schedule
      const uint32_t lo = i == 0u ? d0 : (i == 1u ? d1 : d2), hi = i == 0u ? d1 : (i == 1u ? d2 : d3);
      w = (uint32_t)(((((uint64_t)lo << 32) | (uint64_t)hi) << sh) >> 32);
    }
    const bool isdc = s.k == 0;
    int len;
    const int rs = dc12 ? jh_symbol12(isdc ? dc[ci] : ac[ci], (isdc ? dc12 : ac12) + ci * JH_LOOK12, w, len) : jh_symbol(isdc ? dc[ci] : ac[ci], w, len);
    if (rs < 0 || (isdc && rs > 11)) { n.bad = 1; s.p += 1; continue; }   // invalid under this state: slip a bit and keep going (a guess; an error on the true path)
    const int run = isdc ? 0 : rs >> 4, size = isdc ? rs : rs & 15;
    const int val = size ? jh_extend((int)(((w << len) >> 1) >> (31 - size)), size) : 0;   // len + size <= 31
    s.p += (uint32_t)(len + size);
    const int k = (int)s.k + run;
    const bool over = !isdc && size != 0 && k > 63;                 // a run past the end of the block
    const bool emit = isdc || (size != 0 && k <= 63);
    if (over) n.bad = 1;
    int knew = isdc ? 1 : (size == 0 ? (run == 15 ? (int)s.k + 16 : 64) : (k > 63 ? 64 : k + 1));
    if (knew > 64) knew = 64;   // sixteen zeros past the end of the block end it (as the host decoder's loop does)
    s.k = (uint16_t)knew;
    s.open = isdc ? 1u : s.open + (emit ? 1u : 0u);
    n.entries += emit ? 1u : 0u;
    n.dc_sum[0] += (isdc && ci == 0) ? val : 0;   // three registers and selects: an array indexed by the component lives in private memory on the device
    n.dc_sum[1] += (isdc && ci == 1) ? val : 0;
    n.dc_sum[2] += (isdc && ci == 2) ? val : 0;
    if (emit) E.entry(isdc, ci, k, val);
    if (s.k >= 64) {
      n.blocks++;
      const bool more = E.block_done(ci, s.bi, s.open);
      s.k = 0;
      s.open = 0;
      s.bi = (uint16_t)(s.bi + 1 == D.blocks_per_mcu ? 0 : s.bi + 1);
      if (!more) break;
    }
  }
  return s;
}

struct JHNoEmit {
  JH_HD void entry(bool, int, int, int) {}
  JH_HD bool block_done(int, int, uint32_t) { return true; }
};

// how a segment of `nbits` bits is cut
JH_HD void jh_geometry(uint32_t nbits, uint32_t& C, uint32_t& B) {
  uint32_t c = nbits / JH_MIN_CHUNK_BITS;
  if (c < 1u) c = 1u;
  if (c > (uint32_t)JH_MAX_CHUNKS) c = (uint32_t)JH_MAX_CHUNKS;
  C = c;
  B = (nbits + c - 1u) / c;
}

// table index of block `ordinal` (decode order): its component's first block + row * blocks per row + column
JH_HD uint32_t jh_block_index(const SfJpegLayout& L, const SfJpegHuffGeom& D, uint32_t ordinal, int bi) {
  const uint32_t mcu = ordinal / D.blocks_per_mcu, mx = mcu % D.mcux, my = mcu / D.mcux;
  const int ci = D.comp_of[bi];
  return L.block_off[ci] + (my * L.v[ci] + D.by_of[bi]) * (uint32_t)(L.bw[ci] / 8) + mx * L.h[ci] + D.bx_of[bi];
}

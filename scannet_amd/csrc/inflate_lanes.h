// inflate_lanes.h -- inflate of ONE final fixed-Huffman deflate block as lane programs: compiled for the device (inflate_gpu.hip) and for the host
// (tools/inflate_parallelism/emulate_gpu.cpp: the lock-step emulation the CPU tests run against zlib).
//
// What it replaces, for the frame pipeline: the per-frame zlib inflate of the reference's depth path -- RGBDFrame::decompressDepthAlloc_stb ->
// stbi_zlib_decode_malloc (SensReader/c++/src/sensorData.h:693-709, stb_image.h:3791-3846) -- for exactly the streams the reference's writer
// produces: stbi_zlib_compress emits ONE final block with the fixed Huffman code (stb_image_write.h:733-736: `stbiw__zlib_add(1,1);
// stbiw__zlib_add(1,2);`), and so does this repository's writer.  Anything else (stored / dynamic blocks, several blocks) stays with the host
// inflater (zlib_codec.cpp).
//
// Two kernels per batch of frames:
//   TOKENS  one 1024-lane workgroup per frame, the stream cut into C <= 1024 chunks of >= 2400 bits:
//     stage A  lane c decodes chunk c from where lane c - 1 stopped in the previous round (round 0: from 1024 bits in front of the chunk, counting
//              from the first token that starts inside it) and records where it stops and how many bytes its tokens produce; lanes whose
//              start did not change do nothing.  With a FIXED code a decoder started at
//              an arbitrary bit falls in step with the true tokens after ~20 tokens, lane 0 starts from the truth: the fixed point is the true
//              tokenisation (measured: 2-3 rounds, 1.95 scans per chunk; tools/inflate_parallelism/README.md).
//     stage B  exclusive prefix sum of the chunks' output sizes.
//     stage C  every lane decodes its chunk once more and writes the PLAN, one u16 per OUTPUT BYTE: a literal's value, or 0x8000 | (far - 1) for
//              byte i of a match (length L, distance d) with far = d * (1 + i / d) -- the distance to the byte IN FRONT of the match it
//              repeats, so that a run (d < L) does not refer to itself.
//   COPIES  one 256-lane workgroup per frame walks the plan in groups of 1024 bytes (4 per lane) with the last 32 KiB in an LDS ring: bytes whose
//           source lies in front of the group read the ring; bytes whose source lies inside the group are resolved by pointer jumping over the
//           group (rounds of "take the value if my source has one, else adopt my source's source"); then the group goes to the ring and to
//           memory.  The chain of dependent copies that makes inflate sequential (depth rows copy the rows above them: 128-1641 matches deep)
//           is walked in 600 steps of a few LDS round trips each instead of ~200 000 token steps.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define IL_HD __host__ __device__ inline
#else
#define IL_HD inline
#endif

constexpr int IL_MAX_CHUNKS = 1024;
constexpr uint32_t IL_MIN_CHUNK_BITS = 2400;
constexpr uint32_t IL_NONE = 0xFFFFFFFFu;
constexpr uint32_t IL_GROUP = 1024;      // bytes per step of the copy stage: 4 per lane of a 256-lane workgroup
constexpr uint32_t IL_WINDOW = 32768;    // RFC 1951: distances up to 32 768

enum : uint32_t { IL_FLAG_OK = 0, IL_FLAG_EOB = 1, IL_FLAG_ERR = 2 };   // bits
enum : int32_t {
  IL_ST_OK = 0,
  IL_ST_NOT_FIXED = -1,    // not a zlib stream of one final fixed-Huffman block: the host inflater's business
  IL_ST_BAD_CODE = -2,     // an invalid code on the true token path
  IL_ST_NO_EOB = -3,       // the stream ends without an end-of-block code
  IL_ST_SIZE = -4,         // the block inflates to another size than the caller expects
  IL_ST_BAD_DISTANCE = -5  // a match reaches in front of the output
};

// lit/len table, indexed by the next 9 stream bits (bit 0 = the first bit = the MSB of the code, RFC 1951 3.1.1):
//   bits 0..3 code length | bits 4..5 kind (0 literal, 1 length, 2 end of block, 3 invalid) | bits 6..9 extra bits | bits 16..31 literal / length base
IL_HD uint32_t il_lit_entry(uint32_t i) {
  uint32_t rev = 0;
  for (int b = 0; b < 9; b++) rev |= ((i >> b) & 1u) << (8 - b);
  const uint32_t top7 = rev >> 2, top8 = rev >> 1;
  uint32_t sym, nb;
  if (top7 <= 0x17u) { sym = 256u + top7; nb = 7; }
  else if (top8 >= 0x30u && top8 <= 0xBFu) { sym = top8 - 0x30u; nb = 8; }
  else if (top8 >= 0xC0u && top8 <= 0xC7u) { sym = 280u + (top8 - 0xC0u); nb = 8; }
  else { sym = 144u + (rev - 0x190u); nb = 9; }
  if (sym < 256u) return nb | (0u << 4) | (sym << 16);
  if (sym == 256u) return nb | (2u << 4);
  if (sym > 285u) return nb | (3u << 4);
  const uint32_t L = sym - 257u;
  uint32_t base, ex;
  if (L < 8u) { base = 3u + L; ex = 0; }
  else if (L == 28u) { base = 258u; ex = 0; }
  else { ex = (L >> 2) - 1u; base = 3u + ((4u + (L & 3u)) << ex); }
  return nb | (1u << 4) | (ex << 6) | (base << 16);
}
// distance table, indexed by the next 5 stream bits: bits 0..3 extra bits | bit 4 invalid | bits 16..31 base
IL_HD uint32_t il_dist_entry(uint32_t i) {
  uint32_t code = 0;
  for (int b = 0; b < 5; b++) code |= ((i >> b) & 1u) << (4 - b);
  if (code >= 30u) return 1u << 4;
  if (code < 4u) return (code + 1u) << 16;
  const uint32_t ex = (code >> 1) - 1u;
  return ex | ((1u + ((2u + (code & 1u)) << ex)) << 16);
}

// The block as the lanes see it -- a type S with
//   uint32_t word(uint32_t i)     the deflate data as 32-bit words (the zlib stream from its third byte on; words behind the data read as anything:
//                                 a token that ends behind nbits is an error whatever it decodes to)
//   uint32_t lit(uint32_t i), dist(uint32_t i)    the two tables above
//   uint32_t nbits                bits of deflate data incl. the Adler-32 trailer
// The host reads a plain array (ILStream); the device keeps 128 bytes of every lane's chunk in LDS (inflate_gpu.hip: LaneWindow).
struct ILStream {
  const uint32_t* words;
  uint32_t nwords;
  uint32_t nbits;
  uint32_t word(uint32_t w) const { return w < nwords ? words[w] : 0u; }
  uint32_t lit(uint32_t i) const { return il_lit_entry(i); }     // the host emulation computes the entries; the device holds them in LDS
  uint32_t dist(uint32_t i) const { return il_dist_entry(i); }
};

struct ILBits {
  uint64_t buf;
  uint32_t cnt, word, pos;
};
template <class S>
IL_HD void il_bits_init(S& s, ILBits& b, uint32_t pos) {
  b.pos = pos;
  b.word = pos >> 5;
  const uint32_t sh = pos & 31u;
  b.buf = (uint64_t)(s.word(b.word) >> sh);
  b.cnt = 32u - sh;
  b.word++;
}
template <class S>
IL_HD void il_refill(S& s, ILBits& b) {   // afterwards cnt >= 33 > 31 = the longest token
  if (b.cnt <= 32u) {
    b.buf |= (uint64_t)s.word(b.word) << b.cnt;
    b.cnt += 32u;
    b.word++;
  }
}
IL_HD void il_consume(ILBits& b, uint32_t n) { b.buf >>= n; b.cnt -= n; b.pos += n; }

struct ILToken {
  uint32_t kind;   // 0 literal, 1 match, 2 end of block, 3 invalid
  uint32_t value;  // literal byte / match length
  uint32_t dist;
};
// one token at the reader's position (consumed).  No branch on the kind: the lanes of a wave hold literals and matches side by side.  (The
// code is fixed and could be decoded by arithmetic instead of the two look-ups; measured: 15 % slower -- the kernel is bound by the number of
// instructions it issues, not by the look-ups' latency.)
template <class S>
IL_HD ILToken il_token(S& s, ILBits& b) {
  il_refill(s, b);
  const uint32_t lo = (uint32_t)b.buf;
  const uint32_t e = s.lit(lo & 511u);
  const uint32_t nb = e & 15u, kind = (e >> 4) & 3u, ex = (e >> 6) & 15u;   // ex = 0 unless a length
  const uint32_t at = nb + ex;                                              // <= 13
  const uint32_t rest = (uint32_t)(b.buf >> at);                            // what follows the length: 5 distance bits + <= 13 extra
  const uint32_t d = s.dist(rest & 31u);
  const uint32_t dex = d & 15u;
  const bool match = kind == 1u;
  ILToken t;
  t.value = (e >> 16) + ((lo >> nb) & ((1u << ex) - 1u));
  t.dist = match ? (d >> 16) + ((rest >> 5) & ((1u << dex) - 1u)) : 0u;
  t.kind = (match && (d & 16u)) ? 3u : kind;
  il_consume(b, match ? at + 5u + dex : nb);
  return t;
}

// stage A: the tokens that START in [start, limit): where the first token behind them starts, how many bytes they produce.
// Nothing stops the scan before the limit: a lane that started at a wrong bit must reach the end of its chunk (and resynchronise on the way) even
// if it meets an invalid code or what looks like an end-of-block code -- the flags only count if this turns out to be the true token path, and
// then: bytes are counted up to the first end-of-block code, and invalid codes behind it (the Adler-32 trailer read as tokens) are no error.
// In pieces (begin / run until a bit position / read the result) because the device streams the chunk through LDS 64 bytes at a time.
struct ILScan {
  ILBits b;
  uint32_t out, fl;
  bool off_end;      // ran off the end of the stream
  uint32_t own;      // tokens that start in front of this bit are decoded (to fall in step) but belong to the chunk before: not counted
  uint32_t first;    // where the first counted token starts (IL_NONE: none yet)
};
template <class S>
IL_HD void il_scan_begin(S& s, ILScan& sc, uint32_t start, uint32_t own) {
  il_bits_init(s, sc.b, start);
  sc.out = 0;
  sc.fl = IL_FLAG_OK;
  sc.off_end = false;
  sc.own = own;
  sc.first = IL_NONE;
}
template <class S>
IL_HD void il_scan_run(S& s, ILScan& sc, uint32_t until) {   // the tokens that start in front of bit `until`
  while (!sc.off_end && sc.b.pos < until) {
    const uint32_t at = sc.b.pos;
    const ILToken t = il_token(s, sc.b);
    const bool counted = at >= sc.own;
    sc.first = (counted && sc.first == IL_NONE) ? at : sc.first;
    if (sc.b.pos > s.nbits) {
      if (!(sc.fl & IL_FLAG_EOB)) sc.fl |= IL_FLAG_ERR;
      sc.off_end = true;
      break;
    }
    const bool live = counted && !(sc.fl & IL_FLAG_EOB);   // behind the end of the block only the position matters
    sc.out += live ? (t.kind == 0u ? 1u : (t.kind == 1u ? t.value : 0u)) : 0u;
    sc.fl |= live ? (t.kind == 3u ? IL_FLAG_ERR : (t.kind == 2u ? IL_FLAG_EOB : 0u)) : 0u;
  }
}
IL_HD uint32_t il_scan_first(const ILScan& sc) { return sc.first == IL_NONE ? sc.b.pos : sc.first; }   // where the chunk's own tokens begin on this path
// Round 0 does not know where chunk c's first token starts.  It starts IL_LEAD_BITS in front of the chunk: a decoder started at an arbitrary bit
// is in step with the true tokens after ~20 of them (tools/inflate_parallelism/README.md), so most lanes reach their chunk on the true path and
// the boundary they cross it at is the left neighbour's stop -- no second scan.  The others rescan from the neighbour's stop, as before.
constexpr uint32_t IL_LEAD_BITS = 1024;
IL_HD uint32_t il_guess_start(uint32_t c, uint32_t B) { return (c == 0u || c * B < IL_LEAD_BITS + 3u) ? 3u : c * B - IL_LEAD_BITS; }
template <class S>
IL_HD void il_scan_chunk(S& s, uint32_t start, uint32_t own, uint32_t limit, uint32_t& first, uint32_t& end, uint32_t& out_bytes, uint32_t& flag) {
  ILScan sc;
  il_scan_begin(s, sc, start, own);
  il_scan_run(s, sc, limit);
  first = il_scan_first(sc);
  end = sc.b.pos;
  out_bytes = sc.out;
  flag = sc.fl;
}

// how a block of `nbits` bits is cut: C chunks of B bits
IL_HD void il_geometry(uint32_t nbits, uint32_t& C, uint32_t& B) {
  uint32_t c = nbits / IL_MIN_CHUNK_BITS;
  if (c < 1u) c = 1u;
  if (c > (uint32_t)IL_MAX_CHUNKS) c = (uint32_t)IL_MAX_CHUNKS;
  C = c;
  B = (nbits + c - 1u) / c;
}

// stage C: the chunk's tokens once more, written as the plan from offset o on (o_end: what stage A counted for the chunk) through a sink P with
// P.put(uint16_t) and P.put_run(uint16_t, n) (consecutive offsets) -- the device packs four entries into one 8-byte store.  Returns IL_ST_OK or why the frame is corrupt.
constexpr uint16_t IL_PLAN_COPY = 0x8000u;
struct ILWrite {
  ILBits b;
  uint32_t o;
  int32_t status;
};
template <class S>
IL_HD void il_write_begin(S& s, ILWrite& w, uint32_t start, uint32_t o) {
  il_bits_init(s, w.b, start);
  w.o = o;
  w.status = IL_ST_OK;
}
template <class S, class P>
IL_HD void il_write_run(S& s, ILWrite& w, uint32_t o_end, uint32_t until, P& plan) {   // the tokens that start in front of bit `until`
  while (w.status == IL_ST_OK && w.o < o_end && w.b.pos < until) {
    const ILToken t = il_token(s, w.b);
    if (t.kind == 0u) {
      plan.put((uint16_t)t.value);
      w.o++;
    } else if (t.kind == 1u) {
      if (t.dist > w.o) { w.status = IL_ST_BAD_DISTANCE; break; }
      const uint32_t n = t.value < o_end - w.o ? t.value : o_end - w.o;   // stage A counted whole tokens: n == t.value on the true path
      if (t.dist >= n) {
        // the usual match (a depth row copies the row above it): every byte is the same distance from its source -- one value, n times
        // (entry by entry, the lanes of a wave waited for the longest match among them: the writing pass took 0.8 of the kernel's 1.1 ms)
        plan.put_run((uint16_t)(IL_PLAN_COPY | (t.dist - 1u)), n);
      } else {
        uint32_t far = t.dist - 1u, within = 0;                           // a run that repeats itself: far <= 514
        for (uint32_t i = 0; i < n; i++) {
          plan.put((uint16_t)(IL_PLAN_COPY | far));
          if (++within == t.dist) { within = 0; far += t.dist; }
        }
      }
      w.o += n;
    } else {
      w.status = IL_ST_BAD_CODE;   // stage A saw this chunk clean up to o_end: cannot happen
    }
  }
}
template <class S, class P>
IL_HD int32_t il_write_chunk(S& s, uint32_t start, uint32_t o, uint32_t o_end, P& plan) {
  ILWrite w;
  il_write_begin(s, w, start, o);
  il_write_run(s, w, o_end, 0xFFFFFFFFu, plan);
  return w.status;
}

// ------------------------------------------------------------------------------------------------ the copy stage, one group of IL_GROUP bytes
// Lane l of IL_GROUP / 4 owns bytes 4 l .. 4 l + 3 of the group at output offset `pos`.  M is the workgroup's shared memory:
//   uint8_t  ring(uint32_t i)                     the byte at output offset i (i in [pos - 32 768, pos))
//   uint16_t& gref(uint32_t j), uint8_t& gval(j)  per byte of the group: 0xFFFF + its value once known, else the group index of its source
// The phases are separated by a barrier over the workgroup (the emulation: loops over the lanes).
constexpr uint16_t IL_KNOWN = 0xFFFFu;
struct ILQuad {
  uint32_t v;         // the four bytes, little endian
  uint16_t ref[4];    // IL_KNOWN or the group index (< 4 l + b) of the source
};
// phase 1: literals and copies from in front of the group; n = bytes of the group that exist (256 but for the last group).  The four ring reads
// are unconditional (one wait for all of them); what a byte is decides which value it keeps.
template <class M>
IL_HD bool il_quad_classify(const M& m, uint32_t pos, uint32_t lane, uint32_t n, uint64_t plan4, ILQuad& q, bool& bad) {
  bool open = false;
  uint32_t far[4], e4[4];
  uint8_t r[4];
  for (uint32_t b = 0; b < 4u; b++) {
    e4[b] = (uint32_t)(plan4 >> (16u * b)) & 0xFFFFu;
    far[b] = (e4[b] & 0x7FFFu) + 1u;
    r[b] = m.ring(pos + 4u * lane + b - far[b]);
  }
  q.v = 0;
  for (uint32_t b = 0; b < 4u; b++) {
    const uint32_t j = 4u * lane + b;
    const bool copy = (e4[b] & IL_PLAN_COPY) != 0u && j < n;
    const bool inside = copy && far[b] <= j;
    bad = bad || (copy && far[b] > pos + j);
    q.ref[b] = inside ? (uint16_t)(j - far[b]) : IL_KNOWN;
    open = open || inside;
    q.v |= (uint32_t)(copy ? (inside ? 0u : (uint32_t)r[b]) : (e4[b] & 0xFFu)) << (8u * b);
  }
  return open;
}
// phase 2: what the others may read
template <class M>
IL_HD void il_quad_publish(M& m, uint32_t lane, const ILQuad& q) {
  for (uint32_t b = 0; b < 4u; b++) {
    m.gref(4u * lane + b) = q.ref[b];
    m.gval(4u * lane + b) = (uint8_t)(q.v >> (8u * b));
  }
}
// phase 3: take the value if my source has one, else adopt my source's source; true while a byte of this lane is still open.  The eight
// reads are unconditional (one wait for all of them).
template <class M>
IL_HD bool il_quad_resolve(const M& m, ILQuad& q) {
  uint16_t rr[4];
  uint8_t vv[4];
  for (uint32_t b = 0; b < 4u; b++) {
    const uint32_t r = q.ref[b] == IL_KNOWN ? 0u : q.ref[b];
    rr[b] = m.gref(r);
    vv[b] = m.gval(r);
  }
  bool open = false;
  for (uint32_t b = 0; b < 4u; b++) {
    const bool mine = q.ref[b] != IL_KNOWN, got = mine && rr[b] == IL_KNOWN;
    q.v = got ? (q.v & ~(0xFFu << (8u * b))) | ((uint32_t)vv[b] << (8u * b)) : q.v;
    q.ref[b] = mine ? rr[b] : IL_KNOWN;
    open = open || (mine && !got);
  }
  return open;
}

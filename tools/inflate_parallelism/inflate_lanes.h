// inflate_lanes.h -- chunk-parallel inflate of ONE fixed-Huffman deflate block: the per-lane programs, compiled for the device (inflate_gpu.hip:
// one lane per chunk, one 1024-lane workgroup per depth frame) and for the host (the lock-step emulation the CPU tests run against zlib).
//
// What it replaces: the per-frame zlib inflate of the reference's depth path -- RGBDFrame::decompressDepthAlloc_stb -> stbi_zlib_decode_malloc
// (SensReader/c++/src/sensorData.h:693-709, stb_image.h:3791-3846) -- for exactly the streams the reference's writer produces: stb's
// stbi_zlib_compress emits ONE final block with the fixed Huffman code (stb_image_write.h:733-736: `stbiw__zlib_add(1,1); stbiw__zlib_add(1,2);`),
// and so does this repository's writer.  Anything else (stored / dynamic blocks, several blocks) stays on the host inflater (zlib_codec.cpp).
//
// Why it can be parallel at all: with a FIXED code a decoder that starts at an arbitrary bit resynchronises with the true token boundaries after
// a few tokens, so the stream is cut into C chunks of B >= 2400 bits and
//   stage A  every lane decodes its chunk from a guessed start (the chunk's first bit) up to the chunk's end, recording where it stopped; a lane's
//            stop is the next lane's start; repeated for the lanes whose start changed until nothing changes.  Lane 0's start is the true one, so
//            the fixed point is the true tokenisation (induction over the chunks); in practice two or three rounds, at most C + 1.
//   stage B  exclusive prefix sum of the chunks' output sizes = where every chunk's bytes go.
//   stage C  every lane decodes its chunk once more and writes: literals at once, a match as soon as its source bytes have been written -- by the
//            lane itself or by the lanes of earlier chunks, whose progress is published per chunk.  Dependencies point strictly backwards in the
//            output, so the lane with the earliest unfinished byte can always proceed: no deadlock.  At most 8 bytes per lane and turn, so that the
//            lanes of a wave stay in step.
// A chunk covers >= 2400 bits, i.e. >= 263 output bytes (9 bits per byte at worst, a token may start up to 31 bits late), more than the longest match
// (258): the source bytes of one turn straddle at most two chunks.
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
constexpr uint32_t IL_TURN_BYTES = 8;

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

// The block as the lanes see it: the deflate data as aligned 32-bit words (the caller places the zlib stream so that byte 2 -- the first byte behind
// the zlib header -- is 4-byte aligned), the two tables, the number of bits that belong to the stream.
struct ILStream {
  const uint32_t* words;
  uint32_t nwords;     // words that may be read (the rest of the stream reads as 0)
  uint32_t nbits;      // bits of deflate data incl. the Adler-32 trailer: a token that ends beyond is an error
  const uint32_t* lit;   // 512 entries
  const uint32_t* dist;  // 32 entries
};

struct ILBits {
  uint64_t buf;
  uint32_t cnt, word, pos;
};
IL_HD uint32_t il_word(const ILStream& s, uint32_t w) { return w < s.nwords ? s.words[w] : 0u; }
IL_HD void il_bits_init(const ILStream& s, ILBits& b, uint32_t pos) {
  b.pos = pos;
  b.word = pos >> 5;
  const uint32_t sh = pos & 31u;
  b.buf = (uint64_t)(il_word(s, b.word) >> sh);
  b.cnt = 32u - sh;
  b.word++;
}
IL_HD void il_refill(const ILStream& s, ILBits& b) {   // afterwards cnt >= 33 > 31 = the longest token
  if (b.cnt <= 32u) {
    b.buf |= (uint64_t)il_word(s, b.word) << b.cnt;
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
// one token at the reader's position (consumed)
IL_HD ILToken il_token(const ILStream& s, ILBits& b) {
  il_refill(s, b);
  const uint32_t e = s.lit[(uint32_t)b.buf & 511u];
  const uint32_t nb = e & 15u, kind = (e >> 4) & 3u;
  ILToken t{kind, e >> 16, 0u};
  if (kind != 1u) { il_consume(b, nb); return t; }
  const uint32_t ex = (e >> 6) & 15u;
  t.value += ((uint32_t)(b.buf >> nb)) & ((1u << ex) - 1u);
  const uint32_t at = nb + ex;
  const uint32_t d = s.dist[((uint32_t)(b.buf >> at)) & 31u];
  if (d & 16u) { t.kind = 3u; il_consume(b, at + 5u); return t; }
  const uint32_t dex = d & 15u;
  t.dist = (d >> 16) + (((uint32_t)(b.buf >> (at + 5u))) & ((1u << dex) - 1u));
  il_consume(b, at + 5u + dex);
  return t;
}

// stage A: the tokens that START in [start, limit): where the first token behind them starts, how many bytes they produce.
// Nothing stops the scan before the limit: a lane that started at a wrong bit must reach the end of its chunk (and resynchronise on the way) even
// if it meets an invalid code or what looks like an end-of-block code -- the flags only count if this turns out to be the true token path, and
// then: bytes are counted up to the first end-of-block code, and invalid codes behind it (the Adler-32 trailer read as tokens) are no error.
IL_HD void il_scan_chunk(const ILStream& s, uint32_t start, uint32_t limit, uint32_t& end, uint32_t& out_bytes, uint32_t& flag) {
  ILBits b;
  il_bits_init(s, b, start);
  uint32_t out = 0, fl = IL_FLAG_OK;
  while (b.pos < limit) {
    const ILToken t = il_token(s, b);
    if (b.pos > s.nbits) {   // ran off the end of the stream
      if (!(fl & IL_FLAG_EOB)) fl |= IL_FLAG_ERR;
      break;
    }
    if (fl & IL_FLAG_EOB) continue;   // behind the end of the block: only the position matters
    if (t.kind == 3u) fl |= IL_FLAG_ERR;
    else if (t.kind == 2u) fl |= IL_FLAG_EOB;
    else out += t.kind == 0u ? 1u : t.value;
  }
  end = b.pos;
  out_bytes = out;
  flag = fl;
}

// how a block of `nbits` bits is cut: C chunks of B bits
IL_HD void il_geometry(uint32_t nbits, uint32_t& C, uint32_t& B) {
  uint32_t c = nbits / IL_MIN_CHUNK_BITS;
  if (c < 1u) c = 1u;
  if (c > (uint32_t)IL_MAX_CHUNKS) c = (uint32_t)IL_MAX_CHUNKS;
  C = c;
  B = (nbits + c - 1u) / c;
}

// stage C: one lane's state; turn() is called until done.  M is the memory the lanes share:
//   uint8_t  out_load(uint32_t i) / void out_store(uint32_t i, uint8_t v)     the output bytes
//   uint32_t off(uint32_t k)                                                  first output byte of chunk k (k <= C: off(C) = total)
//   uint32_t progress(uint32_t k) / void publish(uint32_t k, uint32_t o)      output offset up to which chunk k is written AND visible; publish()
//                                                                             is called by the owner once its earlier stores have completed
//   void fail(int32_t status)                                                 the frame is corrupt: every lane stops at its next turn
struct ILLane {
  ILBits bits;
  uint32_t c, limit;      // my chunk, first bit behind it
  uint32_t o, o_end;      // next output byte, end of my chunk's output
  uint32_t len, dist;     // match in progress (len > 0)
  uint32_t src_chunk;     // chunk the match's next source byte lies in (search cache)
  bool done;
};
template <class M>
IL_HD void il_lane_init(const ILStream& s, const M& m, ILLane& L, uint32_t c, uint32_t start, uint32_t limit) {
  L.c = c;
  L.limit = limit;
  L.o = m.off(c);
  L.o_end = m.off(c + 1u);
  L.len = 0;
  L.dist = 0;
  L.src_chunk = c;
  L.done = start == IL_NONE || L.o == L.o_end;
  if (!L.done) il_bits_init(s, L.bits, start);
}
template <class M>
IL_HD void il_lane_turn(const ILStream& s, M& m, ILLane& L) {
  if (L.done) return;
  // everything this lane stored in earlier turns has completed (the caller waits for its stores before the turn): tell the others
  m.publish(L.c, L.o);
  if (L.len == 0u) {
    if (L.o >= L.o_end) { L.done = true; return; }   // every token of the chunk written (the stage-A byte count says so)
    const ILToken t = il_token(s, L.bits);
    if (t.kind == 0u) {
      m.out_store(L.o, (uint8_t)t.value);
      L.o++;
      return;
    }
    if (t.kind != 1u) { m.fail(IL_ST_BAD_CODE); L.done = true; return; }   // stage A saw this chunk clean: cannot happen
    if (t.dist > L.o) { m.fail(IL_ST_BAD_DISTANCE); L.done = true; return; }
    L.len = t.value;
    L.dist = t.dist;
    L.src_chunk = L.c;
  }
  // up to IL_TURN_BYTES of the match: out[o + i] = out[o - dist + (i mod dist)], every source byte in front of o
  const uint32_t n = L.len < IL_TURN_BYTES ? L.len : IL_TURN_BYTES;
  const uint32_t src = L.o - L.dist;
  const uint32_t need_end = (L.dist < n ? L.o : src + n);   // one behind the last source byte read
  uint32_t k = L.src_chunk;
  while (m.off(k) > src) k--;
  L.src_chunk = k;
  const uint32_t k2 = (k < L.c && m.off(k + 1u) < need_end) ? k + 1u : k;
  bool ready = m.progress(k2) >= need_end;
  if (k2 != k) ready = ready && m.progress(k) >= m.off(k + 1u);
  if (!ready) return;   // the bytes are not there yet: try again next turn
  uint8_t v[IL_TURN_BYTES];
  uint32_t wrap = 0;
  for (uint32_t i = 0; i < IL_TURN_BYTES; i++) {
    if (i < n) v[i] = m.out_load(src + wrap);
    wrap++;
    if (wrap == L.dist) wrap = 0;
  }
  for (uint32_t i = 0; i < IL_TURN_BYTES; i++)
    if (i < n) m.out_store(L.o + i, v[i]);
  L.o += n;
  L.len -= n;
}

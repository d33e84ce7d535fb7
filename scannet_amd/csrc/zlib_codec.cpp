// zlib_codec.cpp -- from-scratch zlib (RFC 1950) / DEFLATE (RFC 1951) inflate and deflate for .sens depth blobs.
//
// Replaces, on the read side, stb::stbi_zlib_decode_malloc as called by RGBDFrame::decompressDepthAlloc_stb
// (SensReader/c++/src/sensorData.h:703-709 -> sensorData/stb_image.h:3846) and, on the write side,
// stb::stbi_zlib_compress as called by compressDepth (sensorData.h:659-670 -> stb_image_write.h:721-823).
// Behaviour kept from the reference reader: the zlib header is validated (CM = 8, FCHECK, no preset
// dictionary: stb_image.h:3765-3778), stored / fixed / dynamic blocks are all accepted (the reference writer
// only emits one fixed-Huffman block but real files may hold any valid stream), and the
// trailing Adler-32 is NOT verified (stb never reads it).  The writer emits a correct Adler-32.
//
// Decoder design: 64-bit bit reservoir refilled eight bytes at a time, 11-bit first-level lookup tables for
// literal/length and distance codes with a canonical-code walk for the rare longer codes, output straight
// into the caller's buffer (the frame size is known: W*H*2), wide overlapping-safe match copies.
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr int FAST_BITS = 11;

struct Huffman {
  uint16_t fast[1 << FAST_BITS];  // (symbol << 4) | length, 0 = not a short code
  uint16_t first_code[17];        // canonical first code of each length (left-aligned to 16 bits below)
  uint32_t limit[17];             // left-aligned (16-bit) exclusive upper bound of codes of each length
  uint16_t first_sym[17];
  uint16_t syms[320];
  int max_len;
};

inline uint32_t reverse_bits(uint32_t v, int n) {
  v = ((v & 0xAAAAu) >> 1) | ((v & 0x5555u) << 1);
  v = ((v & 0xCCCCu) >> 2) | ((v & 0x3333u) << 2);
  v = ((v & 0xF0F0u) >> 4) | ((v & 0x0F0Fu) << 4);
  v = ((v & 0xFF00u) >> 8) | ((v & 0x00FFu) << 8);
  return v >> (16 - n);
}

// Build decode tables from code lengths; returns false for an over-subscribed code.
bool build_huffman(Huffman& h, const uint8_t* lengths, int n) {
  int count[17] = {0};
  for (int i = 0; i < n; i++) count[lengths[i]]++;
  count[0] = 0;
  std::memset(h.fast, 0, sizeof(h.fast));
  int code = 0, sym = 0;
  uint16_t next_code[17];
  h.max_len = 0;
  for (int len = 1; len <= 16; len++) {
    if (len <= 15 && count[len]) h.max_len = len;
    next_code[len] = (uint16_t)code;
    h.first_code[len] = (uint16_t)code;
    h.first_sym[len] = (uint16_t)sym;
    code += len <= 15 ? count[len] : 0;
    if (len <= 15 && code > (1 << len)) return false;
    h.limit[len] = (uint32_t)code << (16 - len);
    code <<= 1;
    sym += len <= 15 ? count[len] : 0;
  }
  uint16_t offs[17];
  for (int len = 1; len <= 16; len++) offs[len] = h.first_sym[len];
  for (int i = 0; i < n; i++) {
    const int len = lengths[i];
    if (!len) continue;
    h.syms[offs[len]++] = (uint16_t)i;
    const uint32_t c = next_code[len]++;
    if (len <= FAST_BITS) {
      const uint32_t r = reverse_bits(c, len);
      const uint16_t entry = (uint16_t)((i << 4) | len);
      for (uint32_t k = r; k < (1u << FAST_BITS); k += (1u << len)) h.fast[k] = entry;
    }
  }
  return true;
}

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;
  int overrun = 0;  // zero bytes fed past the end of the input

  inline void refill() {
    if (end - p >= 8) {
      uint64_t w;
      std::memcpy(&w, p, 8);
      buf |= w << cnt;
      p += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56) {
        uint64_t b = 0;
        if (p < end) b = *p++;
        else overrun++;
        buf |= b << cnt;
        cnt += 8;
      }
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
  inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// decode one symbol; returns -1 on an invalid code.  Needs >= 15 valid bits in the reservoir.
inline int decode_sym(BitReader& br, const Huffman& h) {
  const uint16_t e = h.fast[br.peek(FAST_BITS)];
  if (e) {
    br.drop(e & 15);
    return e >> 4;
  }
  const uint32_t k = reverse_bits(br.peek(16), 16);
  for (int len = FAST_BITS + 1; len <= h.max_len; len++) {
    if (k < h.limit[len]) {
      const uint32_t idx = h.first_sym[len] + ((k >> (16 - len)) - h.first_code[len]);
      br.drop(len);
      return h.syms[idx];
    }
  }
  return -1;
}

const uint16_t LEN_BASE[31] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0};
const uint8_t LEN_EXTRA[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
const uint16_t DIST_BASE[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0};
const uint8_t DIST_EXTRA[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};

// Packed first-level tables: everything the inner loop needs about a symbol in ONE load (the generic tables above give a symbol number that
// has to be looked up again for its base value and extra-bit count -- two more dependent loads per match, and a deflated depth frame is ~100 k
// short matches).  Entry for the next FAST_BITS bits of the stream:
//   literal / length table   bits 0..3 code length | 4..6 extra bits of a length | 8..16 literal byte or base length | 29 end of block, 30 length, 31 literal
//   distance table           bits 0..3 code length | 4..7 extra bits | 8..23 base distance
// 0 = the code is longer than FAST_BITS (or not a valid symbol): the generic path decodes it.
constexpr uint32_t PK_LIT = 1u << 31, PK_LEN = 1u << 30, PK_EOB = 1u << 29;
struct PackedTables { uint32_t lit[1 << FAST_BITS], dist[1 << FAST_BITS]; };

void build_packed(PackedTables& pk, const Huffman& lit, const Huffman& dist) {
  for (uint32_t i = 0; i < (1u << FAST_BITS); i++) {
    const uint32_t e = lit.fast[i];
    uint32_t v = 0;
    if (e) {
      const uint32_t sym = e >> 4, len = e & 15;
      if (sym < 256) v = PK_LIT | (sym << 8) | len;
      else if (sym == 256) v = PK_EOB | len;
      else if (sym - 257 < 29) v = PK_LEN | ((uint32_t)LEN_BASE[sym - 257] << 8) | ((uint32_t)LEN_EXTRA[sym - 257] << 4) | len;
    }
    pk.lit[i] = v;
    const uint32_t d = dist.fast[i];
    uint32_t w = 0;
    if (d && (d >> 4) < 30) w = ((uint32_t)DIST_BASE[d >> 4] << 8) | ((uint32_t)DIST_EXTRA[d >> 4] << 4) | (d & 15);
    pk.dist[i] = w;
  }
}

struct FixedTables {
  Huffman lit, dist;
  PackedTables pk;
  FixedTables() {
    uint8_t l[288];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    build_huffman(lit, l, 288);
    uint8_t d[32];
    for (int i = 0; i < 32; i++) d[i] = 5;
    build_huffman(dist, d, 32);
    build_packed(pk, lit, dist);
  }
};

const FixedTables& fixed_tables() {
  static const FixedTables t;
  return t;
}

// the copy of a match (shared by the packed and the generic path); pos + len <= cap has been checked
static inline void copy_match(uint8_t* out, uint64_t cap, uint64_t& pos, uint32_t len, uint32_t d) {
  uint8_t* dst = out + pos;
  const uint8_t* src = dst - d;
  pos += len;
  if (d >= 8 && pos + 8 <= cap) {
    // 8-byte chunks; may write up to 7 bytes past `len`, which the cap check allows for
    for (uint32_t i = 0; i < len; i += 8) std::memcpy(dst + i, src + i, 8);
  } else if (d == 1) {
    std::memset(dst, src[0], len);
  } else if ((d == 2 || d == 4) && pos + 8 <= cap) {
    // the pixel to the left / two to the left (u16 data): the period divides 8, so one 8-byte pattern serves every chunk
    uint64_t pat;
    if (d == 2) { uint16_t h; std::memcpy(&h, src, 2); pat = 0x0001000100010001ull * h; }
    else { uint32_t w; std::memcpy(&w, src, 4); pat = 0x0000000100000001ull * w; }
    for (uint32_t i = 0; i < len; i += 8) std::memcpy(dst + i, &pat, 8);
  } else {
    for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
  }
}

int inflate_block(BitReader& br, const Huffman& lit, const Huffman& dist, const PackedTables& pk, uint8_t* out, uint64_t cap, uint64_t& pos) {
  constexpr uint32_t FMASK = (1u << FAST_BITS) - 1;
  for (;;) {
    // ---- the tight loop: a deflated depth frame is ~110 k matches of ~5 bytes and ~30 k literals.  While the output has room for the longest
    // match plus the chunked copy's overshoot and the input has 8 bytes left (the refill is then one unaligned load), a symbol costs two table
    // loads and a copy -- no per-symbol bounds checks, no tail handling.  Anything else (end of block, a code longer than FAST_BITS, the last
    // bytes of either buffer) leaves the loop for the careful body below, which handles one symbol and comes back.
    while (pos + 274 <= cap && br.end - br.p >= 8) {
      {
        uint64_t w;
        std::memcpy(&w, br.p, 8);
        br.buf |= w << br.cnt;
        br.p += (63 - br.cnt) >> 3;
        br.cnt |= 56;
      }
      uint32_t e = pk.lit[br.buf & FMASK];
      if (e & PK_LIT) {                       // one or two literals (<= 22 bits), then round again -- or a match right behind ONE literal (11 + 40 bits)
        out[pos++] = (uint8_t)(e >> 8);
        br.buf >>= (e & 15);
        br.cnt -= (int)(e & 15);
        e = pk.lit[br.buf & FMASK];
        if (e & PK_LIT) {
          out[pos++] = (uint8_t)(e >> 8);
          br.buf >>= (e & 15);
          br.cnt -= (int)(e & 15);
          continue;
        }
      }
      if (!(e & PK_LEN)) break;
      const uint32_t lb = e & 15, lx = (e >> 4) & 7;
      const uint32_t dd = pk.dist[(br.buf >> (lb + lx)) & FMASK];
      if (!dd) break;
      br.buf >>= lb;
      const uint32_t len = ((e >> 8) & 0x1FF) + (uint32_t)(br.buf & ((1u << lx) - 1));
      const uint32_t db = dd & 15, dx = (dd >> 4) & 15;
      br.buf >>= lx + db;
      const uint32_t d = (dd >> 8) + (uint32_t)(br.buf & ((1u << dx) - 1));
      br.buf >>= dx;
      br.cnt -= (int)(lb + lx + db + dx);
      if (d > pos) return sf::fail(SF_ERR_FORMAT, "inflate: distance %u before start of output", d);
      uint8_t* dst = out + pos;
      const uint8_t* src = dst - d;
      pos += len;
      if (d >= 8) {
        std::memcpy(dst, src, 8);
        for (uint32_t i = 8; i < len; i += 8) std::memcpy(dst + i, src + i, 8);
      } else if (d == 1) {
        std::memset(dst, src[0], len);
      } else if (d == 2 || d == 4) {
        uint64_t pat;
        if (d == 2) { uint16_t h; std::memcpy(&h, src, 2); pat = 0x0001000100010001ull * h; }
        else { uint32_t w; std::memcpy(&w, src, 4); pat = 0x0000000100000001ull * w; }
        for (uint32_t i = 0; i < len; i += 8) std::memcpy(dst + i, &pat, 8);
      } else {
        for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
      }
    }
    br.refill();
    // ---- the packed path: with 8 bytes of room left, literals run without bounds checks (five codes of <= FAST_BITS = 11 bits fit the >= 56 bits
    // just loaded), and a length / distance pair whose two codes are short is decoded from two table loads: 11 + 5 + 11 + 13 = 40 bits
    if (pos + 8 <= cap) {
      uint32_t e = pk.lit[br.buf & FMASK];
      int run = 0;
      while ((e & PK_LIT) && run < 5) {
        out[pos++] = (uint8_t)(e >> 8);
        br.buf >>= (e & 15);
        br.cnt -= (int)(e & 15);
        run++;
        e = pk.lit[br.buf & FMASK];
      }
      if (run == 5) {                          // the reservoir may be down to one bit: top up first
        if (br.overrun > 8) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stream");
        continue;
      }
      if (run && br.cnt < 48) br.refill();     // the low bits of the reservoir, hence `e`, stay what they were
      if (e & PK_LEN) {
        const uint32_t dd = pk.dist[(br.buf >> ((e & 15) + ((e >> 4) & 7))) & FMASK];
        if (dd) {
          const uint32_t lb = e & 15, lx = (e >> 4) & 7;
          br.buf >>= lb;
          const uint32_t len = ((e >> 8) & 0x1FF) + (uint32_t)(br.buf & ((1u << lx) - 1));
          const uint32_t db = dd & 15, dx = (dd >> 4) & 15;
          br.buf >>= lx + db;
          const uint32_t d = (dd >> 8) + (uint32_t)(br.buf & ((1u << dx) - 1));
          br.buf >>= dx;
          br.cnt -= (int)(lb + lx + db + dx);
          if (d > pos) return sf::fail(SF_ERR_FORMAT, "inflate: distance %u before start of output", d);
          if (pos + len > cap) return sf::fail(SF_ERR_BOUNDS, "inflate: output exceeds %llu bytes", (unsigned long long)cap);
          copy_match(out, cap, pos, len, d);
          if (br.overrun > 8) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stream");
          continue;
        }
      }
    }
    int sym = decode_sym(br, lit);
    if (sym < 0) return sf::fail(SF_ERR_FORMAT, "inflate: bad literal/length code");
    if (sym < 256) {
      if (pos >= cap) return sf::fail(SF_ERR_BOUNDS, "inflate: output exceeds %llu bytes", (unsigned long long)cap);
      out[pos++] = (uint8_t)sym;
      // a second symbol without refilling: the reservoir still holds >= 41 bits, a code takes at most 15
      sym = decode_sym(br, lit);
      if (sym < 0) return sf::fail(SF_ERR_FORMAT, "inflate: bad literal/length code");
      if (sym < 256) {
        if (pos >= cap) return sf::fail(SF_ERR_BOUNDS, "inflate: output exceeds %llu bytes", (unsigned long long)cap);
        out[pos++] = (uint8_t)sym;
        continue;
      }
    }
    if (sym == 256) {
      if (br.overrun > 8) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stream");
      return SF_OK;
    }
    sym -= 257;
    if (sym >= 29) return sf::fail(SF_ERR_FORMAT, "inflate: bad length symbol");
    uint32_t len = LEN_BASE[sym] + br.take(LEN_EXTRA[sym]);
    // worst case so far 15 + 15 + 5 of the >= 56 bits; a distance needs up to 15 + 13 more (dynamic codes of maximum length)
    if (br.cnt < 28) br.refill();
    const int ds = decode_sym(br, dist);
    if (ds < 0 || ds >= 30) return sf::fail(SF_ERR_FORMAT, "inflate: bad distance code");
    const uint32_t d = DIST_BASE[ds] + br.take(DIST_EXTRA[ds]);
    if (d > pos) return sf::fail(SF_ERR_FORMAT, "inflate: distance %u before start of output", d);
    if (pos + len > cap) return sf::fail(SF_ERR_BOUNDS, "inflate: output exceeds %llu bytes", (unsigned long long)cap);
    copy_match(out, cap, pos, len, d);
    if (br.overrun > 8) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stream");
  }
}

int inflate_raw(const uint8_t* src, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  BitReader br{src, src + n};
  uint64_t pos = 0;
  static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  for (;;) {
    br.refill();
    const uint32_t final = br.take(1);
    const uint32_t type = br.take(2);
    if (type == 0) {
      br.drop(br.cnt & 7);  // to byte boundary
      br.refill();
      const uint32_t len = br.take(16), nlen = br.take(16);
      if ((len ^ 0xFFFFu) != nlen) return sf::fail(SF_ERR_FORMAT, "inflate: stored block length check failed");
      // un-read the whole bytes still sitting in the reservoir
      const uint8_t* q = br.p - (br.cnt >> 3) + br.overrun;
      if (br.overrun > 0 || q > br.end || (uint64_t)(br.end - q) < len) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stored block");
      if (pos + len > cap) return sf::fail(SF_ERR_BOUNDS, "inflate: output exceeds %llu bytes", (unsigned long long)cap);
      std::memcpy(out + pos, q, len);
      pos += len;
      br.p = q + len;
      br.buf = 0;
      br.cnt = 0;
    } else if (type == 1) {
      const FixedTables& ft = fixed_tables();
      const int rc = inflate_block(br, ft.lit, ft.dist, ft.pk, out, cap, pos);
      if (rc != SF_OK) return rc;
    } else if (type == 2) {
      br.refill();
      const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
      uint8_t cl[19] = {0};
      for (int i = 0; i < hclen; i++) {
        if (br.cnt < 3) br.refill();
        cl[CL_ORDER[i]] = (uint8_t)br.take(3);
      }
      Huffman clh;
      if (!build_huffman(clh, cl, 19)) return sf::fail(SF_ERR_FORMAT, "inflate: bad code-length code");
      uint8_t lens[320 + 16];
      int i = 0;
      while (i < hlit + hdist) {
        br.refill();
        const int s = decode_sym(br, clh);
        if (s < 0) return sf::fail(SF_ERR_FORMAT, "inflate: bad code-length symbol");
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
          int rep;
          uint8_t fill = 0;
          if (s == 16) {
            if (i == 0) return sf::fail(SF_ERR_FORMAT, "inflate: repeat with no previous length");
            rep = 3 + (int)br.take(2);
            fill = lens[i - 1];
          } else if (s == 17) rep = 3 + (int)br.take(3);
          else rep = 11 + (int)br.take(7);
          if (i + rep > hlit + hdist) return sf::fail(SF_ERR_FORMAT, "inflate: code lengths overflow");
          std::memset(lens + i, fill, (size_t)rep);
          i += rep;
        }
      }
      static thread_local Huffman lit, dist;
      static thread_local PackedTables pk;
      if (!build_huffman(lit, lens, hlit) || !build_huffman(dist, lens + hlit, hdist)) return sf::fail(SF_ERR_FORMAT, "inflate: over-subscribed Huffman code");
      build_packed(pk, lit, dist);
      const int rc = inflate_block(br, lit, dist, pk, out, cap, pos);
      if (rc != SF_OK) return rc;
    } else {
      return sf::fail(SF_ERR_FORMAT, "inflate: reserved block type");
    }
    if (br.overrun > 8) return sf::fail(SF_ERR_FORMAT, "inflate: truncated stream");
    if (final) break;
  }
  *out_len = pos;
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------ deflate
struct BitWriter {
  uint8_t* out;
  uint64_t cap, pos = 0;
  uint64_t buf = 0;
  int cnt = 0;
  bool overflow = false;
  inline void put(uint32_t v, int n) {
    buf |= (uint64_t)v << cnt;
    cnt += n;
    while (cnt >= 8) {
      if (pos < cap) out[pos++] = (uint8_t)buf;
      else overflow = true;
      buf >>= 8;
      cnt -= 8;
    }
  }
  inline void flush() {
    if (cnt > 0) put(0, 8 - cnt);
  }
};

struct FixedEnc {
  uint16_t lit_code[288];
  uint8_t lit_len[288];
  uint8_t len_sym[259];  // match length -> symbol - 257
  uint8_t dist_sym_lo[512];
  FixedEnc() {
    for (int i = 0; i < 288; i++) {
      int len, code;
      if (i < 144) { len = 8; code = 0x30 + i; }
      else if (i < 256) { len = 9; code = 0x190 + (i - 144); }
      else if (i < 280) { len = 7; code = i - 256; }
      else { len = 8; code = 0xC0 + (i - 280); }
      lit_len[i] = (uint8_t)len;
      lit_code[i] = (uint16_t)reverse_bits((uint32_t)code, len);
    }
    for (int l = 3; l <= 258; l++) {
      int s = 28;
      while (LEN_BASE[s] > l) s--;
      if (l == 258) s = 28;
      else if (s == 28) s = 27;
      len_sym[l] = (uint8_t)s;
    }
    for (int d = 1; d <= 512; d++) {
      int s = 29;
      while (DIST_BASE[s] > d) s--;
      dist_sym_lo[d - 1] = (uint8_t)s;
    }
  }
  inline int dist_sym(uint32_t d) const {
    if (d <= 512) return dist_sym_lo[d - 1];
    int s = 29;
    while (DIST_BASE[s] > d) s--;
    return s;
  }
};
const FixedEnc& fixed_enc() {
  static const FixedEnc e;
  return e;
}

uint32_t adler32(const uint8_t* p, uint64_t n) {
  uint32_t a = 1, b = 0;
  while (n) {
    const uint64_t k = n < 5552 ? n : 5552;
    for (uint64_t i = 0; i < k; i++) { a += p[i]; b += a; }
    a %= 65521u; b %= 65521u;
    p += k; n -= k;
  }
  return (b << 16) | a;
}

inline uint32_t hash3(const uint8_t* p) {
  const uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  return (v * 2654435761u) >> 17;  // 15 bits
}

}  // namespace

// dst_cap should be >= sf_zlib_deflate_bound(n)
SF_API uint64_t sf_zlib_deflate_bound(uint64_t n) { return n + (n >> 3) + 64; }

SF_API int sf_zlib_deflate(const void* src_, uint64_t n, void* dst, uint64_t dst_cap, uint64_t* out_len) {
  if ((!src_ && n) || !dst || !out_len) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint8_t* src = (const uint8_t*)src_;
  const FixedEnc& fe = fixed_enc();
  BitWriter bw{(uint8_t*)dst, dst_cap};
  bw.put(0x78, 8);
  bw.put(0x5E, 8);
  bw.put(1, 1);  // BFINAL
  bw.put(1, 2);  // fixed Huffman
  constexpr int HSIZE = 1 << 15, WINDOW = 32768, MAX_CHAIN = 16, MIN_MATCH = 3, MAX_MATCH = 258;
  std::vector<int32_t> head(HSIZE, -1), prev(WINDOW, -1);
  uint64_t i = 0;
  auto emit_lit = [&](uint8_t c) { bw.put(fe.lit_code[c], fe.lit_len[c]); };
  while (i < n) {
    int best_len = 0;
    uint32_t best_dist = 0;
    if (i + MIN_MATCH <= n) {
      const uint32_t h = hash3(src + i);
      int32_t cand = head[h];
      const uint64_t max_len = (n - i) < (uint64_t)MAX_MATCH ? (n - i) : (uint64_t)MAX_MATCH;
      for (int chain = 0; cand >= 0 && chain < MAX_CHAIN; chain++) {
        const uint64_t dist = i - (uint64_t)cand;
        if (dist > (uint64_t)WINDOW - 1) break;
        const uint8_t* a = src + cand;
        const uint8_t* b = src + i;
        if (a[best_len] == b[best_len] || best_len == 0) {
          uint64_t l = 0;
          while (l < max_len && a[l] == b[l]) l++;
          if ((int)l > best_len) {
            best_len = (int)l;
            best_dist = (uint32_t)dist;
            if (l == max_len) break;
          }
        }
        cand = prev[cand & (WINDOW - 1)];
      }
      prev[i & (WINDOW - 1)] = head[h];
      head[h] = (int32_t)i;
    }
    if (best_len >= MIN_MATCH) {
      const int ls = fe.len_sym[best_len];
      bw.put(fe.lit_code[257 + ls], fe.lit_len[257 + ls]);
      if (LEN_EXTRA[ls]) bw.put((uint32_t)(best_len - LEN_BASE[ls]), LEN_EXTRA[ls]);
      const int ds = fe.dist_sym(best_dist);
      bw.put(reverse_bits((uint32_t)ds, 5), 5);
      if (DIST_EXTRA[ds]) bw.put(best_dist - DIST_BASE[ds], DIST_EXTRA[ds]);
      // index the skipped positions so later matches can reach them -- all of them for ordinary matches; inside a long one (a run: a
      // constant image region, the zeros of an Up-filtered PNG row) only its first and last 16, which is where the next match starts
      const uint64_t stop = i + (uint64_t)best_len;
      for (uint64_t j = i + 1; j < stop && j + MIN_MATCH <= n; j++) {
        if (best_len >= 64 && j == i + 17) { j = stop - 17; continue; }
        const uint32_t h = hash3(src + j);
        prev[j & (WINDOW - 1)] = head[h];
        head[h] = (int32_t)j;
      }
      i = stop;
    } else {
      emit_lit(src[i]);
      i++;
    }
  }
  bw.put(fe.lit_code[256], fe.lit_len[256]);
  bw.flush();
  const uint32_t ad = adler32(src, n);
  bw.put(ad >> 24, 8); bw.put((ad >> 16) & 0xFF, 8); bw.put((ad >> 8) & 0xFF, 8); bw.put(ad & 0xFF, 8);
  if (bw.overflow) return sf::fail(SF_ERR_BOUNDS, "deflate: output buffer too small (%llu bytes)", (unsigned long long)dst_cap);
  *out_len = bw.pos;
  return SF_OK;
}

SF_API int sf_zlib_inflate(const void* src_, uint64_t n, void* dst, uint64_t dst_cap, uint64_t* out_len) {
  if (!src_ || (!dst && dst_cap) || !out_len) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint8_t* src = (const uint8_t*)src_;
  if (n < 2) return sf::fail(SF_ERR_FORMAT, "zlib: stream shorter than its header");
  const uint32_t cmf = src[0], flg = src[1];
  if ((cmf * 256 + flg) % 31 != 0) return sf::fail(SF_ERR_FORMAT, "zlib: bad header check");
  if (flg & 32) return sf::fail(SF_ERR_FORMAT, "zlib: preset dictionary not allowed");
  if ((cmf & 15) != 8) return sf::fail(SF_ERR_FORMAT, "zlib: compression method is not DEFLATE");
  return inflate_raw(src + 2, n - 2, (uint8_t*)dst, dst_cap, out_len);
}

// inflate_lanes.h (study) -- the REJECTED third stage of a chunk-parallel inflate: every lane writes its chunk in token order and a match waits
// until its source bytes are there (README.md: 30 000 - 95 000 turns per frame, the lanes serialise).  The tokeniser, the tables and stages A / B
// are the product's (scannet_amd/csrc/inflate_lanes.h, included below); what the product does instead of this stage is there too: per-byte back
// references written by the lanes, then ONE wave per frame walking the output in 256-byte groups with the window in LDS.
//   stage C  every lane decodes its chunk once more and writes: literals at once, a match as soon as its source bytes have been written -- by the
//            lane itself or by the lanes of earlier chunks, whose progress is published per chunk.  Dependencies point strictly backwards in the
//            output, so the lane with the earliest unfinished byte can always proceed: no deadlock.  At most 8 bytes per lane and turn, so that the
//            lanes of a wave stay in step.
// A chunk covers >= 2400 bits, i.e. >= 263 output bytes (9 bits per byte at worst, a token may start up to 31 bits late), more than the longest match
// (258): the source bytes of one turn straddle at most two chunks.
#pragma once
#include "../../scannet_amd/csrc/inflate_lanes.h"

constexpr uint32_t IL_TURN_BYTES = 8;

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

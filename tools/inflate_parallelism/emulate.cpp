// inflate_emu.cpp -- the chunk-parallel inflate of inflate_lanes.h run on the HOST, lane by lane in lock step.  Test infrastructure for the algorithm
// (scanfuse_internal.h: sf_inflate_lanes_emulate): the CPU suite runs it against zlib and the reference's stb inflater on the streams the writers
// produce, on corrupted streams and on the adversarial cases (long runs, far matches, tiny frames); the device kernel (inflate_gpu.hip) executes the
// same lane programs, so what is left to the GPU tests is the memory model.  The product path never calls this.
#include <cstring>
#include <vector>

#include "common.h"
#include "inflate_lanes.h"

namespace {

struct HostMem {
  uint8_t* out;
  uint32_t* off_;
  uint32_t* prog;
  int32_t* status;
  uint8_t out_load(uint32_t i) const { return out[i]; }
  void out_store(uint32_t i, uint8_t v) { out[i] = v; }
  uint32_t off(uint32_t k) const { return off_[k]; }
  uint32_t progress(uint32_t k) const { return prog[k]; }
  void publish(uint32_t k, uint32_t o) { prog[k] = o; }
  void fail(int32_t st) { if (*status == IL_ST_OK) *status = st; }
};

}  // namespace

// stats_out (nullable, 4 words): chunks, stage-A rounds, stage-A chunk scans, stage-C turns
SF_API int sf_inflate_lanes_emulate(const void* src_, uint64_t n, void* dst, uint64_t dst_cap, uint64_t* out_len, uint32_t* stats_out) {
  if (!src_ || !dst || !out_len) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint8_t* src = (const uint8_t*)src_;
  *out_len = 0;
  if (n < 8 || (src[0] & 0x0F) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) return sf::fail(SF_ERR_FORMAT, "not a zlib stream");
  if ((src[2] & 7) != 3) return sf::fail(SF_ERR_UNSUPPORTED, "not one final fixed-Huffman block (the host inflater's business)");
  if (n - 2 > (1ull << 28)) return sf::fail(SF_ERR_UNSUPPORTED, "stream too long for 32-bit bit positions");
  const uint32_t nbytes = (uint32_t)(n - 2);
  std::vector<uint32_t> words((nbytes + 3) / 4 + 2, 0u);
  std::memcpy(words.data(), src + 2, nbytes);
  std::vector<uint32_t> lit(512), dist(32);
  for (uint32_t i = 0; i < 512; i++) lit[i] = il_lit_entry(i);
  for (uint32_t i = 0; i < 32; i++) dist[i] = il_dist_entry(i);
  ILStream s{words.data(), (uint32_t)words.size(), nbytes * 8u, lit.data(), dist.data()};
  uint32_t C, B;
  il_geometry(s.nbits, C, B);
  std::vector<uint32_t> start(C), end(C, 0), outb(C, 0), flag(C, IL_FLAG_OK);
  std::vector<char> dirty(C, 1);
  for (uint32_t c = 0; c < C; c++) start[c] = c == 0 ? 3u : c * B;
  uint32_t rounds = 0, scans = 0;
  for (;;) {
    rounds++;
    for (uint32_t c = 0; c < C; c++)
      if (dirty[c]) {
        if (start[c] == IL_NONE) { end[c] = IL_NONE; outb[c] = 0; flag[c] = IL_FLAG_EOB; }
        else { il_scan_chunk(s, start[c], (c + 1 == C) ? s.nbits : (c + 1) * B, end[c], outb[c], flag[c]); scans++; }
      }
    bool any = false;
    for (uint32_t c = 0; c < C; c++) {
      const uint32_t ns = c == 0 ? 3u : (((flag[c - 1] & IL_FLAG_EOB) || end[c - 1] == IL_NONE) ? IL_NONE : end[c - 1]);
      dirty[c] = ns != start[c];
      start[c] = ns;
      any = any || dirty[c];
    }
    if (!any) break;
    if (rounds > C + 2) return sf::fail(SF_ERR_FORMAT, "chunk starts did not converge");   // cannot happen (induction over the chunks)
  }
  // the true token path is known: validate it
  int32_t status = IL_ST_OK;
  bool eob = false;
  for (uint32_t c = 0; c < C && status == IL_ST_OK; c++) {
    if (start[c] == IL_NONE) continue;
    if (flag[c] & IL_FLAG_ERR) status = IL_ST_BAD_CODE;
    else if (flag[c] & IL_FLAG_EOB) eob = true;
  }
  if (status == IL_ST_OK && !eob) status = IL_ST_NO_EOB;
  std::vector<uint32_t> off(C + 1, 0), prog(C + 1, 0);
  for (uint32_t c = 0; c < C; c++) off[c + 1] = off[c] + outb[c];
  if (status == IL_ST_OK && (uint64_t)off[C] > dst_cap) status = IL_ST_SIZE;
  uint32_t turns = 0;
  if (status == IL_ST_OK) {
    for (uint32_t c = 0; c <= C; c++) prog[c] = off[c];
    HostMem m{(uint8_t*)dst, off.data(), prog.data(), &status};
    std::vector<ILLane> lanes(C);
    for (uint32_t c = 0; c < C; c++) il_lane_init(s, m, lanes[c], c, start[c], (c + 1 == C) ? s.nbits : (c + 1) * B);
    for (;;) {
      bool all = true;
      // lock step: every lane sees the memory as the previous turn left it (the loads of a turn happen before its stores on the device too)
      for (uint32_t c = 0; c < C; c++) {
        il_lane_turn(s, m, lanes[c]);
        all = all && lanes[c].done;
      }
      turns++;
      if (all || status != IL_ST_OK) break;
      if (turns > 4u * (off[C] + 64u)) return sf::fail(SF_ERR_FORMAT, "stage C made no progress (deadlock)");   // cannot happen
    }
  }
  if (stats_out) { stats_out[0] = C; stats_out[1] = rounds; stats_out[2] = scans; stats_out[3] = turns; }
  if (status != IL_ST_OK) return sf::fail(SF_ERR_FORMAT, "chunk-parallel inflate: status %d", status);
  *out_len = off[C];
  return SF_OK;
}

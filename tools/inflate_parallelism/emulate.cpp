// emulate.cpp -- the chunk-parallel inflate of inflate_lanes.h run on the HOST, lane by lane in lock step: how many stage-A rounds the chunk starts
// need to settle, and how many turns stage C takes when 1 024 lanes write their chunks concurrently and a match has to wait for its source bytes.
// A STUDY (tools/inflate_parallelism/README.md), not product code: the output is checked against the input's zlib inflate by run.py.
//   emulate <file.z> <expected bytes> [out.bin]
#include <cstdlib>
#include <cstring>
#include <vector>

#include <cstdio>

#include "first_study_lanes.h"

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

static int fail(const char* what) { std::fprintf(stderr, "emulate: %s\n", what); return 1; }

static int emulate(const void* src_, uint64_t n, void* dst, uint64_t dst_cap, uint64_t* out_len, uint32_t* stats_out) {
  const uint8_t* src = (const uint8_t*)src_;
  *out_len = 0;
  if (n < 8 || (src[0] & 0x0F) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) return fail("not a zlib stream");
  if ((src[2] & 7) != 3) return fail("not one final fixed-Huffman block");
  if (n - 2 > (1ull << 28)) return fail("stream too long");
  const uint32_t nbytes = (uint32_t)(n - 2);
  std::vector<uint32_t> words((nbytes + 3) / 4 + 2, 0u);
  std::memcpy(words.data(), src + 2, nbytes);
  ILStream s{words.data(), (uint32_t)words.size(), nbytes * 8u};
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
        else { uint32_t first_; il_scan_chunk(s, start[c], start[c], (c + 1 == C) ? s.nbits : (c + 1) * B, first_, end[c], outb[c], flag[c]); scans++; }
      }
    bool any = false;
    for (uint32_t c = 0; c < C; c++) {
      const uint32_t ns = c == 0 ? 3u : end[c - 1];
      dirty[c] = ns != start[c];
      start[c] = ns;
      any = any || dirty[c];
    }
    if (std::getenv("IL_TRACE")) { unsigned nd = 0; for (uint32_t c = 0; c < C; c++) nd += dirty[c]; std::fprintf(stderr, "round %u: %u chunks restart\n", rounds, nd); }
    if (!any) break;
    if (rounds > C + 2) return fail("chunk starts did not converge");   // cannot happen (induction over the chunks)
  }
  // the true token path is known: validate it
  int32_t status = IL_ST_OK;
  bool eob = false;
  for (uint32_t c = 0; c < C && status == IL_ST_OK; c++) {
    if (eob) { start[c] = IL_NONE; outb[c] = 0; continue; }   // behind the end of the block: the trailer, not tokens
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
      if (turns > 4u * (off[C] + 64u)) return fail("stage C made no progress (deadlock)");   // cannot happen
    }
  }
  if (stats_out) { stats_out[0] = C; stats_out[1] = rounds; stats_out[2] = scans; stats_out[3] = turns; }
  if (status != IL_ST_OK) { std::fprintf(stderr, "emulate: status %d\n", status); return 1; }
  *out_len = off[C];
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return fail("usage: emulate <file.z> <expected bytes> [out.bin]");
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return fail("cannot open input");
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> src((size_t)n);
  if (std::fread(src.data(), 1, (size_t)n, f) != (size_t)n) return fail("short read");
  std::fclose(f);
  const uint64_t want = std::strtoull(argv[2], nullptr, 10);
  std::vector<uint8_t> out(want + 64);
  uint64_t got = 0;
  uint32_t st[4] = {0, 0, 0, 0};
  if (emulate(src.data(), (uint64_t)n, out.data(), want, &got, st)) return 1;
  std::printf("chunks %u | stage A: %u rounds, %.2f scans per chunk | stage C (lanes write in token order, wait for their source bytes): %u turns | %llu bytes\n", st[0], st[1],
              (double)st[2] / st[0], st[3], (unsigned long long)got);
  if (argc > 3) { FILE* o = std::fopen(argv[3], "wb"); if (o) { std::fwrite(out.data(), 1, got, o); std::fclose(o); } }
  return got == want ? 0 : 1;
}

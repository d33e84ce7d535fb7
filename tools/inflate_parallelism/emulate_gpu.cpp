// emulate_gpu.cpp -- the two kernels of scannet_amd/csrc/inflate_gpu.hip run on the HOST with the lane programs of csrc/inflate_lanes.h: 1024 lanes
// in lock step for the token stage (chunk starts to their fixed point, prefix sums, per-byte back references), 256 lanes in lock step for the copy
// stage (groups of 1024 bytes, pointer jumping inside a group, the window as a 32 KiB ring).  The CPU tests run it against zlib
// (tests/test_inflate_lanes.py); the GPU runs the same header.
//   emulate_gpu <file.z> <expected bytes> [out.bin]      exit code 0 + statistics, 2 = not a stream the device takes, 3 = corrupt (status printed)
//   g++ -O2 -std=c++17 -I scannet_amd/csrc tools/inflate_parallelism/emulate_gpu.cpp -o emulate_gpu
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../scannet_amd/csrc/inflate_lanes.h"

namespace {

struct HostWave {
  uint8_t* ring_;
  uint16_t* gref_;
  uint8_t* gval_;
  uint8_t ring(uint32_t i) const { return ring_[i & (IL_WINDOW - 1u)]; }
  uint16_t& gref(uint32_t j) const { return gref_[j]; }
  uint8_t& gval(uint32_t j) const { return gval_[j]; }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: emulate_gpu <file.z> <expected bytes> [out.bin]\n"); return 1; }
  FILE* fi = std::fopen(argv[1], "rb");
  if (!fi) return 1;
  std::fseek(fi, 0, SEEK_END);
  const long n = std::ftell(fi);
  std::fseek(fi, 0, SEEK_SET);
  std::vector<uint8_t> src((size_t)n);
  if (std::fread(src.data(), 1, (size_t)n, fi) != (size_t)n) return 1;
  std::fclose(fi);
  const uint32_t expect = (uint32_t)std::strtoul(argv[2], nullptr, 10);
  if (n < 8 || (src[0] & 0x0F) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20) || (src[2] & 7) != 3 || (expect & 3u)) { std::printf("not taken\n"); return 2; }
  const uint32_t nbytes = (uint32_t)(n - 2);
  std::vector<uint32_t> words((nbytes + 3) / 4 + 2, 0u);
  std::memcpy(words.data(), src.data() + 2, nbytes);
  ILStream S{words.data(), (nbytes + 3u) / 4u, nbytes * 8u};
  uint32_t C, B;
  il_geometry(S.nbits, C, B);
  // ---- k_inflate_tokens
  std::vector<uint32_t> start(C), end(C, 0), outb(C, 0), flag(C, IL_FLAG_OK);
  std::vector<char> dirty(C, 1);
  uint32_t rounds = 0, scans = 0, rescans_round1 = 0;
  for (;;) {
    rounds++;
    for (uint32_t c = 0; c < C; c++)
      if (dirty[c]) {
        const uint32_t limit = (c + 1 == C) ? S.nbits : (c + 1) * B;
        if (rounds == 1) il_scan_chunk(S, il_guess_start(c, B), c == 0 ? 3u : c * B, limit, start[c], end[c], outb[c], flag[c]);   // start[c]: where the chunk's own tokens begin on the guessed path
        else { uint32_t first; il_scan_chunk(S, start[c], start[c], limit, first, end[c], outb[c], flag[c]); }
        scans++;
      }
    bool any = false;
    for (uint32_t c = C; c-- > 1;) { dirty[c] = end[c - 1] != start[c]; start[c] = end[c - 1]; any = any || dirty[c]; }
    dirty[0] = 0;
    if (rounds == 1) for (uint32_t c = 0; c < C; c++) rescans_round1 += dirty[c];
    if (!any || rounds > C + 2) break;
  }
  int32_t status = IL_ST_OK;
  uint32_t n_eob = 0, total = 0;
  std::vector<uint32_t> off(C + 1, 0);
  std::vector<char> live(C, 0);
  bool err = false;
  for (uint32_t c = 0; c < C; c++) {
    live[c] = n_eob == 0;
    if (live[c] && (flag[c] & IL_FLAG_ERR)) err = true;
    off[c] = total;
    if (live[c]) total += outb[c];
    if (flag[c] & IL_FLAG_EOB) n_eob++;
  }
  if (err) status = IL_ST_BAD_CODE;
  else if (n_eob == 0) status = IL_ST_NO_EOB;
  else if (total != expect) status = IL_ST_SIZE;
  std::vector<uint8_t> out((size_t)expect + IL_GROUP, 0xEE);
  std::vector<uint16_t> plan((size_t)expect + IL_GROUP, 0xEEEE);
  struct Sink { uint16_t* p; void put(uint16_t v) { *p++ = v; } void put_run(uint16_t v, uint32_t n) { while (n--) *p++ = v; } };
  for (uint32_t c = 0; c < C && status == IL_ST_OK; c++)
    if (live[c] && outb[c]) { Sink P{plan.data() + off[c]}; status = il_write_chunk(S, start[c], off[c], off[c] + outb[c], P); }
  // ---- k_inflate_copy
  uint32_t steps = 0, jump_rounds = 0, max_rounds = 0;
  bool bad = false;
  if (status == IL_ST_OK) {
    std::vector<uint8_t> ring(IL_WINDOW, 0), gval(IL_GROUP);
    std::vector<uint16_t> gref(IL_GROUP);
    HostWave M{ring.data(), gref.data(), gval.data()};
    constexpr uint32_t LANES = IL_GROUP / 4u;
    ILQuad q[LANES];
    bool open[LANES];
    for (uint32_t pos = 0; pos < expect; pos += IL_GROUP) {
      const uint32_t ng = expect - pos < IL_GROUP ? expect - pos : IL_GROUP;
      steps++;
      bool any = false;
      for (uint32_t l = 0; l < LANES; l++) {
        uint64_t plan4 = 0;
        if (4u * l < ng) std::memcpy(&plan4, &plan[pos + 4u * l], 8);
        open[l] = il_quad_classify(M, pos, l, ng, plan4, q[l], bad);
        any = any || open[l];
      }
      uint32_t r = 0;
      while (any) {
        r++;
        for (uint32_t l = 0; l < LANES; l++) il_quad_publish(M, l, q[l]);
        any = false;
        for (uint32_t l = 0; l < LANES; l++) {
          if (open[l]) open[l] = il_quad_resolve(M, q[l]);
          any = any || open[l];
        }
        if (r > IL_GROUP + 2u) { std::printf("the copy stage did not settle\n"); return 4; }
      }
      jump_rounds += r;
      if (r > max_rounds) max_rounds = r;
      for (uint32_t l = 0; l < LANES; l++)
        if (4u * l < ng) { std::memcpy(&ring[(pos + 4u * l) & (IL_WINDOW - 1u)], &q[l].v, 4); std::memcpy(&out[pos + 4u * l], &q[l].v, 4); }
    }
    if (bad) status = IL_ST_BAD_DISTANCE;
  }
  if (status != IL_ST_OK) { std::printf("status %d\n", status); return 3; }
  std::printf("chunks %u | tokens: %u rounds, %.2f scans per chunk, %u chunks rescanned after round 0 | copies: %u groups, %.2f pointer-jumping rounds per group (max %u) | %u bytes\n", C, rounds, (double)scans / C, rescans_round1, steps,
              (double)jump_rounds / steps, max_rounds, expect);
  if (argc > 3) { FILE* o = std::fopen(argv[3], "wb"); if (o) { std::fwrite(out.data(), 1, expect, o); std::fclose(o); } }
  return 0;
}

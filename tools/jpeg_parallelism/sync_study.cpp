// sync_study.cpp -- does the entropy-coded segment of a baseline JPEG decode in parallel?  A decoder that starts at an arbitrary bit with a
// guessed state (block within the MCU, coefficient index) falls in step with the true decoder once bit position, coefficient index AND the
// block's place in the MCU agree (luma and chroma blocks use different tables); the stream is cut into C chunks, every lane decodes its
// chunk from its predecessor's end state of the previous round, and this tool counts the rounds until nothing changes -- which is the
// sequential result (lane 0 starts from the truth).  A STUDY for the GPU Huffman stage of the colour path (csrc/jpeg_huff_gpu.hip).
//     sync_study <file.jpg> [chunks]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Huff {
  uint8_t bits[17] = {0}, vals[256] = {0};
  uint16_t look[65536];   // (length << 8) | value for the 16-bit prefix; 0 = invalid
  void build() {
    std::memset(look, 0, sizeof(look));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      for (int i = 0; i < bits[l]; i++, k++) {
        const int first = code << (16 - l);
        for (int f = 0; f < (1 << (16 - l)); f++) look[first + f] = (uint16_t)((l << 8) | vals[k]);
        code++;
      }
      code <<= 1;
    }
  }
};

struct State { uint32_t p; uint16_t bi, k; bool operator==(const State& o) const { return p == o.p && bi == o.bi && k == o.k; } };
struct Ctx {
  std::vector<uint8_t> ecs;   // unstuffed
  uint32_t nbits;
  Huff dc[2], ac[2];
  int blocks_per_mcu, tab_of_block[10];   // 0 luma / 1 chroma per block of the MCU
  uint32_t total_blocks;
};
static inline uint32_t peek16(const Ctx& c, uint32_t p) {
  const uint32_t by = p >> 3, sh = p & 7;
  uint32_t w = 0;
  for (int i = 0; i < 3; i++) w = (w << 8) | (by + i < c.ecs.size() ? c.ecs[by + i] : 0);
  return (w >> (8 - sh)) & 0xFFFF;
}
// decode from s until p >= limit (a symbol that starts before the limit is finished); counts blocks completed and non-zero ACs
static State run(const Ctx& c, State s, uint32_t limit, uint32_t& blocks, uint32_t& entries, bool& bad) {
  blocks = entries = 0; bad = false;
  while (s.p < limit && s.p < c.nbits) {
    const int t = c.tab_of_block[s.bi];
    if (s.k == 0) {
      const uint16_t e = c.dc[t].look[peek16(c, s.p)];
      if (!e || (e & 0xFF) > 11) { bad = true; s.p += 1; continue; }   // invalid under this guess: slip a bit, keep going
      s.p += (e >> 8) + (e & 0xFF);
      s.k = 1;
    } else {
      const uint16_t e = c.ac[t].look[peek16(c, s.p)];
      if (!e) { bad = true; s.p += 1; continue; }
      const int run = (e & 0xFF) >> 4, size = e & 15;
      s.p += (e >> 8) + size;
      if (size == 0) {
        if (run == 15) s.k += 16;
        else s.k = 64;   // EOB
      } else {
        s.k += run + 1;
        entries++;
      }
      if (s.k > 64) { bad = true; s.k = 64; }
    }
    if (s.k >= 64) { s.k = 0; s.bi = (uint16_t)((s.bi + 1) % c.blocks_per_mcu); blocks++; }
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d((size_t)n);
  if (std::fread(d.data(), 1, (size_t)n, f) != (size_t)n) return 1;
  std::fclose(f);
  Ctx c;
  int W = 0, H = 0, nc = 0, hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, td[3] = {0}, ta[3] = {0};
  size_t pos = 2;
  while (pos + 4 <= d.size()) {
    if (d[pos] != 0xFF) { pos++; continue; }
    const int m = d[pos + 1];
    const size_t len = (d[pos + 2] << 8) | d[pos + 3], seg = pos + 4, end = pos + 2 + len;
    if (m == 0xC4) {
      size_t q = seg;
      while (q < end) {
        const int tc = d[q] >> 4, th = d[q] & 15; q++;
        Huff& h = tc ? c.ac[th & 1] : c.dc[th & 1];
        int tot = 0;
        for (int i = 1; i <= 16; i++) { h.bits[i] = d[q + i - 1]; tot += h.bits[i]; }
        q += 16; std::memcpy(h.vals, &d[q], (size_t)tot); q += tot; h.build();
      }
    } else if (m == 0xC0) {
      H = (d[seg + 1] << 8) | d[seg + 2]; W = (d[seg + 3] << 8) | d[seg + 4]; nc = d[seg + 5];
      for (int i = 0; i < nc; i++) { hs[i] = d[seg + 7 + 3 * i] >> 4; vs[i] = d[seg + 7 + 3 * i] & 15; }
    } else if (m == 0xDD) { std::fprintf(stderr, "restart intervals: trivially parallel, not this study\n"); return 2; }
    else if (m == 0xDA) {
      for (int i = 0; i < nc; i++) { td[i] = d[seg + 2 + 2 * i] >> 4; ta[i] = d[seg + 2 + 2 * i] & 15; }
      pos = end; break;
    }
    pos = end;
  }
  // unstuff
  for (size_t i = pos; i + 1 < d.size(); i++) {
    if (d[i] == 0xFF) { if (d[i + 1] == 0) { c.ecs.push_back(0xFF); i++; continue; } break; }
    c.ecs.push_back(d[i]);
  }
  c.nbits = (uint32_t)c.ecs.size() * 8;
  c.blocks_per_mcu = 0;
  int hmax = 1, vmax = 1;
  for (int i = 0; i < nc; i++) { hmax = hs[i] > hmax ? hs[i] : hmax; vmax = vs[i] > vmax ? vs[i] : vmax; }
  for (int i = 0; i < nc; i++) for (int b = 0; b < hs[i] * vs[i]; b++) c.tab_of_block[c.blocks_per_mcu++] = td[i] & 1;
  (void)ta;
  const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
  c.total_blocks = (uint32_t)(mcux * mcuy * c.blocks_per_mcu);
  // sequential truth: decode exactly total_blocks blocks
  uint32_t tb = 0, te = 0; bool bad = false;
  State s{0, 0, 0};
  {
    uint32_t b, e;
    // run() stops on bit limit: decode in small steps until the block count is reached
    while (tb < c.total_blocks && s.p < c.nbits) { State s2 = run(c, s, s.p + 1, b, e, bad); tb += b; te += e; s = s2; if (bad) break; }
  }
  std::printf("%dx%d, %d blocks per MCU, %u blocks, %zu bytes of entropy-coded data (%.1f bits per block), %u non-zero AC coefficients, truth ends at bit %u%s\n", W, H,
              c.blocks_per_mcu, c.total_blocks, c.ecs.size(), 8.0 * c.ecs.size() / c.total_blocks, te, s.p, bad ? " BAD" : "");
  const uint32_t true_end = s.p;
  for (uint32_t C : {256u, 1024u, 4096u}) {
    if (argc > 2) C = (uint32_t)std::atoi(argv[2]);
    const uint32_t B = (true_end + C - 1) / C;
    std::vector<State> start(C), end(C);
    std::vector<uint32_t> nb(C), ne(C);
    std::vector<char> dirty(C, 1);
    for (uint32_t k = 0; k < C; k++) start[k] = State{k * B, 0, 0};
    uint32_t rounds = 0, scans = 0;
    for (;;) {
      rounds++;
      for (uint32_t k = 0; k < C; k++) if (dirty[k]) { bool bd; end[k] = run(c, start[k], (k + 1 == C) ? true_end : (k + 1) * B, nb[k], ne[k], bd); scans++; }
      bool any = false;
      uint32_t nd = 0;
      for (uint32_t k = 1; k < C; k++) { dirty[k] = !(end[k - 1] == start[k]); start[k] = end[k - 1]; any = any || dirty[k]; nd += dirty[k]; }
      dirty[0] = 0;
      if (std::getenv("JP_TRACE")) std::fprintf(stderr, "  round %u: %u chunks restart\n", rounds, nd);
      if (!any || rounds > C + 2) break;
    }
    uint64_t sb = 0, se = 0;
    for (uint32_t k = 0; k < C; k++) { sb += nb[k]; se += ne[k]; }
    std::printf("  %4u chunks of %u bits: %u rounds, %.2f scans per chunk; blocks %llu (%s), entries %llu (%s)\n", C, B, rounds, (double)scans / C, (unsigned long long)sb,
                sb == c.total_blocks ? "ok" : "MISMATCH", (unsigned long long)se, se == te ? "ok" : "MISMATCH");
    if (argc > 2) break;
  }
  return 0;
}

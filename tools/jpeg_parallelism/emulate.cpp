// emulate.cpp -- the lane programs of csrc/jpeg_huff.h (the device's JPEG entropy decoder) run on the HOST for up to 1024 lanes in lock step, on
// the payload jpeg_prepare_huff builds, and compared coefficient by coefficient with the host decoder's payload (jpeg_decode_coef).
// A study / development aid and a CPU test (tests/test_jpeg_lanes.py):   emulate <file.jpg> <width> <height>
//   g++ -O2 -std=c++17 -I include -I scannet_amd/csrc tools/jpeg_parallelism/emulate.cpp scannet_amd/csrc/jpeg.cpp tools/jpeg_parallelism/host_stub.cpp -o emulate
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../scannet_amd/csrc/jpeg_huff.h"

int jpeg_prepare_huff(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);
int jpeg_decode_coef(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);
extern "C" const char* sf_last_error(void);

struct Writer {
  const SfJpegLayout& L;
  const SfJpegHuffDesc& D;
  uint32_t* table;
  uint32_t* entries;
  uint32_t ordinal, e;   // next block ordinal, next entry index
  int pred[3];
  void entry(bool isdc, int ci, int k, int v) {
    if (isdc) { pred[ci] += v; entries[e++] = (uint32_t)(uint16_t)(int16_t)pred[ci]; }
    else entries[e++] = ((uint32_t)jh_zigzag(k) << 16) | (uint32_t)(uint16_t)(int16_t)v;
  }
  bool block_done(int, int bi, uint32_t cnt) {
    if (ordinal >= D.total_blocks) return false;
    table[jh_block_index(L, D, ordinal, bi)] = ((e - cnt) << 7) | cnt;
    ordinal++;
    return ordinal < D.total_blocks;
  }
};

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d((size_t)n);
  if (std::fread(d.data(), 1, (size_t)n, f) != (size_t)n) return 1;
  std::fclose(f);
  const uint32_t W = (uint32_t)std::atoi(argv[2]), H = (uint32_t)std::atoi(argv[3]);
  const size_t cap = 64u << 20;
  std::vector<uint32_t> prep(cap / 4), ref(cap / 4), out(cap / 4, 0u);
  if (jpeg_prepare_huff(d.data(), (uint64_t)n, W, H, (uint8_t*)prep.data(), cap)) { std::printf("prepare: %s\n", sf_last_error()); return 2; }
  if (jpeg_decode_coef(d.data(), (uint64_t)n, W, H, (uint8_t*)ref.data(), cap)) { std::printf("host decode: %s\n", sf_last_error()); return 2; }
  const SfJpegLayout& L = *(const SfJpegLayout*)prep.data();
  const SfJpegHuffDesc& D = *(const SfJpegHuffDesc*)((const uint8_t*)prep.data() + sizeof(SfJpegLayout));
  const uint32_t* words = (const uint32_t*)((const uint8_t*)prep.data() + sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc));
  uint32_t C, B;
  jh_geometry(D.ecs_bytes * 8u, C, B);
  std::vector<uint16_t> dc12(3 * JH_LOOK12), ac12(3 * JH_LOOK12);   // the device's 12-bit first-level tables, built the way the kernel builds them
  for (int t = 0; t < 3; t++)
    for (uint32_t i = 0; i < (uint32_t)JH_LOOK12; i++) { dc12[t * JH_LOOK12 + i] = jh_look12_entry(D.dc[t], i); ac12[t * JH_LOOK12 + i] = jh_look12_entry(D.ac[t], i); }
  std::vector<JHState> start(C), end(C);
  std::vector<JHCounts> cnt(C);
  std::vector<char> dirty(C, 1);
  for (uint32_t c = 0; c < C; c++) start[c] = JHState{c * B, 0, 0, 0};
  uint32_t rounds = 0, scans = 0;
  JHNoEmit none;
  for (;;) {
    rounds++;
    for (uint32_t c = 0; c < C; c++)
      if (dirty[c]) { end[c] = jh_run(D, D.dc, D.ac, dc12.data(), ac12.data(), words, start[c], (c + 1 == C) ? D.ecs_bytes * 8u : (c + 1) * B, cnt[c], none); scans++; }
    bool any = false;
    for (uint32_t c = 1; c < C; c++) { dirty[c] = !jh_same(end[c - 1], start[c]); start[c] = end[c - 1]; any = any || dirty[c]; }
    dirty[0] = 0;
    if (!any || rounds > C + 2) break;
  }
  // stage B + C
  std::memcpy(out.data(), prep.data(), sizeof(SfJpegLayout));
  uint32_t* table = out.data() + sizeof(SfJpegLayout) / 4;
  uint32_t* entries = table + L.nblocks;
  uint32_t ord = 0, ent = 0;
  int pred[3] = {0, 0, 0};
  uint32_t bad = 0;
  for (uint32_t c = 0; c < C; c++) {
    Writer wtr{L, D, table, entries, ord, ent, {pred[0], pred[1], pred[2]}};
    JHCounts k;
    k.bad = 0;
    if (ord < D.total_blocks) (void)jh_run(D, D.dc, D.ac, dc12.data(), ac12.data(), words, start[c], (c + 1 == C) ? D.ecs_bytes * 8u : (c + 1) * B, k, wtr);
    bad += k.bad;   // of the writing pass, which stops at the picture's last block: the padding bits behind it are not a code
    ord += cnt[c].blocks; ent += cnt[c].entries;
    for (int i = 0; i < 3; i++) pred[i] += cnt[c].dc_sum[i];
  }
  // compare with the host decoder, block by block, as dense coefficient arrays
  const SfJpegLayout& R = *(const SfJpegLayout*)ref.data();
  const uint32_t* rt = ref.data() + sizeof(SfJpegLayout) / 4;
  const uint32_t* re = rt + R.nblocks;
  uint32_t diff_blocks = 0;
  for (uint32_t b = 0; b < L.nblocks; b++) {
    int16_t a[64] = {0}, g[64] = {0};
    for (uint32_t i = 0; i < (rt[b] & 127u); i++) { const uint32_t w = re[(rt[b] >> 7) + i]; a[(w >> 16) & 63] = (int16_t)(w & 0xFFFF); }
    for (uint32_t i = 0; i < (table[b] & 127u); i++) { const uint32_t w = entries[(table[b] >> 7) + i]; g[(w >> 16) & 63] = (int16_t)(w & 0xFFFF); }
    diff_blocks += std::memcmp(a, g, sizeof(a)) != 0;
  }
  std::printf("%s: %u blocks, %u chunks of %u bits, stage A %u rounds (%.2f scans per chunk), %u blocks / %u entries counted, bad %u; blocks that differ from the host decoder: %u\n",
              argv[1], L.nblocks, C, B, rounds, (double)scans / C, ord, ent, bad, diff_blocks);
  return diff_blocks ? 3 : 0;
}

// dependency_depth.cpp -- what bounds a parallel LZ77 resolution of a deflated depth frame: the depth of the match -> match dependency chains
// (a match whose source bytes were written by another match has to wait for it), the mix of literals / matches / distances, and how quickly a
// decoder started at an arbitrary bit falls in step with the true token boundaries (the fixed Huffman code is self-synchronising).
// A STUDY (tools/inflate_parallelism/README.md), not product code.     dependency_depth <file.z>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "first_study_lanes.h"
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> src(n); fread(src.data(), 1, n, f); fclose(f);
  uint32_t nbytes = n - 2; std::vector<uint32_t> words((nbytes + 3) / 4 + 2, 0u); memcpy(words.data(), src.data() + 2, nbytes);
  ILStream s{words.data(), (uint32_t)words.size(), nbytes * 8u};
  ILBits b; il_bits_init(s, b, 3);
  std::vector<uint16_t> depth; depth.reserve(700000);
  long nlit = 0, nmatch = 0, mbytes = 0; std::vector<long> hist(4096, 0), lenh(300, 0), disth(20, 0);
  // also: speculative resync test
  std::vector<uint32_t> bounds;
  for (;;) {
    bounds.push_back(b.pos);
    ILToken t = il_token(s, b);
    if (t.kind == 2) break;
    if (t.kind == 3) { printf("bad\n"); return 1; }
    if (t.kind == 0) { depth.push_back(0); nlit++; continue; }
    nmatch++; mbytes += t.value; lenh[t.value]++;
    int lg = 0; while ((1u << lg) < t.dist) lg++; disth[lg]++;
    size_t o = depth.size(); uint16_t d = 0;
    for (uint32_t i = 0; i < std::min(t.value, t.dist); i++) d = std::max(d, depth[o - t.dist + i]);
    d++;
    for (uint32_t i = 0; i < t.value; i++) depth.push_back(d);
    hist[std::min<int>(d, 4095)]++;
  }
  printf("bytes %zu literals %ld matches %ld match bytes %ld\n", depth.size(), nlit, nmatch, mbytes);
  long cum = 0; int maxd = 0; for (int i = 0; i < 4096; i++) if (hist[i]) maxd = i;
  printf("max depth %d; matches by depth:", maxd); for (int i = 1; i <= std::min(maxd, 40); i++) printf(" %ld", hist[i]); printf("\n");
  printf("len hist 3..10:"); for (int i = 3; i <= 10; i++) printf(" %ld", lenh[i]); long big = 0; for (int i = 11; i < 300; i++) big += lenh[i]; printf(" >10: %ld\n", big);
  printf("dist log2 hist:"); for (int i = 0; i < 16; i++) printf(" %ld", disth[i]); printf("\n");
  // resync: start at arbitrary bit offsets, count tokens until landing on a true boundary
  std::vector<char> isb(s.nbits + 64, 0); for (uint32_t p : bounds) isb[p] = 1;
  long tot = 0, fails = 0, trials = 0, maxtok = 0; std::vector<long> th(64, 0);
  for (uint32_t start = 100003; start + 5000 < s.nbits; start += 3571) {
    if (isb[start]) continue;
    ILBits q; il_bits_init(s, q, start); int k = 0; bool ok = false;
    while (q.pos < start + 2400) { il_token(s, q); k++; if (isb[q.pos]) { ok = true; break; } }
    trials++; if (!ok) fails++; else { tot += k; maxtok = std::max<long>(maxtok, k); th[std::min(k, 63)]++; }
  }
  printf("resync trials %ld fails(within 2400 bits) %ld avg tokens %.1f max %ld\n", trials, fails, trials > fails ? (double)tot / (trials - fails) : 0.0, maxtok);
  return 0;
}

// TEST INFRASTRUCTURE: ThreadSanitizer run of the host-side thread pools that have no GPU in them (tools/tsan/run_host.sh, tests/test_sanitizers.py):
//   * sf_sens_writer_*        a producer against the background compressor / writer through a two-frame queue, three files in a row
//   * sf_sens_save_to_images  the export pool with its in-order progress
//   * sf_mesh_merge_parts     key ranges merged side by side, then sf_mesh_write_ply's slices pwritten by several threads
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "scanfuse.h"

#define CHECK(call) do { if ((call) != SF_OK) { std::printf("%s: %s\n", #call, sf_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  sf_sens_info h;
  std::memset(&h, 0, sizeof h);
  h.depth_width = 64; h.depth_height = 48; h.color_width = 64; h.color_height = 48; h.depth_compression = 1; h.color_compression = 0; h.depth_shift = 1000;
  std::snprintf(h.sensor_name, sizeof h.sensor_name, "tsan");
  const std::string sens_path = dir + "/stream.sens";
  for (int round = 0; round < 3; round++) {
    sf_sens_writer* w = nullptr;
    CHECK(sf_sens_writer_open(&h, sens_path.c_str(), 1, 2, &w));
    std::vector<uint16_t> d(64 * 48);
    std::vector<uint8_t> c(64 * 48 * 3);
    const float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 200; i++) {
      for (size_t k = 0; k < d.size(); k++) d[k] = (uint16_t)(i * 7 + k);
      for (size_t k = 0; k < c.size(); k++) c[k] = (uint8_t)(i + k);
      CHECK(sf_sens_writer_add_frame(w, c.data(), c.size(), d.data(), pose, i, i));
    }
    uint64_t n = 0;
    CHECK(sf_sens_writer_close(w, &n));
    if (n != 200) { std::printf("wrote %llu frames\n", (unsigned long long)n); return 1; }
  }
  sf_sens* s = nullptr;
  CHECK(sf_sens_open(sens_path.c_str(), &s));
  uint64_t seen = 0;
  struct P { static void tick(uint64_t i, uint64_t, void* u) { *(uint64_t*)u += i + 1; } };
  CHECK(sf_sens_save_to_images(s, (dir + "/images").c_str(), nullptr, &P::tick, &seen));
  sf_sens_close(s);
  if (seen != 200 * 201 / 2) { std::printf("progress calls: %llu\n", (unsigned long long)seen); return 1; }
  std::mt19937_64 rng(1);
  const int K = 3;
  const size_t NV = 300000, NF = 300000;
  std::vector<sf_mesh*> parts;
  for (int p = 0; p < K; p++) {
    std::vector<uint64_t> keys(NV), fk(NF);
    for (auto& k : keys) k = (rng() % 2000000) * 2 + (rng() % 4 == 0 ? 0 : 1) * (uint64_t)p;   // a quarter of the keys can meet another part's
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    for (auto& k : fk) k = (rng() >> 16) * K + p;
    std::sort(fk.begin(), fk.end());
    std::vector<float> xyz(keys.size() * 3, 1.0f);
    std::vector<uint8_t> rgba(keys.size() * 4, 7);
    std::vector<uint32_t> tris(NF * 3);
    for (auto& t : tris) t = (uint32_t)(rng() % keys.size());
    sf_mesh* m = nullptr;
    CHECK(sf_mesh_create_keyed(xyz.data(), rgba.data(), keys.data(), keys.size(), tris.data(), fk.data(), NF, &m));
    parts.push_back(m);
  }
  sf_mesh* out = nullptr;
  CHECK(sf_mesh_merge_parts(parts.data(), K, &out));
  CHECK(sf_mesh_write_ply(out, (dir + "/merged.ply").c_str()));
  uint64_t nv = 0, nf = 0;
  sf_mesh_counts(out, &nv, &nf);
  for (sf_mesh* m : parts) sf_mesh_free(m);
  sf_mesh_free(out);
  std::printf("streamed 3 x 200 frames, exported 200, merged %llu vertices %llu faces\ntsan: clean\n", (unsigned long long)nv, (unsigned long long)nf);
  return 0;
}

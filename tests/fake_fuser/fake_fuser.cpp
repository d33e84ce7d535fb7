// TEST INFRASTRUCTURE (tests/test_zz_depthsensing_ranks.py): a stand-in for the DEVICE half of libscanfuse.so, so that bin/depthsensing's own logic --
// the ranks it starts, the files they exchange, the waiting, the failure protocol, the merge and the file it writes -- runs on a machine without a GPU.
// It is linked with the REAL host half (sens.cpp, params.cpp, the codecs, ply.cpp: sf_mesh_create_keyed / sf_mesh_merge_parts / sf_mesh_write_ply) into a
// library of the same name in a scratch directory, and the tool's source is compiled against that.  Nothing here is shipped, and nothing here fuses:
//
//   the "volume" of a scan of n frames is a fixed pseudo-random set of blocks (x in [-48, 48), y, z in {0, 1, 2}; present iff hash(x, y, z, n) % 4 != 0),
//   each with 4096 bytes of content derived from its coordinates; a fuser owns the blocks of its stripes (the rule of sf_fuser_set_stripes), exports the
//   lowest layer of each stripe, checks the content of what it is handed and keeps what sits right above a layer it owns -- as the real fuser does;
//   the "mesh" has one vertex per present lattice point (key = packed coordinates, position derived from the block's content) and, per OWNED block, one face
//   to its +x and +y neighbours when those are present AND known to this fuser (owned or ghost) -- so a boundary layer that did not arrive, arrived at the
//   wrong rank or arrived damaged changes the merged file, exactly where marching cubes would.
//
// Knobs (environment, read here only): FAKE_FAIL_DEVICE=<d> fails sf_fuse_run on device d; FAKE_SLOW_DEVICE=<d> delays device d by FAKE_SLOW_MS (400) ms
// before its export (the neighbour has to wait); FAKE_CORRUPT_DEVICE=<d>: device d exports one damaged block; FAKE_DEVICES=<n> devices "visible" (default 4).
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <string>
#include <vector>

#include "common.h"
#include "sens.h"

namespace {
uint64_t mix(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}
uint64_t block_hash(int x, int y, int z, uint64_t n) { return mix(((uint64_t)(uint32_t)(x + 1000) << 40) ^ ((uint64_t)(y + 10) << 20) ^ (uint64_t)(z + 10) ^ (n * 0x9e3779b97f4a7c15ull)); }
bool present(int x, int y, int z, uint64_t n) { return x >= -48 && x < 48 && y >= 0 && y < 3 && z >= 0 && z < 3 && block_hash(x, y, z, n) % 4 != 0; }
void content(int x, int y, int z, uint64_t n, uint8_t* out) {
  uint64_t h = block_hash(x, y, z, n);
  for (int i = 0; i < 512; i++) { h = mix(h + i); std::memcpy(out + i * 8, &h, 8); }
}
uint64_t key_of(int x, int y, int z) { return ((uint64_t)(x + 1000) << 42) | ((uint64_t)(y + 1000) << 21) | (uint64_t)(z + 1000); }
int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
}  // namespace

struct sf_fuser {
  int device = 0;
  bool striped = false;
  int origin = 0, thick = 1, world = 1, rank = 0;
  uint64_t frames = 0;
  std::map<std::tuple<int, int, int>, std::vector<uint8_t>> ghosts;
  int owner(int x) const {
    int q = x - origin, s = q >= 0 ? q / thick : -((-q + thick - 1) / thick);
    return ((s % world) + world) % world;
  }
  bool owns(int x) const { return !striped || owner(x) == rank; }
};

SF_API int sf_device_count(int* count) { *count = env_int("FAKE_DEVICES", 4); return SF_OK; }
SF_API int sf_fuser_create(const sf_params*, int device, sf_fuser** out) {
  if (device < 0 || device >= env_int("FAKE_DEVICES", 4)) return sf::fail(SF_ERR_DEVICE, "fake: no device %d", device);
  *out = new sf_fuser();
  (*out)->device = device;
  return SF_OK;
}
SF_API void sf_fuser_destroy(sf_fuser* f) { delete f; }
SF_API int sf_fuser_set_stripes(sf_fuser* f, int axis, int32_t origin, int32_t thick, int world, int rank) {
  if (axis != 0 || thick < 1 || world < 1 || rank < 0 || rank >= world) return sf::fail(SF_ERR_INVALID_ARG, "fake: stripes");
  f->striped = true; f->origin = origin; f->thick = thick; f->world = world; f->rank = rank;
  return SF_OK;
}
SF_API int sf_fuse_run_prepare(const sf_sens*, const sf_params*, int) { return SF_OK; }
SF_API int sf_fuse_run(sf_fuser* f, const sf_sens* s, uint64_t, uint64_t, int, sf_run_stats* rs) {
  if (env_int("FAKE_FAIL_DEVICE", -1) == f->device) return sf::fail(SF_ERR_FORMAT, "fake: device %d was told to fail", f->device);
  sf_sens_info info;
  sf_sens_get_info(s, &info);
  f->frames = info.num_frames;
  if (env_int("FAKE_SLOW_DEVICE", -1) == f->device) usleep(1000 * (useconds_t)env_int("FAKE_SLOW_MS", 400));
  std::memset(rs, 0, sizeof *rs);
  rs->frames_total = rs->frames_integrated = info.num_frames;
  rs->decode_threads = 1;
  rs->seconds_total = 0.001;
  return SF_OK;
}
SF_API int sf_fuser_stats(sf_fuser* f, sf_stats* st) {
  std::memset(st, 0, sizeof *st);
  for (int x = -48; x < 48; x++)
    for (int y = 0; y < 3; y++)
      for (int z = 0; z < 3; z++) st->blocks_allocated += present(x, y, z, f->frames) && f->owns(x);
  return SF_OK;
}
SF_API int sf_fuser_garbage_collect(sf_fuser*, uint32_t* freed) { *freed = 0; return SF_OK; }
SF_API int sf_fuser_export_boundary(sf_fuser* f, int32_t* coords, void* voxels, uint64_t capacity, uint64_t* n, int on_device) {
  if (on_device) return sf::fail(SF_ERR_INVALID_ARG, "fake: host buffers only");
  uint64_t k = 0;
  for (int x = -48; x < 48; x++) {
    if (!f->striped || !f->owns(x) || f->owner(x - 1) == f->rank) continue;   // the lowest layer of one of my stripes
    for (int y = 0; y < 3; y++)
      for (int z = 0; z < 3; z++)
        if (present(x, y, z, f->frames)) {
          if (coords) {
            if (k >= capacity) return sf::fail(SF_ERR_CAPACITY, "fake: capacity");
            coords[k * 3] = x; coords[k * 3 + 1] = y; coords[k * 3 + 2] = z;
            content(x, y, z, f->frames, (uint8_t*)voxels + k * 4096);
            if (k == 3 && env_int("FAKE_CORRUPT_DEVICE", -1) == f->device) ((uint8_t*)voxels)[k * 4096 + 77] ^= 1;
          }
          k++;
        }
  }
  *n = k;
  return SF_OK;
}
SF_API int sf_fuser_import_ghosts(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int on_device, uint64_t* imported) {
  if (on_device) return sf::fail(SF_ERR_INVALID_ARG, "fake: host buffers only");
  uint8_t want[4096];
  uint64_t kept = 0;
  for (uint64_t i = 0; i < n; i++) {
    const int x = coords[i * 3], y = coords[i * 3 + 1], z = coords[i * 3 + 2];
    if (!present(x, y, z, f->frames)) return sf::fail(SF_ERR_FORMAT, "fake: block (%d,%d,%d) does not exist in this scan", x, y, z);
    content(x, y, z, f->frames, want);
    if (std::memcmp(want, (const uint8_t*)voxels + i * 4096, 4096) != 0) return sf::fail(SF_ERR_FORMAT, "fake: block (%d,%d,%d) arrived damaged", x, y, z);
    if (f->owns(x) || !f->owns(x - 1)) continue;   // mine already, or not above a layer of mine
    f->ghosts[{x, y, z}].assign(want, want + 4096);
    kept++;
  }
  if (imported) *imported = kept;
  return SF_OK;
}
// The device-to-device exchange (csrc/exchange.hip) as the tool sees it.  Default: "no transport on every rank" (SF_ERR_UNSUPPORTED), which sends the
// tool down its file exchange -- the fallback is what most tests here run.  FAKE_EXCHANGE=1: a stand-in transport (the same ring shift through files
// of its own in the rendezvous directory) so that the tool's sf_exchange_* branch, its log line and its clean-up are exercised too; FAKE_EXCHANGE=2:
// the set-up fails with a device error (a transport asked for by name must fail the run, not fall back).
struct sf_exchange { std::string dir; int rank, ranks; };
SF_API int sf_exchange_create(const char* dir, int rank, int ranks, int, int transport, sf_exchange** out) {
  const int mode = env_int("FAKE_EXCHANGE", 0);
  if (mode == 2) return sf::fail(SF_ERR_DEVICE, "fake: the transport could not be set up");
  if (mode == 0) return sf::fail(SF_ERR_UNSUPPORTED, "fake: no device-to-device transport (transport %d asked for)", transport);
  *out = new sf_exchange{dir, rank, ranks};
  return SF_OK;
}
SF_API const char* sf_exchange_transport(const sf_exchange*) { return "the stand-in transport"; }
SF_API void sf_exchange_destroy(sf_exchange* x) { delete x; }
SF_API int sf_exchange_boundary(sf_exchange* x, sf_fuser* f, uint64_t* sent, uint64_t* received, uint64_t* kept) {
  uint64_t n = 0, m = 0, k = 0, got = 0;
  if (sf_fuser_export_boundary(f, nullptr, nullptr, 0, &n, 0) != SF_OK) return SF_ERR_DEVICE;
  std::vector<int32_t> coords(n * 3);
  std::vector<uint8_t> voxels(n * 4096);
  if (n && sf_fuser_export_boundary(f, coords.data(), voxels.data(), n, &m, 0) != SF_OK) return SF_ERR_DEVICE;
  const std::string mine = x->dir + "/ipc0_" + std::to_string(x->rank), tmp = mine + ".tmp";
  FILE* fp = std::fopen(tmp.c_str(), "wb");
  if (!fp) return sf::fail(SF_ERR_IO, "fake: exchange file");
  std::fwrite(&n, 8, 1, fp);
  if (n) { std::fwrite(coords.data(), 12, n, fp); std::fwrite(voxels.data(), 4096, n, fp); }
  std::fclose(fp);
  std::rename(tmp.c_str(), mine.c_str());
  const std::string from = x->dir + "/ipc0_" + std::to_string((x->rank + 1) % x->ranks);
  for (int spin = 0;; spin++) {
    if (FILE* t = std::fopen(from.c_str(), "rb")) { fp = t; break; }
    if (FILE* t = std::fopen((x->dir + "/abort").c_str(), "rb")) { std::fclose(t); return sf::fail(SF_ERR_IO, "fake: exchange aborted"); }
    if (spin > 60000) return sf::fail(SF_ERR_IO, "fake: exchange timed out");
    usleep(500);
  }
  bool ok = std::fread(&k, 8, 1, fp) == 1;
  coords.resize(k * 3);
  voxels.resize(k * 4096);
  ok = ok && (k == 0 || (std::fread(coords.data(), 12, k, fp) == k && std::fread(voxels.data(), 4096, k, fp) == k));
  std::fclose(fp);
  if (!ok) return sf::fail(SF_ERR_IO, "fake: short exchange file");
  if (k && sf_fuser_import_ghosts(f, coords.data(), voxels.data(), k, 0, &got) != SF_OK) return SF_ERR_FORMAT;
  if (sent) *sent = n;
  if (received) *received = k;
  if (kept) *kept = got;
  return SF_OK;
}

SF_API int sf_fuser_extract_mesh(sf_fuser* f, sf_mesh** out) {
  auto known = [&](int x, int y, int z) { return present(x, y, z, f->frames) && (f->owns(x) || f->ghosts.count({x, y, z}) != 0); };
  std::map<uint64_t, std::tuple<int, int, int>> verts;   // key -> lattice point, ascending
  struct Face { uint64_t cube; uint64_t a, b, c; };
  std::vector<Face> faces;
  for (int x = -48; x < 48; x++)
    for (int y = 0; y < 3; y++)
      for (int z = 0; z < 3; z++) {
        if (!present(x, y, z, f->frames) || !f->owns(x)) continue;
        if (!known(x + 1, y, z) || !known(x, y + 1, z)) continue;
        faces.push_back({key_of(x, y, z), key_of(x, y, z), key_of(x + 1, y, z), key_of(x, y + 1, z)});
        verts[key_of(x, y, z)] = {x, y, z};
        verts[key_of(x + 1, y, z)] = {x + 1, y, z};
        verts[key_of(x, y + 1, z)] = {x, y + 1, z};
      }
  std::vector<float> xyz;
  std::vector<uint8_t> rgba;
  std::vector<uint64_t> keys, fkeys;
  std::map<uint64_t, uint32_t> index;
  for (const auto& kv : verts) {
    const int x = std::get<0>(kv.second), y = std::get<1>(kv.second), z = std::get<2>(kv.second);
    const uint64_t h = block_hash(x, y, z, f->frames);   // position and colour come from the block's content: both ranks that hold it agree
    index[kv.first] = (uint32_t)keys.size();
    keys.push_back(kv.first);
    xyz.push_back(x + (h & 255) / 256.0f); xyz.push_back(y + ((h >> 8) & 255) / 256.0f); xyz.push_back(z + ((h >> 16) & 255) / 256.0f);
    for (int c = 0; c < 4; c++) rgba.push_back((uint8_t)(h >> (24 + 8 * c)));
  }
  std::sort(faces.begin(), faces.end(), [](const Face& a, const Face& b) { return a.cube < b.cube; });
  std::vector<uint32_t> tris;
  for (const Face& t : faces) {
    tris.push_back(index[t.a]); tris.push_back(index[t.b]); tris.push_back(index[t.c]);
    fkeys.push_back(t.cube);
  }
  return sf_mesh_create_keyed(xyz.data(), rgba.data(), keys.data(), keys.size(), tris.data(), fkeys.data(), fkeys.size(), out);
}

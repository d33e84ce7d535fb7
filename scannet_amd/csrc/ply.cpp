// ply.cpp -- PLY / OBJ ingest and the PLY writer (see include/scanfuse.h for the reference interfaces replaced).
//
// The reader maps the file and walks it once with fixed strides (tinyply dispatches a std::function per
// property per element: 63 % of the reference Segmentator's run time).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <thread>
#include <exception>
#include <mutex>
#include <memory>
#include <vector>

#include "mesh.h"

namespace {

struct Prop {
  std::string name;
  int type = -1;       // index into TYPES
  bool is_list = false;
  int count_type = -1;
};
struct Elem {
  std::string name;
  uint64_t size = 0;
  std::vector<Prop> props;
};

const char* const TYPE_NAMES[][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                     {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
const int TYPE_SIZE[] = {1, 1, 2, 2, 4, 4, 4, 8};

int type_from(const std::string& s) {
  for (int i = 0; i < 8; i++)
    if (s == TYPE_NAMES[i][0] || s == TYPE_NAMES[i][1]) return i;
  return -1;
}

struct MapFile {
  const uint8_t* p = nullptr;
  uint64_t n = 0;
  int fd = -1;
  ~MapFile() {
    if (p) munmap((void*)p, (size_t)n);
    if (fd >= 0) ::close(fd);
  }
  int open_file(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return sf::fail(SF_ERR_IO, "could not open file %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) return sf::fail(SF_ERR_IO, "could not stat %s", path);
    n = (uint64_t)st.st_size;
    if (n == 0) { p = nullptr; return SF_OK; }
    void* m = mmap(nullptr, (size_t)n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return sf::fail(SF_ERR_IO, "mmap of %s failed", path);
    p = (const uint8_t*)m;
    return SF_OK;
  }
};

template <typename T>
inline T load(const uint8_t* p, bool swap) {
  uint8_t b[sizeof(T)];
  if (swap) for (size_t i = 0; i < sizeof(T); i++) b[i] = p[sizeof(T) - 1 - i];
  else std::memcpy(b, p, sizeof(T));
  T v;
  std::memcpy(&v, b, sizeof(T));
  return v;
}

inline uint64_t load_uint(const uint8_t* p, int type, bool swap) {
  switch (type) {
    case 0: return (uint64_t)(int64_t)load<int8_t>(p, swap);
    case 1: return load<uint8_t>(p, swap);
    case 2: return (uint64_t)(int64_t)load<int16_t>(p, swap);
    case 3: return load<uint16_t>(p, swap);
    case 4: return (uint64_t)(int64_t)load<int32_t>(p, swap);
    case 5: return load<uint32_t>(p, swap);
    default: return 0;
  }
}

int read_ply(const char* path, sf_mesh* m) {
  MapFile f;
  int rc = f.open_file(path);
  if (rc != SF_OK) return rc;
  // ---- header
  std::vector<Elem> elems;
  bool binary = false, big = false, got_end = false;
  uint64_t pos = 0;
  while (pos < f.n) {
    uint64_t e = pos;
    while (e < f.n && f.p[e] != '\n') e++;
    std::string line((const char*)f.p + pos, (size_t)(e - pos));
    pos = e < f.n ? e + 1 : e;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string tok;
    ls >> tok;
    if (tok == "ply" || tok == "PLY" || tok.empty()) continue;
    if (tok == "comment" || tok == "obj_info") continue;
    if (tok == "format") {
      std::string s;
      ls >> s;
      if (s == "binary_little_endian") binary = true;
      else if (s == "binary_big_endian") binary = big = true;
    } else if (tok == "element") {
      Elem el;
      ls >> el.name >> el.size;
      elems.push_back(el);
    } else if (tok == "property") {
      if (elems.empty()) return sf::fail(SF_ERR_FORMAT, "%s: property before any element", path);
      Prop pr;
      std::string t;
      ls >> t;
      if (t == "list") {
        std::string ct;
        ls >> ct >> t;
        pr.is_list = true;
        pr.count_type = type_from(ct);
        if (pr.count_type < 0 || pr.count_type > 5) return sf::fail(SF_ERR_FORMAT, "%s: bad list count type '%s'", path, ct.c_str());
      }
      pr.type = type_from(t);
      if (pr.type < 0) return sf::fail(SF_ERR_FORMAT, "%s: unknown property type '%s'", path, t.c_str());
      ls >> pr.name;
      elems.back().props.push_back(pr);
    } else if (tok == "end_header") {
      got_end = true;
      break;
    } else {
      return sf::fail(SF_ERR_FORMAT, "%s: file is not ply or encountered junk in header", path);  // tinyply.cpp:56-59
    }
  }
  if (!got_end) return sf::fail(SF_ERR_FORMAT, "%s: file is not ply or encountered junk in header", path);

  // ---- body
  const uint8_t* p = f.p + pos;
  const uint8_t* end = f.p + f.n;
  auto need = [&](uint64_t k) { return (uint64_t)(end - p) >= k; };
  // ascii tokenizer
  auto next_token = [&](const char*& a, const char*& b) -> bool {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    if (p >= end) return false;
    a = (const char*)p;
    while (p < end && !(*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    b = (const char*)p;
    return true;
  };
  for (const Elem& el : elems) {
    const bool is_vertex = el.name == "vertex", is_face = el.name == "face";
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, ia = -1, il = -1;
    for (size_t k = 0; k < el.props.size(); k++) {
      const Prop& pr = el.props[k];
      if (is_vertex && !pr.is_list) {
        if (pr.name == "x") ix = (int)k; else if (pr.name == "y") iy = (int)k; else if (pr.name == "z") iz = (int)k;
        else if (pr.name == "red") ir = (int)k; else if (pr.name == "green") ig = (int)k; else if (pr.name == "blue") ib = (int)k;
        else if (pr.name == "alpha") ia = (int)k;
      }
      if (is_face && pr.is_list && il < 0 && pr.name == "vertex_indices") il = (int)k;
    }
    if (is_face && il < 0)
      for (size_t k = 0; k < el.props.size(); k++)
        if (el.props[k].is_list && el.props[k].name == "vertex_index") { il = (int)k; break; }  // segmentator.cpp:136-139
    if (is_vertex && ix >= 0 && iy >= 0 && iz >= 0) {
      for (int q : {ix, iy, iz})
        if (TYPE_SIZE[el.props[q].type] != 4) return sf::fail(SF_ERR_FORMAT, "%s: destination vector is wrongly typed to hold this property (x/y/z must be 4-byte floats)", path);
      m->pos.assign(el.size * 3, 0.0f);   // (mesh arrays do not zero-fill on resize: mesh.h)
      const bool has_col = ir >= 0 && ig >= 0 && ib >= 0 && TYPE_SIZE[el.props[ir].type] == 1 && TYPE_SIZE[el.props[ig].type] == 1 && TYPE_SIZE[el.props[ib].type] == 1;
      if (has_col) m->col.assign(el.size * 4, 255);
    }
    if (is_face && il >= 0 && TYPE_SIZE[el.props[il].type] != 4)
      return sf::fail(SF_ERR_FORMAT, "%s: destination vector is wrongly typed to hold this property (face indices must be 4-byte integers)", path);
    if (is_face && il >= 0) m->tri.assign(el.size * 3, 0u);
    const bool want_v = is_vertex && !m->pos.empty();
    const bool want_f = is_face && il >= 0;
    const bool has_col = want_v && !m->col.empty();
    if (binary) {
      // fixed-stride fast path when the element has no list property
      bool fixed = true;
      uint64_t stride = 0;
      std::vector<uint64_t> off(el.props.size());
      for (size_t k = 0; k < el.props.size(); k++) {
        if (el.props[k].is_list) fixed = false;
        off[k] = stride;
        stride += TYPE_SIZE[el.props[k].type];
      }
      if (fixed) {
        if (stride && el.size > (uint64_t)(end - p) / stride) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
        if (want_v) {
          for (uint64_t i = 0; i < el.size; i++) {
            const uint8_t* r = p + i * stride;
            m->pos[3 * i] = load<float>(r + off[ix], big);
            m->pos[3 * i + 1] = load<float>(r + off[iy], big);
            m->pos[3 * i + 2] = load<float>(r + off[iz], big);
            if (has_col) {
              m->col[4 * i] = r[off[ir]]; m->col[4 * i + 1] = r[off[ig]]; m->col[4 * i + 2] = r[off[ib]];
              if (ia >= 0 && TYPE_SIZE[el.props[ia].type] == 1) m->col[4 * i + 3] = r[off[ia]];
            }
          }
        }
        p += el.size * stride;
      } else {
        for (uint64_t i = 0; i < el.size; i++) {
          for (size_t k = 0; k < el.props.size(); k++) {
            const Prop& pr = el.props[k];
            const int ts = TYPE_SIZE[pr.type];
            if (!pr.is_list) {
              if (!need((uint64_t)ts)) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
              if (want_v) {
                if ((int)k == ix) m->pos[3 * i] = load<float>(p, big);
                else if ((int)k == iy) m->pos[3 * i + 1] = load<float>(p, big);
                else if ((int)k == iz) m->pos[3 * i + 2] = load<float>(p, big);
              }
              p += ts;
            } else {
              const int cs = TYPE_SIZE[pr.count_type];
              if (!need((uint64_t)cs)) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
              const uint64_t cnt = load_uint(p, pr.count_type, big);
              p += cs;
              if (cnt > (uint64_t)(end - p) / (uint64_t)ts) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
              if (want_f && (int)k == il) {
                if (cnt != 3) return sf::fail(SF_ERR_FORMAT, "%s: face %llu has %llu vertices; only triangle meshes are supported", path, (unsigned long long)i, (unsigned long long)cnt);
                for (int q = 0; q < 3; q++) m->tri[3 * i + q] = load<uint32_t>(p + 4 * q, big);
              }
              p += cnt * (uint64_t)ts;
            }
          }
        }
      }
    } else {
      const char *a, *b;
      for (uint64_t i = 0; i < el.size; i++) {
        for (size_t k = 0; k < el.props.size(); k++) {
          const Prop& pr = el.props[k];
          if (!pr.is_list) {
            if (!next_token(a, b)) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
            if (want_v && ((int)k == ix || (int)k == iy || (int)k == iz)) {
              const float v = std::strtof(std::string(a, b).c_str(), nullptr);
              m->pos[3 * i + ((int)k == ix ? 0 : ((int)k == iy ? 1 : 2))] = v;
            } else if (has_col && ((int)k == ir || (int)k == ig || (int)k == ib || (int)k == ia)) {
              const int v = std::atoi(std::string(a, b).c_str());
              m->col[4 * i + ((int)k == ir ? 0 : ((int)k == ig ? 1 : ((int)k == ib ? 2 : 3)))] = (uint8_t)v;
            }
          } else {
            if (!next_token(a, b)) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
            const long cnt = std::atol(std::string(a, b).c_str());
            if (cnt < 0) return sf::fail(SF_ERR_FORMAT, "%s: negative list size", path);
            if (want_f && (int)k == il && cnt != 3) return sf::fail(SF_ERR_FORMAT, "%s: face %llu has %ld vertices; only triangle meshes are supported", path, (unsigned long long)i, cnt);
            for (long q = 0; q < cnt; q++) {
              if (!next_token(a, b)) return sf::fail(SF_ERR_FORMAT, "%s: truncated '%s' element", path, el.name.c_str());
              if (want_f && (int)k == il) m->tri[3 * i + q] = (uint32_t)std::strtoll(std::string(a, b).c_str(), nullptr, 10);
            }
          }
        }
      }
    }
  }
  const uint64_t nv = m->pos.size() / 3;
  for (uint32_t t : m->tri)
    if (t >= nv) return sf::fail(SF_ERR_FORMAT, "%s: face index %u out of range (%llu vertices)", path, t, (unsigned long long)nv);
  return SF_OK;
}

// Minimal Wavefront OBJ reader with tiny_obj_loader's shape rules as the Segmentator relies on them
// (segmentator.cpp:142-174): all `v` records, faces of the FIRST shape only.
int read_obj(const char* path, sf_mesh* m, bool* multi) {
  FILE* fp = std::fopen(path, "r");
  if (!fp) return sf::fail(SF_ERR_IO, "could not open file %s", path);
  char buf[4096];
  int shape = 0;
  bool shape_has_faces = false;
  *multi = false;
  auto fix = [&](long idx) -> long { const long nv = (long)(m->pos.size() / 3); return idx > 0 ? idx - 1 : (idx < 0 ? nv + idx : -1); };
  while (std::fgets(buf, sizeof(buf), fp)) {
    char* s = buf;
    while (*s == ' ' || *s == '\t') s++;
    if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
      float x = 0, y = 0, z = 0;
      std::sscanf(s + 2, "%f %f %f", &x, &y, &z);
      m->pos.push_back(x); m->pos.push_back(y); m->pos.push_back(z);
    } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
      std::vector<long> ids;
      char* q = s + 2;
      while (*q) {
        while (*q == ' ' || *q == '\t') q++;
        if (*q == 0 || *q == '\n' || *q == '\r') break;
        ids.push_back(fix(std::strtol(q, &q, 10)));
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;  // skip /vt/vn
      }
      if (shape == 0) {
        if (ids.size() != 3) { std::fclose(fp); return sf::fail(SF_ERR_FORMAT, "%s: only triangle faces are supported", path); }
        for (long id : ids) m->tri.push_back((uint32_t)id);
      } else *multi = true;
      shape_has_faces = true;
    } else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || s[1] == '\n')) {
      if (shape_has_faces) { shape++; shape_has_faces = false; }
    }
  }
  std::fclose(fp);
  const uint64_t nv = m->pos.size() / 3;
  for (uint32_t t : m->tri)
    if (t >= nv) return sf::fail(SF_ERR_FORMAT, "%s: face index out of range", path);
  return SF_OK;
}

bool ends_with(const std::string& v, const std::string& e) { return e.size() <= v.size() && std::equal(e.rbegin(), e.rend(), v.rbegin()); }

}  // namespace

int mesh_read_any(const char* path, sf_mesh* m, bool* obj_multi) {
  const std::string s(path);
  if (obj_multi) *obj_multi = false;
  if (ends_with(s, ".ply") || ends_with(s, ".PLY")) return read_ply(path, m);
  if (ends_with(s, ".obj") || ends_with(s, ".OBJ")) { bool mm = false; const int rc = read_obj(path, m, &mm); if (obj_multi) *obj_multi = mm; return rc; }
  // the reference silently segments an empty mesh for other extensions (segmentator.cpp:130-175)
  return SF_OK;
}

SF_API int sf_ply_read(const char* path, sf_mesh** out) {
  if (!path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_mesh* m = new sf_mesh();
  const int rc = mesh_read_any(path, m, nullptr);
  if (rc != SF_OK) { delete m; return rc; }
  *out = m;
  return SF_OK;
}

SF_API int sf_mesh_create(const float* xyz, const uint8_t* rgba, uint64_t nv, const uint32_t* tris, uint64_t nf, sf_mesh** out) {
  if ((!xyz && nv) || (!tris && nf) || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  for (uint64_t i = 0; i < nf * 3; i++)
    if (tris[i] >= nv) return sf::fail(SF_ERR_BOUNDS, "face index %u out of range (%llu vertices)", tris[i], (unsigned long long)nv);
  try {
    std::unique_ptr<sf_mesh> m(new sf_mesh());
    m->pos.assign(xyz, xyz + nv * 3);
    if (rgba) m->col.assign(rgba, rgba + nv * 4);
    m->tri.assign(tris, tris + nf * 3);
    *out = m.release();
  } catch (...) { return sf::fail(SF_ERR_IO, "out of memory for a mesh of %llu vertices, %llu faces", (unsigned long long)nv, (unsigned long long)nf); }
  return SF_OK;
}

SF_API int sf_mesh_create_keyed(const float* xyz, const uint8_t* rgba, const uint64_t* keys, uint64_t nv, const uint32_t* tris, const uint64_t* face_keys,
                                uint64_t nf, sf_mesh** out) {
  if (!out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!keys && nv) return sf::fail(SF_ERR_INVALID_ARG, "NULL vertex keys");
  sf_mesh* m = nullptr;
  const int rc = sf_mesh_create(xyz, rgba, nv, tris, nf, &m);
  if (rc != SF_OK) return rc;
  try {
    m->keys.assign(keys, keys + nv);
    if (face_keys) m->tkeys.assign(face_keys, face_keys + nf);
  } catch (...) {
    delete m;
    return sf::fail(SF_ERR_IO, "out of memory for the keys of a mesh of %llu vertices, %llu faces", (unsigned long long)nv, (unsigned long long)nf);
  }
  *out = m;
  return SF_OK;
}

// The meshes of a partitioned scan -> the mesh one fuser would have extracted (scannet_amd/partition.py merge_slab_meshes is the same rule in numpy;
// tests/test_partition_merge.py holds the two against each other).  Every vertex carries the key of the lattice edge it sits on, every face the key of its cube:
//   vertices: unique by key, in key order; of the copies of a key (an edge shared by cubes of two ranks: same key, same position, same colour on
//             both) the one of the lowest part is kept;
//   faces:    re-indexed through the keys, parts concatenated, then -- when every part has face keys -- stably sorted by cube key (each cube belongs to
//             one rank, so that is the one-fuser order whatever the partition); without face keys the order stays "part after part" (contiguous slabs).
namespace {

template <class F>
void parallel_for(int n_threads, F&& fn) {   // fn(t) for t in [0, n_threads); the caller's thread takes t = 0
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex mu;
  auto guarded = [&](int t) {   // an exception inside a std::thread is std::terminate: carry the first one to the caller's thread
    try { fn(t); } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!err) err = std::current_exception(); }
  };
  try {
    for (int t = 1; t < n_threads; t++) th.emplace_back(guarded, t);
  } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!err) err = std::current_exception(); }
  guarded(0);
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

// k sorted runs (run p = positions [lo[p], hi[p]) of part p, key(p, i) ascending in i) merged in ascending (key, part, position) order: take(p, i, key).
// The smallest head, the lowest part among equals -- k is the number of GPUs, a scan over the heads beats a heap -- and everything of that run below the
// next-best head goes out in one go (the runs of one rank's stripe are long).
template <class Key, class Take>
void merge_runs(int k, std::vector<size_t> at, const std::vector<size_t>& hi, Key&& key, Take&& take) {
  for (;;) {
    int best = -1;
    uint64_t bk = 0;
    for (int p = 0; p < k; p++)
      if (at[p] < hi[p]) {
        const uint64_t h = key(p, at[p]);
        if (best < 0 || h < bk) { best = p; bk = h; }
      }
    if (best < 0) return;
    uint64_t limit = 0;
    bool limited = false;
    for (int p = 0; p < k; p++)
      if (p != best && at[p] < hi[p]) {
        const uint64_t h = key(p, at[p]);
        const uint64_t lim = p < best ? h : h + 1;   // a later part with the same key waits for this one; an earlier part cannot hold it (it would be `best`)
        if (!limited || lim < limit) { limit = lim; limited = true; }
      }
    size_t i = at[best];
    do { take(best, i, key(best, i)); i++; } while (i < hi[best] && (!limited || key(best, i) < limit));
    at[best] = i;
  }
}

// T - 1 keys that cut the union of the runs into T ranges of about equal size (quantiles of a sample), and where each range begins in each run
template <class Key>
std::vector<std::vector<size_t>> split_runs(int k, const std::vector<size_t>& size, int T, Key&& key) {
  std::vector<uint64_t> sample;
  for (int p = 0; p < k; p++)
    for (int j = 0; j < 256 && size[p]; j++) sample.push_back(key(p, (size_t)((size[p] - 1) * (double)j / 255.0)));
  std::sort(sample.begin(), sample.end());
  std::vector<std::vector<size_t>> cut(T + 1, std::vector<size_t>(k, 0));
  for (int p = 0; p < k; p++) cut[T][p] = size[p];
  for (int t = 1; t < T; t++) {
    const uint64_t split = sample.empty() ? 0 : sample[sample.size() * (size_t)t / T];
    for (int p = 0; p < k; p++) {   // first position of run p whose key is >= split (equal keys of all runs land in one range)
      size_t lo = cut[t - 1][p], hi = size[p];
      while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (key(p, mid) < split) lo = mid + 1;
        else hi = mid;
      }
      cut[t][p] = lo;
    }
  }
  return cut;
}

}  // namespace

static int mesh_merge_parts(const sf_mesh* const* parts, int n_parts, sf_mesh** out);
SF_API int sf_mesh_merge_parts(const sf_mesh* const* parts, int n_parts, sf_mesh** out) {
  if (!out || n_parts < 0 || (!parts && n_parts)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  try { return mesh_merge_parts(parts, n_parts, out); }   // no exception crosses the C ABI
  catch (const std::exception& e) { return sf::fail(SF_ERR_IO, "sf_mesh_merge_parts: %s", e.what()); }
  catch (...) { return sf::fail(SF_ERR_IO, "sf_mesh_merge_parts: unknown exception"); }
}
static int mesh_merge_parts(const sf_mesh* const* parts, int n_parts, sf_mesh** out) {
  const int k = n_parts;
  uint64_t nv = 0, nf = 0;
  bool colour = false, face_keys = k > 0;
  for (int p = 0; p < k; p++) {
    const sf_mesh* m = parts[p];
    if (!m) return sf::fail(SF_ERR_INVALID_ARG, "NULL part %d", p);
    if (m->keys.size() * 3 != m->pos.size()) return sf::fail(SF_ERR_INVALID_ARG, "part %d has no vertex keys (not a marching-cubes mesh)", p);
    nv += m->keys.size();
    nf += m->tri.size() / 3;
    colour |= !m->col.empty();
    face_keys &= m->tkeys.size() * 3 == m->tri.size();
  }
  if (nv >> 32) return sf::fail(SF_ERR_BOUNDS, "%llu vertices in the parts: face indices are 32 bits", (unsigned long long)nv);
  const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)sf::usable_cpus(), 32, (nv + nf) / 200000 + 1}));
  // marching cubes hands every part over with its vertices ascending by key and its faces ascending by cube key: then both merges are k-way merges of
  // sorted runs, cut into T key ranges that T threads merge side by side (a scan-sized mesh has tens of millions of each; one thread sorting them again
  // cost twenty times the extraction).  Parts in any other order are sorted below.  The same sweep checks the face indices.
  std::vector<int> bad_part(T, -1), unsorted(T, 0);
  parallel_for(T, [&](int t) {
    for (int p = 0; p < k; p++) {
      const sf_mesh* m = parts[p];
      const size_t pv = m->keys.size(), n3 = m->tri.size();
      for (size_t i = n3 * t / T; i < n3 * (t + 1) / T; i++)
        if (m->tri[i] >= pv) bad_part[t] = p;
      for (size_t i = std::max<size_t>(1, pv * t / T); i < pv * (t + 1) / T; i++) unsorted[t] |= m->keys[i - 1] > m->keys[i];
      if (face_keys)
        for (size_t i = std::max<size_t>(1, n3 / 3 * t / T); i < n3 / 3 * (t + 1) / T; i++) unsorted[t] |= m->tkeys[i - 1] > m->tkeys[i];
    }
  });
  bool sorted = true;
  for (int t = 0; t < T; t++) {
    if (bad_part[t] >= 0) return sf::fail(SF_ERR_BOUNDS, "part %d: a face index is out of range (%llu vertices)", bad_part[t], (unsigned long long)parts[bad_part[t]]->keys.size());
    sorted &= !unsorted[t];
  }
  std::vector<uint64_t> base(k + 1, 0);
  std::vector<size_t> vsize(k), fsize(k);
  for (int p = 0; p < k; p++) {
    base[p + 1] = base[p] + parts[p]->keys.size();
    vsize[p] = parts[p]->keys.size();
    fsize[p] = parts[p]->tri.size() / 3;
  }
  std::unique_ptr<sf_mesh> r_owner(new sf_mesh());
  sf_mesh* const r = r_owner.get();
  std::vector<uint32_t> remap(nv);   // concatenated vertex index -> merged index
  auto put_vertex = [&](uint64_t u, uint64_t key, int p, size_t v) {
    const sf_mesh* m = parts[p];
    r->keys[u] = key;
    std::memcpy(&r->pos[u * 3], &m->pos[v * 3], 12);
    if (colour) {
      if (m->col.empty()) std::memset(&r->col[u * 4], 255, 4);
      else std::memcpy(&r->col[u * 4], &m->col[v * 4], 4);
    }
  };
  auto size_vertices = [&](uint64_t nu) {
    r->keys.resize(nu);
    r->pos.resize(nu * 3);
    if (colour) r->col.resize(nu * 4);
  };
  auto vkey = [&](int p, size_t i) { return (uint64_t)parts[p]->keys[i]; };
  auto fkey = [&](int p, size_t i) { return (uint64_t)parts[p]->tkeys[i]; };
  if (sorted) {
    const auto cut = split_runs(k, vsize, T, vkey);
    std::vector<uint64_t> first(T + 1, 0);   // merged index of the first vertex of range t: a pass that counts, a pass that writes
    parallel_for(T, [&](int t) {
      uint64_t n = 0, last = 0;
      merge_runs(k, cut[t], cut[t + 1], vkey, [&](int, size_t, uint64_t key) { n += (n == 0 || key != last); last = key; });
      first[t + 1] = n;
    });
    for (int t = 0; t < T; t++) first[t + 1] += first[t];
    size_vertices(first[T]);
    parallel_for(T, [&](int t) {
      uint64_t u = first[t], last = 0;
      bool any = false;
      merge_runs(k, cut[t], cut[t + 1], vkey, [&](int p, size_t i, uint64_t key) {
        if (!any || key != last) put_vertex(u++, key, p, i);
        remap[base[p] + i] = (uint32_t)(u - 1);
        last = key;
        any = true;
      });
    });
  } else {
    struct KV { uint64_t key; uint32_t part, idx; };
    std::vector<KV> all;
    all.reserve(nv);
    for (int p = 0; p < k; p++)
      for (size_t i = 0; i < vsize[p]; i++) all.push_back({parts[p]->keys[i], (uint32_t)p, (uint32_t)i});
    std::sort(all.begin(), all.end(), [](const KV& a, const KV& b) { return a.key != b.key ? a.key < b.key : a.part != b.part ? a.part < b.part : a.idx < b.idx; });
    uint64_t nu = 0;
    for (size_t i = 0; i < all.size(); i++) nu += i == 0 || all[i].key != all[i - 1].key;
    size_vertices(nu);
    uint64_t u = 0;
    for (size_t i = 0; i < all.size(); i++) {
      if (i == 0 || all[i].key != all[i - 1].key) put_vertex(u++, all[i].key, (int)all[i].part, all[i].idx);
      remap[base[all[i].part] + all[i].idx] = (uint32_t)(u - 1);
    }
  }
  r->tri.resize(nf * 3);
  if (face_keys) r->tkeys.resize(nf);
  auto put_face = [&](uint64_t dst, int p, size_t f) {
    const sf_mesh* m = parts[p];
    for (int c = 0; c < 3; c++) r->tri[dst * 3 + c] = remap[base[p] + m->tri[f * 3 + c]];
    if (face_keys) r->tkeys[dst] = m->tkeys[f];
  };
  if (face_keys && k > 1 && sorted) {
    const auto cut = split_runs(k, fsize, T, fkey);
    parallel_for(T, [&](int t) {
      uint64_t dst = 0;
      for (int p = 0; p < k; p++) dst += cut[t][p];
      merge_runs(k, cut[t], cut[t + 1], fkey, [&](int p, size_t i, uint64_t) { put_face(dst++, p, i); });
    });
  } else if (face_keys && k > 1) {
    std::vector<std::pair<uint32_t, uint32_t>> src;   // (part, face) in part order, stably sorted by cube key
    src.reserve(nf);
    for (int p = 0; p < k; p++)
      for (size_t f = 0; f < fsize[p]; f++) src.push_back({(uint32_t)p, (uint32_t)f});
    std::stable_sort(src.begin(), src.end(), [&](const auto& a, const auto& b) { return parts[a.first]->tkeys[a.second] < parts[b.first]->tkeys[b.second]; });
    for (uint64_t dst = 0; dst < nf; dst++) put_face(dst, (int)src[dst].first, src[dst].second);
  } else {   // part after part: cut evenly over the threads
    std::vector<uint64_t> fbase(k + 1, 0);
    for (int p = 0; p < k; p++) fbase[p + 1] = fbase[p] + fsize[p];
    parallel_for(T, [&](int t) {
      for (int p = 0; p < k; p++)
        for (size_t f = fsize[p] * t / T; f < fsize[p] * (t + 1) / T; f++) put_face(fbase[p] + f, p, f);
    });
  }
  *out = r_owner.release();
  return SF_OK;
}

SF_API int sf_mesh_counts(const sf_mesh* m, uint64_t* nv, uint64_t* nf) {
  if (!m) return sf::fail(SF_ERR_INVALID_ARG, "NULL mesh");
  if (nv) *nv = m->pos.size() / 3;
  if (nf) *nf = m->tri.size() / 3;
  return SF_OK;
}

SF_API int sf_mesh_copy_face_keys(const sf_mesh* m, uint64_t* face_keys) {
  if (!m || !face_keys) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (m->tkeys.size() * 3 != m->tri.size()) return sf::fail(SF_ERR_INVALID_ARG, "mesh has no face keys");
  if (!m->tkeys.empty()) std::memcpy(face_keys, m->tkeys.data(), m->tkeys.size() * 8);
  return SF_OK;
}

SF_API int sf_mesh_copy(const sf_mesh* m, float* xyz, uint8_t* rgba, uint32_t* tris, uint64_t* keys) {
  if (!m) return sf::fail(SF_ERR_INVALID_ARG, "NULL mesh");
  // an empty mesh has empty vectors: data() may be NULL, which memcpy must not be handed even for 0 bytes (UBSan, tools/sanitize.py)
  if (xyz && !m->pos.empty()) std::memcpy(xyz, m->pos.data(), m->pos.size() * 4);
  if (rgba) {
    if (m->col.empty()) { if (!m->pos.empty()) std::memset(rgba, 255, m->pos.size() / 3 * 4); }
    else std::memcpy(rgba, m->col.data(), m->col.size());
  }
  if (tris && !m->tri.empty()) std::memcpy(tris, m->tri.data(), m->tri.size() * 4);
  if (keys) {
    if (m->keys.empty()) return sf::fail(SF_ERR_INVALID_ARG, "mesh has no vertex keys");
    std::memcpy(keys, m->keys.data(), m->keys.size() * 8);
  }
  return SF_OK;
}

// The file is header + nv records of 16 bytes + nf records of 13 bytes at known offsets: T threads format their slices into 4 MB buffers and pwrite them
// (one thread filling one zero-initialised vector per element wrote 0.26 GB/s: 0.4 s for the mesh of a scan that is fused in 0.2 s).
SF_API int sf_mesh_write_ply(const sf_mesh* m, const char* path) {
  if (!m || !path) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return sf::fail(SF_ERR_IO, "unable to open file for writing: %s", path);
  const uint64_t nv = m->pos.size() / 3, nf = m->tri.size() / 3;
  char header[512];
  const int hl = std::snprintf(header, sizeof header,
                               "ply\nformat binary_little_endian 1.0\ncomment scanfuse-mi355x\nelement vertex %llu\nproperty float x\nproperty float y\nproperty float z\n"
                               "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nelement face %llu\n"
                               "property list uchar int vertex_indices\nend_header\n",
                               (unsigned long long)nv, (unsigned long long)nf);
  const bool seekable = ::lseek(fd, 0, SEEK_CUR) != (off_t)-1;   // a pipe or a terminal (/dev/stdout): one thread, plain writes in file order
  auto write_at = [fd, seekable](const uint8_t* src, size_t n, uint64_t off) {
    while (n) {
      const ssize_t w = seekable ? ::pwrite(fd, src, n, (off_t)off) : ::write(fd, src, n);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) return false;
      src += w; n -= (size_t)w; off += (uint64_t)w;
    }
    return true;
  };
  std::atomic<bool> ok{write_at((const uint8_t*)header, (size_t)hl, 0)};
  const uint64_t off_v = (uint64_t)hl, off_f = off_v + nv * 16;
  const int T = !seekable ? 1 : (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)sf::usable_cpus(), 16, (nv + nf) / 500000 + 1}));
  const uint64_t CHUNK = 1 << 18;   // records per buffer
  parallel_for(T, [&](int t) {
    sf::mesh_vec<uint8_t> buf(CHUNK * 16);
    for (uint64_t a0 = nv * t / T, a1 = nv * (t + 1) / T; a0 < a1 && ok.load(std::memory_order_relaxed); a0 += CHUNK) {
      const uint64_t n = std::min(CHUNK, a1 - a0);
      for (uint64_t i = 0; i < n; i++) {
        std::memcpy(&buf[i * 16], &m->pos[3 * (a0 + i)], 12);
        if (m->col.empty()) std::memset(&buf[i * 16 + 12], 255, 4);
        else std::memcpy(&buf[i * 16 + 12], &m->col[4 * (a0 + i)], 4);
      }
      if (!write_at(buf.data(), n * 16, off_v + a0 * 16)) ok = false;
    }
    for (uint64_t a0 = nf * t / T, a1 = nf * (t + 1) / T; a0 < a1 && ok.load(std::memory_order_relaxed); a0 += CHUNK) {
      const uint64_t n = std::min(CHUNK, a1 - a0);
      for (uint64_t i = 0; i < n; i++) {
        buf[i * 13] = 3;
        std::memcpy(&buf[i * 13 + 1], &m->tri[3 * (a0 + i)], 12);
      }
      if (!write_at(buf.data(), n * 13, off_f + a0 * 13)) ok = false;
    }
  });
  if (::close(fd) != 0) ok = false;
  if (!ok) return sf::fail(SF_ERR_IO, "write to %s failed", path);
  return SF_OK;
}

SF_API void sf_mesh_free(sf_mesh* m) { delete m; }

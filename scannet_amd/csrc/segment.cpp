// segment.cpp -- mesh over-segmentation (Felzenszwalb-Huttenlocher on vertex normals), host side.
//
// Restates Segmentator/segmentator.cpp: vertex normals as the running mean of unit face normals in FACE ORDER
// (:185-208, cross :107-112 normalises and yields NaN for zero-area faces, lerp :113-116), edge weights
// w = 1 - n_a.n_b, squared on convex edges (:211-229), sort + threshold sweep (:71-92) on a union-find with
// union by rank and one-step path shortening (:24-60), small-segment merge in sorted-edge order (:237-243),
// labels = union-find roots (:246-250).  The labels are bit-exact with the reference binary because
//   * every float operation is a separately rounded IEEE fp32 op in the reference's order (built with
//     -ffp-contract=off; x86-64 -O0 and -O2 builds of the reference agree bit for bit);
//   * the edge order is the permutation libstdc++'s std::sort produces for the weight-only comparator
//     (:67-69) -- ties and NaN weights included -- so the same std::sort is called on the same records.
// What changes is everything around it: parse-free PLY ingest (ply.cpp), flat arrays, no per-element
// std::function dispatch.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "mesh.h"

int mesh_read_any(const char* path, sf_mesh* m, bool* obj_multi);  // ply.cpp

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------
// Layout.  The graph has one edge per face corner; edge 3f + c of face (i, j, k) joins  c = 0: i-j,  c = 1: i-k,  c = 2: k-j  (the order
// segmentator.cpp:199-204 pushes them in).  Nothing but the weight has to travel through the sort: a key is {weight, edge number} = 8 bytes
// -- std::sort's permutation is a function of the comparison results alone (the comparator looks at the weight only, :67-69), so sorting
// 8-byte keys gives the permutation the reference gets for its 12-byte records, ties and NaNs included, with a third less memory moved -- and
// the end points are gathered behind the sort, by all cores, into an array the two sweeps then read front to back.
// Everything touched at random is packed so that one access is one cache line: a vertex's running normal and face count (16 bytes), a set's
// size / threshold / rank (12 bytes, read at roots only); the parent links, which the root walk chases, are an array of their own.
// What is sequential by definition stays sequential: the running mean of the face normals (its value depends on the order of a vertex's
// faces), the sort (ties must fall as libstdc++'s introsort lets them: which vertex ends up as a set's root -- its label -- depends on it)
// and the sweeps.  What is independent per element runs on the cores the process may use: edge weights, end-point gather, label look-up.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct WeightKey {
  float w;
  uint32_t edge;
};
inline bool operator<(const WeightKey& l, const WeightKey& r) { return l.w < r.w; }

struct EdgeEnds {
  uint32_t u, v;
};
inline EdgeEnds edge_ends(const uint32_t* tri, uint32_t e) {
  const uint32_t* t = tri + 3 * (size_t)(e / 3u);
  const uint32_t c = e % 3u;
  return EdgeEnds{c == 2u ? t[2] : t[0], c == 1u ? t[2] : t[1]};
}

struct VertexNormal {
  float x, y, z;
  uint32_t faces;   // incident faces blended in so far
};

// disjoint sets over the vertices: union by rank exactly as segmentator.cpp:43-54 (x under y on a tie, y's rank grows) -- which root survives
// a join decides the label that comes out -- but the look-up is free to shorten paths as it likes (the reference re-points only the node it
// started from, :36-42): a root is a root.  Path halving here.
struct SetInfo {
  uint32_t members;
  float limit;      // the sweep's merge threshold of the set (:82-88), kept beside the size it is computed from
  uint32_t rank;
};
struct VertexSets {
  std::vector<uint32_t> up;
  std::vector<SetInfo> info;   // valid at roots
  VertexSets(size_t n, float limit0) : up(n), info(n, SetInfo{1u, limit0, 0u}) {
    for (size_t i = 0; i < n; i++) up[i] = (uint32_t)i;
  }
  uint32_t root(uint32_t x) {
    while (up[x] != x) {
      up[x] = up[up[x]];
      x = up[x];
    }
    return x;
  }
  uint32_t root_readonly(uint32_t x) const {
    while (up[x] != x) x = up[x];
    return x;
  }
  uint32_t unite(uint32_t x, uint32_t y) {   // both roots; returns the surviving root
    if (info[x].rank > info[y].rank) {
      up[y] = x;
      info[x].members += info[y].members;
      return x;
    }
    up[x] = y;
    info[y].members += info[x].members;
    if (info[x].rank == info[y].rank) info[y].rank++;
    return y;
  }
};

// fn(begin, end) over [0, n) on up to 8 of the CPUs this process may use
template <class Fn>
void in_parallel(size_t n, size_t grain, Fn fn) {
  const size_t want = std::min<size_t>((size_t)std::max(1, std::min(8, sf::usable_cpus())), (n + grain - 1) / std::max<size_t>(grain, 1));
  if (want <= 1) { fn((size_t)0, n); return; }
  const size_t per = (n + want - 1) / want;
  std::vector<std::thread> team;
  for (size_t t = 1; t < want; t++) {
    const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    if (hi > lo) team.emplace_back([=] { fn(lo, hi); });
  }
  fn((size_t)0, std::min(n, per));
  for (std::thread& th : team) th.join();
}

// keys_in: the 3 F weight keys already made (segment_gpu.hip: normals and weights on the GPU), or nullptr: made here
void segment_core(const float* xyz, size_t nv, const uint32_t* tri, size_t nf, float kthr, int min_verts, int32_t* out, std::vector<WeightKey>* keys_in = nullptr) {
  const size_t ne = nf * 3;
  std::vector<WeightKey> keys_own;
  std::vector<WeightKey>& keys = keys_in ? *keys_in : keys_own;
  if (!keys_in) {
  // ---- vertex normals: running mean of the unit face normals in face order (:185-208).  Per face: the unit normal (cross product divided
  // by its length: a zero-area face gives NaN, :107-112), then for each corner n <- t * fn + (1 - t) * n with t = 1 / (faces seen so far + 1)
  // (:113-116; the counts move only after all three corners, :205-207 -- a face that names a vertex twice blends it twice with the same t).
  std::vector<VertexNormal> vn(nv, VertexNormal{0.0f, 0.0f, 0.0f, 0u});
  for (size_t f = 0; f < nf; f++) {
    const uint32_t* t = tri + 3 * f;
    const float* A = xyz + 3 * (size_t)t[0];
    const float* B = xyz + 3 * (size_t)t[1];
    const float* Cc = xyz + 3 * (size_t)t[2];
    const float ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2];
    const float vx = Cc[0] - A[0], vy = Cc[1] - A[1], vz = Cc[2] - A[2];
    float fx = uy * vz - uz * vy, fy = uz * vx - ux * vz, fz = ux * vy - uy * vx;
    const float flen = sqrtf(fx * fx + fy * fy + fz * fz);
    fx /= flen; fy /= flen; fz /= flen;
    for (int c = 0; c < 3; c++) {
      VertexNormal& n = vn[t[c]];
      const float wnew = 1.0f / ((float)n.faces + 1.0f), wold = 1.0f - wnew;
      n.x = wnew * fx + wold * n.x;
      n.y = wnew * fy + wold * n.y;
      n.z = wnew * fz + wold * n.z;
    }
    vn[t[0]].faces++; vn[t[1]].faces++; vn[t[2]].faces++;
  }
  // ---- edge weights (:211-229): 1 - n_u . n_v, squared where the edge is convex (n_v leans along u -> v).  Independent per edge.
  keys.resize(ne);
  in_parallel(ne, 1 << 16, [&](size_t lo, size_t hi) {
    for (size_t e = lo; e < hi; e++) {
      const EdgeEnds ends = edge_ends(tri, (uint32_t)e);
      const float* P = xyz + 3 * (size_t)ends.u;
      const float* Q = xyz + 3 * (size_t)ends.v;
      const VertexNormal &nu = vn[ends.u], &nw = vn[ends.v];
      float ex = Q[0] - P[0], ey = Q[1] - P[1], ez = Q[2] - P[2];
      const float elen = sqrtf(ex * ex + ey * ey + ez * ez);
      ex /= elen; ey /= elen; ez /= elen;
      const float across = nu.x * nw.x + nu.y * nw.y + nu.z * nw.z;
      const float along = nw.x * ex + nw.y * ey + nw.z * ez;
      float w = 1.0f - across;
      if (along > 0) w = w * w;
      keys[e] = WeightKey{w, (uint32_t)e};
    }
  });
  }   // !keys_in
  // ---- the reference's sort call on the reference's comparator (:74); see "Layout" for why the keys may be smaller than its records
  std::sort(keys.begin(), keys.end());
  std::vector<EdgeEnds> sorted_ends(ne);
  in_parallel(ne, 1 << 16, [&](size_t lo, size_t hi) {
    for (size_t q = lo; q < hi; q++) sorted_ends[q] = edge_ends(tri, keys[q].edge);
  });
  // ---- sweep (:76-90): join two sets when the edge is no heavier than either set's threshold; the survivor's threshold becomes the edge
  // weight + k / its size
  VertexSets sets(nv, kthr);
  for (size_t q = 0; q < ne; q++) {
    const uint32_t ru = sets.root(sorted_ends[q].u), rv = sets.root(sorted_ends[q].v);
    if (ru == rv) continue;
    const float w = keys[q].w;
    if (w <= sets.info[ru].limit && w <= sets.info[rv].limit) {
      const uint32_t r = sets.unite(ru, rv);
      sets.info[r].limit = w + (kthr / (float)sets.info[r].members);
    }
  }
  // ---- sets smaller than segMinVerts are joined across any edge, in sorted-edge order (:237-243).  Every vertex is pointed at its root first
  // (in parallel: the links only get shorter), so the look-ups of this pass are one read unless one of its own joins intervenes.
  in_parallel(nv, 1 << 16, [&](size_t lo, size_t hi) {
    for (size_t q = lo; q < hi; q++) {
      // roots are never written here; a non-root's link only ever moves to a node further up its own root path, so a concurrent walker that
      // reads either value arrives at the same root (relaxed atomic accesses: the links are shared between the threads of this loop)
      uint32_t x = (uint32_t)q, p = __atomic_load_n(&sets.up[x], __ATOMIC_RELAXED);
      if (p == x) continue;
      for (uint32_t pp; (pp = __atomic_load_n(&sets.up[p], __ATOMIC_RELAXED)) != p; p = pp) {}
      __atomic_store_n(&sets.up[q], p, __ATOMIC_RELAXED);
    }
  });
  for (size_t q = 0; q < ne; q++) {
    const uint32_t ru = sets.root(sorted_ends[q].u), rv = sets.root(sorted_ends[q].v);
    if (ru != rv && ((int)sets.info[ru].members < min_verts || (int)sets.info[rv].members < min_verts)) sets.unite(ru, rv);
  }
  // ---- label of a vertex = its root (:246-250)
  in_parallel(nv, 1 << 16, [&](size_t lo, size_t hi) {
    for (size_t q = lo; q < hi; q++) out[q] = (int32_t)sets.root_readonly((uint32_t)q);
  });
}

}  // namespace

SF_API int sf_segment_mesh(const float* xyz, uint64_t nv, const uint32_t* tris, uint64_t nf, float kThresh, int segMinVerts, int32_t* out) {
  if ((!xyz && nv) || (!tris && nf) || (!out && nv)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (nv > 0x7FFFFFFFull || nf * 3 > 0x7FFFFFFFull) return sf::fail(SF_ERR_INVALID_ARG, "mesh too large for 32-bit vertex / edge indices");
  for (uint64_t i = 0; i < nf * 3; i++)
    if (tris[i] >= nv) return sf::fail(SF_ERR_BOUNDS, "face index %u out of range (%llu vertices)", tris[i], (unsigned long long)nv);
  segment_core(xyz, (size_t)nv, tris, (size_t)nf, kThresh, segMinVerts, out);
  return SF_OK;
}

int segment_weight_keys_gpu(const float* xyz, size_t nv, const uint32_t* tri, size_t nf, int device, void* keys_out);   // segment_gpu.hip

// The same labels with the vertex normals and the edge weights computed on GPU `device` (segment_gpu.hip: a lane per vertex walks its faces in face order, a
// lane per edge): the weight keys are the host's bit for bit, the sort and the sweeps are the host's.  No fallback: without a device the call fails.
SF_API int sf_segment_mesh_gpu(const float* xyz, uint64_t nv, const uint32_t* tris, uint64_t nf, float kThresh, int segMinVerts, int device, int32_t* out) {
  if ((!xyz && nv) || (!tris && nf) || (!out && nv)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (nv > 0x7FFFFFFFull || nf * 3 > 0x7FFFFFFFull) return sf::fail(SF_ERR_INVALID_ARG, "mesh too large for 32-bit vertex / edge indices");
  for (uint64_t i = 0; i < nf * 3; i++)
    if (tris[i] >= nv) return sf::fail(SF_ERR_BOUNDS, "face index %u out of range (%llu vertices)", tris[i], (unsigned long long)nv);
  try {
    std::vector<WeightKey> keys((size_t)nf * 3);
    static_assert(sizeof(WeightKey) == 8, "the device writes {float weight, uint32 edge} records");
    const int rc = segment_weight_keys_gpu(xyz, (size_t)nv, tris, (size_t)nf, device, keys.data());
    if (rc != SF_OK) return rc;
    segment_core(xyz, (size_t)nv, tris, (size_t)nf, kThresh, segMinVerts, out, &keys);
  } catch (const std::bad_alloc&) {
    return sf::fail(SF_ERR_IO, "sf_segment_mesh_gpu: out of memory");
  }
  return SF_OK;
}

// JSON surface, segmentator.cpp:253-266: no whitespace, kThresh through ostream<<float, ints as decimal
static int write_segs_json(const std::string& file, const std::string& scene, float kthr, int min_verts, const std::vector<int32_t>& seg) {
  std::string body;
  body.reserve(seg.size() * 8 + 256);
  {
    std::ostringstream hs;
    hs << "{" << "\"params\":{\"kThresh\":" << kthr << ",\"segMinVerts\":" << min_verts << "},"
       << "\"sceneId\":\"" << scene << "\"," << "\"segIndices\":[";
    body = hs.str();
  }
  char num[16];
  for (size_t i = 0; i < seg.size(); i++) {
    if (i) body.push_back(',');
    const int n = std::snprintf(num, sizeof(num), "%d", seg[i]);
    body.append(num, (size_t)n);
  }
  body += "]}";
  std::ofstream ofs(file, std::ios::binary);
  if (!ofs) return sf::fail(SF_ERR_IO, "unable to open file for writing: %s", file.c_str());
  ofs.write(body.data(), (std::streamsize)body.size());
  ofs.close();
  if (!ofs) return sf::fail(SF_ERR_IO, "write to %s failed", file.c_str());
  return SF_OK;
}

// shared with the drop-in CLI (tool_segmentator.cpp) through the C ABI below
static int segment_file(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                        uint64_t* counts4 /* vertexCount, verts.size, faceCount, faces.size */, char* out_path, uint64_t out_path_cap,
                        int* obj_multi, int device /* < 0: the host path */) {
  if (!mesh_path) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_mesh m;
  bool multi = false;
  int rc = mesh_read_any(mesh_path, &m, &multi);
  if (rc != SF_OK) return rc;
  if (obj_multi) *obj_multi = multi ? 1 : 0;
  const uint64_t nv = m.pos.size() / 3, nf = m.tri.size() / 3;
  if (counts4) { counts4[0] = nv; counts4[1] = m.pos.size(); counts4[2] = nf; counts4[3] = m.tri.size(); }
  std::vector<int32_t> seg(nv);
  rc = device < 0 ? sf_segment_mesh(m.pos.data(), nv, m.tri.data(), nf, kThresh, segMinVerts, seg.data())
                  : sf_segment_mesh_gpu(m.pos.data(), nv, m.tri.data(), nf, kThresh, segMinVerts, device, seg.data());
  if (rc != SF_OK) return rc;
  if (num_segments) {
    std::unordered_set<int32_t> ids(seg.begin(), seg.end());
    *num_segments = ids.size();
  }
  const std::string ply(mesh_path);
  // naming + sceneId exactly as segmentator.cpp:281-285 (sceneId keeps the leading '/' when the path has one)
  const std::string base = ply.substr(0, ply.find_last_of("."));
  const int lastslash = (int)ply.find_last_of("/");
  const std::string scene = lastslash > 0 ? base.substr((size_t)lastslash) : base;
  const std::string file = out_json ? std::string(out_json) : base + "." + std::to_string(kThresh) + ".segs.json";
  if (out_path && out_path_cap) { std::strncpy(out_path, file.c_str(), (size_t)out_path_cap - 1); out_path[out_path_cap - 1] = 0; }
  return write_segs_json(file, scene, kThresh, segMinVerts, seg);
}

SF_API int sf_segment_file_ex(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                              uint64_t* counts4, char* out_path, uint64_t out_path_cap, int* obj_multi) {
  return segment_file(mesh_path, kThresh, segMinVerts, out_json, num_segments, counts4, out_path, out_path_cap, obj_multi, -1);
}
// ... with the normals and the edge weights on GPU `device` (sf_segment_mesh_gpu): the same file
SF_API int sf_segment_file_gpu(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                               uint64_t* counts4, char* out_path, uint64_t out_path_cap, int* obj_multi, int device) {
  if (device < 0) return sf::fail(SF_ERR_INVALID_ARG, "sf_segment_file_gpu: device %d", device);
  return segment_file(mesh_path, kThresh, segMinVerts, out_json, num_segments, counts4, out_path, out_path_cap, obj_multi, device);
}

SF_API int sf_segment_file(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments) {
  return sf_segment_file_ex(mesh_path, kThresh, segMinVerts, out_json, num_segments, nullptr, nullptr, 0, nullptr);
}

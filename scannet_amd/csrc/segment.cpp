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
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "mesh.h"

int mesh_read_any(const char* path, sf_mesh* m, bool* obj_multi);  // ply.cpp

namespace {

struct GraphEdge {
  float w;
  int a, b;
};
inline bool operator<(const GraphEdge& l, const GraphEdge& r) { return l.w < r.w; }

struct Node {
  int rank, parent, size;
};

struct Forest {
  std::vector<Node> n;
  explicit Forest(int count) : n((size_t)count) {
    for (int i = 0; i < count; i++) n[i] = Node{0, i, 1};
  }
  int find(int x) {
    int r = x;
    while (r != n[r].parent) r = n[r].parent;
    n[x].parent = r;  // only the start node is re-pointed (segmentator.cpp:36-42)
    return r;
  }
  void join(int x, int y) {
    if (n[x].rank > n[y].rank) {
      n[y].parent = x;
      n[x].size += n[y].size;
    } else {
      n[x].parent = y;
      n[y].size += n[x].size;
      if (n[x].rank == n[y].rank) n[y].rank++;
    }
  }
};

struct V3 {
  float x, y, z;
};
inline V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 unit_cross(const V3& u, const V3& v) {
  V3 c{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  const float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
  c.x /= len; c.y /= len; c.z /= len;
  return c;
}
inline V3 blend(const V3& a, const V3& b, float t) {
  const float s = 1.0f - t;
  return V3{t * b.x + s * a.x, t * b.y + s * a.y, t * b.z + s * a.z};
}

void segment_core(const float* xyz, size_t nv, const uint32_t* tri, size_t nf, float kthr, int min_verts, int32_t* out) {
  std::vector<V3> points(nv, V3{0, 0, 0}), normals(nv, V3{0, 0, 0});
  std::vector<int> counts(nv, 0);
  const size_t ne = nf * 3;
  std::vector<GraphEdge> edges(ne);
  for (size_t f = 0; f < nf; f++) {
    const uint32_t i1 = tri[3 * f], i2 = tri[3 * f + 1], i3 = tri[3 * f + 2];
    const V3 p1{xyz[3 * (size_t)i1], xyz[3 * (size_t)i1 + 1], xyz[3 * (size_t)i1 + 2]};
    const V3 p2{xyz[3 * (size_t)i2], xyz[3 * (size_t)i2 + 1], xyz[3 * (size_t)i2 + 2]};
    const V3 p3{xyz[3 * (size_t)i3], xyz[3 * (size_t)i3 + 1], xyz[3 * (size_t)i3 + 2]};
    points[i1] = p1; points[i2] = p2; points[i3] = p3;
    edges[3 * f].a = (int)i1;     edges[3 * f].b = (int)i2;
    edges[3 * f + 1].a = (int)i1; edges[3 * f + 1].b = (int)i3;
    edges[3 * f + 2].a = (int)i3; edges[3 * f + 2].b = (int)i2;
    const V3 fn = unit_cross(sub(p2, p1), sub(p3, p1));
    normals[i1] = blend(normals[i1], fn, 1.0f / (counts[i1] + 1.0f));
    normals[i2] = blend(normals[i2], fn, 1.0f / (counts[i2] + 1.0f));
    normals[i3] = blend(normals[i3], fn, 1.0f / (counts[i3] + 1.0f));
    counts[i1]++; counts[i2]++; counts[i3]++;
  }
  for (size_t e = 0; e < ne; e++) {
    const V3& n1 = normals[edges[e].a];
    const V3& n2 = normals[edges[e].b];
    const V3& p1 = points[edges[e].a];
    const V3& p2 = points[edges[e].b];
    float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
    const float dd = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= dd; dy /= dd; dz /= dd;
    const float dot = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
    const float dot2 = n2.x * dx + n2.y * dy + n2.z * dz;
    float ww = 1.0f - dot;
    if (dot2 > 0) ww = ww * ww;
    edges[e].w = ww;
  }
  std::sort(edges.begin(), edges.end());
  Forest u((int)nv);
  {
    std::vector<float> thr(nv, kthr);
    for (size_t e = 0; e < ne; e++) {
      int a = u.find(edges[e].a);
      const int b = u.find(edges[e].b);
      if (a != b && edges[e].w <= thr[a] && edges[e].w <= thr[b]) {
        u.join(a, b);
        a = u.find(a);
        thr[a] = edges[e].w + (kthr / u.n[a].size);
      }
    }
  }
  for (size_t e = 0; e < ne; e++) {
    const int a = u.find(edges[e].a), b = u.find(edges[e].b);
    if (a != b && (u.n[a].size < min_verts || u.n[b].size < min_verts)) u.join(a, b);
  }
  for (size_t q = 0; q < nv; q++) out[q] = u.find((int)q);
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
SF_API int sf_segment_file_ex(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                              uint64_t* counts4 /* vertexCount, verts.size, faceCount, faces.size */, char* out_path, uint64_t out_path_cap,
                              int* obj_multi) {
  if (!mesh_path) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_mesh m;
  bool multi = false;
  int rc = mesh_read_any(mesh_path, &m, &multi);
  if (rc != SF_OK) return rc;
  if (obj_multi) *obj_multi = multi ? 1 : 0;
  const uint64_t nv = m.pos.size() / 3, nf = m.tri.size() / 3;
  if (counts4) { counts4[0] = nv; counts4[1] = m.pos.size(); counts4[2] = nf; counts4[3] = m.tri.size(); }
  std::vector<int32_t> seg(nv);
  rc = sf_segment_mesh(m.pos.data(), nv, m.tri.data(), nf, kThresh, segMinVerts, seg.data());
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

SF_API int sf_segment_file(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments) {
  return sf_segment_file_ex(mesh_path, kThresh, segMinVerts, out_json, num_segments, nullptr, nullptr, 0, nullptr);
}

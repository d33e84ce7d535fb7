// clean.cpp -- the mesh-cleaning filters between `<id>_vh.ply` and `<id>_vh_clean.ply`, host side.
//
// Replaces `meshlabserver -i X_vh.ply -o X_vh_clean.ply -m vc -s clean.mlx` (Server/scan_processor.py:143) for the
// filter scripts the pipeline ships (Server/tools/meshclean/clean.mlx:3-10, cleanLoRes.mlx:3-10):
//   1. "Merge Close Vertices"  Threshold (RichAbsPerc, ABSOLUTE value 0.0010689)
//   2. "Remove Duplicate Faces"
//   3. "Remove Isolated pieces (wrt Face Num.)"  MinComponentSize 7500 (clean.mlx) / 1000 (cleanLoRes.mlx)
//   4. "Remove Unreferenced Vertex"
// MeshLab / VCG are not in the reference tree and no version is pinned (Server/config.py:19 names an installed
// `VCG\MeshLab\meshlabserver.exe`), so the semantics below restate the published VCG algorithms
// (vcg/complex/algorithms/clean.h) -- PARITY UNPINNED; the checker is oracle/clean_oracle.py:
//   1. tri::Clean::MergeCloseVertex = ClusterVertex + RemoveDuplicateVertex(removeDegenerate = true): vertices are
//      visited in index order; an unvisited vertex becomes a cluster centre and every still unvisited vertex at
//      Euclidean distance < threshold (float: sqrt(dx*dx + dy*dy + dz*dz), strict) is moved onto it and marked
//      visited -- greedy, not transitive.  Vertices with identical positions are then merged into the lowest index
//      (the centre, which also keeps its colour), faces are re-indexed and faces with a repeated vertex are dropped.
//   2. tri::Clean::RemoveDuplicateFace: of the faces with the same vertex SET (any orientation) one survives; VCG's
//      choice among them follows an unstable sort, here the lowest face index survives.
//   3. tri::Clean::RemoveSmallConnectedComponentsSize: components of the face-face adjacency (two faces are adjacent
//      when they share an edge, non-manifold edges connect all their faces); components with FEWER than
//      MinComponentSize faces are deleted.
//   4. tri::Clean::RemoveUnreferencedVertex, then compaction in index order (what the PLY exporter writes).
// The clustering sweep and the component labelling are sequential / pointer-chasing work on a ~1 M face mesh
// (milliseconds on the host); there is no bandwidth-bound kernel here to move to the GPU.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "mesh.h"

namespace {

struct GridKey {
  int64_t x, y, z;
  bool operator==(const GridKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct GridHash {
  size_t operator()(const GridKey& k) const {
    uint64_t h = (uint64_t)k.x * 0x9E3779B97F4A7C15ull ^ ((uint64_t)k.y * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull) ^ ((uint64_t)k.z * 0xD6E8FEB86659FD93ull);
    h ^= h >> 29;
    return (size_t)(h * 0xBF58476D1CE4E5B9ull);
  }
};

uint32_t uf_find(std::vector<uint32_t>& p, uint32_t x) {
  while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
  return x;
}

// filter 1: returns for every vertex the index of the vertex it is merged into (itself for survivors)
void merge_close(const std::vector<float>& pos, float radius, std::vector<uint32_t>& target, uint64_t* merged) {
  const size_t nv = pos.size() / 3;
  target.resize(nv);
  std::iota(target.begin(), target.end(), 0u);
  if (nv == 0) return;
  std::vector<float> p(pos);  // positions move while clustering (members take the centre's position)
  if (radius > 0.0f) {
    // uniform grid with cell = radius: the candidates of a centre are in its 27-neighbourhood
    const double inv = 1.0 / (double)radius;
    std::unordered_map<GridKey, std::vector<uint32_t>, GridHash> grid;
    grid.reserve(nv);
    auto cell = [&](const float* q) { return GridKey{(int64_t)std::floor((double)q[0] * inv), (int64_t)std::floor((double)q[1] * inv), (int64_t)std::floor((double)q[2] * inv)}; };
    for (size_t i = 0; i < nv; i++) grid[cell(&pos[3 * i])].push_back((uint32_t)i);
    std::vector<uint8_t> visited(nv, 0);
    for (size_t i = 0; i < nv; i++) {
      if (visited[i]) continue;
      visited[i] = 1;
      const float cx = p[3 * i], cy = p[3 * i + 1], cz = p[3 * i + 2];
      const GridKey c = cell(&pos[3 * i]);  // an unvisited vertex still sits at its original position
      for (int64_t dz = -1; dz <= 1; dz++)
        for (int64_t dy = -1; dy <= 1; dy++)
          for (int64_t dx = -1; dx <= 1; dx++) {
            auto it = grid.find(GridKey{c.x + dx, c.y + dy, c.z + dz});
            if (it == grid.end()) continue;
            for (uint32_t j : it->second) {
              if (visited[j]) continue;
              const float ex = cx - p[3 * j], ey = cy - p[3 * j + 1], ez = cz - p[3 * j + 2];
              const float dist = std::sqrt(ex * ex + ey * ey + ez * ez);
              if (dist < radius) {
                visited[j] = 1;
                p[3 * j] = cx; p[3 * j + 1] = cy; p[3 * j + 2] = cz;
                if (merged) (*merged)++;
              }
            }
          }
    }
  }
  // RemoveDuplicateVertex: identical positions collapse into the lowest index
  std::vector<uint32_t> order(nv);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const int c = std::memcmp(&p[3 * a], &p[3 * b], 12);  // any total order on the bit patterns groups equal positions
    return c != 0 ? c < 0 : a < b;
  });
  for (size_t i = 0; i < nv;) {
    size_t j = i + 1;
    while (j < nv && p[3 * order[j]] == p[3 * order[i]] && p[3 * order[j] + 1] == p[3 * order[i] + 1] && p[3 * order[j] + 2] == p[3 * order[i] + 2]) j++;
    // -0.0 == +0.0 compares equal but memcmp separates them: scan the (tiny) run for the minimum index instead of trusting the sort
    uint32_t lo = order[i];
    for (size_t k = i; k < j; k++) lo = std::min(lo, order[k]);
    for (size_t k = i; k < j; k++) target[order[k]] = lo;
    i = j;
  }
}

}  // namespace

SF_API int sf_mesh_clean(const sf_mesh* in, float merge_distance, uint32_t min_component_faces, sf_mesh** out, sf_clean_stats* stats) {
  if (!in || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!(merge_distance >= 0.0f)) return sf::fail(SF_ERR_INVALID_ARG, "merge distance must be >= 0");
  const size_t nv = in->pos.size() / 3, nf = in->tri.size() / 3;
  sf_clean_stats st;
  std::memset(&st, 0, sizeof(st));
  st.vertices_in = nv;
  st.faces_in = nf;
  // ---- 1. merge close vertices (+ degenerate faces)
  std::vector<uint32_t> target;
  uint64_t moved = 0;
  merge_close(in->pos, merge_distance, target, &moved);
  for (size_t i = 0; i < nv; i++) st.vertices_merged += target[i] != i;
  std::vector<uint32_t> tri;
  tri.reserve(in->tri.size());
  for (size_t f = 0; f < nf; f++) {
    const uint32_t a = target[in->tri[3 * f]], b = target[in->tri[3 * f + 1]], c = target[in->tri[3 * f + 2]];
    if (a == b || b == c || a == c) { st.faces_degenerate++; continue; }
    tri.push_back(a); tri.push_back(b); tri.push_back(c);
  }
  // ---- 2. duplicate faces: same vertex set, lowest face index survives
  {
    const size_t n = tri.size() / 3;
    struct Key { uint32_t v[3]; uint32_t f; };
    std::vector<Key> keys(n);
    for (size_t f = 0; f < n; f++) {
      uint32_t v[3] = {tri[3 * f], tri[3 * f + 1], tri[3 * f + 2]};
      std::sort(v, v + 3);
      keys[f] = {{v[0], v[1], v[2]}, (uint32_t)f};
    }
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
      if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
      if (a.v[1] != b.v[1]) return a.v[1] < b.v[1];
      if (a.v[2] != b.v[2]) return a.v[2] < b.v[2];
      return a.f < b.f;
    });
    std::vector<uint8_t> dead(n, 0);
    for (size_t i = 1; i < n; i++)
      if (keys[i].v[0] == keys[i - 1].v[0] && keys[i].v[1] == keys[i - 1].v[1] && keys[i].v[2] == keys[i - 1].v[2]) { dead[keys[i].f] = 1; st.faces_duplicate++; }
    size_t w = 0;
    for (size_t f = 0; f < n; f++)
      if (!dead[f]) { tri[3 * w] = tri[3 * f]; tri[3 * w + 1] = tri[3 * f + 1]; tri[3 * w + 2] = tri[3 * f + 2]; w++; }
    tri.resize(3 * w);
  }
  // ---- 3. small connected components (faces adjacent across shared edges)
  {
    const size_t n = tri.size() / 3;
    std::vector<uint32_t> parent(n);
    std::iota(parent.begin(), parent.end(), 0u);
    struct Edge { uint32_t a, b, f; };
    std::vector<Edge> edges(3 * n);
    for (size_t f = 0; f < n; f++)
      for (int e = 0; e < 3; e++) {
        uint32_t a = tri[3 * f + e], b = tri[3 * f + (e + 1) % 3];
        if (a > b) std::swap(a, b);
        edges[3 * f + e] = {a, b, (uint32_t)f};
      }
    std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.f < y.f); });
    for (size_t i = 1; i < edges.size(); i++)
      if (edges[i].a == edges[i - 1].a && edges[i].b == edges[i - 1].b) {
        const uint32_t ra = uf_find(parent, edges[i].f), rb = uf_find(parent, edges[i - 1].f);
        if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
      }
    std::vector<uint32_t> size(n, 0);
    for (size_t f = 0; f < n; f++) size[uf_find(parent, (uint32_t)f)]++;
    for (size_t f = 0; f < n; f++)
      if (parent[f] == f) { st.components_in++; if (size[f] < min_component_faces) st.components_removed++; }
    size_t w = 0;
    for (size_t f = 0; f < n; f++) {
      if (size[uf_find(parent, (uint32_t)f)] < min_component_faces) { st.faces_small_component++; continue; }
      tri[3 * w] = tri[3 * f]; tri[3 * w + 1] = tri[3 * f + 1]; tri[3 * w + 2] = tri[3 * f + 2]; w++;
    }
    tri.resize(3 * w);
  }
  // ---- 4. unreferenced vertices, compaction in index order
  std::vector<uint32_t> remap(nv, 0xFFFFFFFFu);
  {
    std::vector<uint8_t> used(nv, 0);
    for (uint32_t v : tri) used[v] = 1;
    uint32_t w = 0;
    for (size_t i = 0; i < nv; i++)
      if (used[i]) remap[i] = w++;
    st.vertices_out = w;
  }
  sf_mesh* m = new sf_mesh();
  m->pos.resize((size_t)st.vertices_out * 3);
  if (!in->col.empty()) m->col.resize((size_t)st.vertices_out * 4);
  for (size_t i = 0; i < nv; i++) {
    if (remap[i] == 0xFFFFFFFFu) continue;
    std::memcpy(&m->pos[3 * (size_t)remap[i]], &in->pos[3 * i], 12);  // a surviving vertex is a cluster centre: it never moved
    if (!in->col.empty()) std::memcpy(&m->col[4 * (size_t)remap[i]], &in->col[4 * i], 4);
  }
  m->tri.resize(tri.size());
  for (size_t i = 0; i < tri.size(); i++) m->tri[i] = remap[tri[i]];
  st.faces_out = tri.size() / 3;
  st.vertices_unreferenced = nv - st.vertices_merged - st.vertices_out;
  if (stats) *stats = st;
  *out = m;
  return SF_OK;
}

// ---- MeshLab filter scripts (.mlx): the subset the pipeline ships -------------------------------------------------
namespace {

std::string attr(const std::string& tag, const char* name) {
  const std::string pat = std::string(name) + "=\"";
  size_t p = 0;
  while ((p = tag.find(pat, p)) != std::string::npos) {
    if (p == 0 || tag[p - 1] == ' ' || tag[p - 1] == '\t' || tag[p - 1] == '\n') {
      const size_t b = p + pat.size(), e = tag.find('"', b);
      return e == std::string::npos ? std::string() : tag.substr(b, e - b);
    }
    p += pat.size();
  }
  return std::string();
}

}  // namespace

SF_API int sf_mlx_load(const char* path, sf_clean_script* out) {
  if (!path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::ifstream in(path);
  if (!in) return sf::fail(SF_ERR_IO, "could not open filter script %s", path);
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  if (text.find("<FilterScript") == std::string::npos) return sf::fail(SF_ERR_FORMAT, "%s is not a MeshLab FilterScript", path);
  std::memset(out, 0, sizeof(*out));
  int stage = 0;  // the four filters must come in the shipped order; each may appear at most once
  size_t p = 0;
  std::string current;
  while ((p = text.find('<', p)) != std::string::npos) {
    const size_t e = text.find('>', p);
    if (e == std::string::npos) break;
    const std::string tag = text.substr(p + 1, e - p - 1);
    p = e + 1;
    if (tag.compare(0, 7, "filter ") == 0) {
      current = attr(tag, "name");
      int want;
      if (current == "Merge Close Vertices") { want = 1; out->merge_close_vertices = 1; }
      else if (current == "Remove Duplicate Faces") { want = 2; out->remove_duplicate_faces = 1; }
      else if (current == "Remove Isolated pieces (wrt Face Num.)") { want = 3; out->remove_small_components = 1; }
      else if (current == "Remove Unreferenced Vertex") { want = 4; out->remove_unreferenced = 1; }
      else return sf::fail(SF_ERR_UNSUPPORTED, "filter \"%s\" is not implemented (only the clean.mlx / cleanLoRes.mlx filters are)", current.c_str());
      if (want <= stage) return sf::fail(SF_ERR_UNSUPPORTED, "filter \"%s\" out of the clean.mlx order", current.c_str());
      stage = want;
    } else if (tag.compare(0, 6, "Param ") == 0) {
      const std::string name = attr(tag, "name"), value = attr(tag, "value");
      if (current == "Merge Close Vertices" && name == "Threshold") out->merge_distance = (float)std::atof(value.c_str());
      if (current == "Remove Isolated pieces (wrt Face Num.)" && name == "MinComponentSize") out->min_component_faces = (uint32_t)std::atol(value.c_str());
    }
  }
  if (stage == 0) return sf::fail(SF_ERR_FORMAT, "%s holds no filter", path);
  return SF_OK;
}

SF_API int sf_mesh_clean_script(const sf_mesh* in, const sf_clean_script* s, sf_mesh** out, sf_clean_stats* stats) {
  if (!s) return sf::fail(SF_ERR_INVALID_ARG, "NULL script");
  // a filter that is absent from the script degenerates to a no-op parameter
  const float dist = s->merge_close_vertices ? s->merge_distance : -1.0f;
  if (!s->remove_duplicate_faces || !s->remove_unreferenced)
    return sf::fail(SF_ERR_UNSUPPORTED, "scripts without \"Remove Duplicate Faces\" / \"Remove Unreferenced Vertex\" are not supported");
  if (dist < 0.0f) return sf::fail(SF_ERR_UNSUPPORTED, "scripts without \"Merge Close Vertices\" are not supported");
  return sf_mesh_clean(in, dist, s->remove_small_components ? s->min_component_faces : 0u, out, stats);
}

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
// The clustering sweep (greedy in index order) and the component labelling are sequential / pointer-chasing work:
// ~0.5 s on the host for a 3.4 M face mesh, 1.7 s for 7.9 M (SF_CLEAN_TIMING=1 prints the split).  clean_gpu.hip holds the same filters
// as sorts, a round-by-round resolution of the greedy clustering and a lock-free union-find (sf_mesh_clean_gpu: identical output).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "mesh.h"

namespace {

// open-addressing map u64 -> u32 (keys are never ~0)
struct FlatMap {
  std::vector<uint64_t> keys;
  std::vector<uint32_t> vals;
  std::vector<uint64_t> home;   // one bit per slot: some key has this slot as its home -- a cache-resident filter in front of the
                                // table (most look-ups of the clustering sweep are for cells that do not exist: each one a DRAM miss otherwise)
  uint64_t mask = 0;
  explicit FlatMap(size_t n) {
    size_t cap = 64;
    while (cap < 2 * n + 2) cap <<= 1;
    keys.assign(cap, ~0ull);
    vals.assign(cap, 0u);
    home.assign(cap / 64, 0ull);
    mask = cap - 1;
  }
  static uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
  // returns the slot of `k`, inserting (k, v) when absent; *fresh tells which
  size_t find_or_insert(uint64_t k, uint32_t v, bool* fresh) {
    size_t i = (size_t)(mix(k) & mask);
    for (;;) {
      if (keys[i] == k) { *fresh = false; return i; }
      if (keys[i] == ~0ull) {
        keys[i] = k; vals[i] = v; *fresh = true;
        const size_t h = (size_t)(mix(k) & mask);
        home[h >> 6] |= 1ull << (h & 63);
        return i;
      }
      i = (i + 1) & mask;
    }
  }
  const uint32_t* find(uint64_t k) const {
    size_t i = (size_t)(mix(k) & mask);
    if (!((home[i >> 6] >> (i & 63)) & 1ull)) return nullptr;
    for (;;) {
      if (keys[i] == k) return &vals[i];
      if (keys[i] == ~0ull) return nullptr;
      i = (i + 1) & mask;
    }
  }
};

uint32_t uf_find(std::vector<uint32_t>& p, uint32_t x) {
  while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
  return x;
}

// filter 1: returns for every vertex the index of the vertex it is merged into (itself for survivors)
int merge_close(const sf::mesh_vec<float>& pos, float radius, std::vector<uint32_t>& target) {
  const bool timing = std::getenv("SF_CLEAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::printf("clean:   merge: %-19s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  const size_t nv = pos.size() / 3;
  target.resize(nv);
  std::iota(target.begin(), target.end(), 0u);
  if (nv == 0) return SF_OK;
  if (radius > 0.0f) {
    // uniform grid with cell = 2 * radius: the ball of radius r around a point touches at most the 2 x 2 x 2 cells on the
    // point's side of its cell centre (8 look-ups instead of 27); vertices bucketed by a counting sort on the packed
    // cell key, one flat hash from cell key to bucket
    // (cells 1e-5 larger than 2 r: the float distance test below may accept a point whose exact distance is a few 1e-7
    // relative beyond r -- it must still fall inside the 8 cells)
    const double inv = 0.5 / ((double)radius * (1.0 + 1e-5));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < nv; i++)
      for (int c = 0; c < 3; c++) {
        const double q = (double)pos[3 * i + c];
        if (!(q == q) || q > 1e30 || q < -1e30) return sf::fail(SF_ERR_FORMAT, "vertex %zu has a non-finite coordinate", i);
        lo[c] = std::min(lo[c], q); hi[c] = std::max(hi[c], q);
      }
    int64_t base[3];
    for (int c = 0; c < 3; c++) {
      base[c] = (int64_t)std::floor(lo[c] * inv) - 1;
      if ((int64_t)std::floor(hi[c] * inv) + 1 - base[c] >= (1ll << 21))
        return sf::fail(SF_ERR_UNSUPPORTED, "mesh extent / merge distance exceeds 2^21 cells per axis");
    }
    // cell index and which half of the cell (-1 / +1) per axis
    auto cell = [&](const float* q, int64_t* c3, int64_t* side) {
      for (int c = 0; c < 3; c++) {
        const double u = (double)q[c] * inv, fl = std::floor(u);
        c3[c] = (int64_t)fl - base[c];
        side[c] = (u - fl) < 0.5 ? -1 : 1;
      }
    };
    auto pack = [](int64_t x, int64_t y, int64_t z) { return ((uint64_t)z << 42) | ((uint64_t)y << 21) | (uint64_t)x; };
    FlatMap cells(nv);
    std::vector<uint32_t> count;  // per cell id
    std::vector<uint32_t> cell_id(nv);
    for (size_t i = 0; i < nv; i++) {
      int64_t c3[3], side[3];
      cell(&pos[3 * i], c3, side);
      bool fresh;
      const size_t slot = cells.find_or_insert(pack(c3[0], c3[1], c3[2]), (uint32_t)count.size(), &fresh);
      if (fresh) count.push_back(0);
      cell_id[i] = cells.vals[slot];
      count[cell_id[i]]++;
    }
    lap("cells");
    std::vector<uint32_t> start(count.size() + 1, 0);
    for (size_t c = 0; c < count.size(); c++) start[c + 1] = start[c] + count[c];
    std::vector<uint32_t> members(nv), fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < nv; i++) members[fill[cell_id[i]]++] = (uint32_t)i;  // index order inside a cell
    lap("buckets");
    std::vector<uint8_t> visited(nv, 0);
    for (size_t i = 0; i < nv; i++) {
      if (visited[i]) continue;
      visited[i] = 1;
      const float cx = pos[3 * i], cy = pos[3 * i + 1], cz = pos[3 * i + 2];  // a centre never moved
      int64_t c3[3], side[3];
      cell(&pos[3 * i], c3, side);
      for (int oz = 0; oz < 2; oz++)
        for (int oy = 0; oy < 2; oy++)
          for (int ox = 0; ox < 2; ox++) {
            const uint32_t* id = (ox | oy | oz) == 0 ? &cell_id[i] : cells.find(pack(c3[0] + ox * side[0], c3[1] + oy * side[1], c3[2] + oz * side[2]));
            if (!id) continue;
            for (uint32_t k = start[*id]; k < start[*id + 1]; k++) {
              const uint32_t j = members[k];
              if (visited[j]) continue;  // unvisited vertices still sit at their original position
              const float ex = cx - pos[3 * j], ey = cy - pos[3 * j + 1], ez = cz - pos[3 * j + 2];
              const float dist = std::sqrt(ex * ex + ey * ey + ez * ez);
              if (dist < radius) { visited[j] = 1; target[j] = (uint32_t)i; }
            }
          }
    }
    lap("sweep");
    // RemoveDuplicateVertex: members now share their centre's position; two centres never coincide (distance 0 < radius
    // would have clustered them), so the cluster assignment IS the duplicate-vertex merge.
    return SF_OK;
  }
  // radius == 0: only bit-for-bit coincident vertices merge (RemoveDuplicateVertex), lowest index survives
  std::vector<uint32_t> order(nv);
  std::iota(order.begin(), order.end(), 0u);
  auto canon = [&](uint32_t v, float* q) { for (int c = 0; c < 3; c++) q[c] = pos[3 * v + c] + 0.0f; };  // -0.0 -> +0.0
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    float p[3], q[3];
    canon(x, p); canon(y, q);
    const int c = std::memcmp(p, q, 12);
    return c != 0 ? c < 0 : x < y;
  });
  for (size_t i = 0; i < nv;) {
    size_t j = i + 1;
    float p[3], q[3];
    canon(order[i], p);
    while (j < nv && (canon(order[j], q), std::memcmp(p, q, 12) == 0)) j++;
    for (size_t k = i; k < j; k++) target[order[k]] = order[i];  // ties sorted by index: order[i] is the lowest
    i = j;
  }
  return SF_OK;
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
  const bool timing = std::getenv("SF_CLEAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::printf("clean: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // ---- 1. merge close vertices (+ degenerate faces)
  std::vector<uint32_t> target;
  {
    const int rc = merge_close(in->pos, merge_distance, target);
    if (rc != SF_OK) return rc;
  }
  for (size_t i = 0; i < nv; i++) st.vertices_merged += target[i] != i;
  std::vector<uint32_t> tri;
  tri.reserve(in->tri.size());
  for (size_t f = 0; f < nf; f++) {
    const uint32_t a = target[in->tri[3 * f]], b = target[in->tri[3 * f + 1]], c = target[in->tri[3 * f + 2]];
    if (a == b || b == c || a == c) { st.faces_degenerate++; continue; }
    tri.push_back(a); tri.push_back(b); tri.push_back(c);
  }
  lap("merge close vertices");
  // ---- 2. duplicate faces: same vertex set, lowest face index survives (faces are visited in index order)
  {
    const size_t n = tri.size() / 3;
    // 96-bit key -> two-level: hash the sorted triple to 64 bits for the table, confirm on the triple itself
    std::vector<uint32_t> head(1, 0);
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    std::vector<uint32_t> table(cap, 0xFFFFFFFFu);  // face index of the first face with that vertex set
    auto sorted3 = [&](size_t f, uint32_t* v) { v[0] = tri[3 * f]; v[1] = tri[3 * f + 1]; v[2] = tri[3 * f + 2]; std::sort(v, v + 3); };
    size_t w = 0;
    for (size_t f = 0; f < n; f++) {
      uint32_t v[3];
      sorted3(f, v);
      size_t i = (size_t)(FlatMap::mix(((uint64_t)v[0] << 32 | v[1]) ^ FlatMap::mix(v[2])) & (cap - 1));
      bool dup = false;
      for (;;) {
        if (table[i] == 0xFFFFFFFFu) { table[i] = (uint32_t)w; break; }
        uint32_t u[3];
        sorted3(table[i], u);  // already compacted position: tri[] below w holds survivors only
        if (u[0] == v[0] && u[1] == v[1] && u[2] == v[2]) { dup = true; break; }
        i = (i + 1) & (cap - 1);
      }
      if (dup) { st.faces_duplicate++; continue; }
      if (w != f) { tri[3 * w] = tri[3 * f]; tri[3 * w + 1] = tri[3 * f + 1]; tri[3 * w + 2] = tri[3 * f + 2]; }
      w++;
    }
    tri.resize(3 * w);
  }
  lap("duplicate faces");
  // ---- 3. small connected components (faces adjacent across shared edges; a non-manifold edge connects all its faces)
  {
    const size_t n = tri.size() / 3;
    std::vector<uint32_t> parent(n);
    std::iota(parent.begin(), parent.end(), 0u);
    // faces sharing an edge: every directed half-edge is filed under its lower vertex (counting sort: sequential passes, no hashing), then
    // inside a vertex's short list the half-edges with the same other end point are linked (the lists average six entries: a quadratic
    // scan of one list touches a cache line or two, where looking the edge up through the faces around the vertex chased 3 x 6 pointers)
    std::vector<uint32_t> estart(nv + 1, 0);
    for (size_t f = 0; f < n; f++)
      for (int e = 0; e < 3; e++) estart[std::min(tri[3 * f + e], tri[3 * f + (e + 1) % 3]) + 1]++;
    for (size_t v = 0; v < nv; v++) estart[v + 1] += estart[v];
    std::vector<uint32_t> ehi(tri.size()), eface(tri.size()), efill(estart.begin(), estart.end() - 1);
    for (size_t f = 0; f < n; f++)   // faces in index order: inside a list the faces ascend
      for (int e = 0; e < 3; e++) {
        const uint32_t a = tri[3 * f + e], b = tri[3 * f + (e + 1) % 3];
        const uint32_t k = efill[std::min(a, b)]++;
        ehi[k] = std::max(a, b);
        eface[k] = (uint32_t)f;
      }
    auto link = [&](uint32_t fa, uint32_t fb) {
      const uint32_t ra = uf_find(parent, fa), rb = uf_find(parent, fb);
      if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
    };
    std::vector<std::pair<uint32_t, uint32_t>> big;
    for (size_t v = 0; v < nv; v++) {
      const uint32_t lo = estart[v], hi = estart[v + 1];
      if (hi - lo <= 48) {
        for (uint32_t i = lo; i < hi; i++)
          for (uint32_t j = i + 1; j < hi; j++)
            if (ehi[i] == ehi[j] && eface[i] != eface[j]) link(eface[i], eface[j]);
      } else {   // a hub vertex: sort its list by the other end point instead of comparing all pairs
        big.clear();
        for (uint32_t i = lo; i < hi; i++) big.emplace_back(ehi[i], eface[i]);
        std::sort(big.begin(), big.end());
        for (size_t i = 1; i < big.size(); i++)
          if (big[i].first == big[i - 1].first && big[i].second != big[i - 1].second) link(big[i].second, big[i - 1].second);
      }
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
  lap("connected components");
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
  lap("compaction");
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
  out->simplify_device = -1;   // the sequential host filter unless the caller opts into the GPU one
  out->clean_device = -1;      // likewise the cleaning filters
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
      if (current == "Quadric Edge Collapse Decimation") {
        if (stage != 0) return sf::fail(SF_ERR_UNSUPPORTED, "\"Quadric Edge Collapse Decimation\" must be the first filter (simplify.mlx order)");
        if (out->simplify) return sf::fail(SF_ERR_UNSUPPORTED, "more than one \"Quadric Edge Collapse Decimation\"");
        out->simplify = 1;
        sf_simplify_default_params(&out->simplify_params);
        continue;
      }
      if (current == "Merge Close Vertices") { want = 1; out->merge_close_vertices = 1; }
      else if (current == "Remove Duplicate Faces") { want = 2; out->remove_duplicate_faces = 1; }
      else if (current == "Remove Isolated pieces (wrt Face Num.)") { want = 3; out->remove_small_components = 1; }
      else if (current == "Remove Unreferenced Vertex") { want = 4; out->remove_unreferenced = 1; }
      else return sf::fail(SF_ERR_UNSUPPORTED, "filter \"%s\" is not implemented (only the clean.mlx / cleanLoRes.mlx filters are)", current.c_str());
      if (want <= stage) return sf::fail(SF_ERR_UNSUPPORTED, "filter \"%s\" out of the clean.mlx order", current.c_str());
      stage = want;
    } else if (tag.compare(0, 6, "Param ") == 0) {
      const std::string name = attr(tag, "name"), value = attr(tag, "value");
      if (current == "Quadric Edge Collapse Decimation") {
        sf_simplify_params& sp = out->simplify_params;
        const bool on = value == "true";
        if (name == "TargetFaceNum") sp.target_faces = (uint64_t)std::atoll(value.c_str());
        else if (name == "TargetPerc") sp.target_perc = (float)std::atof(value.c_str());
        else if (name == "QualityThr") sp.quality_thr = (float)std::atof(value.c_str());
        else if (name == "PreserveBoundary") sp.preserve_boundary = on;
        else if (name == "BoundaryWeight") sp.boundary_weight = (float)std::atof(value.c_str());
        else if (name == "PreserveNormal") sp.preserve_normal = on;
        else if (name == "PreserveTopology") sp.preserve_topology = on;
        else if (name == "OptimalPlacement") sp.optimal_placement = on;
        else if (name == "PlanarQuadric") sp.planar_quadric = on;
        else if (name == "QualityWeight") sp.quality_weight = on;
        else if (name == "AutoClean") sp.auto_clean = on;
        else if (name == "Selected" && on) return sf::fail(SF_ERR_UNSUPPORTED, "Selected = true: there is no face selection in a batch run");
      }
      if (current == "Merge Close Vertices" && name == "Threshold") out->merge_distance = (float)std::atof(value.c_str());
      if (current == "Remove Isolated pieces (wrt Face Num.)" && name == "MinComponentSize") out->min_component_faces = (uint32_t)std::atol(value.c_str());
    }
  }
  if (stage == 0 && !out->simplify) return sf::fail(SF_ERR_FORMAT, "%s holds no filter", path);
  return SF_OK;
}

SF_API int sf_mesh_clean_script(const sf_mesh* in, sf_clean_script* s, sf_mesh** out, sf_clean_stats* stats) {
  if (!s) return sf::fail(SF_ERR_INVALID_ARG, "NULL script");
  sf_mesh* simplified = nullptr;
  if (s->simplify) {
    const int rc = s->simplify_device >= 0 ? sf_mesh_simplify_gpu(in, &s->simplify_params, s->simplify_device, &simplified, &s->simplify_stats)
                                           : sf_mesh_simplify(in, &s->simplify_params, &simplified, &s->simplify_stats);
    if (rc != SF_OK) return rc;
    in = simplified;
    if (!s->merge_close_vertices && !s->remove_duplicate_faces && !s->remove_small_components && !s->remove_unreferenced) {
      if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->vertices_in = stats->vertices_out = simplified->pos.size() / 3;
        stats->faces_in = stats->faces_out = simplified->tri.size() / 3;
      }
      *out = simplified;
      return SF_OK;
    }
  }
  // a filter that is absent from the script degenerates to a no-op parameter
  const float dist = s->merge_close_vertices ? s->merge_distance : -1.0f;
  int rc = SF_OK;
  if (!s->remove_duplicate_faces || !s->remove_unreferenced)
    rc = sf::fail(SF_ERR_UNSUPPORTED, "scripts without \"Remove Duplicate Faces\" / \"Remove Unreferenced Vertex\" are not supported");
  else if (dist < 0.0f) rc = sf::fail(SF_ERR_UNSUPPORTED, "scripts without \"Merge Close Vertices\" are not supported");
  else if (s->clean_device >= 0) rc = sf_mesh_clean_gpu(in, dist, s->remove_small_components ? s->min_component_faces : 0u, s->clean_device, out, stats);
  else rc = sf_mesh_clean(in, dist, s->remove_small_components ? s->min_component_faces : 0u, out, stats);
  if (simplified) sf_mesh_free(simplified);
  return rc;
}

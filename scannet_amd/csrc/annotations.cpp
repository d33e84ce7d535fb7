// annotations.cpp -- host side of the ProjectAnnotations tool: which instance / label id every mesh vertex carries, and the transfer of
// those ids from the decimated mesh to the high-resolution one.
// Reference: AnnotationTools/ProjectAnnotations/Visualizer.cpp:259-377, common/Aggregation.h:47-82, common/Segmentation.h:56-75,
// ProjectAnnotations/LabelUtil.h:40-84.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

bool slurp(const char* path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

struct SegGroup {
  unsigned id = 0;
  std::string label;
  std::vector<unsigned> segments;
};

// position just after `"key":` among the top-level members of the JSON object text o, or npos
size_t member(const std::string& o, const char* key) {
  int d = 0;
  bool s = false;
  const std::string k = std::string("\"") + key + "\"";
  for (size_t j = 0; j < o.size(); j++) {
    const char ch = o[j];
    if (s) { if (ch == '\\') j++; else if (ch == '"') s = false; continue; }
    if (ch == '{' || ch == '[') d++;
    else if (ch == '}' || ch == ']') d--;
    else if (ch == '"') {
      if (d == 1 && o.compare(j, k.size(), k) == 0) {
        size_t q = j + k.size();
        while (q < o.size() && (o[q] == ' ' || o[q] == '\t' || o[q] == '\n' || o[q] == '\r')) q++;
        if (q < o.size() && o[q] == ':') return q + 1;
      }
      s = true;
    }
  }
  return std::string::npos;
}

// Aggregation::getUINT (Aggregation.h:120-127): ints, or strings holding ints; null -> (unsigned)-1
unsigned json_uint(const std::string& o, size_t& p) {
  while (p < o.size() && (o[p] == ' ' || o[p] == '\t' || o[p] == '\n' || o[p] == '\r')) p++;
  if (p < o.size() && o[p] == '"') p++;
  if (o.compare(p, 4, "null") == 0) { p += 4; return (unsigned)-1; }
  char* end = nullptr;
  const long long v = std::strtoll(o.c_str() + p, &end, 10);
  p = (size_t)(end - o.c_str());
  if (p < o.size() && o[p] == '"') p++;
  return (unsigned)v;
}

// Aggregation::loadFromJSONFile (Aggregation.h:47-82): segGroups[i].{id, label, segments}
bool parse_seg_groups(const std::string& t, std::vector<SegGroup>& out) {
  size_t p = t.find("\"segGroups\"");
  if (p == std::string::npos) return false;
  p = t.find('[', p);
  if (p == std::string::npos) return false;
  int depth = 0;
  size_t obj_start = 0;
  bool in_str = false;
  for (size_t i = p; i < t.size(); i++) {
    const char c = t[i];
    if (in_str) { if (c == '\\') i++; else if (c == '"') in_str = false; continue; }
    if (c == '"') { in_str = true; continue; }
    if (c == '{') { if (depth == 1) obj_start = i; depth++; }
    else if (c == '[') depth++;
    else if (c == ']') { depth--; if (depth == 0) return true; }
    else if (c == '}') {
      depth--;
      if (depth != 1) continue;
      const std::string o = t.substr(obj_start, i - obj_start + 1);
      size_t pi = member(o, "id"), pl = member(o, "label"), ps = member(o, "segments");
      if (pi == std::string::npos || pl == std::string::npos || ps == std::string::npos) return false;
      SegGroup g;
      g.id = json_uint(o, pi);
      size_t q = o.find('"', pl);
      if (q == std::string::npos) return false;
      for (q++; q < o.size() && o[q] != '"'; q++) { if (o[q] == '\\' && q + 1 < o.size()) q++; g.label.push_back(o[q]); }
      ps = o.find('[', ps);
      if (ps == std::string::npos) return false;
      for (ps++; ps < o.size();) {
        while (ps < o.size() && (o[ps] == ' ' || o[ps] == ',' || o[ps] == '\n' || o[ps] == '\r' || o[ps] == '\t')) ps++;
        if (ps >= o.size() || o[ps] == ']') break;
        const size_t before = ps;
        g.segments.push_back(json_uint(o, ps));
        if (ps == before) return false;
      }
      out.push_back(std::move(g));
    }
  }
  return false;
}

// Segmentation::loadFromFile (Segmentation.h:56-75): the segIndices array
bool parse_seg_indices(const std::string& t, std::vector<unsigned>& out) {
  size_t p = t.find("\"segIndices\"");
  if (p == std::string::npos) return false;
  p = t.find('[', p);
  if (p == std::string::npos) return false;
  for (p++; p < t.size();) {
    while (p < t.size() && (t[p] == ' ' || t[p] == ',' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) p++;
    if (p >= t.size()) return false;
    if (t[p] == ']') return true;
    const size_t before = p;
    out.push_back(json_uint(t, p));
    if (p == before) return false;
  }
  return false;
}

// LabelUtil::getLabelMappingFromFile with labelName "category", idName "" (LabelUtil.h:40-84): id = 1-based line number
bool parse_label_map(const char* path, std::unordered_map<std::string, unsigned short>& out) {
  std::ifstream f(path);
  std::string line;
  if (!f || !std::getline(f, line)) return false;
  auto split = [](const std::string& s) {
    std::vector<std::string> v;
    std::string cur;
    for (char c : s) { if (c == '\t') { v.push_back(cur); cur.clear(); } else if (c != '\r') cur.push_back(c); }
    v.push_back(cur);
    return v;
  };
  const std::vector<std::string> header = split(line);
  int col = -1;
  for (size_t i = 0; i < header.size(); i++) if (header[i] == "category") col = (int)i;
  if (col < 0) return false;
  unsigned line_count = 1;
  while (std::getline(f, line)) {
    const std::vector<std::string> parts = split(line);
    if ((int)parts.size() > col && !parts[(size_t)col].empty()) {
      if (line_count > 65535) continue;   // "skip" (:72-73) -- without advancing the line counter
      out[parts[(size_t)col]] = (unsigned short)line_count;
    }
    ++line_count;
  }
  return true;
}

struct V3 { float x, y, z; };

// area-weighted vertex normals: face cross products accumulated in face order, then normalised (MeshData::computeVertexNormals)
std::vector<V3> vertex_normals(const float* xyz, uint64_t V, const uint32_t* tris, uint64_t F) {
  std::vector<V3> n(V, V3{0.0f, 0.0f, 0.0f});
  for (uint64_t f = 0; f < F; f++) {
    const uint32_t a = tris[3 * f], b = tris[3 * f + 1], c = tris[3 * f + 2];
    if (a >= V || b >= V || c >= V) continue;
    const float ux = xyz[3 * b] - xyz[3 * a], uy = xyz[3 * b + 1] - xyz[3 * a + 1], uz = xyz[3 * b + 2] - xyz[3 * a + 2];
    const float vx = xyz[3 * c] - xyz[3 * a], vy = xyz[3 * c + 1] - xyz[3 * a + 1], vz = xyz[3 * c + 2] - xyz[3 * a + 2];
    const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    for (uint32_t v : {a, b, c}) { n[v].x += cx; n[v].y += cy; n[v].z += cz; }
  }
  for (uint64_t i = 0; i < V; i++) {
    const float len = std::sqrt(n[i].x * n[i].x + n[i].y * n[i].y + n[i].z * n[i].z);
    if (len > 0.0f) { n[i].x /= len; n[i].y /= len; n[i].z /= len; }
  }
  return n;
}

}  // namespace

// Visualizer::computeObjectIdsAndColorsPerVertex (:259-295) on the decimated mesh: vertex -> (instance, label), 0 = unannotated.
// As in the reference the colour table is keyed by LABEL id (:271-276), so every object of a category carries the instance value
// (index + 1) of the first object with that category.
SF_API int sf_annotation_vertex_ids(const char* segs_json, const char* aggregation_json, const char* label_map_tsv, uint64_t num_vertices,
                                    uint8_t* vertex_instance, uint16_t* vertex_label, uint32_t* num_labels) {
  if (!segs_json || !aggregation_json || !label_map_tsv || !vertex_instance || !vertex_label) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::string t;
  std::vector<unsigned> seg;
  if (!slurp(segs_json, t)) return sf::fail(SF_ERR_IO, "[parse] failed to open file %s", segs_json);
  if (!parse_seg_indices(t, seg)) return sf::fail(SF_ERR_FORMAT, "no segIndices array in %s", segs_json);
  if (seg.size() != num_vertices)
    return sf::fail(SF_ERR_FORMAT, "%s holds %llu segment ids for a mesh of %llu vertices", segs_json, (unsigned long long)seg.size(), (unsigned long long)num_vertices);
  std::vector<SegGroup> groups;
  if (!slurp(aggregation_json, t)) return sf::fail(SF_ERR_IO, "failed to open file %s", aggregation_json);
  if (!parse_seg_groups(t, groups)) return sf::fail(SF_ERR_FORMAT, "no segGroups array in %s", aggregation_json);
  std::unordered_map<std::string, unsigned short> label_ids;
  if (!parse_label_map(label_map_tsv, label_ids)) return sf::fail(SF_ERR_IO, "error reading label mapping file %s", label_map_tsv);
  std::unordered_map<unsigned, std::string> id_to_label;   // m_objectIdsToLabels, keyed by the group's "id"
  for (const SegGroup& g : groups) id_to_label[g.id] = g.label;
  struct Ids { uint8_t inst; uint16_t label; };
  std::unordered_map<unsigned short, Ids> colour;            // objectColors
  std::unordered_map<unsigned, unsigned short> object_label; // objectIdsToLabelIds
  for (unsigned i = 0; i < groups.size(); i++) {
    const auto itl = id_to_label.find(i);
    if (itl == id_to_label.end()) return sf::fail(SF_ERR_FORMAT, "%s: no segGroup with id %u (ids must cover 0..%zu)", aggregation_json, i, groups.size() - 1);
    const auto lid = label_ids.find(itl->second);
    if (lid == label_ids.end()) continue;
    object_label[i] = lid->second;
    if (colour.find(lid->second) == colour.end()) colour[lid->second] = Ids{(uint8_t)(i + 1), lid->second};
  }
  std::unordered_map<unsigned, std::vector<uint32_t>> verts_of_seg;
  for (uint64_t v = 0; v < num_vertices; v++) verts_of_seg[seg[v]].push_back((uint32_t)v);
  std::memset(vertex_instance, 0, num_vertices);
  std::memset(vertex_label, 0, 2 * num_vertices);
  for (unsigned i = 0; i < groups.size(); i++) {
    const auto itl = object_label.find(i);
    if (itl == object_label.end()) continue;
    const Ids c = colour[itl->second];
    for (unsigned s : groups[i].segments) {
      const auto it = verts_of_seg.find(s);
      if (it == verts_of_seg.end()) continue;
      for (uint32_t v : it->second) { vertex_instance[v] = c.inst; vertex_label[v] = c.label; }
    }
  }
  if (num_labels) *num_labels = (uint32_t)colour.size();
  return SF_OK;
}

// Visualizer::propagateAnnotations (:297-377): every vertex of the high-resolution mesh takes the ids of one of its three nearest
// annotated vertices of the decimated mesh -- the first (by distance) within maxThresh whose normal is within normal_thresh radians,
// else the nearest one's ids if all three are within maxThresh and agree on the instance, else none.  The reference asks FLANN
// (randomised kd-trees, 100 checks: approximate, not reproducible) for the neighbours; here they are the exact three nearest, ties
// broken by vertex index.  Distances are squared (FLANN's L2) and compared with maxThresh = max(1 % of the bounding box extent, 0.05).
SF_API int sf_annotation_propagate(const float* src_xyz, uint64_t src_vertices, const uint32_t* src_tris, uint64_t src_triangles, const uint8_t* src_instance,
                                   const uint16_t* src_label, const float* dst_xyz, uint64_t dst_vertices, const uint32_t* dst_tris, uint64_t dst_triangles,
                                   float normal_thresh, uint8_t* dst_instance, uint16_t* dst_label) {
  if (!src_xyz || !src_tris || !src_instance || !src_label || !dst_xyz || !dst_tris || !dst_instance || !dst_label) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::memset(dst_instance, 0, dst_vertices);
  std::memset(dst_label, 0, 2 * dst_vertices);
  if (src_vertices == 0 || dst_vertices == 0) return SF_OK;
  const std::vector<V3> nsrc = vertex_normals(src_xyz, src_vertices, src_tris, src_triangles);
  const std::vector<V3> ndst = vertex_normals(dst_xyz, dst_vertices, dst_tris, dst_triangles);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint64_t i = 0; i < src_vertices; i++)
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], src_xyz[3 * i + k]); hi[k] = std::max(hi[k], src_xyz[3 * i + k]); }
  const float extent = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  const float max_thresh = std::max(extent * 0.01f, 0.05f);
  std::vector<uint32_t> search;   // searchIndices: annotated source vertices, in index order
  for (uint64_t i = 0; i < src_vertices; i++) if (src_label[i] > 0) search.push_back((uint32_t)i);
  if (search.empty()) return SF_OK;
  // uniform grid with cells of the search radius: the candidates of a query lie in its 27 neighbouring cells
  const float cell = std::sqrt(max_thresh) * 1.0001f;
  auto cell_of = [&](const float* p, int k) { return (long long)std::floor((p[k] - lo[k]) / cell); };
  const long long gx = cell_of(hi, 0) + 1, gy = cell_of(hi, 1) + 1, gz = cell_of(hi, 2) + 1;
  if (gx * gy * gz > (1ll << 28)) return sf::fail(SF_ERR_INVALID_ARG, "mesh extent too large for the neighbour grid");
  std::vector<uint32_t> start((size_t)(gx * gy * gz) + 1, 0), order(search.size());
  auto cell_index = [&](long long x, long long y, long long z) { return (size_t)((z * gy + y) * gx + x); };
  for (uint32_t s : search) start[cell_index(cell_of(src_xyz + 3 * (size_t)s, 0), cell_of(src_xyz + 3 * (size_t)s, 1), cell_of(src_xyz + 3 * (size_t)s, 2)) + 1]++;
  for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
  {
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (uint32_t k = 0; k < search.size(); k++) {
      const float* p = src_xyz + 3 * (size_t)search[k];
      order[fill[cell_index(cell_of(p, 0), cell_of(p, 1), cell_of(p, 2))]++] = k;   // k ascending inside a cell
    }
  }
  auto work = [&](uint64_t b, uint64_t e) {
    for (uint64_t i = b; i < e; i++) {
      const float* p = dst_xyz + 3 * i;
      struct Hit { float d; uint32_t k; };
      Hit best[3] = {{INFINITY, 0}, {INFINITY, 0}, {INFINITY, 0}};
      int found = 0;
      const long long cx = cell_of(p, 0), cy = cell_of(p, 1), cz = cell_of(p, 2);
      for (long long z = cz - 1; z <= cz + 1; z++)
        for (long long y = cy - 1; y <= cy + 1; y++)
          for (long long x = cx - 1; x <= cx + 1; x++) {
            if (x < 0 || y < 0 || z < 0 || x >= gx || y >= gy || z >= gz) continue;
            const size_t c = cell_index(x, y, z);
            for (uint32_t o = start[c]; o < start[c + 1]; o++) {
              const uint32_t k = order[o];
              const float* s = src_xyz + 3 * (size_t)search[k];
              const float dx = p[0] - s[0], dy = p[1] - s[1], dz = p[2] - s[2];
              const float d = dx * dx + dy * dy + dz * dz;
              if (!(d < max_thresh)) continue;
              Hit h{d, k};
              for (int m = 0; m < 3; m++)
                if (h.d < best[m].d || (h.d == best[m].d && best[m].d != INFINITY && h.k < best[m].k)) std::swap(h, best[m]);
              if (found < 3) found++;
            }
          }
      if (found == 0) continue;   // nearest neighbour beyond maxThresh: allSame = false, no normal match
      const uint32_t v_first = search[best[0].k];
      bool all_same = found == 3;
      int hit = -1;
      for (int m = 0; m < found; m++) {
        const uint32_t v = search[best[m].k];
        float dot = nsrc[v].x * ndst[i].x + nsrc[v].y * ndst[i].y + nsrc[v].z * ndst[i].z;
        dot = dot < -1.0f ? -1.0f : (dot > 1.0f ? 1.0f : dot);
        if (std::acos(dot) < normal_thresh) { hit = m; break; }
        if (src_instance[v] != src_instance[v_first]) all_same = false;
      }
      if (hit >= 0) { dst_instance[i] = src_instance[search[best[hit].k]]; dst_label[i] = src_label[search[best[hit].k]]; }
      else if (all_same) { dst_instance[i] = src_instance[v_first]; dst_label[i] = src_label[v_first]; }
    }
  };
  unsigned nt = std::max(1u, std::min(64u, (unsigned)sf::usable_cpus()));
  if (dst_vertices < 4096) nt = 1;
  std::vector<std::thread> pool;
  const uint64_t chunk = (dst_vertices + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const uint64_t b = t * chunk, e = std::min(dst_vertices, b + chunk);
    if (b < e) pool.emplace_back(work, b, e);
  }
  for (std::thread& th : pool) th.join();
  return SF_OK;
}

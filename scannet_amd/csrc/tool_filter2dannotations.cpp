// tool_filter2dannotations.cpp -- the 2-D annotation filter as a command-line tool: the per-scene body of
// AnnotationTools/Filter2dAnnotations/Filter2dAnnotations.cpp process() (:257-414) on top of libscanfuse's C ABI.
//
//     filter2dannotations <annotations dir> <scan.sens> <aggregation.json> <label map .tsv> <output dir>
//
// The reference compiles its paths in (Filter2dAnnotations.cpp:89-95: ../annotations-2d/<scene>, ../../data/scans/<scene>/<scene>.sens,
// <scene>.aggregation.json, ../../data/tasks/scannet-labels.combined.tsv, ../annotations-2d-filtered/<scene>) and loops over a scene
// list; here one scene per invocation with the five paths as arguments.  As there:
//   * <annotations dir>/instance/<frame>.png (8-bit) and label/<frame>.png (16-bit) are the projected annotations (:262-265, :340-341);
//     every file of label/ is a frame to process, its name is the frame index (:312-314);
//   * object ids come from the aggregation's segGroups (common/Aggregation.h:62-80), their label ids from the `category` column of the
//     label map, line number = id (LabelUtil.h:40-84); instance value = object id + 1, histogram bins in object-id order (:293-309;
//     the reference walks an unordered_map, i.e. an unspecified order -- it only matters for exact vote ties);
//   * a frame whose pose is -inf gets all-zero images (:315-321); a scene whose outputs are complete is skipped (:272-279);
//   * results go to <output dir>/instance/<frame>.png and label/<frame>.png (:400-401).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "scanfuse.h"

namespace {

bool is_dir(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
void make_dir(const std::string& p) { ::mkdir(p.c_str(), 0777); }
std::vector<std::string> list_files(const std::string& dir) {
  std::vector<std::string> out;
  if (DIR* d = ::opendir(dir.c_str())) {
    while (dirent* e = ::readdir(d))
      if (e->d_name[0] != '.') out.push_back(e->d_name);
    ::closedir(d);
  }
  std::sort(out.begin(), out.end());
  return out;
}
std::string slash(std::string p) { if (p.empty() || (p.back() != '/' && p.back() != '\\')) p.push_back('/'); return p; }

// segGroups[i].id / .label out of an aggregation JSON (a scanner for exactly that shape; strings may hold escaped quotes)
bool parse_aggregation(const std::string& path, std::map<unsigned, std::string>& out) {
  std::ifstream f(path);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string t = ss.str();
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
    else if (c == ']') { depth--; if (depth == 0) break; }
    else if (c == '}') {
      depth--;
      if (depth == 1) {  // one segGroup object: top-level keys only
        const std::string o = t.substr(obj_start, i - obj_start + 1);
        auto key_pos = [&](const char* key) {
          int d = 0; bool s = false;
          const std::string k = std::string("\"") + key + "\"";
          for (size_t j = 0; j < o.size(); j++) {
            const char ch = o[j];
            if (s) { if (ch == '\\') j++; else if (ch == '"') s = false; continue; }
            if (ch == '{' || ch == '[') d++;
            else if (ch == '}' || ch == ']') d--;
            else if (ch == '"') {
              if (d == 1 && o.compare(j, k.size(), k) == 0) { size_t q = o.find(':', j + k.size()); if (q != std::string::npos) return q + 1; }
              s = true;
            }
          }
          return std::string::npos;
        };
        const size_t pi = key_pos("id"), pl = key_pos("label");
        if (pi == std::string::npos || pl == std::string::npos) return false;
        const unsigned id = (unsigned)std::strtoul(o.c_str() + pi, nullptr, 10);
        size_t q = o.find('"', pl);
        if (q == std::string::npos) return false;
        std::string label;
        for (q++; q < o.size() && o[q] != '"'; q++) { if (o[q] == '\\' && q + 1 < o.size()) q++; label.push_back(o[q]); }
        out[id] = label;
      }
    }
  }
  return true;
}

// LabelUtil::getLabelMappingFromFile with labelName "category", idName "" (LabelUtil.h:40-84): id = 1-based line number
bool parse_label_map(const std::string& path, std::map<std::string, unsigned short>& out) {
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
    if ((int)parts.size() > col && !parts[(size_t)col].empty() && line_count <= 65535) out[parts[(size_t)col]] = (unsigned short)line_count;
    ++line_count;
  }
  return true;
}

int die(const char* what) { std::fprintf(stderr, "%s: %s\n", what, sf_last_error()); return 1; }

}  // namespace

int main(int argc, const char** argv) {
  if (argc != 6) {
    std::fprintf(stderr, "usage: filter2dannotations <annotations dir> <scan.sens> <aggregation.json> <label map .tsv> <output dir>\n");
    return 1;
  }
  const std::string path = slash(argv[1]), sens_path = argv[2], agg_path = argv[3], tsv_path = argv[4], out_path = slash(argv[5]);
  const std::string inst_dir = path + "instance/", label_dir = path + "label/";
  if (!is_dir(inst_dir) || !is_dir(label_dir)) { std::fprintf(stderr, "instance/label dir does not exist for %s\n", path.c_str()); return 1; }
  std::printf("%s\n", path.c_str());
  sf_sens* sd = nullptr;
  if (sf_sens_open(sens_path.c_str(), &sd) != SF_OK) return die("sens");
  sf_sens_info info;
  sf_sens_get_info(sd, &info);
  const std::string out_inst = out_path + "instance/", out_label = out_path + "label/";
  if (is_dir(out_inst) && is_dir(out_label) && list_files(out_inst).size() == info.num_frames && list_files(out_label).size() == info.num_frames) {
    std::printf("  ==> skipping, already exists\n");
    return 0;
  }
  std::map<unsigned, std::string> objects;
  if (!parse_aggregation(agg_path, objects)) { std::fprintf(stderr, "failed to open file %s\n", agg_path.c_str()); return 1; }
  std::map<std::string, unsigned short> label_ids;
  if (!parse_label_map(tsv_path, label_ids)) { std::fprintf(stderr, "error reading label mapping file %s\n", tsv_path.c_str()); return 1; }
  std::printf("read %zu labels\n", label_ids.size());
  make_dir(out_path); make_dir(out_inst); make_dir(out_label);
  // Filter2dAnnotations.cpp:293-309
  uint8_t to_idx[256], to_inst[80];
  uint16_t to_label[256];
  std::memset(to_idx, 255, sizeof(to_idx));
  std::memset(to_inst, 255, sizeof(to_inst));
  for (uint16_t& v : to_label) v = 65535;
  to_idx[0] = 0; to_inst[0] = 0; to_label[0] = 0;
  unsigned idx = 1;
  for (const auto& a : objects) {
    if (a.first + 1 > 255 || idx >= 80) { std::fprintf(stderr, "more than %d annotated objects (MAX_NUM_LABELS_PER_SCENE)\n", 79); return 1; }
    const auto it = label_ids.find(a.second);
    to_label[a.first + 1] = it == label_ids.end() ? 0 : it->second;
    to_idx[a.first + 1] = (uint8_t)idx;
    to_inst[idx] = (uint8_t)(a.first + 1);
    idx++;
  }
  const uint32_t cw = info.color_width, ch = info.color_height, dw = info.depth_width, dh = info.depth_height;
  sf_filter2d* filt = nullptr;
  const int device = std::getenv("SF_DEVICE") ? std::atoi(std::getenv("SF_DEVICE")) : 0;
  if (sf_filter2d_create((int)dw, (int)dh, (int)cw, (int)ch, device, &filt) != SF_OK) return die("filter");
  if (sf_filter2d_set_tables(filt, to_idx, to_inst, to_label) != SF_OK) return die("tables");
  const std::vector<std::string> files = list_files(label_dir);
  for (const std::string& f : files) {
    const uint64_t frame = std::strtoull(f.c_str(), nullptr, 10);
    if (frame >= info.num_frames) { std::fprintf(stderr, "%s names frame %llu of %llu\n", f.c_str(), (unsigned long long)frame, (unsigned long long)info.num_frames); return 1; }
  }
  // Frames in chunks: while the GPU filters chunk k (one frame after the other), threads decode the inputs of chunk k+1 (depth inflate,
  // colour JPEG, instance PNG) and write the PNGs of chunk k-1 -- the reference does all of it on one thread.
  struct Frame {
    std::string name;
    bool valid = false;
    std::vector<uint16_t> depth, label_out;
    std::vector<uint8_t> rgb, inst_in, inst_out;
    std::string error;
  };
  const size_t chunk_frames = 16;
  struct Chunk { std::vector<Frame> frames; std::vector<std::thread> loaders, writers; size_t n = 0; };
  Chunk chunks[2];
  for (Chunk& c : chunks) {
    c.frames.resize(chunk_frames);
    for (Frame& f : c.frames) {
      f.depth.resize((size_t)dw * dh); f.label_out.resize((size_t)cw * ch);
      f.rgb.resize((size_t)cw * ch * 3); f.inst_in.resize((size_t)cw * ch); f.inst_out.resize((size_t)cw * ch);
    }
  }
  auto start_load = [&](Chunk& c, size_t first) {
    c.n = std::min(chunk_frames, files.size() - first);
    for (size_t k = 0; k < c.n; k++) {
      Frame* fr = &c.frames[k];
      fr->name = files[first + k];
      fr->error.clear();
      c.loaders.emplace_back([&, fr] {
        const uint64_t frame = std::strtoull(fr->name.c_str(), nullptr, 10);
        float pose[16];
        int valid = 0;
        sf_sens_pose(sd, frame, pose, &valid);
        fr->valid = valid != 0;
        if (!fr->valid) return;
        if (sf_sens_decode_depth(sd, frame, fr->depth.data()) != SF_OK || sf_sens_decode_color(sd, frame, fr->rgb.data()) != SF_OK) {
          fr->error = std::string("frame decode: ") + sf_last_error();
          return;
        }
        uint32_t w = 0, h = 0;
        int c2 = 0, b2 = 0;
        void* data = nullptr;
        if (sf_png_read((inst_dir + fr->name).c_str(), &w, &h, &c2, &b2, &data) != SF_OK) { fr->error = std::string("instance image: ") + sf_last_error(); return; }
        if (w != cw || h != ch || c2 != 1 || b2 != 8) fr->error = inst_dir + fr->name + ": expected an 8-bit grey " + std::to_string(cw) + "x" + std::to_string(ch) + " image";
        else std::memcpy(fr->inst_in.data(), data, (size_t)cw * ch);
        sf_free(data);
      });
    }
  };
  auto join = [](std::vector<std::thread>& v) { for (std::thread& t : v) t.join(); v.clear(); };
  size_t done = 0;
  double kernel_ms = 0;
  int cur = 0;
  if (!files.empty()) start_load(chunks[0], 0);
  for (size_t first = 0; first < files.size(); first += chunk_frames, cur ^= 1) {
    Chunk& c = chunks[cur];
    Chunk& other = chunks[cur ^ 1];
    join(c.loaders);
    join(other.writers);
    for (size_t k = 0; k < other.n; k++) if (!other.frames[k].error.empty()) { std::fprintf(stderr, "%s\n", other.frames[k].error.c_str()); return 1; }
    if (first + chunk_frames < files.size()) start_load(other, first + chunk_frames);
    for (size_t k = 0; k < c.n; k++) {
      Frame& fr = c.frames[k];
      if (!fr.error.empty()) { join(other.loaders); std::fprintf(stderr, "%s\n", fr.error.c_str()); return 1; }
      if (!fr.valid) {
        std::fill(fr.inst_out.begin(), fr.inst_out.end(), 0);
        std::fill(fr.label_out.begin(), fr.label_out.end(), 0);
      } else {
        float us = 0;
        if (sf_filter2d_frame(filt, fr.depth.data(), fr.rgb.data(), fr.inst_in.data(), fr.inst_out.data(), fr.label_out.data(), &us) != SF_OK) {
          join(other.loaders);
          return die("filter frame");
        }
        kernel_ms += us * 1e-3;
      }
      if (done % 10 == 0 || done + 1 == files.size()) { std::printf("\r[ %zu | %zu ]", done, files.size()); std::fflush(stdout); }
      ++done;
    }
    for (size_t k = 0; k < c.n; k++) {
      Frame* fr = &c.frames[k];
      c.writers.emplace_back([&, fr] {
        if (sf_png_write_gray((out_inst + fr->name).c_str(), fr->inst_out.data(), cw, ch, 8) != SF_OK ||
            sf_png_write_gray((out_label + fr->name).c_str(), fr->label_out.data(), cw, ch, 16) != SF_OK)
          fr->error = std::string("output image: ") + sf_last_error();
      });
    }
  }
  for (Chunk& c : chunks) { join(c.loaders); join(c.writers); }
  for (Chunk& c : chunks)
    for (size_t k = 0; k < c.n; k++) if (!c.frames[k].error.empty()) { std::fprintf(stderr, "%s\n", c.frames[k].error.c_str()); return 1; }
  std::printf("\n%zu frames, %.1f ms of GPU kernels\n", done, kernel_ms);
  sf_filter2d_destroy(filt);
  sf_sens_close(sd);
  return 0;
}

// projectannotations -- drop-in for AnnotationTools/ProjectAnnotations (main.cpp:5-52, Visualizer.cpp:8-193):
//     projectannotations [<zParametersScan.txt> <scan dir>]
// Reads <scan dir>/<scan>.sens, <scan>_vh_clean_2.ply, <scan>_vh_clean_2.0.010000.segs.json, <scan>.aggregation.json (and
// <scan>_vh_clean.ply when s_useHiResMesh), labels the mesh vertices, draws the mesh into every s_frameSkip-th frame on the GPU and
// writes <s_outDir>/<scan>/instance/<frame>.png (8-bit) and label/<frame>.png (16-bit).  The reference is a Direct3D 11 window
// application that renders one frame per message-loop iteration; here frames go to the GPU eight at a time while the previous eight
// are being written as PNGs.
//   * like the reference the tool insists on <scan dir>/<scan>.txt holding colorWidth / colorHeight (main.cpp:29-48; it sizes the window
//     with them); the render target takes the .sens colour size (Visualizer.cpp:51)
//   * s_outputDebugImages (random-colour visualisations, :188-225) is not implemented
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "scanfuse.h"

namespace {

bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
void make_dir(const std::string& p) { if (!is_dir(p)) ::mkdir(p.c_str(), 0777); }

// mLib ParameterFile: `name = value;` lines, // comments, optional quotes
bool read_parameter_file(const std::string& path, std::map<std::string, std::string>& out) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    const size_t c = line.find("//");
    if (c != std::string::npos) line.resize(c);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    auto trim = [](std::string s) {
      const char* ws = " \t\r\n;";
      const size_t a = s.find_first_not_of(ws), b = s.find_last_not_of(ws);
      s = a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
      if (s.size() >= 2 && s.front() == '"' && s.back() == '"') s = s.substr(1, s.size() - 2);
      return s;
    };
    out[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
  }
  return true;
}
bool as_bool(const std::string& v) { return v == "true" || v == "1"; }

int die(const char* what) { std::fprintf(stderr, "%s: %s\n", what, sf_last_error()); return 1; }

struct HostMesh {
  std::vector<float> xyz;
  std::vector<uint32_t> tris;
  uint64_t V = 0, F = 0;
};
bool load_mesh(const std::string& path, HostMesh& m) {
  sf_mesh* h = nullptr;
  if (sf_ply_read(path.c_str(), &h) != SF_OK) return false;
  sf_mesh_counts(h, &m.V, &m.F);
  m.xyz.resize(3 * m.V);
  m.tris.resize(3 * m.F);
  const int rc = sf_mesh_copy(h, m.xyz.data(), nullptr, m.tris.data(), nullptr);
  sf_mesh_free(h);
  return rc == SF_OK;
}

}  // namespace

int main(int argc, const char** argv) {
  std::string param_file = "zParametersScan.txt", scan_dir;
  if (argc == 3) { param_file = argv[1]; scan_dir = argv[2]; }   // main.cpp:11-17
  std::printf("fileNameDescGlobalApp = %s\n", param_file.c_str());
  std::map<std::string, std::string> gp;
  if (!read_parameter_file(param_file, gp)) { std::fprintf(stderr, "could not read parameter file %s\n", param_file.c_str()); return 1; }
  auto get = [&](const char* k, const char* dflt) { const auto it = gp.find(k); return it == gp.end() ? std::string(dflt) : it->second; };
  if (scan_dir.empty()) scan_dir = get("s_scanDir", "");
  for (char& c : scan_dir) if (c == '\\') c = '/';
  if (scan_dir.empty()) { std::fprintf(stderr, "no scan directory (s_scanDir)\n"); return 1; }
  if (scan_dir.back() != '/') scan_dir.push_back('/');
  std::string scan_name = scan_dir.substr(0, scan_dir.size() - 1);
  scan_name = scan_name.substr(scan_name.find_last_of('/') == std::string::npos ? 0 : scan_name.find_last_of('/') + 1);
  const float depth_min = (float)std::atof(get("s_depthMin", "0.1").c_str()), depth_max = (float)std::atof(get("s_depthMax", "15.0").c_str());
  const float dist_thresh = (float)std::atof(get("s_depthDistThresh", "0.2").c_str()), normal_thresh = (float)std::atof(get("s_propagateNormalThresh", "0.5").c_str());
  const bool use_hi = as_bool(get("s_useHiResMesh", "true")), filter_orig = as_bool(get("s_filterUsingOrigialDepthImage", "false"));
  const unsigned frame_skip = (unsigned)std::max(1, std::atoi(get("s_frameSkip", "1").c_str()));
  std::string out_dir = get("s_outDir", "output/");
  const std::string label_map = get("s_labelMappingFile", "");

  const std::string meta_file = scan_dir + scan_name + ".txt";   // main.cpp:29-48
  std::map<std::string, std::string> meta;
  if (!exists(meta_file) || !read_parameter_file(meta_file, meta)) { std::printf("ERROR: meta-file (%s) does not exist! \n", meta_file.c_str()); return 255; }
  for (const char* k : {"colorWidth", "colorHeight"})
    if (meta.find(k) == meta.end()) { std::fprintf(stderr, "ERROR: failed to read \"%s\" param from %s\n", k, meta_file.c_str()); return 255; }

  // Visualizer::init (:8-55)
  std::printf("[ProjectAnnotations] %s\n", scan_dir.c_str());
  const std::string sens_file = scan_dir + scan_name + ".sens", mesh_file = scan_dir + scan_name + "_vh_clean_2.ply",
                    segs_file = scan_dir + scan_name + "_vh_clean_2.0.010000.segs.json", agg_file = scan_dir + scan_name + ".aggregation.json",
                    hi_file = scan_dir + scan_name + "_vh_clean.ply";
  if (!(exists(sens_file) && exists(mesh_file) && exists(segs_file) && exists(agg_file) && (!use_hi || exists(hi_file)))) {
    std::printf("WARNING: no sens/mesh/segs/aggregation file, skipping\n");
    return 0;
  }
  if (!exists(label_map)) { std::fprintf(stderr, "%s does not exist!\n", label_map.c_str()); return 1; }
  std::printf("loading scan info... ");
  std::fflush(stdout);
  sf_sens* sd = nullptr;
  if (sf_sens_open(sens_file.c_str(), &sd) != SF_OK) return die("sens");
  sf_sens_info info;
  sf_sens_get_info(sd, &info);
  HostMesh lo, hi;
  if (!load_mesh(mesh_file, lo)) return die("mesh");
  std::vector<uint8_t> inst(lo.V);
  std::vector<uint16_t> label(lo.V);
  uint32_t num_labels = 0;
  if (sf_annotation_vertex_ids(segs_file.c_str(), agg_file.c_str(), label_map.c_str(), lo.V, inst.data(), label.data(), &num_labels) != SF_OK) return die("annotations");
  const HostMesh* draw = &lo;
  if (use_hi) {
    if (!load_mesh(hi_file, hi)) return die("hi-res mesh");
    std::vector<uint8_t> hinst(hi.V);
    std::vector<uint16_t> hlabel(hi.V);
    if (sf_annotation_propagate(lo.xyz.data(), lo.V, lo.tris.data(), lo.F, inst.data(), label.data(), hi.xyz.data(), hi.V, hi.tris.data(), hi.F, normal_thresh,
                                hinst.data(), hlabel.data()) != SF_OK)
      return die("propagate");
    inst.swap(hinst);
    label.swap(hlabel);
    draw = &hi;
  }
  std::printf("done! (%u labels, %llu vertices, %llu triangles)\n", num_labels, (unsigned long long)draw->V, (unsigned long long)draw->F);

  sf_project_params pp;
  pp.color_width = info.color_width; pp.color_height = info.color_height; pp.depth_width = info.depth_width; pp.depth_height = info.depth_height;
  pp.fx = info.color_intrinsic[0]; pp.fy = info.color_intrinsic[5];
  pp.depth_min = depth_min; pp.depth_max = depth_max; pp.depth_dist_thresh = dist_thresh; pp.filter_using_original_depth = filter_orig ? 1 : 0;
  const int device = std::getenv("SF_DEVICE") ? std::atoi(std::getenv("SF_DEVICE")) : 0;
  sf_projector* proj = nullptr;
  if (sf_projector_create(&pp, device, &proj) != SF_OK) return die("projector");
  if (sf_projector_set_mesh(proj, draw->xyz.data(), draw->V, draw->tris.data(), draw->F, inst.data(), label.data()) != SF_OK) return die("mesh upload");

  if (out_dir.back() != '/' && out_dir.back() != '\\') out_dir.push_back('/');
  make_dir(out_dir);
  out_dir += scan_name + "/";
  const std::string out_inst = out_dir + "instance/", out_label = out_dir + "label/";
  make_dir(out_dir); make_dir(out_inst); make_dir(out_label);

  const int B = sf_projector_max_batch();
  const size_t np = (size_t)info.color_width * info.color_height, dn = (size_t)info.depth_width * info.depth_height;
  struct Set { uint8_t* inst = nullptr; uint16_t* label = nullptr; std::vector<uint64_t> frames; std::vector<std::thread> writers; bool failed = false; };
  Set sets[2];
  uint16_t* depth = nullptr;
  for (Set& s : sets)
    if (sf_host_alloc(B * np, (void**)&s.inst) != SF_OK || sf_host_alloc(B * np * 2, (void**)&s.label) != SF_OK) return die("host buffers");
  if (sf_host_alloc(B * dn * 2, (void**)&depth) != SF_OK) return die("host buffers");
  std::vector<float> poses(B * 16);
  std::vector<uint64_t> todo;
  for (uint64_t f = 0; f < info.num_frames; f += frame_skip) todo.push_back(f);
  double kernel_ms = 0;
  int cur = 0;
  for (size_t b0 = 0; b0 < todo.size(); b0 += (size_t)B, cur ^= 1) {
    Set& s = sets[cur];
    for (std::thread& t : s.writers) t.join();
    s.writers.clear();
    if (s.failed) return die("output image");
    const int n = (int)std::min((size_t)B, todo.size() - b0);
    s.frames.assign(todo.begin() + b0, todo.begin() + b0 + n);
    std::vector<std::thread> dec;
    std::vector<int> rc(n, SF_OK);
    for (int k = 0; k < n; k++) {
      int valid = 0;
      sf_sens_pose(sd, s.frames[k], &poses[16 * k], &valid);
      if (!valid) { poses[16 * k] = -INFINITY; std::memset(&depth[k * dn], 0, 2 * dn); continue; }
      dec.emplace_back([&, k] { rc[k] = sf_sens_decode_depth(sd, s.frames[k], &depth[k * dn]); });
    }
    for (std::thread& t : dec) t.join();
    for (int k = 0; k < n; k++) if (rc[k] != SF_OK) { std::fprintf(stderr, "depth frame %llu could not be decoded\n", (unsigned long long)s.frames[k]); return 1; }
    float us = 0;
    if (sf_projector_run(proj, n, poses.data(), depth, s.inst, s.label, nullptr, &us) != SF_OK) return die("render");
    kernel_ms += us * 1e-3;
    for (int k = 0; k < n; k++) {
      const std::string name = std::to_string(s.frames[k]) + ".png";
      s.writers.emplace_back([&s, k, np, name, out_inst, &info] { if (sf_png_write_gray((out_inst + name).c_str(), &s.inst[k * np], info.color_width, info.color_height, 8) != SF_OK) s.failed = true; });
      s.writers.emplace_back([&s, k, np, name, out_label, &info] { if (sf_png_write_gray((out_label + name).c_str(), &s.label[k * np], info.color_width, info.color_height, 16) != SF_OK) s.failed = true; });
    }
    std::printf("\r[ %llu | %llu ]", (unsigned long long)(s.frames.back() + 1), (unsigned long long)info.num_frames);
    std::fflush(stdout);
  }
  for (Set& s : sets) {
    for (std::thread& t : s.writers) t.join();
    if (s.failed) return die("output image");
  }
  std::printf("\ndone\n%zu frames, %.1f ms of GPU kernels\n", todo.size(), kernel_ms);
  for (Set& s : sets) { sf_host_free(s.inst); sf_host_free(s.label); }
  sf_host_free(depth);
  sf_projector_destroy(proj);
  sf_sens_close(sd);
  return 0;
}

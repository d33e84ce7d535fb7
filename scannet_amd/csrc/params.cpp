// params.cpp -- error state, version string and the reconstruction-parameter surface.
//
// sf_params_load_file reads the mLib "ParameterFile" text format the reference passes to its native
// tools as argv[1] (Server/scan_processor.py:34-35,138; the shipped instance is
// Server/tools/recons/zParametersScanNet.txt): one `name = value [value ...];` statement per line,
// `//` comments, `f` suffix on floats, quoted strings, true/false.  mLib itself is not in the reference
// tree (.gitmodules:1-3), the grammar is taken from the shipped files and from the in-tree X-macro
// readers (Alignment/src/globalAppState.h:8-41).
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <algorithm>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace sf {
std::string& last_error_ref() {
  static thread_local std::string err;
  return err;
}
}  // namespace sf

SF_API const char* sf_last_error(void) { return sf::last_error_ref().c_str(); }
SF_API int sf_params_upstream_preset(sf_params* p, int which) {
  if (!p) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (which < 0 || which > 2) return sf::fail(SF_ERR_INVALID_ARG, "upstream preset %d (0 = SURVEY App. C, 1 = VoxelHashing / DepthSensing.exe, 2 = BundleFusion / FriedLiver.exe)", which);
  const int on = which != 0;
  p->frustum_mode = p->colour_round = p->colour_first = p->weight_wrap = on;
  p->weight_mode = which == 1;
  return SF_OK;
}

SF_API const char* sf_version(void) { return "scanfuse 0.1 (gfx950)"; }

SF_API void sf_params_default(sf_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->depth_width = 640;
  p->depth_height = 480;
  p->fx = 577.87f; p->fy = 577.87f; p->mx = 319.5f; p->my = 239.5f;  // SURVEY.md 8d camera
  p->depth_shift = 1000.0f;                                            // sensorData.h:895
  p->depth_min = 0.1f; p->depth_max = 6.0f;                            // zParametersScanNet.txt:34-35
  p->voxel_size = 0.004f;                                              // BASELINE.json configs[1] (file value: 0.010, :47)
  p->trunc_base = 0.06f; p->trunc_scale = 0.02f;                       // :49-50
  p->max_integration_dist = 4.0f;                                      // :51
  p->weight_sample = 1;                                                // :52
  p->weight_max = 255;                                                 // :53 says 99999999; a uchar weight saturates
  p->mc_thresh_factor = 10.0f;                                         // :48
  p->hash_num_buckets = 1u << 19;                                      // BASELINE.json configs[1] (file value: 800000, :56)
  p->hash_bucket_size = 10;                                            // HASH_BUCKET_SIZE, :55
  p->num_sdf_blocks = 1u << 20;                                        // 4 GiB of 4 KiB tiles; file value 600000 (:57)
  p->mc_max_triangles = 0;
  p->gc_enabled = 0;
  p->color_width = 0; p->color_height = 0;
  p->cfx = p->cfy = p->cmx = p->cmy = 0.0f;
  p->integration_width = p->integration_height = 0;   // integrate at the input resolution (BASELINE.json's 640x480)
  p->frustum_mode = p->colour_round = p->colour_first = p->weight_mode = p->weight_wrap = 0;   // SURVEY App. C throughout (scanfuse.h: upstream-conformance switches)
}

namespace {

std::string strip(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) a++;
  while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

bool parse_float(const std::string& tok, float* out) {
  std::string t = tok;
  if (!t.empty() && (t.back() == 'f' || t.back() == 'F')) t.pop_back();
  if (t.empty()) return false;
  char* end = nullptr;
  const double v = std::strtod(t.c_str(), &end);
  if (end == t.c_str() || *end != 0) return false;
  *out = (float)v;
  return true;
}

}  // namespace

SF_API int sf_params_load_file(const char* path, sf_params* p) {
  if (!path || !p) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::ifstream in(path);
  if (!in) return sf::fail(SF_ERR_IO, "could not open parameter file %s", path);
  std::map<std::string, std::vector<std::string>> kv;
  std::string line;
  int lineno = 0;
  while (std::getline(in, line)) {
    lineno++;
    // strip // comments (not inside quotes)
    bool inq = false;
    size_t cut = std::string::npos;
    for (size_t i = 0; i + 1 < line.size(); i++) {
      if (line[i] == '"') inq = !inq;
      if (!inq && line[i] == '/' && line[i + 1] == '/') { cut = i; break; }
    }
    if (cut != std::string::npos) line = line.substr(0, cut);
    line = strip(line);
    if (line.empty()) continue;
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;  // mLib ignores lines without an assignment
    std::string name = strip(line.substr(0, eq));
    std::string val = strip(line.substr(eq + 1));
    const size_t semi = val.find(';');
    if (semi != std::string::npos) val = strip(val.substr(0, semi));
    std::vector<std::string> toks;
    if (!val.empty() && val[0] == '"') {
      const size_t q = val.find('"', 1);
      toks.push_back(q == std::string::npos ? val.substr(1) : val.substr(1, q - 1));
    } else {
      std::istringstream ss(val);
      std::string t;
      while (ss >> t) toks.push_back(t);
    }
    if (name.empty()) return sf::fail(SF_ERR_FORMAT, "%s:%d: empty parameter name", path, lineno);
    kv[name] = toks;
  }
  auto getf = [&](const char* k, float* dst) -> int {
    auto it = kv.find(k);
    if (it == kv.end()) return SF_OK;
    if (it->second.empty() || !parse_float(it->second[0], dst)) return sf::fail(SF_ERR_FORMAT, "%s: bad value for %s", path, k);
    return SF_OK;
  };
  auto geti = [&](const char* k, int64_t* dst) -> int {
    auto it = kv.find(k);
    if (it == kv.end()) return SF_OK;
    if (it->second.empty()) return sf::fail(SF_ERR_FORMAT, "%s: bad value for %s", path, k);
    const std::string& t = it->second[0];
    if (t == "true") { *dst = 1; return SF_OK; }
    if (t == "false") { *dst = 0; return SF_OK; }
    float fv;
    if (!parse_float(t, &fv)) return sf::fail(SF_ERR_FORMAT, "%s: bad value for %s", path, k);
    *dst = (int64_t)std::strtoll(t.c_str(), nullptr, 10);
    return SF_OK;
  };
  int rc;
#define GETF(key, field) if ((rc = getf(key, &p->field)) != SF_OK) return rc
  GETF("s_sensorDepthMax", depth_max);
  GETF("s_sensorDepthMin", depth_min);
  GETF("s_SDFVoxelSize", voxel_size);
  GETF("s_SDFMarchingCubeThreshFactor", mc_thresh_factor);
  GETF("s_SDFTruncation", trunc_base);
  GETF("s_SDFTruncationScale", trunc_scale);
  GETF("s_SDFMaxIntegrationDistance", max_integration_dist);
#undef GETF
  int64_t v;
  v = p->weight_sample; if ((rc = geti("s_SDFIntegrationWeightSample", &v)) != SF_OK) return rc; p->weight_sample = (int32_t)v;
  v = p->weight_max;    if ((rc = geti("s_SDFIntegrationWeightMax", &v)) != SF_OK) return rc;    p->weight_max = (int32_t)(v > 0x7FFFFFFF ? 0x7FFFFFFF : v);   // sf_fuser_create clamps to 255 unless weight_wrap
  v = p->hash_num_buckets; if ((rc = geti("s_hashNumBuckets", &v)) != SF_OK) return rc; p->hash_num_buckets = (uint32_t)v;
  v = p->num_sdf_blocks;   if ((rc = geti("s_hashNumSDFBlocks", &v)) != SF_OK) return rc; p->num_sdf_blocks = (uint32_t)v;
  v = p->mc_max_triangles; if ((rc = geti("s_marchingCubesMaxNumTriangles", &v)) != SF_OK) return rc; p->mc_max_triangles = (uint32_t)v;
  v = p->gc_enabled;       if ((rc = geti("s_garbageCollectionEnabled", &v)) != SF_OK) return rc; p->gc_enabled = (int32_t)v;
  // zParametersScanNet.txt:20-21: the size the input depth is resampled to (the input size itself comes from the .sens file)
  v = p->integration_width;  if ((rc = geti("s_integrationWidth", &v)) != SF_OK) return rc; p->integration_width = (int32_t)v;
  v = p->integration_height; if ((rc = geti("s_integrationHeight", &v)) != SF_OK) return rc; p->integration_height = (int32_t)v;
  // upstream-conformance switches (scanfuse.h): not keys of the upstream tools, which ignore names they do not know
  v = -1; if ((rc = geti("s_scanfuseUpstream", &v)) != SF_OK) return rc;
  if (v >= 0 && (rc = sf_params_upstream_preset(p, (int)v)) != SF_OK) return rc;
  v = p->frustum_mode; if ((rc = geti("s_scanfuseFrustumMode", &v)) != SF_OK) return rc; p->frustum_mode = (int32_t)v;
  v = p->colour_round; if ((rc = geti("s_scanfuseColourRound", &v)) != SF_OK) return rc; p->colour_round = (int32_t)v;
  v = p->colour_first; if ((rc = geti("s_scanfuseColourFirst", &v)) != SF_OK) return rc; p->colour_first = (int32_t)v;
  v = p->weight_mode;  if ((rc = geti("s_scanfuseWeightMode", &v)) != SF_OK) return rc;  p->weight_mode = (int32_t)v;
  v = p->weight_wrap;  if ((rc = geti("s_scanfuseWeightWrap", &v)) != SF_OK) return rc;  p->weight_wrap = (int32_t)v;
  if (!(p->voxel_size > 0) || p->hash_num_buckets == 0 || p->num_sdf_blocks == 0)
    return sf::fail(SF_ERR_FORMAT, "%s: non-positive voxel size / hash size", path);
  return SF_OK;
}

namespace {
// CPUs this process may actually use: the cgroup CPU quota when there is one (a container that shows 256 logical CPUs may be allowed
// the time of 16: threads beyond that only add contention), else the hardware concurrency
int usable_cpus_impl() {
  int hw = std::max(1, (int)std::thread::hardware_concurrency());
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
    char q[64] = {0};
    long long period = 0;
    if (std::fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
      const long long quota = std::atoll(q);
      if (quota > 0) hw = std::min(hw, (int)std::max<long long>(1, (quota + period - 1) / period));
    }
    std::fclose(f);
  } else {
    long long quota = -1, period = 0;
    if (FILE* a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(a, "%lld", &quota) != 1) quota = -1; std::fclose(a); }
    if (FILE* b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(b, "%lld", &period) != 1) period = 0; std::fclose(b); }
    if (quota > 0 && period > 0) hw = std::min(hw, (int)std::max<long long>(1, (quota + period - 1) / period));
  }
  return hw;
}
}  // namespace

namespace sf {
int usable_cpus() { return usable_cpus_impl(); }
}  // namespace sf

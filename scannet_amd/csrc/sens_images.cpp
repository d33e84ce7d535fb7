// sens_images.cpp -- a folder of images -> a .sens in memory: SensorData::loadFromImages(sourceFolder, basename = "frame-", colorEnding = "png")
// (SensReader/c++/src/sensorData.h:1468-1559, "7-scenes format"; compiled only with FreeImage there).  Host code, no GPU.
//
//   <folder>/info.txt            the header as saveToImages writes it (:1385-1404): `name = value` lines, the four 4x4 matrices as 16 numbers.  The
//                                reference WRITES `_info.txt` and READS `info.txt` (:1385 against :1476): both names are tried here.
//   <folder>/<basename>%06d.color.<jpg|png>   stored as the frame's colour blob, bytes untouched (:1514-1521,1540-1541)
//   <folder>/<basename>%06d.depth.png         16-bit grey PNG -> the frame's depth (:1524-1526), compressed with zlib like any added frame (:1539);
//                                             where it is missing, <basename>%06d.depth.pgm -- what saveToImages actually writes (:1460) -- is read
//                                             instead, so that `bin/sens` out and `bin/sens --from-images` back is a round trip
//   <folder>/<basename>%06d.pose.txt          camera-to-world, 16 numbers (:1546-1547,1691-1704); "-inf" is the tracking-lost mark
// Frames are taken until one of a frame's three files is missing (:1509-1512); time stamps are 0.  Differences from the reference, all on the tolerant
// side: a sensor name with blanks is read to the end of its line (the reference's `>>` stops at the first blank and derails, :1481), the colour ending
// may be left to what frame 0 has, the file names count as StringCounter does.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "sens.h"

namespace {

std::string counted(const std::string& base, unsigned current, const std::string& ending) {   // StringCounter::getCurrent (:1317-1328), 6 digits
  std::stringstream ss;
  ss << base;
  for (unsigned i = std::max(1u, (unsigned)ceilf(log10f((float)current + 1))); i < 6; i++) ss << "0";
  ss << current << ending;
  return ss.str();
}
bool exists(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return (bool)f;
}
bool slurp(const std::string& p, std::vector<uint8_t>* out) {
  std::ifstream f(p, std::ios::binary | std::ios::ate);
  if (!f) return false;
  const std::streamoff n = f.tellg();
  f.seekg(0, std::ios::beg);
  out->resize((size_t)n);
  return n == 0 || (bool)f.read((char*)out->data(), n);
}
// "name = v v v ..." -> the text after " = " (to the end of the line)
bool value_of(const std::vector<std::string>& lines, const char* name, std::string* out) {
  const std::string key = std::string(name) + " =";
  for (const std::string& ln : lines)
    if (ln.compare(0, key.size(), key) == 0) {
      size_t a = key.size();
      while (a < ln.size() && ln[a] == ' ') a++;
      size_t b = ln.size();
      while (b > a && (ln[b - 1] == ' ' || ln[b - 1] == '\r')) b--;
      *out = ln.substr(a, b - a);
      return true;
    }
  return false;
}
bool floats_of(const std::string& text, float* dst, int n) {
  const char* p = text.c_str();
  for (int i = 0; i < n; i++) {
    char* end = nullptr;
    dst[i] = std::strtof(p, &end);   // "inf" / "-inf" / "nan" included
    if (end == p) return false;
    p = end;
  }
  return true;
}
// binary PGM with 16-bit big-endian samples (saveAsPGM, :1342-1359): "P5", comment lines, width height, maxval, one white-space byte, samples
bool read_pgm16(const std::vector<uint8_t>& file, uint32_t w, uint32_t h, uint16_t* dst) {
  size_t at = 0;
  auto token = [&](std::string* t) {
    for (;;) {
      while (at < file.size() && std::isspace(file[at])) at++;
      if (at < file.size() && file[at] == '#') { while (at < file.size() && file[at] != '\n') at++; continue; }
      break;
    }
    t->clear();
    while (at < file.size() && !std::isspace(file[at])) t->push_back((char)file[at++]);
    return !t->empty();
  };
  std::string t;
  if (!token(&t) || t != "P5") return false;
  unsigned long dims[3];
  for (unsigned long& d : dims) { if (!token(&t)) return false; d = std::strtoul(t.c_str(), nullptr, 10); }
  if (dims[0] != w || dims[1] != h || dims[2] != 65535 || at >= file.size()) return false;
  at++;   // the single white-space byte behind maxval
  if (file.size() - at < (size_t)w * h * 2) return false;
  for (size_t i = 0; i < (size_t)w * h; i++) dst[i] = (uint16_t)((file[at + 2 * i] << 8) | file[at + 2 * i + 1]);
  return true;
}

}  // namespace

SF_API int sf_sens_load_from_images(const char* folder_, const char* basename_, const char* color_ending_, sf_sens** out) {
  if (!folder_ || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const std::string folder = folder_, base = folder + "/" + (basename_ ? basename_ : "frame-");
  std::ifstream meta(folder + "/info.txt");
  if (!meta) meta.open(folder + "/_info.txt");
  if (!meta) return sf::fail(SF_ERR_IO, "no info.txt (or _info.txt) in %s", folder.c_str());
  std::vector<std::string> lines;
  for (std::string ln; std::getline(meta, ln);) lines.push_back(ln);
  sf_sens_info h;
  std::memset(&h, 0, sizeof h);
  std::string v;
  auto u32 = [&](const char* name, uint32_t* dst) { if (!value_of(lines, name, &v)) return false; *dst = (uint32_t)std::strtoul(v.c_str(), nullptr, 10); return true; };
  if (!u32("m_colorWidth", &h.color_width) || !u32("m_colorHeight", &h.color_height) || !u32("m_depthWidth", &h.depth_width) || !u32("m_depthHeight", &h.depth_height))
    return sf::fail(SF_ERR_FORMAT, "info.txt: frame sizes missing");
  if (!value_of(lines, "m_depthShift", &v) || !floats_of(v, &h.depth_shift, 1)) return sf::fail(SF_ERR_FORMAT, "info.txt: m_depthShift missing");
  if (value_of(lines, "m_sensorName", &v)) std::snprintf(h.sensor_name, sizeof h.sensor_name, "%s", v.c_str());
  const struct { const char* name; float* m; } mats[4] = {{"m_calibrationColorIntrinsic", h.color_intrinsic}, {"m_calibrationColorExtrinsic", h.color_extrinsic},
                                                         {"m_calibrationDepthIntrinsic", h.depth_intrinsic}, {"m_calibrationDepthExtrinsic", h.depth_extrinsic}};
  for (const auto& m : mats)
    if (!value_of(lines, m.name, &v) || !floats_of(v, m.m, 16)) return sf::fail(SF_ERR_FORMAT, "info.txt: %s missing or short", m.name);
  if ((uint64_t)h.depth_width * h.depth_height == 0 || (uint64_t)h.depth_width * h.depth_height > (1ull << 28)) return sf::fail(SF_ERR_FORMAT, "info.txt: depth frame size");
  std::string ending = color_ending_ ? color_ending_ : "";
  if (ending.empty()) ending = exists(counted(base, 0, ".color.jpg")) ? "jpg" : "png";
  if (ending != "png" && ending != "jpg") return sf::fail(SF_ERR_INVALID_ARG, "invalid color format %s", ending.c_str());
  h.color_compression = ending == "jpg" ? 2 : 1;
  h.depth_compression = 1;
  sf_sens* s = nullptr;
  int rc = sf_sens_create(&h, &s);
  if (rc != SF_OK) return rc;
  std::vector<uint16_t> depth((size_t)h.depth_width * h.depth_height);
  std::vector<uint8_t> color, file;
  for (unsigned i = 0;; i++) {
    const std::string cf = counted(base, i, ".color." + ending), df = counted(base, i, ".depth.png"), dg = counted(base, i, ".depth.pgm"), pf = counted(base, i, ".pose.txt");
    const bool png = exists(df);
    if (!exists(cf) || (!png && !exists(dg)) || !exists(pf)) break;   // "DONE" (:1509-1512)
    if (!slurp(cf, &color)) { rc = sf::fail(SF_ERR_IO, "cannot read %s", cf.c_str()); break; }
    if (png) {
      uint32_t w = 0, hh = 0;
      int ch = 0, bits = 0;
      void* data = nullptr;
      rc = sf_png_read(df.c_str(), &w, &hh, &ch, &bits, &data);
      if (rc != SF_OK) break;
      const bool fits = w == h.depth_width && hh == h.depth_height && ch == 1 && bits == 16;
      if (fits) std::memcpy(depth.data(), data, depth.size() * 2);
      sf_free(data);
      if (!fits) { rc = sf::fail(SF_ERR_FORMAT, "%s is not a %ux%u 16-bit grey image", df.c_str(), h.depth_width, h.depth_height); break; }
    } else if (!slurp(dg, &file) || !read_pgm16(file, h.depth_width, h.depth_height, depth.data())) {
      rc = sf::fail(SF_ERR_FORMAT, "%s is not a %ux%u binary PGM with 16-bit samples", dg.c_str(), h.depth_width, h.depth_height);
      break;
    }
    float pose[16];
    std::ifstream pin(pf);
    std::stringstream ptext;
    ptext << pin.rdbuf();
    if (!floats_of(ptext.str(), pose, 16)) { rc = sf::fail(SF_ERR_FORMAT, "%s does not hold 16 numbers", pf.c_str()); break; }
    rc = sf_sens_add_frame(s, color.empty() ? nullptr : color.data(), color.size(), depth.data(), pose, 0, 0);
    if (rc != SF_OK) break;
  }
  if (rc != SF_OK) { sf_sens_close(s); return rc; }
  *out = s;
  return SF_OK;
}

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
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
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

// ---- the other direction: SensorData::saveToImages(outputFolder, basename = "frame-") (sensorData.h:1380-1466) ---------------------------------------
//     <folder>/_info.txt                       :1383-1407  names " = " values; the four 4x4 matrices row-major, 16 numbers and a trailing blank
//     <folder>/<basename>%06d.color.jpg | .png :1431-1450  the stored JPEG / PNG blob as it is; a TYPE_RAW frame as a PNG made here (the reference needs its
//                                                          Windows-only encoder for that one, :576-593: off Windows it throws on the first frame)
//     <folder>/<basename>%06d.depth.pgm        :1342-1359  binary PGM, comment line with the depth shift, 16-bit samples big-endian
//     <folder>/<basename>%06d.pose.txt         :1706-1714  camera-to-world, four rows, no newline after the last
// Every number goes through the iostream formatting the reference uses, the names count as its StringCounter does: the folder is the compiled
// reference's byte for byte (tests/test_sens_export.py).  The frames are independent (three files each): a pool of threads takes them in index order;
// progress(i, n, user) is called on the CALLER's thread, in index order, before frame i is waited for -- where the reference prints its progress line.
SF_API int sf_sens_save_to_images(const sf_sens* s, const char* folder_, const char* basename_, void (*progress)(uint64_t, uint64_t, void*), void* user) {
  if (!s || !folder_) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const std::string out_dir = folder_, base = out_dir + "/" + (basename_ ? basename_ : "frame-");
  sf_sens_info info;
  sf_sens_get_info(s, &info);
  struct stat st;
  if (::stat(out_dir.c_str(), &st) != 0) ::mkdir(out_dir.c_str(), 0777);   // one level, as ml::util::makeDirectory
  {
    std::ofstream meta(out_dir + "/_info.txt");
    if (!meta) return sf::fail(SF_ERR_IO, "cannot open file %s/_info.txt", out_dir.c_str());
    meta << "m_versionNumber = " << info.version << '\n';
    meta << "m_sensorName = " << info.sensor_name << '\n';
    meta << "m_colorWidth = " << info.color_width << '\n';
    meta << "m_colorHeight = " << info.color_height << '\n';
    meta << "m_depthWidth = " << info.depth_width << '\n';
    meta << "m_depthHeight = " << info.depth_height << '\n';
    meta << "m_depthShift = " << info.depth_shift << '\n';
    const struct { const char* name; const float* m; } mats[4] = {{"m_calibrationColorIntrinsic", info.color_intrinsic}, {"m_calibrationColorExtrinsic", info.color_extrinsic},
                                                                 {"m_calibrationDepthIntrinsic", info.depth_intrinsic}, {"m_calibrationDepthExtrinsic", info.depth_extrinsic}};
    for (const auto& m : mats) {
      meta << m.name << " = ";
      for (int i = 0; i < 16; i++) meta << m.m[i] << " ";
      meta << "\n";
    }
    meta << "m_frames.size = " << info.num_frames << "\n";
  }
  const uint64_t n = info.num_frames;
  if (n == 0) return SF_OK;
  const std::string color_ending = info.color_compression == 2 ? "jpg" : "png";
  std::vector<char> done(n, 0);
  std::vector<std::string> error(n);
  std::atomic<uint64_t> next{0};
  std::atomic<bool> failed{false};
  std::mutex mu;
  std::condition_variable cv;
  auto work = [&]() {
    std::vector<uint16_t> depth((size_t)info.depth_width * info.depth_height);
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= n) return;
      std::string err;
      if (!failed.load()) {
        const std::string color_file = counted(base, (unsigned)i, ".color." + color_ending), pose_file = counted(base, (unsigned)i, ".pose.txt"),
                          pgm_file = counted(base, (unsigned)i, ".depth.pgm");
        const uint8_t *cblob = nullptr, *dblob = nullptr;
        uint64_t cbytes = 0, dbytes = 0;
        if (sf_sens_frame_blobs(s, i, &cblob, &cbytes, &dblob, &dbytes) != SF_OK) err = sf_last_error();
        else if (info.color_compression == 0 && cbytes != 0) {   // TYPE_RAW pixels: a PNG of them
          if (cbytes != (uint64_t)info.color_width * info.color_height * 3) err = "raw colour frame of " + std::to_string(cbytes) + " bytes";
          else if (sf_png_write(color_file.c_str(), cblob, info.color_width, info.color_height, 3, 8) != SF_OK) err = "cannot open file " + color_file;
        } else if (info.color_compression == 0) {
          // a TYPE_RAW file without colour (depth only): no colour file
        } else if (info.color_compression == 1 || info.color_compression == 2) {
          FILE* fp = std::fopen(color_file.c_str(), "wb");
          const bool ok = fp && (cbytes == 0 || std::fwrite(cblob, 1, (size_t)cbytes, fp) == cbytes);
          if (fp) std::fclose(fp);
          if (!ok) err = "cannot open file " + color_file;
        } else {
          err = "unknown format";
        }
        if (err.empty() && sens_decode_depth(s, i, depth.data()) != SF_OK) err = sf_last_error();
        if (err.empty()) {
          std::ofstream of(pgm_file, std::ios::binary);
          std::stringstream ss;
          ss << "P5\n";
          ss << "# data values are 16-bit each; depth shift is " << info.depth_shift << "\n";
          ss << info.depth_width << " " << info.depth_height << "\n";
          ss << std::numeric_limits<unsigned short>::max() << "\n";
          of << ss.str();
          for (uint16_t& v : depth) v = (uint16_t)((v << 8) | (v >> 8));   // PGM samples are big-endian
          of.write((const char*)depth.data(), (std::streamsize)(depth.size() * 2));
          const float* m = s->frames[i].pose;
          std::ofstream pf(pose_file);
          pf << m[0] << " " << m[1] << " " << m[2] << " " << m[3] << "\n"
             << m[4] << " " << m[5] << " " << m[6] << " " << m[7] << "\n"
             << m[8] << " " << m[9] << " " << m[10] << " " << m[11] << "\n"
             << m[12] << " " << m[13] << " " << m[14] << " " << m[15];
          if (!of || !pf) err = "cannot write " + pgm_file;
        }
        if (!err.empty()) failed.store(true);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        error[i] = err;
        done[i] = 1;
      }
      cv.notify_all();
    }
  };
  const uint64_t T = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)sf::usable_cpus(), 16), n));
  std::vector<std::thread> pool;
  for (uint64_t t = 0; t < T; t++) pool.emplace_back(work);
  std::string first_error;
  for (uint64_t i = 0; i < n && first_error.empty(); i++) {
    if (progress) progress(i, n, user);
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done[i] != 0; });
    first_error = error[i];
  }
  for (auto& t : pool) t.join();
  if (!first_error.empty()) return sf::fail(SF_ERR_IO, "%s", first_error.c_str());
  return SF_OK;
}

// SensorData::saveToPointCloud(filename, frameFrom, frameTo) (sensorData.h:1564-1602; compiled only where mLib is present -- SURVEY 8a row a6 cites it as the
// reference's statement of the unprojection): every valid depth pixel of frames [from, to) as a world-space point with the colour the colour camera sees
// there.  Per pixel, in fp32 and in the reference's order:
//     d = (float)depth / depthShift;  cam = K_depth^-1 * (x d, y d, d, 0);  world = camToWorld * cam   (identity when the pose is -inf or starts with 0, :1573)
//     c = K_colour * (E_depth * cam);  u = c.x / c.z, v = c.y / c.z;  pixel = round(u), round(v);  inside the colour image: its rgb, alpha 255; else (0, 0, 0, 0)
// K^-1 by cofactors and one division of the determinant (the public mLib's Matrix4x4::getInverse; mLib itself is not in the reference tree, so the last bits
// of that inverse are not pinned).  Output: binary little-endian PLY, vertex = float x, y, z + uchar red, green, blue, alpha -- the layout the pipeline's
// other PLY files have (README.md:45-46).  frame_to = 0: one frame.
namespace {
bool invert4(const float* m, float* inv) {
  float t[16];
  t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
  if (det == 0.0f) return false;
  const float r = 1.0f / det;
  for (int i = 0; i < 16; i++) inv[i] = t[i] * r;
  return true;
}
inline void mul_point(const float* m, const float* p, float* o) {   // rows 0..2 of m * (p, 1)
  for (int r = 0; r < 3; r++) o[r] = m[4 * r] * p[0] + m[4 * r + 1] * p[1] + m[4 * r + 2] * p[2] + m[4 * r + 3];
}
}  // namespace

SF_API int sf_sens_save_point_cloud(const sf_sens* s, const char* ply_path, uint64_t frame_from, uint64_t frame_to, uint64_t* n_points) {
  if (!s || !ply_path) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  try {
    sf_sens_info info;
    sf_sens_get_info(s, &info);
    if (frame_to == 0) frame_to = frame_from + 1;
    if (frame_from >= frame_to || frame_to > info.num_frames) return sf::fail(SF_ERR_BOUNDS, "frames [%llu, %llu) of %llu", (unsigned long long)frame_from, (unsigned long long)frame_to, (unsigned long long)info.num_frames);
    float kinv[16];
    if (!invert4(info.depth_intrinsic, kinv)) return sf::fail(SF_ERR_FORMAT, "the depth intrinsic is singular");
    const uint32_t W = info.depth_width, H = info.depth_height, CW = info.color_width, CH = info.color_height;
    std::vector<uint16_t> depth((size_t)W * H);
    std::vector<uint8_t> color((size_t)CW * CH * 3);
    struct Vtx { float x, y, z; uint8_t r, g, b, a; };
    static_assert(sizeof(Vtx) == 16, "PLY vertex record");
    std::vector<Vtx> pts;
    for (uint64_t f = frame_from; f < frame_to; f++) {
      int rc = sf_sens_decode_depth(s, f, depth.data());
      if (rc != SF_OK) return rc;
      sf_sens_frame_meta_t meta;
      sf_sens_frame_meta(s, f, &meta);
      const bool has_color = meta.color_bytes != 0 && CW != 0 && CH != 0;
      if (has_color && (rc = sf_sens_decode_color(s, f, color.data())) != SF_OK) return rc;
      float T[16];
      int valid = 0;
      sf_sens_pose(s, f, T, &valid);
      if (T[0] == -std::numeric_limits<float>::infinity() || T[0] == 0.0f) {
        std::memset(T, 0, sizeof T);
        T[0] = T[5] = T[10] = T[15] = 1.0f;
      }
      for (uint32_t i = 0; i < W * H; i++) {
        if (depth[i] == 0) continue;
        const uint32_t x = i % W, y = i / W;
        const float d = (float)depth[i] / info.depth_shift;
        const float v4[3] = {(float)x * d, (float)y * d, d};   // w = 0: the fourth column of K^-1 does not take part
        float cam[3], world[3], cf[3], cc[3];
        for (int r = 0; r < 3; r++) cam[r] = kinv[4 * r] * v4[0] + kinv[4 * r + 1] * v4[1] + kinv[4 * r + 2] * v4[2];
        mul_point(T, cam, world);
        mul_point(info.depth_extrinsic, cam, cf);
        mul_point(info.color_intrinsic, cf, cc);
        Vtx v{world[0], world[1], world[2], 0, 0, 0, 0};
        if (has_color) {
          const float u = cc[0] / cc[2], w = cc[1] / cc[2];
          const long long px = (long long)std::floor(u + 0.5f), py = (long long)std::floor(w + 0.5f);
          if (px >= 0 && px < (long long)CW && py >= 0 && py < (long long)CH) {
            const uint8_t* c = &color[3 * ((size_t)py * CW + (size_t)px)];
            v.r = c[0]; v.g = c[1]; v.b = c[2]; v.a = 255;
          }
        }
        pts.push_back(v);
      }
    }
    FILE* fp = std::fopen(ply_path, "wb");
    if (!fp) return sf::fail(SF_ERR_IO, "cannot open file %s", ply_path);
    std::fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\n"
                     "property uchar blue\nproperty uchar alpha\nend_header\n", pts.size());
    const bool ok = pts.empty() || std::fwrite(pts.data(), sizeof(Vtx), pts.size(), fp) == pts.size();
    if (std::fclose(fp) != 0 || !ok) return sf::fail(SF_ERR_IO, "short write to %s", ply_path);
    if (n_points) *n_points = pts.size();
    return SF_OK;
  } catch (...) { return sf::fail(SF_ERR_IO, "out of memory building the point cloud"); }
}

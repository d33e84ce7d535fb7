// occipital.cpp -- the ScannerApp's depth stream: Occipital shift codec, shift -> millimetre table, `.depth` / `.imu` /
// `.txt` capture files, and the `convert` stage that turns a capture into a `.sens`.
//
// Replaces (host C++, as the reference):
//   uplinksimple::decode / encode      ScannerApp/depth2pgm/uplinksimple_image-codecs.h:157-249, :253-396 (bit reader :120-148,
//                                      bit writer :38-66); the variable-length code is documented at :160-176
//   uplinksimple::shift2depth          ScannerApp/depth2pgm/uplinksimple_shift2depth.h:9-90
//   the per-frame loop of the Converter  Converter/main.cpp:72-103 (u32 size + stream, decode, shift2depth, values >=
//                                      shift2depth(0xffff) -> 0), time stamps :118-133, IMU records :136-154, MetaData
//                                      Converter/src/metaData.h:17-60; the capture layout the ScannerApp writes:
//                                      ScannerApp/Scanner/ViewController+Sensor.mm:52-96,796-805, ViewController.mm:531-574
// Checker (tests/test_occipital.py): the reference headers themselves, compiled where they lie by the test infrastructure,
// on every table entry and on random / adversarial streams.
//
// The code: last value starts at 0;  00 same | 11 +1 | 10 -1 | 010 bbbbb: N+5 repeats of the current value | 0111 + 11 bits:
// new value | 01101 +2 | 01100 -2;  values are 16-bit with wrap-around (lastVal - 1 at 0 is 0xFFFF, as the reference's uint16_t).
// Unlike the reference (asserts only, reads past the buffer on a truncated stream) the decoder is bounds-checked: a stream
// that ends early is SF_ERR_FORMAT.  A run that overshoots the frame is clipped (the reference would write past its buffer).
//
// shift2depth: the reference ships a 1105-entry table.  Every entry 1..1104 equals floor(300000 / (1134.8335 - shift)) --
// the sensor's disparity model -- so the table is GENERATED here from that closed form in exact integer arithmetic
// (3e9 / (11348335 - 10000 shift)); entry 0 is 0 and shifts >= 1105 saturate at the last entry, 9729.  The generated
// table is compared with the compiled reference for all 65536 inputs in tests/test_occipital.py.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "sens.h"

namespace {

constexpr int OCC_TABLE = 1105;

struct ShiftTable {   // built once (thread-safe function-local static: decode pools call this from many threads)
  uint16_t t[OCC_TABLE];
  ShiftTable() {
    t[0] = 0;
    for (int s = 1; s < OCC_TABLE; s++) t[s] = (uint16_t)(3000000000ll / (11348335ll - 10000ll * s));
  }
};
const uint16_t* shift_table() {
  static const ShiftTable table;
  return table.t;
}

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;
  int have = 0;
  bool ok = true;
  BitReader(const uint8_t* b, uint64_t n) : p(b), end(b + n) {}
  // n <= 16
  uint32_t get(int n) {
    while (have < n) {
      if (p == end) { ok = false; return 0; }
      acc = (acc << 8) | *p++;
      have += 8;
    }
    have -= n;
    return (uint32_t)(acc >> have) & ((1u << n) - 1u);
  }
};

struct BitWriter {
  uint8_t* p;
  uint8_t* end;
  uint64_t acc = 0;
  int have = 0;
  bool ok = true;
  BitWriter(uint8_t* b, uint64_t n) : p(b), end(b + n) {}
  void put(uint32_t bits, int n) {
    acc = (acc << n) | (bits & ((1u << n) - 1u));
    have += n;
    while (have >= 8) {
      if (p == end) { ok = false; have -= 8; continue; }
      *p++ = (uint8_t)(acc >> (have - 8));
      have -= 8;
    }
  }
  // the reference's bs_flush + bs_bytes_used: a partly filled byte is written (zero padded) and counted
  uint8_t* finish() {
    if (have > 0) {
      if (p == end) ok = false;
      else *p++ = (uint8_t)(acc << (8 - have));
      have = 0;
    }
    return p;
  }
};

void burn_zeros(BitWriter& w, int& zeros) {
  while (zeros > 0) {
    if (zeros <= 4) { w.put(0, 2 * zeros); zeros = 0; }
    else {
      const int n = std::min(zeros - 5, 31);
      w.put(0x2, 3);
      w.put((uint32_t)n, 5);
      zeros -= n + 5;
    }
  }
}

}  // namespace

SF_API int sf_occ_decode(const uint8_t* stream, uint64_t stream_bytes, uint64_t num_elements, uint16_t* out) {
  if ((!stream && stream_bytes) || (!out && num_elements)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  BitReader r(stream, stream_bytes);
  uint16_t last = 0, cur = 0;
  uint64_t i = 0;
  while (i < num_elements) {
    const uint32_t b01 = r.get(2);
    if (!r.ok) break;
    if (b01 == 0) { cur = last; out[i++] = cur; }                                  // 00
    else if (b01 & 2) { cur = (uint16_t)(last + ((b01 & 1) ? 1 : -1)); out[i++] = cur; last = cur; }  // 11 / 10
    else {                                                                         // 01...
      if (r.get(1) == 0) {                                                         // 010 bbbbb: run of the CURRENT value
        uint64_t n = (uint64_t)r.get(5) + 5;
        if (!r.ok) break;
        n = std::min(n, num_elements - i);
        for (uint64_t k = 0; k < n; k++) out[i + k] = cur;
        i += n;
      } else if (r.get(1) == 0) {                                                  // 0110 d: +-2
        cur = (uint16_t)(last + (r.get(1) ? 2 : -2));
        if (!r.ok) break;
        out[i++] = cur; last = cur;
      } else {                                                                     // 0111 + 11 bits
        cur = (uint16_t)r.get(11);
        if (!r.ok) break;
        out[i++] = cur; last = cur;
      }
    }
  }
  if (!r.ok) return sf::fail(SF_ERR_FORMAT, "Occipital depth stream ends after %llu of %llu values", (unsigned long long)i, (unsigned long long)num_elements);
  return SF_OK;
}

SF_API uint64_t sf_occ_encode_bound(uint64_t num_elements) { return 2 * num_elements + 16; }  // worst case 15 bits per value

SF_API int sf_occ_encode(const uint16_t* in, uint64_t num_elements, uint8_t* out, uint64_t out_capacity, uint64_t* out_bytes) {
  if ((!in && num_elements) || !out || !out_bytes) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  for (uint64_t i = 0; i < num_elements; i++)
    if (in[i] > 2047) return sf::fail(SF_ERR_INVALID_ARG, "sf_occ_encode: value %u at %llu does not fit the code's 11 bits", in[i], (unsigned long long)i);
  BitWriter w(out, out_capacity);
  int zeros = 0, last = 0;
  for (uint64_t i = 0; i < num_elements; i++) {
    const int cur = in[i], delta = cur - last;
    if (delta == 0) zeros++;
    else {
      burn_zeros(w, zeros);
      if (delta == 1 || delta == -1) w.put(delta == 1 ? 0x3 : 0x2, 2);
      else if (delta == 2 || delta == -2) w.put(delta == 2 ? 0xD : 0xC, 5);
      else { w.put(0x7, 4); w.put((uint32_t)cur >> 8, 3); w.put((uint32_t)cur, 8); }  // 11 bits (the reference corrupts its stream on larger values: rejected above)
    }
    last = cur;
  }
  burn_zeros(w, zeros);
  const uint8_t* e = w.finish();
  if (!w.ok) return sf::fail(SF_ERR_BOUNDS, "sf_occ_encode: output buffer of %llu bytes is too small", (unsigned long long)out_capacity);
  *out_bytes = (uint64_t)(e - out);
  return SF_OK;
}

SF_API uint16_t sf_occ_shift2depth(uint16_t shift) {
  const uint16_t* t = shift_table();
  return shift < OCC_TABLE ? t[shift] : t[OCC_TABLE - 1];
}

// In place: shift -> millimetres; with zero_invalid != 0 values >= shift2depth(0xffff) become 0 (Converter/main.cpp:89-93).
SF_API int sf_occ_shift2depth_buffer(uint16_t* buf, uint64_t n, int zero_invalid) {
  if (!buf && n) return sf::fail(SF_ERR_INVALID_ARG, "NULL buffer");
  const uint16_t* t = shift_table();
  const uint16_t top = t[OCC_TABLE - 1];
  for (uint64_t i = 0; i < n; i++) {
    uint16_t d = buf[i] < OCC_TABLE ? t[buf[i]] : top;
    if (zero_invalid && d >= top) d = 0;
    buf[i] = d;
  }
  return SF_OK;
}

// ---- capture files ------------------------------------------------------------------------------------------------
struct sf_capture {
  std::string base;                 // path without extension
  sf_capture_meta meta;
  std::vector<uint8_t> depth_file;  // whole .depth file
  std::vector<uint64_t> frame_off;  // offset of each frame's stream
  std::vector<uint32_t> frame_len;
  std::vector<double> ts_depth;     // seconds
  std::vector<uint8_t> imu;         // .imu file (16 doubles per record)
};

namespace {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return false;
  const std::streamsize n = f.tellg();
  f.seekg(0);
  out.resize((size_t)n);
  if (n && !f.read((char*)out.data(), n)) return false;
  return true;
}

std::string strip_ext(const std::string& p) {
  const size_t slash = p.find_last_of('/');
  const size_t dot = p.find('.', slash == std::string::npos ? 0 : slash + 1);  // mLib removeExtensions: everything after the first dot of the name
  return dot == std::string::npos ? p : p.substr(0, dot);
}

// mLib ParameterFile as the ScannerApp writes it: `name = value` lines, CRLF, values may hold spaces (the 16 extrinsics)
int parse_meta(const std::string& path, sf_capture_meta& m) {
  std::ifstream f(path);
  if (!f) return sf::fail(SF_ERR_IO, "file not found %s", path.c_str());
  std::memset(&m, 0, sizeof(m));
  for (int i = 0; i < 4; i++) m.color_to_depth_extrinsics[5 * i] = 1.0f;
  std::string line;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ' || line.back() == ';')) line.pop_back();
    const size_t c = line.find("//");
    if (c != std::string::npos) line.resize(c);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string name = line.substr(0, eq), value = line.substr(eq + 1);
    auto trim = [](std::string& s) {
      size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
      s = a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    trim(name); trim(value);
    const double v = std::atof(value.c_str());
    if (name == "numColorFrames") m.num_color_frames = (uint32_t)v;
    else if (name == "numDepthFrames") m.num_depth_frames = (uint32_t)v;
    else if (name == "numIMUmeasurements") m.num_imu = (uint32_t)v;
    else if (name == "colorWidth") m.color_width = (uint32_t)v;
    else if (name == "colorHeight") m.color_height = (uint32_t)v;
    else if (name == "depthWidth") m.depth_width = (uint32_t)v;
    else if (name == "depthHeight") m.depth_height = (uint32_t)v;
    else if (name == "fx_color") m.fx_color = (float)v;
    else if (name == "fy_color") m.fy_color = (float)v;
    else if (name == "mx_color") m.mx_color = (float)v;
    else if (name == "my_color") m.my_color = (float)v;
    else if (name == "fx_depth") m.fx_depth = (float)v;
    else if (name == "fy_depth") m.fy_depth = (float)v;
    else if (name == "mx_depth") m.mx_depth = (float)v;
    else if (name == "my_depth") m.my_depth = (float)v;
    else if (name == "colorToDepthExtrinsics") {
      std::istringstream is(value);
      float e[16];
      int k = 0;
      while (k < 16 && (is >> e[k])) k++;
      if (k == 16) { std::memcpy(m.color_to_depth_extrinsics, e, sizeof(e)); m.has_extrinsics = 1; }
    }
  }
  if (!m.depth_width || !m.depth_height) return sf::fail(SF_ERR_FORMAT, "%s: depthWidth / depthHeight missing", path.c_str());
  return SF_OK;
}

// 4x4 inverse (row-major), double arithmetic, cofactor expansion
bool invert4(const float* a, float* out) {
  double m[16], inv[16];
  for (int i = 0; i < 16; i++) m[i] = a[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return false;
  for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] / det);
  return true;
}

uint64_t seconds_to_us(double d) { return (uint64_t)(d * 1000.0 * 1000.0); }  // Converter/main.cpp:11-13

}  // namespace

SF_API int sf_capture_open(const char* any_capture_file, sf_capture** out) {
  if (!any_capture_file || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_capture* c = new sf_capture();
  c->base = strip_ext(any_capture_file);
  int rc = parse_meta(c->base + ".txt", c->meta);
  if (rc != SF_OK) { delete c; return rc; }
  if (!read_file(c->base + ".depth", c->depth_file)) { const std::string b = c->base; delete c; return sf::fail(SF_ERR_IO, "file not found %s.depth", b.c_str()); }
  // index the frames: u32 size + stream, numDepthFrames times; then numDepthFrames doubles (depth time stamps)
  const uint64_t n = c->meta.num_depth_frames, total = c->depth_file.size();
  uint64_t off = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (off + 4 > total) { delete c; return sf::fail(SF_ERR_FORMAT, ".depth file ends inside frame %llu of %llu", (unsigned long long)i, (unsigned long long)n); }
    uint32_t len;
    std::memcpy(&len, &c->depth_file[off], 4);
    off += 4;
    if (off + len > total) { delete c; return sf::fail(SF_ERR_FORMAT, ".depth file ends inside frame %llu of %llu", (unsigned long long)i, (unsigned long long)n); }
    c->frame_off.push_back(off);
    c->frame_len.push_back(len);
    off += len;
  }
  c->ts_depth.assign(n, 0.0);
  if (n != 0 && off + 8 * n <= total) std::memcpy(c->ts_depth.data(), &c->depth_file[off], 8 * n);  // a capture cut short has none: zeros (n = 0: nothing to copy, and no NULL into memcpy)
  if (!read_file(c->base + ".imu", c->imu)) c->imu.clear();  // optional here; the convert tool insists on it as the reference does
  *out = c;
  return SF_OK;
}
SF_API void sf_capture_close(sf_capture* c) { delete c; }
SF_API int sf_capture_get_meta(const sf_capture* c, sf_capture_meta* out) {
  if (!c || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  *out = c->meta;
  return SF_OK;
}
// depth frame `frame` in millimetres (decode + shift2depth + invalid -> 0), and its time stamp in microseconds
SF_API int sf_capture_decode_depth(const sf_capture* c, uint64_t frame, uint16_t* dst, uint64_t* timestamp_us) {
  if (!c || !dst) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= c->frame_off.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu of %zu", (unsigned long long)frame, c->frame_off.size());
  const uint64_t n = (uint64_t)c->meta.depth_width * c->meta.depth_height;
  const int rc = sf_occ_decode(&c->depth_file[c->frame_off[frame]], c->frame_len[frame], n, dst);
  if (rc != SF_OK) return rc;
  sf_occ_shift2depth_buffer(dst, n, 1);
  if (timestamp_us) *timestamp_us = seconds_to_us(c->ts_depth[frame]);
  return SF_OK;
}

// The convert stage (Converter/main.cpp:16-179 minus ffmpeg): capture -> .sens with TYPE_ZLIB_USHORT depth, depthShift 1000,
// sensor name "StructureSensor", identity poses, colour frames passed in by the caller's callback (JPEG blobs or raw RGB) or
// absent.  Depth frames are decoded and deflated by `threads` workers, appended in order.
SF_API int sf_capture_convert(const sf_capture* c, const char* out_sens, sf_capture_color_fn color_fn, void* color_user, int color_compression,
                              int threads, sf_convert_stats* stats) {
  if (!c || !out_sens) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const sf_capture_meta& m = c->meta;
  sf_sens_info h;
  std::memset(&h, 0, sizeof(h));
  h.version = 4;
  h.color_width = m.color_width; h.color_height = m.color_height;
  h.depth_width = m.depth_width; h.depth_height = m.depth_height;
  h.color_compression = color_fn ? color_compression : 2;  // the reference always declares TYPE_JPEG (main.cpp:39)
  h.depth_compression = 1;                                  // TYPE_ZLIB_USHORT (:40)
  h.depth_shift = 1000.0f;                                  // :41
  std::snprintf(h.sensor_name, sizeof(h.sensor_name), "StructureSensor");  // :42
  auto intr = [](float* k, float fx, float fy, float mx, float my) {
    std::memset(k, 0, 64);
    k[0] = fx; k[2] = mx; k[5] = fy; k[6] = my; k[10] = 1.0f; k[15] = 1.0f;  // metaData.h:31-37
  };
  intr(h.color_intrinsic, m.fx_color, m.fy_color, m.mx_color, m.my_color);
  intr(h.depth_intrinsic, m.fx_depth, m.fy_depth, m.mx_depth, m.my_depth);
  for (int i = 0; i < 4; i++) h.color_extrinsic[5 * i] = h.depth_extrinsic[5 * i] = 1.0f;
  if (m.has_extrinsics && !invert4(m.color_to_depth_extrinsics, h.depth_extrinsic))  // metaData.h:47-48: depthToColor = inverse
    return sf::fail(SF_ERR_FORMAT, "colorToDepthExtrinsics is singular");
  sf_sens* s = nullptr;
  int rc = sf_sens_create(&h, &s);
  if (rc != SF_OK) return rc;
  uint64_t n = std::min(m.num_depth_frames, m.num_color_frames ? m.num_color_frames : m.num_depth_frames);  // main.cpp:62-63
  n = std::min<uint64_t>(n, c->frame_off.size());
  const uint64_t npx = (uint64_t)m.depth_width * m.depth_height;
  int nthreads = threads > 0 ? threads : sf::usable_cpus();
  nthreads = std::max(1, std::min(nthreads, 64));
  // decode in parallel batches, append in order
  const uint64_t batch = (uint64_t)nthreads * 4;
  std::vector<uint16_t> depth(batch * npx);
  std::vector<int> rcs(batch);
  std::vector<std::string> errs(batch);
  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  uint64_t done = 0, depth_bytes_in = 0;
  std::vector<uint8_t> color;
  while (done < n && rc == SF_OK) {
    const uint64_t cnt = std::min(batch, n - done);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
      for (;;) {
        const uint64_t k = next.fetch_add(1);
        if (k >= cnt) return;
        rcs[k] = sf_capture_decode_depth(c, done + k, &depth[k * npx], nullptr);
        if (rcs[k] != SF_OK) errs[k] = sf_last_error();
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads && (uint64_t)t < cnt; t++) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
    for (uint64_t k = 0; k < cnt && rc == SF_OK; k++) {
      if (rcs[k] != SF_OK) { rc = sf::fail(rcs[k], "frame %llu: %s", (unsigned long long)(done + k), errs[k].c_str()); break; }
      const uint8_t* cptr = nullptr;
      uint64_t cbytes = 0;
      if (color_fn) {
        const int crc = color_fn(color_user, done + k, &cptr, &cbytes);
        if (crc != SF_OK) { rc = sf::fail(crc, "colour frame %llu unavailable", (unsigned long long)(done + k)); break; }
      }
      const uint64_t ts = seconds_to_us(c->ts_depth[done + k]);  // the depth time stamp serves both (main.cpp:121-127)
      rc = sf_sens_add_frame(s, cptr, cbytes, &depth[k * npx], identity, ts, ts);
      depth_bytes_in += c->frame_len[done + k];
    }
    done += cnt;
  }
  uint64_t imu_kept = 0, imu_skipped = 0;
  if (rc == SF_OK) {
    const uint64_t rec = 16 * 8, nrec = std::min<uint64_t>(m.num_imu, c->imu.size() / rec);
    for (uint64_t i = 0; i < nrec; i++) {
      double d[16];
      std::memcpy(d, &c->imu[i * rec], rec);
      const uint64_t ts = seconds_to_us(d[0]);
      if (ts == 0) { imu_skipped++; continue; }  // "invalid IMUFrame -> skipping" (main.cpp:149-152)
      uint8_t frame[128];
      std::memcpy(frame, &d[1], 120);            // rotationRate, acceleration, magneticField, attitude, gravity (5 x vec3d)
      std::memcpy(frame + 120, &ts, 8);          // sensorData.h:796-803
      s->imu.insert(s->imu.end(), frame, frame + 128);
      imu_kept++;
    }
    rc = sf_sens_save(s, out_sens);
  }
  sf_sens_close(s);
  if (rc != SF_OK) return rc;
  if (stats) {
    stats->frames = n;
    stats->depth_frames_in_capture = c->frame_off.size();
    stats->imu_frames = imu_kept;
    stats->imu_skipped = imu_skipped;
    stats->depth_stream_bytes = depth_bytes_in;
    stats->threads = (uint32_t)nthreads;
  }
  return SF_OK;
}

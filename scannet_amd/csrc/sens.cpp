// sens.cpp -- from-scratch reader / writer of the .sens v4 container (little-endian, packed).
//
// Layout restated from the reference codec, SensReader/c++/src/sensorData.h:
//   header  :1257-1273  u32 version(=4) . u64 strlen . name . 4 x mat4f (colour K, colour E, depth K, depth E)
//                       . i32 colourCompr . i32 depthCompr . u32 cw,ch,dw,dh . f32 depthShift
//   frames  :1275-1280, :743-754   u64 numFrames ; per frame mat4f camToWorld . u64 tsColor . u64 tsDepth
//                       . u64 colourBytes . u64 depthBytes . colour blob . depth blob
//   IMU     :1282-1289, :796-803   u64 numIMU ; per IMU frame 5 x vec3d + u64 = 128 bytes
// The Python reader's struct formats (SensReader/python/SensorData.py:14-20,54-74) say the same.
// Unlike the reference (whole file malloc'ed frame by frame, 7 istream::read calls per frame) the file is
// memory-mapped once and frames are views into the mapping.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <thread>
#include <limits>

#include "sens.h"

int jpeg_decode_rgb(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h);  // jpeg.cpp
int png_decode_rgb(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h);   // png.cpp

namespace {

struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if ((uint64_t)(end - p) < sizeof(T)) { ok = false; return v; }
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  const uint8_t* take(uint64_t n) {
    if ((uint64_t)(end - p) < n) { ok = false; return nullptr; }
    const uint8_t* q = p;
    p += n;
    return q;
  }
};

}  // namespace

SF_API int sf_sens_open(const char* path, sf_sens** out) {
  if (!path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) return sf::fail(SF_ERR_IO, "could not open file %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 4) { ::close(fd); return sf::fail(SF_ERR_FORMAT, "%s: too short to be a .sens file", path); }
  void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (map == MAP_FAILED) { ::close(fd); return sf::fail(SF_ERR_IO, "mmap of %s failed", path); }
  sf_sens* s = new sf_sens();
  s->map = map; s->map_bytes = (uint64_t)st.st_size; s->fd = fd;
  auto bail = [&](int code, const char* what) { sf_sens_close(s); return sf::fail(code, "%s: %s", path, what); };
  Cursor c{(const uint8_t*)map, (const uint8_t*)map + st.st_size};
  sf_sens_info& h = s->info;
  std::memset(&h, 0, sizeof(h));
  h.version = c.get<uint32_t>();
  if (!c.ok) return bail(SF_ERR_FORMAT, "truncated header");
  if (h.version != 4) {  // assertVersionNumber, sensorData.h:883-886
    const uint32_t v = h.version;
    sf_sens_close(s);
    return sf::fail(SF_ERR_FORMAT, "%s: invalid file version -- found %u but expected 4", path, v);
  }
  const uint64_t slen = c.get<uint64_t>();
  const uint8_t* name = c.take(slen);
  if (!c.ok) return bail(SF_ERR_FORMAT, "truncated header (sensor name)");
  std::memcpy(h.sensor_name, name, slen < sizeof(h.sensor_name) - 1 ? slen : sizeof(h.sensor_name) - 1);
  float* mats[4] = {h.color_intrinsic, h.color_extrinsic, h.depth_intrinsic, h.depth_extrinsic};
  for (float* m : mats) {
    const uint8_t* q = c.take(64);
    if (!c.ok) return bail(SF_ERR_FORMAT, "truncated header (calibration)");
    std::memcpy(m, q, 64);
  }
  h.color_compression = c.get<int32_t>();
  h.depth_compression = c.get<int32_t>();
  h.color_width = c.get<uint32_t>(); h.color_height = c.get<uint32_t>();
  h.depth_width = c.get<uint32_t>(); h.depth_height = c.get<uint32_t>();
  h.depth_shift = c.get<float>();
  h.num_frames = c.get<uint64_t>();
  if (!c.ok) return bail(SF_ERR_FORMAT, "truncated header");
  if (h.num_frames > (uint64_t)st.st_size / 96) return bail(SF_ERR_FORMAT, "frame count larger than the file allows");
  s->frames.resize(h.num_frames);
  for (uint64_t i = 0; i < h.num_frames; i++) {
    SensFrame& f = s->frames[i];
    const uint8_t* q = c.take(64);
    if (!c.ok) return bail(SF_ERR_FORMAT, "truncated frame record");
    std::memcpy(f.pose, q, 64);
    f.ts_color = c.get<uint64_t>(); f.ts_depth = c.get<uint64_t>();
    f.color_bytes = c.get<uint64_t>(); f.depth_bytes = c.get<uint64_t>();
    if (!c.ok) return bail(SF_ERR_FORMAT, "truncated frame record");
    f.color = c.take(f.color_bytes);
    f.depth = c.ok ? c.take(f.depth_bytes) : nullptr;
    if (!c.ok) return bail(SF_ERR_FORMAT, "truncated frame data");
  }
  // files written by LiveSensorDataWriter before close() may end right after the frames (sensorData.h:1146-1156)
  h.num_imu = 0;
  if (c.p < c.end) {
    h.num_imu = c.get<uint64_t>();
    if (!c.ok) return bail(SF_ERR_FORMAT, "truncated IMU count");
    if (h.num_imu > (uint64_t)(c.end - c.p) / 128) return bail(SF_ERR_FORMAT, "truncated IMU frames");
    const uint8_t* q = c.take(h.num_imu * 128);
    s->imu.assign(q, q + h.num_imu * 128);
  }
  *out = s;
  return SF_OK;
}

SF_API void sf_sens_close(sf_sens* s) {
  if (!s) return;
  if (s->map) munmap(s->map, (size_t)s->map_bytes);
  if (s->fd >= 0) ::close(s->fd);
  delete s;
}

SF_API int sf_sens_get_info(const sf_sens* s, sf_sens_info* out) {
  if (!s || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  *out = s->info;
  out->num_frames = s->frames.size();
  out->num_imu = s->imu.size() / 128;
  return SF_OK;
}

int sens_decode_depth(const sf_sens* s, uint64_t i, uint16_t* dst) {
  if (i >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)i, s->frames.size());
  const SensFrame& f = s->frames[i];
  const uint64_t want = (uint64_t)s->info.depth_width * s->info.depth_height * 2;
  switch (s->info.depth_compression) {
    case 0:  // TYPE_RAW_USHORT, sensorData.h:724-730
      if (f.depth_bytes != want) return sf::fail(SF_ERR_FORMAT, "raw depth frame %llu has %llu bytes, expected %llu", (unsigned long long)i, (unsigned long long)f.depth_bytes, (unsigned long long)want);
      std::memcpy(dst, f.depth, want);
      return SF_OK;
    case 1: {  // TYPE_ZLIB_USHORT, sensorData.h:703-709
      if (!f.depth || f.depth_bytes == 0) return sf::fail(SF_ERR_FORMAT, "frame %llu has no depth data", (unsigned long long)i);
      uint64_t got = 0;
      const int rc = sf_zlib_inflate(f.depth, f.depth_bytes, dst, want, &got);
      if (rc != SF_OK) return rc;
      if (got != want) return sf::fail(SF_ERR_FORMAT, "depth frame %llu inflates to %llu bytes, expected %llu", (unsigned long long)i, (unsigned long long)got, (unsigned long long)want);
      return SF_OK;
    }
    case 2: {  // TYPE_OCCI_USHORT, sensorData.h:711-722: uplinksimple::decode + shift2depth (built only under _USE_UPLINK_COMPRESSION there)
      if (!f.depth || f.depth_bytes == 0) return sf::fail(SF_ERR_FORMAT, "frame %llu has no depth data", (unsigned long long)i);
      const uint64_t npx = (uint64_t)s->info.depth_width * s->info.depth_height;
      const int rc = sf_occ_decode(f.depth, f.depth_bytes, npx, dst);
      if (rc != SF_OK) return rc;
      return sf_occ_shift2depth_buffer(dst, npx, 0);
    }
    default:
      return sf::fail(SF_ERR_FORMAT, "unknown depth compression type %d", s->info.depth_compression);
  }
}

SF_API int sf_sens_decode_depth(const sf_sens* s, uint64_t frame, uint16_t* dst) {
  if (!s || !dst) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  return sens_decode_depth(s, frame, dst);
}

SF_API int sf_sens_decode_color(const sf_sens* s, uint64_t frame, uint8_t* dst) {
  if (!s || !dst) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  const SensFrame& f = s->frames[frame];
  const uint64_t want = (uint64_t)s->info.color_width * s->info.color_height * 3;
  if (!f.color || f.color_bytes == 0) return sf::fail(SF_ERR_FORMAT, "frame %llu has no colour data", (unsigned long long)frame);  // sensorData.h:607,646
  switch (s->info.color_compression) {
    case 0:  // TYPE_RAW, sensorData.h:644-650
      if (f.color_bytes != want) return sf::fail(SF_ERR_FORMAT, "raw colour frame has %llu bytes, expected %llu", (unsigned long long)f.color_bytes, (unsigned long long)want);
      std::memcpy(dst, f.color, want);
      return SF_OK;
    case 2:  // TYPE_JPEG, sensorData.h:609-616
      return jpeg_decode_rgb(f.color, f.color_bytes, dst, s->info.color_width, s->info.color_height);
    case 1:  // TYPE_PNG: the same stb call as JPEG in the reference (sensorData.h:609-616)
      return png_decode_rgb(f.color, f.color_bytes, dst, s->info.color_width, s->info.color_height);
    default:
      return sf::fail(SF_ERR_FORMAT, "unknown colour compression type %d", s->info.color_compression);
  }
}

SF_API int sf_sens_pose(const sf_sens* s, uint64_t frame, float out16[16], int* valid) {
  if (!s || !out16) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  std::memcpy(out16, s->frames[frame].pose, 64);
  if (valid) *valid = out16[0] != -std::numeric_limits<float>::infinity();
  return SF_OK;
}

SF_API int sf_sens_frame_meta(const sf_sens* s, uint64_t frame, sf_sens_frame_meta_t* out) {
  if (!s || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  const SensFrame& f = s->frames[frame];
  out->timestamp_color = f.ts_color; out->timestamp_depth = f.ts_depth;
  out->color_bytes = f.color_bytes; out->depth_bytes = f.depth_bytes;
  return SF_OK;
}

// RGBDFrame::getColorCompressed / getDepthCompressed / get*SizeBytes (sensorData.h:418-429): the frame's blobs where they lie (in the mapped
// file or in the writer's storage); valid until the handle is closed or, for a file under construction, until the next frame is added
SF_API int sf_sens_frame_blobs(const sf_sens* s, uint64_t frame, const uint8_t** color, uint64_t* color_bytes, const uint8_t** depth, uint64_t* depth_bytes) {
  if (!s) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  const SensFrame& f = s->frames[frame];
  if (color) *color = f.color;
  if (color_bytes) *color_bytes = f.color_bytes;
  if (depth) *depth = f.depth;
  if (depth_bytes) *depth_bytes = f.depth_bytes;
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------ writer
SF_API int sf_sens_create(const sf_sens_info* header, sf_sens** out) {
  if (!header || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (header->depth_compression < 0 || header->depth_compression > 2)
    return sf::fail(SF_ERR_UNSUPPORTED, "writer supports depth compression 0 (raw), 1 (zlib) and 2 (Occipital shift code) only");
  sf_sens* s = new sf_sens();
  s->info = *header;
  s->info.version = 4;
  s->info.num_frames = 0;
  s->info.num_imu = 0;
  s->info.sensor_name[sizeof(s->info.sensor_name) - 1] = 0;
  *out = s;
  return SF_OK;
}

// n depth-only frames at once, compressed on `threads` threads (0 = the CPUs this process may use) and appended in order: what
// sf_sens_add_frame does n times, at the speed of the machine instead of one core (the reference's writer compresses frame by frame on the
// caller's thread, sensorData.h:1101-1109; a 5 578-frame scan of real-entropy depth is ~30 s of deflate).
SF_API int sf_sens_add_depth_frames(sf_sens* s, const uint16_t* depth, uint64_t frame_stride_bytes, uint64_t n, const float* poses,
                                    uint64_t timestamp0_us, uint64_t timestamp_step_us, int threads) {
  if (!s || (n && (!depth || !poses))) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (s->info.depth_compression != 0 && s->info.depth_compression != 1) return sf::fail(SF_ERR_UNSUPPORTED, "sf_sens_add_depth_frames: raw or zlib depth only");
  const uint64_t raw = (uint64_t)s->info.depth_width * s->info.depth_height * 2;
  if (frame_stride_bytes < raw) return sf::fail(SF_ERR_INVALID_ARG, "frame stride smaller than a frame");
  if (n == 0) return SF_OK;
  // all or nothing: a failure in a later chunk takes the earlier chunks' frames back out (their blob pointers were never fixed up)
  const size_t frames_before = s->frames.size();
  auto fix_up = [&]() {
    for (SensFrame& q : s->frames)
      if (!q.owned.empty()) { q.color = q.owned.data(); q.depth = q.owned.data() + q.color_bytes; }
  };
  auto undo = [&](int rc) { s->frames.resize(frames_before); fix_up(); return rc; };
  // no exception may cross the C ABI (std::thread's constructor and every allocation below can throw), and the frames are compressed in
  // CHUNKS: only a chunk's worth of deflate-bound-sized buffers is ever alive beside the frames already shrunk to their compressed size
  try {
    const int nt = (int)std::min<uint64_t>(n, (uint64_t)(threads > 0 ? threads : sf::usable_cpus()));
    const uint64_t chunk = std::max<uint64_t>(64, 8 * (uint64_t)nt);
    s->frames.reserve(s->frames.size() + (size_t)n);
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
      const uint64_t cn = std::min(chunk, n - c0);
      std::vector<SensFrame> made((size_t)cn);
      std::atomic<uint64_t> next{0};
      std::atomic<int> failed{0};
      auto work = [&]() {
        try {
          for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= cn || failed.load()) return;
            const uint64_t i = c0 + k;
            SensFrame& f = made[(size_t)k];
            std::memcpy(f.pose, poses + 16 * i, 64);
            f.ts_color = 0;
            f.ts_depth = timestamp0_us + i * timestamp_step_us;
            const uint16_t* src = (const uint16_t*)((const uint8_t*)depth + i * frame_stride_bytes);
            uint64_t dbytes = raw;
            if (s->info.depth_compression == 0) {
              f.owned.assign((const uint8_t*)src, (const uint8_t*)src + raw);
            } else {
              f.owned.resize(sf_zlib_deflate_bound(raw));
              if (sf_zlib_deflate(src, raw, f.owned.data(), f.owned.size(), &dbytes) != SF_OK) { failed.store(1); return; }
              f.owned.resize(dbytes);
              f.owned.shrink_to_fit();
            }
            f.color_bytes = 0;
            f.depth_bytes = dbytes;
          }
        } catch (...) {
          failed.store(2);   // bad_alloc on a worker: reported below, never thrown across the thread boundary
        }
      };
      std::vector<std::thread> pool;
      const int ct = (int)std::min<uint64_t>(cn, (uint64_t)nt);
      try {
        for (int t = 1; t < ct; t++) pool.emplace_back(work);
      } catch (...) {
        // could not start another thread: the ones that run (and this one) finish the chunk
      }
      work();
      for (auto& t : pool) t.join();
      if (failed.load() == 1) return undo(sf::fail(SF_ERR_FORMAT, "sf_sens_add_depth_frames: deflate failed"));
      if (failed.load()) return undo(sf::fail(SF_ERR_IO, "sf_sens_add_depth_frames: out of memory"));
      for (auto& f : made) s->frames.push_back(std::move(f));
    }
  } catch (const std::exception& e) {
    return undo(sf::fail(SF_ERR_IO, "sf_sens_add_depth_frames: %s", e.what()));
  }
  fix_up();
  return SF_OK;
}

// A frame whose blobs are ALREADY in the container's compression (what loadFromFile keeps per frame, sensorData.h:743-754): transcoding,
// merging files, or pairing depth streams some other writer compressed with colour pictures -- the bytes are stored as given.
SF_API int sf_sens_add_frame_blobs(sf_sens* s, const uint8_t* color, uint64_t color_bytes, const uint8_t* depth, uint64_t depth_bytes,
                                   const float pose[16], uint64_t ts_color, uint64_t ts_depth) {
  if (!s || !pose || (color_bytes && !color) || (depth_bytes && !depth)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint64_t raw = (uint64_t)s->info.depth_width * s->info.depth_height * 2;
  if (depth_bytes && s->info.depth_compression == 0 && depth_bytes != raw) return sf::fail(SF_ERR_INVALID_ARG, "raw depth frame must be depthWidth*depthHeight*2 bytes");
  if (color_bytes && s->info.color_compression == 0 && color_bytes != (uint64_t)s->info.color_width * s->info.color_height * 3)
    return sf::fail(SF_ERR_INVALID_ARG, "raw colour frame must be colorWidth*colorHeight*3 bytes");
  try {
    SensFrame f;
    std::memcpy(f.pose, pose, 64);
    f.ts_color = ts_color; f.ts_depth = ts_depth;
    f.owned.resize(color_bytes + depth_bytes);
    if (color_bytes) std::memcpy(f.owned.data(), color, color_bytes);
    if (depth_bytes) std::memcpy(f.owned.data() + color_bytes, depth, depth_bytes);
    f.color_bytes = color_bytes;
    f.depth_bytes = depth_bytes;
    s->frames.push_back(std::move(f));
  } catch (const std::exception& e) {
    return sf::fail(SF_ERR_IO, "sf_sens_add_frame_blobs: %s", e.what());
  }
  for (SensFrame& q : s->frames)
    if (!q.owned.empty()) { q.color = q.owned.data(); q.depth = q.owned.data() + q.color_bytes; }
  return SF_OK;
}

SF_API int sf_sens_add_frame(sf_sens* s, const uint8_t* color, uint64_t color_bytes, const uint16_t* depth, const float pose[16],
                             uint64_t ts_color, uint64_t ts_depth) {
  if (!s || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!color) color_bytes = 0;
  if (color_bytes && s->info.color_compression == 0 && color_bytes != (uint64_t)s->info.color_width * s->info.color_height * 3)
    return sf::fail(SF_ERR_INVALID_ARG, "raw colour frame must be colorWidth*colorHeight*3 bytes");
  SensFrame f;
  std::memcpy(f.pose, pose, 64);
  f.ts_color = ts_color; f.ts_depth = ts_depth;
  const uint64_t raw = (uint64_t)s->info.depth_width * s->info.depth_height * 2;
  uint64_t dbytes = 0;
  if (depth) {
    if (s->info.depth_compression == 0) {
      f.owned.resize(color_bytes + raw);
      std::memcpy(f.owned.data() + color_bytes, depth, raw);
      dbytes = raw;
    } else if (s->info.depth_compression == 2) {
      // TYPE_OCCI_USHORT as the reference writes it (sensorData.h:672-684): the values are coded AS GIVEN -- they are sensor shift
      // values, the reader maps them to millimetres through the shift table (sensorData.h:715-716)
      const uint64_t bound = sf_occ_encode_bound(raw / 2);
      f.owned.resize(color_bytes + bound);
      const int rc = sf_occ_encode(depth, raw / 2, f.owned.data() + color_bytes, bound, &dbytes);
      if (rc != SF_OK) return rc;
      f.owned.resize(color_bytes + dbytes);
    } else {
      const uint64_t bound = sf_zlib_deflate_bound(raw);
      f.owned.resize(color_bytes + bound);
      const int rc = sf_zlib_deflate(depth, raw, f.owned.data() + color_bytes, bound, &dbytes);
      if (rc != SF_OK) return rc;
      f.owned.resize(color_bytes + dbytes);
    }
  } else {
    f.owned.resize(color_bytes);
  }
  if (color_bytes) std::memcpy(f.owned.data(), color, color_bytes);
  f.color_bytes = color_bytes;
  f.depth_bytes = dbytes;
  s->frames.push_back(std::move(f));
  SensFrame& g = s->frames.back();
  g.color = g.owned.data();
  g.depth = g.owned.data() + color_bytes;
  // earlier frames' vectors may have moved with the push_back: re-point all writer-owned frames
  for (SensFrame& q : s->frames)
    if (!q.owned.empty()) { q.color = q.owned.data(); q.depth = q.owned.data() + q.color_bytes; }
  return SF_OK;
}

SF_API int sf_sens_set_pose(sf_sens* s, uint64_t frame, const float pose[16]) {
  if (!s || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  std::memcpy(s->frames[frame].pose, pose, 64);
  return SF_OK;
}

SF_API int sf_sens_add_imu(sf_sens* s, const void* frame128) {
  if (!s || !frame128) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint8_t* b = (const uint8_t*)frame128;
  s->imu.insert(s->imu.end(), b, b + 128);
  return SF_OK;
}

namespace {
// a frame's blobs rebuilt in writer-owned storage (colour then depth), whatever they were before (in the mapped file or owned)
int set_blobs(sf_sens* s, SensFrame& f, const uint8_t* color, uint64_t color_bytes, const uint8_t* depth, uint64_t depth_bytes) {
  try {
    std::vector<uint8_t> next(color_bytes + depth_bytes);
    if (color_bytes) std::memcpy(next.data(), color, color_bytes);       // the sources may lie in f.owned: copy before the swap
    if (depth_bytes) std::memcpy(next.data() + color_bytes, depth, depth_bytes);
    f.owned.swap(next);
  } catch (const std::exception& e) {
    return sf::fail(SF_ERR_IO, "out of memory: %s", e.what());
  }
  f.color_bytes = color_bytes;
  f.depth_bytes = depth_bytes;
  f.color = f.owned.data();
  f.depth = f.owned.data() + color_bytes;
  (void)s;
  return SF_OK;
}
}  // namespace

// SensorData::replaceDepth(frameIdx, depth) (sensorData.h:948-955 -> RGBDFrame::replaceDepth :499-502): the frame's depth compressed anew with the file's
// depth compression type; colour, pose and the COLOUR time stamp stay -- the depth time stamp goes to 0, as freeDepth() leaves it (:516-521).
// Works on an opened file too (the Calibrate stage's use, Calibrate/src/calibration.h:303): the frame then lives in writer-owned memory.
static int sens_replace_depth(sf_sens* s, uint64_t frame, const uint16_t* depth);
SF_API int sf_sens_replace_depth(sf_sens* s, uint64_t frame, const uint16_t* depth) {
  if (!s || !depth) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  try { return sens_replace_depth(s, frame, depth); }   // no exception crosses the C ABI
  catch (...) { return sf::fail(SF_ERR_IO, "out of memory replacing the depth of frame %llu", (unsigned long long)frame); }
}
static int sens_replace_depth(sf_sens* s, uint64_t frame, const uint16_t* depth) {
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  const uint64_t raw = (uint64_t)s->info.depth_width * s->info.depth_height * 2;
  std::vector<uint8_t> blob;
  uint64_t n = raw;
  if (s->info.depth_compression == 0) blob.assign((const uint8_t*)depth, (const uint8_t*)depth + raw);
  else if (s->info.depth_compression == 2) {
    blob.resize(sf_occ_encode_bound(raw / 2));
    const int rc = sf_occ_encode(depth, raw / 2, blob.data(), blob.size(), &n);
    if (rc != SF_OK) return rc;
  } else {
    blob.resize(sf_zlib_deflate_bound(raw));
    const int rc = sf_zlib_deflate(depth, raw, blob.data(), blob.size(), &n);
    if (rc != SF_OK) return rc;
  }
  SensFrame& f = s->frames[frame];
  const int rc = set_blobs(s, f, f.color, f.color_bytes, blob.data(), n);
  if (rc == SF_OK) f.ts_depth = 0;
  return rc;
}

// SensorData::replaceColor(frameIdx, color) (:957-964 -> RGBDFrame::replaceColor :505-508): `color` as sf_sens_add_frame takes it -- W*H*3 RGB bytes for a
// TYPE_RAW file, an encoded JPEG / PNG blob otherwise (the reference encodes the pixels itself, with its Windows-only encoder, :576-593); the colour time
// stamp goes to 0 (freeColor, :510-515).
SF_API int sf_sens_replace_color(sf_sens* s, uint64_t frame, const uint8_t* color, uint64_t color_bytes) {
  if (!s || (!color && color_bytes)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  if (color_bytes && s->info.color_compression == 0 && color_bytes != (uint64_t)s->info.color_width * s->info.color_height * 3)
    return sf::fail(SF_ERR_INVALID_ARG, "raw colour frame must be colorWidth*colorHeight*3 bytes");
  SensFrame& f = s->frames[frame];
  const int rc = set_blobs(s, f, color, color_bytes, f.depth, f.depth_bytes);
  if (rc == SF_OK) f.ts_color = 0;
  return rc;
}

// SensorData::append(second) (:1605-1624): the frames of `other` (blobs, poses, time stamps) behind this file's; its IMU frames are not taken, as in the
// reference.  The reference means to refuse incompatible files but joins its six tests with && (it throws only when ALL of them differ): here any
// difference in frame sizes or compression types is refused.
SF_API int sf_sens_append(sf_sens* s, const sf_sens* other) {
  if (!s || !other) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const sf_sens_info &a = s->info, &b = other->info;
  if (a.color_width != b.color_width || a.color_height != b.color_height || a.depth_width != b.depth_width || a.depth_height != b.depth_height ||
      a.color_compression != b.color_compression || a.depth_compression != b.depth_compression)
    return sf::fail(SF_ERR_INVALID_ARG, "sensor data incompatible");
  const size_t n = other->frames.size();   // `other` may be `s` itself
  try {
    s->frames.reserve(s->frames.size() + n);
    for (size_t i = 0; i < n; i++) {
      const SensFrame& o = other->frames[i];
      SensFrame f;
      std::memcpy(f.pose, o.pose, 64);
      f.ts_color = o.ts_color; f.ts_depth = o.ts_depth;
      f.owned.resize(o.color_bytes + o.depth_bytes);
      if (o.color_bytes) std::memcpy(f.owned.data(), o.color, o.color_bytes);
      if (o.depth_bytes) std::memcpy(f.owned.data() + o.color_bytes, o.depth, o.depth_bytes);
      f.color_bytes = o.color_bytes; f.depth_bytes = o.depth_bytes;
      s->frames.push_back(std::move(f));
    }
  } catch (const std::exception& e) {
    return sf::fail(SF_ERR_IO, "sf_sens_append: %s", e.what());
  }
  for (SensFrame& q : s->frames)
    if (!q.owned.empty()) { q.color = q.owned.data(); q.depth = q.owned.data() + q.color_bytes; }
  return SF_OK;
}

// SensorData::computeDepthImage(frameIdx) (:968-982, behind _HAS_MLIB): the frame in metres -- (float)depth / depthShift, the "no measurement" 0 as 0.0f
// (DepthImage32's invalid value there, :971).  The fusion path does this conversion in its pre-pass on the GPU; this is the reference's host convenience.
SF_API int sf_sens_depth_image(const sf_sens* s, uint64_t frame, float* dst) {
  if (!s || !dst) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const size_t n = (size_t)s->info.depth_width * s->info.depth_height;
  std::vector<uint16_t> d(n);
  const int rc = sf_sens_decode_depth(s, frame, d.data());
  if (rc != SF_OK) return rc;
  const float shift = s->info.depth_shift;
  for (size_t i = 0; i < n; i++) dst[i] = d[i] == 0 ? 0.0f : (float)d[i] / shift;
  return SF_OK;
}

// SensorData::applyTransform(t) (:1047-1054, behind _HAS_MLIB): every tracked frame's camera-to-world becomes t * m (row-major 4x4 product, float sums in
// index order; mLib itself is not in the reference tree, so the rounding of its product is not pinned); frames whose pose is the all -inf "tracking lost"
// mark (m(0,0) == -inf) stay as they are.  What the pipeline's alignment step does to a scan's trajectory (Alignment/src/alignment.h, out of scope here).
SF_API int sf_sens_apply_transform(sf_sens* s, const float t[16]) {
  if (!s || !t) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  for (SensFrame& f : s->frames) {
    if (f.pose[0] == -std::numeric_limits<float>::infinity()) continue;
    float r[16];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        float acc = t[i * 4 + 0] * f.pose[0 * 4 + j];
        for (int k = 1; k < 4; k++) acc = acc + t[i * 4 + k] * f.pose[k * 4 + j];
        r[i * 4 + j] = acc;
      }
    std::memcpy(f.pose, r, 64);
  }
  return SF_OK;
}

// SensorData::operator== (:1626-1650): version, sensor name, both calibrations, compression types, sizes, depth shift, every frame (blob sizes, time
// stamps, the pose compared as floats -- an all -inf pose equals itself, a NaN nothing --, blob bytes: RGBDFrame::operator== :756-771) and every IMU
// frame (doubles compared as doubles, :813-821).
SF_API int sf_sens_equal(const sf_sens* x, const sf_sens* y, int* equal) {
  if (!x || !y || !equal) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  *equal = 0;
  const sf_sens_info &a = x->info, &b = y->info;
  if (a.version != b.version || std::strncmp(a.sensor_name, b.sensor_name, sizeof a.sensor_name) != 0) return SF_OK;
  for (int i = 0; i < 16; i++)
    if (a.color_intrinsic[i] != b.color_intrinsic[i] || a.color_extrinsic[i] != b.color_extrinsic[i] || a.depth_intrinsic[i] != b.depth_intrinsic[i] ||
        a.depth_extrinsic[i] != b.depth_extrinsic[i])
      return SF_OK;
  if (a.color_compression != b.color_compression || a.depth_compression != b.depth_compression || a.color_width != b.color_width ||
      a.color_height != b.color_height || a.depth_width != b.depth_width || a.depth_height != b.depth_height || a.depth_shift != b.depth_shift)
    return SF_OK;
  if (x->frames.size() != y->frames.size() || x->imu.size() != y->imu.size()) return SF_OK;
  for (size_t i = 0; i < x->frames.size(); i++) {
    const SensFrame &f = x->frames[i], &g = y->frames[i];
    if (f.color_bytes != g.color_bytes || f.depth_bytes != g.depth_bytes || f.ts_color != g.ts_color || f.ts_depth != g.ts_depth) return SF_OK;
    for (int k = 0; k < 16; k++)
      if (f.pose[k] != g.pose[k]) return SF_OK;
    if ((f.color_bytes && std::memcmp(f.color, g.color, f.color_bytes) != 0) || (f.depth_bytes && std::memcmp(f.depth, g.depth, f.depth_bytes) != 0)) return SF_OK;
  }
  for (size_t i = 0; i < x->imu.size() / 128; i++) {
    double p[15], q[15];
    std::memcpy(p, &x->imu[i * 128], 120);
    std::memcpy(q, &y->imu[i * 128], 120);
    for (int k = 0; k < 15; k++)
      if (p[k] != q[k]) return SF_OK;
    if (std::memcmp(&x->imu[i * 128 + 120], &y->imu[i * 128 + 120], 8) != 0) return SF_OK;
  }
  *equal = 1;
  return SF_OK;
}

// m_IMUFrames[index] (sensorData.h:1691): the 128 bytes as stored
SF_API int sf_sens_imu(const sf_sens* s, uint64_t index, void* frame128) {
  if (!s || !frame128) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (index >= s->imu.size() / 128) return sf::fail(SF_ERR_BOUNDS, "IMU frame %llu out of bounds (%zu frames)", (unsigned long long)index, s->imu.size() / 128);
  std::memcpy(frame128, &s->imu[index * 128], 128);
  return SF_OK;
}

// SensorData::findClosestIMUFrame(frameIdx, basedOnRGB) (sensorData.h:1000-1044): the IMU frame nearest in time to the frame's colour (or depth) time
// stamp -- the first / last one outside the recorded span, an exact hit as it is, else the nearer of the two neighbours with the LATER one on a tie
// (`<`, :1035).  The reference's bisection reads one element past the end when the key equals the last time stamp (:1033-1038 with end == size);
// here that case returns the last frame, which is what it finds when the read happens to succeed.
SF_API int sf_sens_find_closest_imu(const sf_sens* s, uint64_t frame, int based_on_rgb, void* frame128, uint64_t* index) {
  if (!s) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (frame >= s->frames.size()) return sf::fail(SF_ERR_BOUNDS, "frame %llu out of bounds (%zu frames)", (unsigned long long)frame, s->frames.size());
  const size_t n = s->imu.size() / 128;
  if (n == 0) return sf::fail(SF_ERR_INVALID_ARG, "no imu data available");
  auto stamp = [&](size_t i) { uint64_t t; std::memcpy(&t, &s->imu[i * 128 + 120], 8); return t; };
  const uint64_t key = based_on_rgb ? s->frames[frame].ts_color : s->frames[frame].ts_depth;
  size_t found;
  if (key < stamp(0)) found = 0;
  else if (key > stamp(n - 1)) found = n - 1;
  else {
    size_t begin = 0, end = n;
    bool exact = false;
    while (begin + 1 < end) {
      const size_t middle = begin + (end - begin) / 2;
      if (stamp(middle) == key) { begin = middle; exact = true; break; }
      if (stamp(middle) > key) end = middle;
      else begin = middle;
    }
    if (exact || end == n) found = begin;
    else found = key - stamp(begin) < stamp(end) - key ? begin : end;
  }
  if (frame128) std::memcpy(frame128, &s->imu[found * 128], 128);
  if (index) *index = found;
  return SF_OK;
}

SF_API int sf_sens_save(const sf_sens* s, const char* path) {
  if (!s || !path) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  FILE* fp = std::fopen(path, "wb");
  if (!fp) return sf::fail(SF_ERR_IO, "unable to open file for writing: %s", path);
  bool ok = true;
  auto put = [&](const void* p, size_t n) { if (n && std::fwrite(p, 1, n, fp) != n) ok = false; };
  const sf_sens_info& h = s->info;
  const uint32_t version = 4;
  const uint64_t slen = std::strlen(h.sensor_name);
  put(&version, 4); put(&slen, 8); put(h.sensor_name, slen);
  put(h.color_intrinsic, 64); put(h.color_extrinsic, 64); put(h.depth_intrinsic, 64); put(h.depth_extrinsic, 64);
  put(&h.color_compression, 4); put(&h.depth_compression, 4);
  put(&h.color_width, 4); put(&h.color_height, 4); put(&h.depth_width, 4); put(&h.depth_height, 4);
  put(&h.depth_shift, 4);
  const uint64_t nf = s->frames.size();
  put(&nf, 8);
  for (const SensFrame& f : s->frames) {
    put(f.pose, 64); put(&f.ts_color, 8); put(&f.ts_depth, 8); put(&f.color_bytes, 8); put(&f.depth_bytes, 8);
    put(f.color, f.color_bytes); put(f.depth, f.depth_bytes);
  }
  const uint64_t ni = s->imu.size() / 128;
  put(&ni, 8); put(s->imu.data(), s->imu.size());
  if (std::fclose(fp) != 0) ok = false;
  if (!ok) return sf::fail(SF_ERR_IO, "write to %s failed", path);
  return SF_OK;
}

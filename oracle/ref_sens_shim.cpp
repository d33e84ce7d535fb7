// oracle/ref_sens_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper around the REFERENCE's own header-only .sens codec,
// /root/reference/SensReader/c++/src/sensorData.h (class ml::SensorData).  The header is
// compiled from where it lies under /root/reference (see oracle/Makefile, -I path); no
// reference source is copied into this repository.  The resulting oracle/_ref/libref_sens.so
// is the black-box oracle for: header/frame layout, inflate output, pose bytes and the
// reference writer (stb zlib deflate).  It is also the "reference" CPU decode baseline.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "sensorData.h"

extern "C" {

struct ref_sens_info {
  uint32_t version;
  uint32_t color_width, color_height, depth_width, depth_height;
  int32_t color_compression, depth_compression;
  float depth_shift;
  uint64_t num_frames, num_imu;
  float color_intrinsic[16], color_extrinsic[16], depth_intrinsic[16], depth_extrinsic[16];
  char sensor_name[256];
};

void* ref_sens_open(const char* path) {
  try { return new ml::SensorData(std::string(path)); } catch (...) { return nullptr; }
}
void ref_sens_close(void* h) { delete (ml::SensorData*)h; }

void ref_sens_get_info(void* h, ref_sens_info* o) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  std::memset(o, 0, sizeof(*o));
  o->version = sd.m_versionNumber;
  o->color_width = sd.m_colorWidth; o->color_height = sd.m_colorHeight;
  o->depth_width = sd.m_depthWidth; o->depth_height = sd.m_depthHeight;
  o->color_compression = (int32_t)sd.m_colorCompressionType;
  o->depth_compression = (int32_t)sd.m_depthCompressionType;
  o->depth_shift = sd.m_depthShift;
  o->num_frames = sd.m_frames.size(); o->num_imu = sd.m_IMUFrames.size();
  std::memcpy(o->color_intrinsic, sd.m_calibrationColor.m_intrinsic.matrix, 64);
  std::memcpy(o->color_extrinsic, sd.m_calibrationColor.m_extrinsic.matrix, 64);
  std::memcpy(o->depth_intrinsic, sd.m_calibrationDepth.m_intrinsic.matrix, 64);
  std::memcpy(o->depth_extrinsic, sd.m_calibrationDepth.m_extrinsic.matrix, 64);
  std::strncpy(o->sensor_name, sd.m_sensorName.c_str(), sizeof(o->sensor_name) - 1);
}

// decompressDepthAlloc (sensorData.h:943-946): returns 0 on success
int ref_sens_decode_depth(void* h, uint64_t frame, uint16_t* dst) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  if (frame >= sd.m_frames.size()) return -1;
  try {
    unsigned short* d = sd.decompressDepthAlloc((size_t)frame);
    if (!d) return -2;
    std::memcpy(dst, d, (size_t)sd.m_depthWidth * sd.m_depthHeight * 2);
    std::free(d);
    return 0;
  } catch (...) { return -3; }
}

// decompressColorAlloc (sensorData.h:933-936)
int ref_sens_decode_color(void* h, uint64_t frame, uint8_t* dst) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  if (frame >= sd.m_frames.size()) return -1;
  try {
    ml::vec3uc* c = sd.decompressColorAlloc((size_t)frame);
    if (!c) return -2;
    std::memcpy(dst, c, (size_t)sd.m_colorWidth * sd.m_colorHeight * 3);
    std::free(c);
    return 0;
  } catch (...) { return -3; }
}

int ref_sens_pose(void* h, uint64_t frame, float* out16) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  if (frame >= sd.m_frames.size()) return -1;
  std::memcpy(out16, sd.m_frames[frame].getCameraToWorld().matrix, 64);
  return 0;
}
int ref_sens_frame_meta(void* h, uint64_t frame, uint64_t* ts_color, uint64_t* ts_depth, uint64_t* color_bytes, uint64_t* depth_bytes) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  if (frame >= sd.m_frames.size()) return -1;
  *ts_color = sd.m_frames[frame].getTimeStampColor(); *ts_depth = sd.m_frames[frame].getTimeStampDepth();
  *color_bytes = sd.m_frames[frame].getColorSizeBytes(); *depth_bytes = sd.m_frames[frame].getDepthSizeBytes();
  return 0;
}

// Writer: initDefault + addFrame + saveToFile (sensorData.h:891-921, 1101-1109).  Colour is
// TYPE_RAW (JPEG encode needs uplinksimple, Windows only: sensorData.h:576-593); rgb may be NULL.
void* ref_sens_create(uint32_t cw, uint32_t ch, uint32_t dw, uint32_t dh, const float* color_intr, const float* depth_intr,
                      float depth_shift, const char* name) {
  ml::SensorData* sd = new ml::SensorData();
  ml::SensorData::CalibrationData cc, cd;
  std::memcpy(cc.m_intrinsic.matrix, color_intr, 64);
  std::memcpy(cd.m_intrinsic.matrix, depth_intr, 64);
  sd->initDefault(cw, ch, dw, dh, cc, cd, ml::SensorData::TYPE_RAW, ml::SensorData::TYPE_ZLIB_USHORT, depth_shift, std::string(name));
  return sd;
}
int ref_sens_add_frame(void* h, const uint8_t* rgb, const uint16_t* depth, const float* pose16, uint64_t ts_color, uint64_t ts_depth) {
  ml::SensorData& sd = *(ml::SensorData*)h;
  ml::mat4f m; std::memcpy(m.matrix, pose16, 64);
  try { sd.addFrame((const ml::vec3uc*)rgb, depth, m, ts_color, ts_depth); return 0; } catch (...) { return -1; }
}
// n frames at once: every frame is built by the reference's own SensorData::createFrame (sensorData.h:915-917 -> RGBDFrame(...) ->
// compressDepth :648-670 -> stb::stbi_zlib_compress, quality 8) -- a const method that touches nothing shared -- on `threads` threads, then
// moved into m_frames in order: the file addFrame x n would write (:919-922), at the speed of the machine (31 ms of stb deflate per 640x480 frame).
// rgb may be NULL (no colour), else n frames of colorWidth x colorHeight x 3 bytes (TYPE_RAW).
int ref_sens_add_frames_mt(void* h, const uint8_t* rgb, const uint16_t* depth, uint64_t n, const float* poses16, uint64_t ts0, uint64_t ts_step, int threads) {
  ml::SensorData& sd = *(ml::SensorData*)h;
  const size_t dpx = (size_t)sd.m_depthWidth * sd.m_depthHeight, cpx = (size_t)sd.m_colorWidth * sd.m_colorHeight;
  std::vector<ml::SensorData::RGBDFrame*> made((size_t)n, (ml::SensorData::RGBDFrame*)NULL);   // the move CONSTRUCTOR is public (:404), the assignments are not (:524,546)
  std::atomic<uint64_t> next(0);
  std::atomic<int> failed(0);
  auto work = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= n || failed.load()) return;
      ml::mat4f m; std::memcpy(m.matrix, poses16 + 16 * i, 64);
      try {
        made[(size_t)i] = new ml::SensorData::RGBDFrame(sd.createFrame(rgb ? (const ml::vec3uc*)(rgb + 3 * cpx * i) : (const ml::vec3uc*)NULL, depth + dpx * i, m, ts0 + i * ts_step, ts0 + i * ts_step));
      } catch (...) { failed.store(1); }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads && (uint64_t)t < n; t++) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (!failed.load())
    for (auto* f : made) sd.m_frames.push_back(std::move(*f));
  for (auto* f : made) delete f;
  return failed.load() ? -1 : 0;
}

// RGBDFrame::getDepthCompressed / getColorCompressed (sensorData.h:418-429): the blobs as the reference holds them
const uint8_t* ref_sens_depth_blob(void* h, uint64_t frame, uint64_t* bytes) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  if (frame >= sd.m_frames.size()) return NULL;
  *bytes = sd.m_frames[frame].getDepthSizeBytes();
  return sd.m_frames[frame].getDepthCompressed();
}

// IMU frames: addIMUFrame (sensorData.h:923-926) and findClosestIMUFrame(frameIdx, basedOnRGB) (:1042-1044) -> the index of the frame it returns
// (-1: it threw).  The caller keeps keys away from the last time stamp: there the reference's bisection reads m_IMUFrames[size] (:1033-1038).
int ref_sens_add_imu(void* h, const void* frame128) {
  ml::SensorData& sd = *(ml::SensorData*)h;
  ml::SensorData::IMUFrame f;
  const uint8_t* b = (const uint8_t*)frame128;
  std::memcpy(&f.rotationRate, b, 24); std::memcpy(&f.acceleration, b + 24, 24); std::memcpy(&f.magneticField, b + 48, 24);
  std::memcpy(&f.attitude, b + 72, 24); std::memcpy(&f.gravity, b + 96, 24); std::memcpy(&f.timeStamp, b + 120, 8);
  sd.addIMUFrame(f);
  return 0;
}
int64_t ref_sens_find_closest_imu(void* h, uint64_t frame, int based_on_rgb) {
  const ml::SensorData& sd = *(ml::SensorData*)h;
  try {
    const ml::SensorData::IMUFrame& f = sd.findClosestIMUFrame((size_t)frame, based_on_rgb != 0);
    return (int64_t)(&f - &sd.m_IMUFrames[0]);
  } catch (...) { return -1; }
}

// SensorData::append (:1605-1624), operator== (:1626-1650), replaceDepth (:948-955) on the reference's own objects
int ref_sens_append(void* h, void* second) {
  try { ((ml::SensorData*)h)->append(*(const ml::SensorData*)second); return 0; } catch (...) { return -1; }
}
int ref_sens_equal(void* a, void* b) { return *(const ml::SensorData*)a == *(const ml::SensorData*)b ? 1 : 0; }
int ref_sens_replace_depth(void* h, uint64_t frame, const uint16_t* depth) {
  try { ((ml::SensorData*)h)->replaceDepth((size_t)frame, depth); return 0; } catch (...) { return -1; }
}

int ref_sens_save(void* h, const char* path) {
  try { ((ml::SensorData*)h)->saveToFile(std::string(path)); return 0; } catch (...) { return -1; }
}

}  // extern "C"

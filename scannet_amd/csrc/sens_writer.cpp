// sens_writer.cpp -- frames streamed into a .sens as they arrive: SensorData::LiveSensorDataWriter (SensReader/c++/src/sensorData.h:1112-1246, _HAS_MLIB
// builds only there).  The in-memory writer (sf_sens_create / add_frame / save) holds a whole scan's blobs -- 2 GB for 5 578 frames -- until the end; a
// capture or a conversion that produces frames one by one writes them through this one with `cache` frames of memory.  Host code, no GPU.
//
//   open   header written at once, the frame count as 0 (patched at close, :1196-1203); a path that exists is overwritten or -- overwrite = 0 -- left
//          alone and the name's numeric suffix counted up until it is free ("scan.sens" -> "scan1.sens" -> "scan2.sens", :1118-1131);
//   add    the caller's buffers are COPIED into a bounded queue (the reference takes ownership of malloc'ed frames and frees them, :1159-1175; across a C
//          ABI the copy is the clean form) -- the call blocks while `cache` frames wait (:1166-1170); one background thread compresses (depth: the
//          header's type; colour: as given, see sf_sens_add_frame) and writes, in order (:1206-1225);
//   close  drains the queue, writes "0 IMU frames" and the frame count (:1146-1157: "does not work with IMU frames").
// The file is byte for byte what sf_sens_create + sf_sens_add_frame x n + sf_sens_save write (tests/test_sens.py), and the reference's reader reads it.
#include <sys/stat.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sens.h"

struct sf_sens_writer {
  struct Item {
    std::vector<uint8_t> color;
    std::vector<uint16_t> depth;
    bool blobs = false;               // depth holds an already compressed blob (bytes in depth_blob)
    std::vector<uint8_t> depth_blob;
    float pose[16];
    uint64_t ts_color = 0, ts_depth = 0;
  };
  FILE* fp = nullptr;
  std::string path;
  sf_sens* scratch = nullptr;   // a one-frame SensorData: the compressors of sf_sens_add_frame
  long count_pos = 0;
  uint64_t written = 0;
  size_t cache = 500;
  std::deque<Item> queue;
  std::mutex mu;
  std::condition_variable cv_room, cv_work;
  bool closing = false;
  int error = SF_OK;
  std::string error_text;
  std::thread bg;
};

namespace {

bool file_exists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}
// "dir/scan12.sens" -> "dir/scan13.sens"; no numeric suffix counts as 0 (:1120-1130)
std::string next_name(const std::string& p) {
  const size_t slash = p.find_last_of('/');
  const std::string dir = slash == std::string::npos ? "" : p.substr(0, slash + 1), file = slash == std::string::npos ? p : p.substr(slash + 1);
  const size_t dot = file.find_last_of('.');
  std::string stem = dot == std::string::npos ? file : file.substr(0, dot);
  const std::string ext = dot == std::string::npos ? "" : file.substr(dot);
  size_t digits = stem.size();
  while (digits > 0 && stem[digits - 1] >= '0' && stem[digits - 1] <= '9') digits--;
  const unsigned long num = digits < stem.size() ? std::strtoul(stem.c_str() + digits, nullptr, 10) : 0;
  return dir + stem.substr(0, digits) + std::to_string(num + 1) + ext;
}

void run(sf_sens_writer* w) {
  for (;;) {
    sf_sens_writer::Item it;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv_work.wait(lk, [&] { return w->closing || !w->queue.empty(); });
      if (w->queue.empty()) return;
      it = std::move(w->queue.front());
      w->queue.pop_front();
    }
    w->cv_room.notify_one();
    if (w->error != SF_OK) continue;   // drain without writing once something failed
    int rc;
    if (it.blobs)
      rc = sf_sens_add_frame_blobs(w->scratch, it.color.empty() ? nullptr : it.color.data(), it.color.size(), it.depth_blob.empty() ? nullptr : it.depth_blob.data(),
                                   it.depth_blob.size(), it.pose, it.ts_color, it.ts_depth);
    else
      rc = sf_sens_add_frame(w->scratch, it.color.empty() ? nullptr : it.color.data(), it.color.size(), it.depth.empty() ? nullptr : it.depth.data(), it.pose, it.ts_color,
                             it.ts_depth);
    bool ok = rc == SF_OK;
    std::string text = ok ? "" : sf_last_error();
    if (ok) {
      const SensFrame& f = w->scratch->frames.back();
      auto put = [&](const void* p, size_t n) { if (n && std::fwrite(p, 1, n, w->fp) != n) ok = false; };
      put(f.pose, 64); put(&f.ts_color, 8); put(&f.ts_depth, 8); put(&f.color_bytes, 8); put(&f.depth_bytes, 8);
      put(f.color, f.color_bytes); put(f.depth, f.depth_bytes);
      if (!ok) { rc = SF_ERR_IO; text = "write to " + w->path + " failed"; }
    }
    w->scratch->frames.clear();
    std::lock_guard<std::mutex> lk(w->mu);
    if (ok) w->written++;
    else if (w->error == SF_OK) { w->error = rc; w->error_text = text; }
  }
}

int enqueue(sf_sens_writer* w, sf_sens_writer::Item&& it) {
  std::unique_lock<std::mutex> lk(w->mu);
  if (w->closing) return sf::fail(SF_ERR_INVALID_ARG, "the writer is closed");
  if (w->error != SF_OK) return sf::fail(w->error, "%s", w->error_text.c_str());   // an earlier frame failed: say so now, not only at close
  w->cv_room.wait(lk, [&] { return w->queue.size() < w->cache; });
  w->queue.push_back(std::move(it));
  lk.unlock();
  w->cv_work.notify_one();
  return SF_OK;
}

}  // namespace

SF_API int sf_sens_writer_open(const sf_sens_info* header, const char* path, int overwrite, uint32_t cache_frames, sf_sens_writer** out) {
  if (!header || !path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_sens* scratch = nullptr;
  const int rc = sf_sens_create(header, &scratch);   // checks the compression types
  if (rc != SF_OK) return rc;
  std::string actual = path;
  if (!overwrite)
    while (file_exists(actual)) actual = next_name(actual);
  FILE* fp = std::fopen(actual.c_str(), "wb");
  if (!fp) { sf_sens_close(scratch); return sf::fail(SF_ERR_IO, "Unable to open file for writing: %s", path); }
  sf_sens_writer* w = new sf_sens_writer();
  w->fp = fp; w->path = actual; w->scratch = scratch; w->cache = cache_frames ? cache_frames : 500;
  const sf_sens_info& h = scratch->info;
  bool ok = true;
  auto put = [&](const void* p, size_t n) { if (n && std::fwrite(p, 1, n, fp) != n) ok = false; };
  const uint32_t version = 4;
  const uint64_t slen = std::strlen(h.sensor_name), zero = 0;
  put(&version, 4); put(&slen, 8); put(h.sensor_name, slen);
  put(h.color_intrinsic, 64); put(h.color_extrinsic, 64); put(h.depth_intrinsic, 64); put(h.depth_extrinsic, 64);
  put(&h.color_compression, 4); put(&h.depth_compression, 4);
  put(&h.color_width, 4); put(&h.color_height, 4); put(&h.depth_width, 4); put(&h.depth_height, 4);
  put(&h.depth_shift, 4);
  w->count_pos = std::ftell(fp);
  put(&zero, 8);
  if (!ok) { std::fclose(fp); sf_sens_close(scratch); delete w; return sf::fail(SF_ERR_IO, "write to %s failed", actual.c_str()); }
  w->bg = std::thread(run, w);
  *out = w;
  return SF_OK;
}

SF_API const char* sf_sens_writer_path(const sf_sens_writer* w) { return w ? w->path.c_str() : ""; }

// color / depth as sf_sens_add_frame takes them (raw RGB or an encoded blob, W*H u16; either may be NULL)
SF_API int sf_sens_writer_add_frame(sf_sens_writer* w, const uint8_t* color, uint64_t color_bytes, const uint16_t* depth, const float pose[16], uint64_t ts_color,
                                    uint64_t ts_depth) {
  if (!w || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const sf_sens_info& h = w->scratch->info;
  if (!color) color_bytes = 0;
  if (color_bytes && h.color_compression == 0 && color_bytes != (uint64_t)h.color_width * h.color_height * 3)
    return sf::fail(SF_ERR_INVALID_ARG, "raw colour frame must be colorWidth*colorHeight*3 bytes");
  sf_sens_writer::Item it;
  try {
    if (color_bytes) it.color.assign(color, color + color_bytes);
    if (depth) it.depth.assign(depth, depth + (size_t)h.depth_width * h.depth_height);
  } catch (const std::exception& e) {
    return sf::fail(SF_ERR_IO, "out of memory: %s", e.what());
  }
  std::memcpy(it.pose, pose, 64);
  it.ts_color = ts_color; it.ts_depth = ts_depth;
  return enqueue(w, std::move(it));
}

// both blobs already compressed (sf_sens_add_frame_blobs): transcoding, merging files
SF_API int sf_sens_writer_add_frame_blobs(sf_sens_writer* w, const uint8_t* color, uint64_t color_bytes, const uint8_t* depth, uint64_t depth_bytes, const float pose[16],
                                          uint64_t ts_color, uint64_t ts_depth) {
  if (!w || !pose || (color_bytes && !color) || (depth_bytes && !depth)) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_sens_writer::Item it;
  it.blobs = true;
  try {
    if (color_bytes) it.color.assign(color, color + color_bytes);
    if (depth_bytes) it.depth_blob.assign(depth, depth + depth_bytes);
  } catch (const std::exception& e) {
    return sf::fail(SF_ERR_IO, "out of memory: %s", e.what());
  }
  std::memcpy(it.pose, pose, 64);
  it.ts_color = ts_color; it.ts_depth = ts_depth;
  return enqueue(w, std::move(it));
}

// drains, patches the frame count, closes and frees the handle whatever happened; the first error of any frame is the result
SF_API int sf_sens_writer_close(sf_sens_writer* w, uint64_t* frames_written) {
  if (!w) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->closing = true;
  }
  w->cv_work.notify_all();
  if (w->bg.joinable()) w->bg.join();
  bool ok = true;
  const uint64_t zero = 0;
  if (std::fwrite(&zero, 1, 8, w->fp) != 8) ok = false;                      // number of IMU frames
  if (std::fseek(w->fp, w->count_pos, SEEK_SET) != 0 || std::fwrite(&w->written, 1, 8, w->fp) != 8) ok = false;
  if (std::fclose(w->fp) != 0) ok = false;
  if (frames_written) *frames_written = w->written;
  int rc = w->error;
  const std::string text = w->error_text, path = w->path;
  sf_sens_close(w->scratch);
  delete w;
  if (rc != SF_OK) return sf::fail(rc, "%s", text.c_str());
  if (!ok) return sf::fail(SF_ERR_IO, "write to %s failed", path.c_str());
  return SF_OK;
}

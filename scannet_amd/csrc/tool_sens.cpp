// tool_sens.cpp -- drop-in for the reference's .sens exporter, SensReader/c++ `sens <sensFile> <outputDir>` (src/main.cpp:35-106, README.txt:3-9):
// every frame spat out as an image pair and a pose, the header as text --
//     <outputDir>/_info.txt                     sensorData.h:1383-1407  (names " = " values; the four 4x4 matrices row-major, 16 numbers and a trailing blank)
//     <outputDir>/frame-%06d.color.jpg | .png   :1431-1450  the stored JPEG / PNG blob as it is; a TYPE_RAW frame as a PNG made here (the reference needs
//                                                           its Windows-only encoder for that one, :576-593: off Windows it throws after _info.txt)
//     <outputDir>/frame-%06d.depth.pgm          :1342-1359  binary PGM, comment line with the depth shift, 16-bit samples big-endian
//     <outputDir>/frame-%06d.pose.txt           :1706-1714  camera-to-world, four rows, no newline after the last
// The file names count as the reference's StringCounter does (:1317-1328), the numbers are written by the same iostream formatting, stdout follows
// main.cpp (header dump :1941-1955, "[ processing frame i of n ]" progress, "All done :)"), failures print "Exception caught! ..." and exit non-zero.
// tests/test_sens_export.py holds the output directory against the compiled reference's, byte for byte.
// Thin C++ host over libscanfuse.so's C ABI (sf_sens_open / frame_blobs / decode_depth / pose); no GPU involved.
#include <sys/stat.h>

#include <atomic>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "scanfuse.h"

namespace {

struct Counter {   // StringCounter (sensorData.h:1292-1337): base + zero padding to `digits` + count + ending
  std::string base, ending;
  unsigned digits, current = 0;
  Counter(const std::string& b, const std::string& e, unsigned d) : base(b), ending(e[0] == '.' ? e : "." + e), digits(d) {}
  std::string next() {
    std::stringstream ss;
    ss << base;
    for (unsigned i = std::max(1u, (unsigned)ceilf(log10f((float)current + 1))); i < digits; i++) ss << "0";
    ss << current++ << ending;
    return ss.str();
  }
};

int fail(const std::string& what) {   // main.cpp:82-90
  std::cout << "Exception caught! " << what << std::endl;
  return EXIT_FAILURE;
}

bool write_blob(const std::string& path, const void* data, size_t n) {
  FILE* fp = std::fopen(path.c_str(), "wb");
  if (!fp) return false;
  const bool ok = n == 0 || std::fwrite(data, 1, n, fp) == n;
  return std::fclose(fp) == 0 && ok;
}

}  // namespace

int main(int argc, char** argv) {
  // sens --from-images <folder> <out.sens> [jpg|png]: the way back (not an argument of the reference's tool; its library has the function,
  // SensorData::loadFromImages, sensorData.h:1468-1559, in FreeImage builds): a folder as this tool writes it -> a .sens
  if (argc >= 2 && std::string(argv[1]) == "--from-images") {
    if (argc < 4) { std::cout << "run ./sens --from-images <folder> <out.sens> [jpg|png]" << std::endl; return EXIT_FAILURE; }
    sf_sens* s = nullptr;
    if (sf_sens_load_from_images(argv[2], nullptr, argc >= 5 ? argv[4] : nullptr, &s) != SF_OK) return fail(sf_last_error());
    sf_sens_info info;
    sf_sens_get_info(s, &info);
    if (sf_sens_save(s, argv[3]) != SF_OK) return fail(sf_last_error());
    std::cout << "DONE" << std::endl;   // loadFromImages' own last word (:1510)
    std::cout << info.num_frames << " frames from " << argv[2] << " written to " << argv[3] << std::endl;
    sf_sens_close(s);
    return 0;
  }
  std::string filename = "scene0001_00.sens", out_dir = "./out/";
  if (argc >= 2) filename = argv[1];
  else {
    std::cout << "run ./sens <sensfilename>.sens";
    std::cout << "type in filename manually: ";
    std::cin >> filename;
  }
  if (argc >= 3) out_dir = argv[2];
  std::cout << "filename =\t" << filename << std::endl;
  std::cout << "outDir =\t" << out_dir << std::endl;
  std::cout << "loading from file... ";
  sf_sens* s = nullptr;
  if (sf_sens_open(filename.c_str(), &s) != SF_OK) return fail(sf_last_error());
  std::cout << "done!" << std::endl;
  sf_sens_info info;
  sf_sens_get_info(s, &info);
  std::cout << "CalibratedSensorData:\n"
            << "\tsensorData.m_versionNumber=" << info.version << '\n'
            << "\tsensorData.m_sensorName=" << info.sensor_name << '\n'
            << "\tsensorData.m_colorWidth=" << info.color_width << '\n'
            << "\tsensorData.m_colorHeight=" << info.color_height << '\n'
            << "\tsensorData.m_depthWidth=" << info.depth_width << '\n'
            << "\tsensorData.m_depthHeight=" << info.depth_height << '\n'
            << "\tsensorData.m_depthShift=" << info.depth_shift << '\n'
            << "\tsensorData.m_frames.size()=" << info.num_frames << '\n'
            << "\tsensorData.m_IMUFrames.size()=" << info.num_imu << '\n'
            << std::endl;
  struct stat st;
  if (::stat(out_dir.c_str(), &st) != 0) ::mkdir(out_dir.c_str(), 0777);   // one level, as ml::util::makeDirectory
  {
    std::ofstream meta(out_dir + "/_info.txt");
    if (!meta) return fail("cannot open file " + out_dir + "/_info.txt");
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
    if (info.num_imu > 0) std::cout << "warning sensor has imu frames; but writing is not implemented here" << std::endl;
  }
  if (info.num_frames != 0) {
    const std::string color_ending = info.color_compression == 2 ? "jpg" : "png";
    const uint64_t n = info.num_frames;
    std::vector<std::string> color_file(n), pose_file(n), pgm_file(n);
    {
      Counter color(out_dir + "/frame-", "color." + color_ending, 6), pose(out_dir + "/frame-", ".pose.txt", 6), pgm(out_dir + "/frame-", "depth.pgm", 6);
      for (uint64_t i = 0; i < n; i++) { color_file[i] = color.next(); pose_file[i] = pose.next(); pgm_file[i] = pgm.next(); }
    }
    std::cout << std::endl;
    // The frames are independent (three files each): a pool takes them in index order, the progress lines go out in index order as the frames
    // complete -- stdout stays the reference's, the wall time divides by the cores (the reference: one thread, ~2 ms of stb inflate + the writes per frame).
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
          const uint8_t *cblob = nullptr, *dblob = nullptr;
          uint64_t cbytes = 0, dbytes = 0;
          if (sf_sens_frame_blobs(s, i, &cblob, &cbytes, &dblob, &dbytes) != SF_OK) err = sf_last_error();
          else if (info.color_compression == 0 && cbytes != 0) {   // TYPE_RAW pixels: a PNG of them
            if (cbytes != (uint64_t)info.color_width * info.color_height * 3) err = "raw colour frame of " + std::to_string(cbytes) + " bytes";
            else if (sf_png_write(color_file[i].c_str(), cblob, info.color_width, info.color_height, 3, 8) != SF_OK) err = "cannot open file " + color_file[i];
          } else if (info.color_compression == 0) {
            // a TYPE_RAW file without colour (depth only): no colour file (the reference throws on the first frame of any TYPE_RAW file off Windows)
          } else if (info.color_compression == 1 || info.color_compression == 2) {
            if (!write_blob(color_file[i], cblob, (size_t)cbytes)) err = "cannot open file " + color_file[i];
          } else {
            err = "unknown format";
          }
          if (err.empty() && sf_sens_decode_depth(s, i, depth.data()) != SF_OK) err = sf_last_error();
          if (err.empty()) {
            std::ofstream of(pgm_file[i], std::ios::binary);
            std::stringstream ss;
            ss << "P5\n";
            ss << "# data values are 16-bit each; depth shift is " << info.depth_shift << "\n";
            ss << info.depth_width << " " << info.depth_height << "\n";
            ss << std::numeric_limits<unsigned short>::max() << "\n";
            of << ss.str();
            for (uint16_t& v : depth) v = (uint16_t)((v << 8) | (v >> 8));   // PGM samples are big-endian
            of.write((const char*)depth.data(), (std::streamsize)(depth.size() * 2));
            float m[16];
            int valid = 0;
            if (sf_sens_pose(s, i, m, &valid) != SF_OK) err = sf_last_error();
            else {
              std::ofstream pf(pose_file[i]);
              pf << m[0] << " " << m[1] << " " << m[2] << " " << m[3] << "\n"
                 << m[4] << " " << m[5] << " " << m[6] << " " << m[7] << "\n"
                 << m[8] << " " << m[9] << " " << m[10] << " " << m[11] << "\n"
                 << m[12] << " " << m[13] << " " << m[14] << " " << m[15];
            }
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
    const unsigned hw = std::thread::hardware_concurrency();
    const uint64_t T = std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(hw ? hw : 1), 16, n}));
    std::vector<std::thread> pool;
    for (uint64_t t = 0; t < T; t++) pool.emplace_back(work);
    std::string first_error;
    for (uint64_t i = 0; i < n && first_error.empty(); i++) {
      std::cout << "\r[ processing frame " << std::to_string(i) << " of " << std::to_string(n) << " ]";
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return done[i] != 0; });
      first_error = error[i];
    }
    for (auto& t : pool) t.join();
    if (!first_error.empty()) return fail(first_error);
  }
  std::cout << std::endl;
  std::cout << "All done :)" << std::endl;
  sf_sens_close(s);
  return 0;
}

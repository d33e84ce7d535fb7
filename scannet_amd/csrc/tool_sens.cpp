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
// Thin C++ host over libscanfuse.so's C ABI (sf_sens_open, sf_sens_save_to_images = SensorData::saveToImages, scannet_amd/csrc/sens_images.cpp); no GPU involved.
#include <cstdlib>
#include <iostream>
#include <string>

#include "scanfuse.h"

namespace {

int fail(const std::string& what) {   // main.cpp:82-90
  std::cout << "Exception caught! " << what << std::endl;
  return EXIT_FAILURE;
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
  if (info.num_imu > 0) std::cout << "warning sensor has imu frames; but writing is not implemented here" << std::endl;
  if (info.num_frames != 0) std::cout << std::endl;
  // SensorData::saveToImages in the library (sf_sens_save_to_images): the progress lines are printed here, in index order
  struct P {
    static void line(uint64_t i, uint64_t n, void*) { std::cout << "\r[ processing frame " << std::to_string(i) << " of " << std::to_string(n) << " ]"; }
  };
  if (sf_sens_save_to_images(s, out_dir.c_str(), nullptr, &P::line, nullptr) != SF_OK) return fail(sf_last_error());
  std::cout << std::endl;
  std::cout << "All done :)" << std::endl;
  sf_sens_close(s);
  return 0;
}

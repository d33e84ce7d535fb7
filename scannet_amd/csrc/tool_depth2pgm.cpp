// tool_depth2pgm.cpp -- drop-in for ScannerApp/depth2pgm (depth2pgm.cpp:41-102, README.md): the first frames of a ScannerApp `.depth` capture as PGMs --
//     depth2pgm path/to/file.depth pgm_seq_basename [numDepthFrames = 1]      ->  <basename>_<frame>.pgm, "frame i: read n [bytes] " per frame on stdout
// per frame u32 size + Occipital stream (uplinksimple::decode), shift -> millimetres (uplinksimple::shift2depth), values >= shift2depth(0xffff) -> 0,
// binary PGM with 16-bit big-endian samples (:10-24); 640x480 as the tool assumes (:42).  Output files and stdout are the compiled reference's byte for
// byte (tests/test_occipital.py).  Where the reference asserts or reads past the end of a short file, this one says what is wrong and exits non-zero.
// Thin C++ host over libscanfuse.so's C ABI (sf_occ_decode / sf_occ_shift2depth_buffer); no GPU involved.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "scanfuse.h"

int main(int argc, char* argv[]) {
  if (argc < 3) {
    std::cerr << "Usage: depth2pgm path/to/file.depth pgm_seq_basename [numDepthFrames]" << std::endl;
    return 0;   // as the reference does (:100-101)
  }
  const std::string depth_file(argv[1]), out_base(argv[2]);
  const int frames = argc >= 4 ? std::atoi(argv[3]) : 1;
  const size_t W = 640, H = 480;
  std::ifstream in(depth_file, std::ios::binary);
  if (!in) { std::cerr << "depth2pgm: cannot open " << depth_file << std::endl; return 1; }
  std::vector<uint8_t> stream;
  std::vector<uint16_t> depth(W * H);
  for (int frame = 0; frame < frames; frame++) {
    uint32_t bytes = 0;
    if (!in.read((char*)&bytes, 4)) { std::cerr << "depth2pgm: " << depth_file << " ends before frame " << frame << std::endl; return 1; }
    stream.resize(bytes);
    if (bytes && !in.read((char*)stream.data(), bytes)) { std::cerr << "depth2pgm: " << depth_file << " ends inside frame " << frame << std::endl; return 1; }
    if (sf_occ_decode(stream.data(), bytes, W * H, depth.data()) != SF_OK || sf_occ_shift2depth_buffer(depth.data(), W * H, 1) != SF_OK) {
      std::cerr << "depth2pgm: frame " << frame << ": " << sf_last_error() << std::endl;
      return 1;
    }
    const std::string filename = out_base + "_" + std::to_string(frame) + ".pgm";
    std::ofstream of(filename, std::ios::binary);
    std::stringstream ss;
    ss << "P5\n";
    ss << "# data values are 16-bit each" << "\n";
    ss << W << " " << H << "\n";
    ss << std::numeric_limits<unsigned short>::max() << "\n";
    of << ss.str();
    for (uint16_t& v : depth) v = (uint16_t)((v << 8) | (v >> 8));
    of.write((const char*)depth.data(), (std::streamsize)(W * H * 2));
    std::cout << "frame " << frame << ": read " << bytes << " [bytes] " << std::endl;
  }
  return 0;
}

// sens.h -- internal C++ view of an opened .sens container (shared by sens.cpp and pipeline.hip)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

struct SensFrame {
  float pose[16];
  uint64_t ts_color = 0, ts_depth = 0;
  const uint8_t* color = nullptr;  // compressed blobs: point into the mmap (reader) or into `owned` (writer)
  uint64_t color_bytes = 0;
  const uint8_t* depth = nullptr;
  uint64_t depth_bytes = 0;
  std::vector<uint8_t> owned;      // storage for frames added through the writer API (color then depth)
};

struct sf_sens {
  sf_sens_info info;
  std::vector<SensFrame> frames;
  std::vector<uint8_t> imu;  // numIMU * 128 bytes, kept verbatim
  void* map = nullptr;       // mmap of the source file
  uint64_t map_bytes = 0;
  int fd = -1;
};

// inflate / copy frame `i`'s depth into dst (W*H u16); thread-safe (no shared mutable state)
int sens_decode_depth(const sf_sens* s, uint64_t i, uint16_t* dst);

// mesh.h -- internal mesh container behind the sf_mesh handle (PLY I/O, marching-cubes output, Segmentator input)
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

struct sf_mesh {
  std::vector<float> pos;      // 3 per vertex
  std::vector<uint8_t> col;    // 4 per vertex (r,g,b,a); empty if the source had no colour
  std::vector<uint32_t> tri;   // 3 per face
  std::vector<uint64_t> keys;  // optional: canonical edge key per vertex (marching-cubes output)
  std::vector<uint64_t> tkeys; // optional: cube key per face, ascending (marching-cubes output): merging the meshes of a partitioned scan
};

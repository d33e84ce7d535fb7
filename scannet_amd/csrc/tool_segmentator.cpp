// tool_segmentator.cpp -- drop-in for the reference `Segmentator` executable of the `segment` stage
// (Server/scan_processor.py:155-156; Segmentator/segmentator.cpp:268-289, Segmentator/README.md:8-11):
//   segmentator input.ply [kThresh=0.01] [segMinVerts=20]  ->  <input minus ext>.<kThresh %f>.segs.json
// Same stdout lines as the reference, nothing on stderr on success (util.call logs stderr as an error,
// Server/util.py:42-44).  A thin host over libscanfuse.so's C ABI.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scanfuse.h"



int main(int argc, const char** argv) {
  if (argc < 2) {
    std::printf("Usage: ./segmentator input.ply [kThresh] [segMinVerts] (defaults: kThresh=0.01 segMinVerts=20)\n");
    return 255;  // the reference calls exit(-1)
  }
  // `--gpu [device]` behind the reference's arguments (not a flag of the reference's): vertex normals and edge weights on that GPU, the same file
  int device = -1;
  for (int i = 2; i < argc; i++)
    if (std::strcmp(argv[i], "--gpu") == 0) {
      device = (i + 1 < argc && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9') ? std::atoi(argv[i + 1]) : 0;
      argc = i;
      break;
    }
  const char* mesh = argv[1];
  const float kthr = argc > 2 ? (float)std::atof(argv[2]) : 0.01f;
  const int min_verts = argc > 3 ? std::atoi(argv[3]) : 20;
  std::printf("Segmenting %s with kThresh=%f, segMinVerts=%d ...\n", mesh, kthr, min_verts);
  uint64_t nseg = 0, counts[4] = {0, 0, 0, 0};
  char out[4096];
  int multi = 0;
  const int rc = device < 0 ? sf_segment_file_ex(mesh, kthr, min_verts, nullptr, &nseg, counts, out, sizeof(out), &multi)
                            : sf_segment_file_gpu(mesh, kthr, min_verts, nullptr, &nseg, counts, out, sizeof(out), &multi, device);
  if (rc != SF_OK) {
    std::fprintf(stderr, "%s\n", sf_last_error());
    return 1;
  }
  if (multi) std::fprintf(stderr, "Warning: only single mesh OBJ supported, segmenting first mesh\n");
  std::printf("Read mesh with vertexCount %lu %lu, faceCount %lu %lu\n", (unsigned long)counts[0], (unsigned long)counts[1],
              (unsigned long)counts[2], (unsigned long)counts[3]);
  std::printf("Segmentation written to %s with %lu segments\n", out, (unsigned long)nseg);
  return 0;
}

// tool_converter.cpp -- drop-in for the `convert` stage's converter.exe (Converter/main.cpp:182-241):
//     converter <staging folder of one ScannerApp capture> <out.sens>
// The capture is <folder>/<folder name>.{txt,depth,imu,h264} (main.cpp:186-193).  An existing output is left alone
// ("sensFile already available ... skipping folder", :198-201).  Colour: the reference shells out to ffmpeg to turn the
// .h264 into numbered images (:44-58); this tool uses already extracted frames when it finds them --
// <folder>/color/frame-%06d.color.jpg (JPEG blobs go into the .sens untouched, TYPE_JPEG) -- runs `ffmpeg` to produce them
// when it is on PATH and the .h264 exists, and otherwise writes the frames without colour (0-byte colour blobs) and says so
// on stdout.  Progress on stdout, failures on stderr with a non-zero exit (Server/util.py:38-50).
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "scanfuse.h"

namespace {
bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

struct ColorDir {
  std::string dir;
  std::vector<uint8_t> blob;
};
int color_cb(void* user, uint64_t frame, const uint8_t** blob, uint64_t* bytes) {
  ColorDir* c = (ColorDir*)user;
  char name[64];
  std::snprintf(name, sizeof(name), "/frame-%06llu.color.jpg", (unsigned long long)(frame + 1));  // ffmpeg's %6d starts at 1
  std::ifstream f(c->dir + name, std::ios::binary | std::ios::ate);
  if (!f) return SF_ERR_IO;
  const std::streamsize n = f.tellg();
  f.seekg(0);
  c->blob.resize((size_t)n);
  if (n && !f.read((char*)c->blob.data(), n)) return SF_ERR_IO;
  *blob = c->blob.data();
  *bytes = (uint64_t)n;
  return SF_OK;
}
}  // namespace

int main(int argc, const char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "requires the path and output file as a command line arguments\n");  // main.cpp:219
    return 1;
  }
  std::string folder = argv[1];
  for (char& ch : folder) if (ch == '\\') ch = '/';  // main.cpp:184
  while (folder.size() > 1 && folder.back() == '/') folder.pop_back();
  const std::string out = argv[2];
  std::printf("%s\n%s\n", folder.c_str(), out.c_str());
  const size_t slash = folder.find_last_of('/');
  const std::string base = folder + "/" + (slash == std::string::npos ? folder : folder.substr(slash + 1));
  std::printf("converting: %s\n", base.c_str());
  if (exists(out)) {
    std::printf("sensFile already available: %s\n\t -> skipping folder\n", out.c_str());
    return 0;
  }
  for (const char* ext : {".txt", ".depth", ".imu"})  // the reference also insists on the .h264 (main.cpp:26-29); colour is optional here
    if (!exists(base + ext)) { std::fprintf(stderr, "file not found %s%s\n", base.c_str(), ext); return 1; }
  sf_capture* cap = nullptr;
  if (sf_capture_open((base + ".depth").c_str(), &cap) != SF_OK) { std::fprintf(stderr, "%s\n", sf_last_error()); return 1; }
  sf_capture_meta meta;
  sf_capture_get_meta(cap, &meta);
  ColorDir cd;
  cd.dir = folder + "/color";
  bool have_color = exists(cd.dir + "/frame-000001.color.jpg");
  if (!have_color && exists(base + ".h264") && std::system("ffmpeg -version > /dev/null 2>&1") == 0) {
    const std::string cmd = "mkdir -p '" + cd.dir + "' && ffmpeg -i '" + base + ".h264' -q:v 2 '" + cd.dir + "/frame-%6d.color.jpg' > /dev/null 2>&1";
    std::printf("running: %s\n", cmd.c_str());
    have_color = std::system(cmd.c_str()) == 0 && exists(cd.dir + "/frame-000001.color.jpg");
  }
  if (!have_color) std::printf("no colour frames (no %s/frame-%%06d.color.jpg, no ffmpeg): writing depth-only frames\n", cd.dir.c_str());
  sf_convert_stats st;
  if (sf_capture_convert(cap, out.c_str(), have_color ? color_cb : nullptr, &cd, 2, 0, &st) != SF_OK) {
    std::fprintf(stderr, "%s\n", sf_last_error());
    sf_capture_close(cap);
    return 1;
  }
  if (st.frames != meta.num_depth_frames || st.frames != meta.num_color_frames)
    std::printf("frame counts are different: converted(%llu) meta.numDepthImages(%u) meta.numColorImages(%u)\n", (unsigned long long)st.frames,
                meta.num_depth_frames, meta.num_color_frames);
  std::printf("%llu frames, %llu IMU frames (%llu skipped), %llu depth-stream bytes, %u threads -> %s\n", (unsigned long long)st.frames,
              (unsigned long long)st.imu_frames, (unsigned long long)st.imu_skipped, (unsigned long long)st.depth_stream_bytes, st.threads, out.c_str());
  sf_capture_close(cap);
  return 0;
}

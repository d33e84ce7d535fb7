// tool_calibrate.cpp -- drop-in for the `calibrate` stage's calibrate.exe (Calibrate/src/main.cpp:41-80):
//     calibrate <input.sens> <output.sens> <device calibration map .csv> <device calibration directory>
// as Server/scan_processor.py:118 calls it.  The scan's meta file <dir of input>/<dir name>.txt names the deviceId
// (main.cpp:7-14); the CSV (header row with the columns `id` and `calibration_name`, :16-37) maps it to a calibration name;
// <dir>/<name>.txt and <dir>/<name>.lut are the parameter file and the distance table (:56-57).  No calibration name:
// "no calibration name found" and success (:59), as the reference.  Progress on stdout, failures on stderr + non-zero exit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "scanfuse.h"

namespace {
std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) {
    if (c == sep) { out.push_back(cur); cur.clear(); }
    else if (c != '\r') cur.push_back(c);
  }
  out.push_back(cur);
  return out;
}
std::string trim(std::string s) {
  const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n;");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
}  // namespace

int main(int argc, const char** argv) {
  (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);   // this process's streams on hardware queues of their own (the application's decision; an exported value wins)
  if (argc != 5) {
    std::fprintf(stderr, "requires the input sens filepath, output sens filepath, parameter file, and input undistortion table as a command line arguments\n");
    return 1;
  }
  const std::string in = argv[1], out = argv[2], csv = argv[3];
  std::string dir = argv[4];
  if (dir.empty() || (dir.back() != '/' && dir.back() != '\\')) dir.push_back('/');
  // the scan directory = the directory of the input file; its meta file carries the device id
  std::string scan = in;
  for (char& c : scan) if (c == '\\') c = '/';
  const size_t slash = scan.find_last_of('/');
  scan = slash == std::string::npos ? std::string(".") : scan.substr(0, slash);
  while (scan.size() > 1 && scan.back() == '/') scan.pop_back();
  const size_t s2 = scan.find_last_of('/');
  const std::string meta = scan + "/" + (s2 == std::string::npos ? scan : scan.substr(s2 + 1)) + ".txt";
  std::string name;
  {
    std::ifstream mf(meta);
    if (mf) {
      std::string line, device_id;
      while (std::getline(mf, line)) {
        const size_t eq = line.find('=');
        if (eq != std::string::npos && trim(line.substr(0, eq)) == "deviceId") device_id = trim(line.substr(eq + 1));
      }
      if (device_id.empty()) { std::fprintf(stderr, "no device id in meta file: %s\n", meta.c_str()); return 1; }
      std::ifstream cf(csv);
      std::string header;
      if (!cf || !std::getline(cf, header)) { std::fprintf(stderr, "failed to read device calibration map csv: %s\n", csv.c_str()); return 1; }
      const std::vector<std::string> h = split(header, ',');
      int id_col = -1, name_col = -1;
      for (size_t i = 0; i < h.size(); i++) {
        if (h[i] == "id") id_col = (int)i;
        if (h[i] == "calibration_name") name_col = (int)i;
      }
      if (id_col < 0 || name_col < 0) { std::fprintf(stderr, "unable to find device id/calibration name in device calibration map cs file: %s\n", csv.c_str()); return 1; }
      while (std::getline(cf, line)) {
        const std::vector<std::string> e = split(line, ',');
        if ((int)e.size() > std::max(id_col, name_col) && e[(size_t)id_col] == device_id) {
          name = e[(size_t)name_col];
          std::printf("\tdevice id: %s, calibration name: %s\n", device_id.c_str(), name.c_str());
          break;
        }
      }
    }
  }
  if (name.empty()) { std::printf("no calibration name found\n"); return 0; }
  const std::string params = dir + name + ".txt", lut = dir + name + ".lut";
  std::printf("loading .sens file %s... \n", in.c_str());
  sf_calibrate_stats st;
  if (sf_calibrate_sens(in.c_str(), out.c_str(), params.c_str(), lut.c_str(), 0, 0, &st) != SF_OK) {
    std::fprintf(stderr, "%s\n", sf_last_error());
    return 1;
  }
  if (st.skipped_existing) std::printf("output sens file %s already exists, skipping\n", out.c_str());
  else if (st.already_aligned) std::printf("color and depth is already aligned -- cannot further calibrate .sens file -> exiting\n");
  else std::printf("calibrateScan: %llu frames (%llu with colour) in %.2f s, %u threads\nsaving .sens file %s... done!\n", (unsigned long long)st.frames,
                   (unsigned long long)st.frames_with_colour, st.seconds_total, st.threads, out.c_str());
  return 0;
}

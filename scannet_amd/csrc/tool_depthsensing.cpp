// tool_depthsensing.cpp -- drop-in for the `improve` stage executable of the reference pipeline:
//     DepthSensing.exe zParametersScanNet.txt zParametersTrackingDefault.txt <abs path>.sens
// (Server/scan_processor.py:34-35,137-138; cwd = tool directory, Server/util.py:34-38).  Product: <dir>/<id>_vh.ply
// next to the .sens file (scan_processor.py:141; Server/config/scan_stages.json:33-42 checks existence only).
// Protocol kept: progress on stdout (captured into process.log, util.py:38-41), NOTHING on stderr on success
// (util.py:42-44 logs any stderr as an error), non-zero exit + stderr message on failure.
// Thin C++ host over libscanfuse.so's C ABI; device selected with SF_DEVICE (one process per GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "scanfuse.h"

static int die(const char* what) {
  std::fprintf(stderr, "%s: %s\n", what, sf_last_error());
  return 1;
}

int main(int argc, const char** argv_in) {
  // This PROCESS drives sf_fuse_run's seven streams: ask the HIP runtime for a hardware queue each before its first call (default 4: kernels of
  // streams that share a queue run one after the other).  The application's decision, not the library's; a value the user exported wins.
  (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
  // --upstream[=voxelhashing|bundlefusion]: the upstream-conformance preset (scanfuse.h sf_params_upstream_preset; default: SURVEY App. C).  Not an
  // argument of the tool this replaces -- the pipeline's command line (scan_processor.py:138) stays valid -- and it may stand anywhere.
  const char* argv[8];
  int upstream = 0, n = 0;
  for (int i = 0; i < argc; i++) {
    if (!std::strcmp(argv_in[i], "--upstream") || !std::strcmp(argv_in[i], "--upstream=voxelhashing")) upstream = 1;
    else if (!std::strcmp(argv_in[i], "--upstream=bundlefusion")) upstream = 2;
    else if (n < 8) argv[n++] = argv_in[i];
  }
  argc = n;
  if (argc < 4) {
    std::printf("Usage: depthsensing [--upstream[=voxelhashing|bundlefusion]] <zParameters.txt> <zParametersTracking.txt> <scan.sens> [out.ply]\n");
    return 255;
  }
  const char* sens_path = argv[3];
  sf_params p;
  sf_params_default(&p);
  if (upstream && sf_params_upstream_preset(&p, upstream) != SF_OK) return die("preset");
  if (sf_params_load_file(argv[1], &p) != SF_OK) return die("parameter file");   // s_scanfuse* keys of the file override the preset
  if (p.frustum_mode | p.colour_round | p.colour_first | p.weight_mode | p.weight_wrap)
    std::printf("Upstream-conformance switches: frustum_mode %d, colour_round %d, colour_first %d, weight_mode %d, weight_wrap %d\n", p.frustum_mode, p.colour_round,
                p.colour_first, p.weight_mode, p.weight_wrap);
  // the second parameter file holds tracking settings only; it must exist (the reference tool reads it) but nothing in it concerns fusion
  if (FILE* fp = std::fopen(argv[2], "r")) std::fclose(fp);
  else { std::fprintf(stderr, "could not open parameter file %s\n", argv[2]); return 1; }
  sf_sens* sens = nullptr;
  if (sf_sens_open(sens_path, &sens) != SF_OK) return die("sens");
  sf_sens_info info;
  sf_sens_get_info(sens, &info);
  std::printf("Loaded %s: %llu frames, depth %ux%u, color %ux%u, sensor '%s'\n", sens_path, (unsigned long long)info.num_frames, info.depth_width,
              info.depth_height, info.color_width, info.color_height, info.sensor_name);
  // the frames come at the file's depth resolution with the file's calibration (row-major intrinsic, sensorData.h:305-312); the fuser
  // resamples them to s_integrationWidth x s_integrationHeight when the parameter file asks for another size (zParametersScanNet.txt:20-21)
  p.depth_width = (int32_t)info.depth_width;
  p.depth_height = (int32_t)info.depth_height;
  if (p.integration_width > 0 && p.integration_height > 0 && (p.integration_width != p.depth_width || p.integration_height != p.depth_height))
    std::printf("Depth resampled to s_integrationWidth x s_integrationHeight = %d x %d\n", p.integration_width, p.integration_height);
  p.fx = info.depth_intrinsic[0]; p.fy = info.depth_intrinsic[5]; p.mx = info.depth_intrinsic[2]; p.my = info.depth_intrinsic[6];
  p.depth_shift = info.depth_shift;
  if ((info.color_width != info.depth_width || info.color_height != info.depth_height) && info.color_width > 0 && info.color_height > 0 &&
      (info.color_compression == 0 || info.color_compression == 2) && info.color_intrinsic[0] > 0) {
    // real ScanNet scans: 1296x968 colour over 640x480 depth -- sample the colour under each depth pixel's ray
    p.color_width = (int32_t)info.color_width; p.color_height = (int32_t)info.color_height;
    p.cfx = info.color_intrinsic[0]; p.cfy = info.color_intrinsic[5]; p.cmx = info.color_intrinsic[2]; p.cmy = info.color_intrinsic[6];
  }
  const int device = std::getenv("SF_DEVICE") ? std::atoi(std::getenv("SF_DEVICE")) : 0;
  sf_fuser* fuser = nullptr;
  if (sf_fuser_create(&p, device, &fuser) != SF_OK) return die("fuser");
  sf_run_stats rs;
  if (sf_fuse_run(fuser, sens, 0, 0, 0, &rs) != SF_OK) return die("fuse");
  sf_stats st;
  sf_fuser_stats(fuser, &st);
  std::printf("Integrated %llu frames (%llu skipped: invalid pose) in %.3f s = %.1f frames/s with %u decode threads; %u SDF blocks, heapFreeCount = %u\n",
              (unsigned long long)rs.frames_integrated, (unsigned long long)rs.frames_skipped, rs.seconds_total,
              rs.seconds_total > 0 ? (double)rs.frames_total / rs.seconds_total : 0.0, rs.decode_threads, st.blocks_allocated, st.heap_free);
  if (st.alloc_failures) std::printf("WARNING: %u block allocations failed (s_hashNumSDFBlocks / s_hashNumBuckets too small)\n", st.alloc_failures);
  if (p.gc_enabled) {
    uint32_t freed = 0;
    if (sf_fuser_garbage_collect(fuser, &freed) != SF_OK) return die("garbage collection");
    std::printf("Garbage collection freed %u blocks\n", freed);
  }
  sf_mesh* mesh = nullptr;
  if (sf_fuser_extract_mesh(fuser, &mesh) != SF_OK) return die("marching cubes");
  uint64_t nv = 0, nf = 0;
  sf_mesh_counts(mesh, &nv, &nf);
  std::string out;
  if (argc > 4) out = argv[4];
  else {
    out = sens_path;
    const size_t dot = out.find_last_of('.');
    if (dot != std::string::npos) out = out.substr(0, dot);
    out += "_vh.ply";
  }
  if (sf_mesh_write_ply(mesh, out.c_str()) != SF_OK) return die("ply");
  std::printf("Mesh with %llu vertices, %llu faces written to %s\n", (unsigned long long)nv, (unsigned long long)nf, out.c_str());
  sf_mesh_free(mesh);
  sf_fuser_destroy(fuser);
  sf_sens_close(sens);
  return 0;
}

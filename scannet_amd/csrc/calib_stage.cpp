// calib_stage.cpp -- the `calibrate` stage end to end: <in>.sens -> threaded decode -> GPU image operations (calibrate.hip) ->
// re-encode -> <out>.sens.  Replaces Calibration::calibrateScan(inSens, outSens, params, table), Calibrate/src/calibration.h:87-137
// (file handling and header rewrite) around the per-frame body :253-307.
//   * missing input but existing output: "already exists, skipping" (:89-93); missing parameter / table file: error (:98)
//   * a file whose depth extrinsic is the identity is "already aligned": moved to the output name untouched (:112-116)
//   * image dimensions must match the calibration (:254)
//   * header of the result (:119-129): sensor name + " (calibrated)", both extrinsics identity, colour intrinsic = the
//     calibration's, depth intrinsic = the colour intrinsic scaled to the depth resolution (fx, fy by W_d / W_c, H_d / H_c;
//     mx, my by (W_d - 1) / (W_c - 1), (H_d - 1) / (H_c - 1))
//   * frames keep their poses; colour is re-encoded in the file's colour type (replaceColor :268 -> compressColor), depth in its
//     depth type (replaceDepth :302); the input file is deleted once the output is written when the names differ (:135)
// One deliberate difference: the reference's replaceColor / replaceDepth go through freeColor / freeDepth, which ZERO the frame's
// time stamps (SensReader/c++/src/sensorData.h:510-523); here the time stamps are kept.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "jpeg_idct.h"
#include "sens.h"

int jpeg_encode_rgb(const uint8_t* rgb, uint32_t width, uint32_t height, int quality, int subsample, std::vector<uint8_t>& out);  // jpeg_enc.cpp
int jpeg_decode_coef(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);  // jpeg.cpp
size_t calibrator_payload_capacity(const sf_calibrator* c);                                                                            // calibrate.hip
int calibrator_run_payload(sf_calibrator* c, int n, const uint8_t* const* rgb_in, const uint8_t* const* payload, const uint32_t* payload_bytes,
                           uint8_t* const* rgb_out, const uint16_t* const* depth_in, uint16_t* const* depth_out);                      // calibrate.hip

namespace {
bool file_exists(const char* p) { struct stat st; return ::stat(p, &st) == 0; }
}  // namespace

SF_API int sf_calibrate_sens(const char* in_sens, const char* out_sens, const char* params_txt, const char* lut_path, int device, int threads,
                             sf_calibrate_stats* stats) {
  if (!in_sens || !out_sens || !params_txt) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  sf_calibrate_stats st;
  std::memset(&st, 0, sizeof(st));
  const auto t_start = std::chrono::steady_clock::now();
  if (!file_exists(in_sens)) {
    if (file_exists(out_sens)) { st.skipped_existing = 1; if (stats) *stats = st; return SF_OK; }
    return sf::fail(SF_ERR_IO, "no sens file: %s", in_sens);
  }
  if (!file_exists(params_txt) || (lut_path && !file_exists(lut_path)))
    return sf::fail(SF_ERR_IO, "no calibration param file(s): %s / %s", params_txt, lut_path ? lut_path : "(none)");
  sf_calib_params cp;
  int rc = sf_calib_params_load(params_txt, &cp);
  if (rc != SF_OK) return rc;
  sf_lut lut;
  std::memset(&lut, 0, sizeof(lut));
  if (lut_path && (rc = sf_lut_load(lut_path, &lut)) != SF_OK) return rc;
  sf_sens* in = nullptr;
  if ((rc = sf_sens_open(in_sens, &in)) != SF_OK) { sf_lut_free(&lut); return rc; }
  sf_sens* out = nullptr;
  sf_calibrator* cal = nullptr;
  auto done = [&](int code) {
    if (cal) sf_calibrator_destroy(cal);
    if (out) sf_sens_close(out);
    if (in) sf_sens_close(in);
    sf_lut_free(&lut);
    return code;
  };
  const sf_sens_info hi = in->info;
  {
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (std::memcmp(hi.depth_extrinsic, id, 64) == 0) {  // calibration.h:112-116
      sf_sens_close(in);
      in = nullptr;
      if (std::strcmp(in_sens, out_sens) != 0 && std::rename(in_sens, out_sens) != 0) return done(sf::fail(SF_ERR_IO, "could not move %s to %s", in_sens, out_sens));
      st.already_aligned = 1;
      if (stats) *stats = st;
      return done(SF_OK);
    }
  }
  if (cp.depth_width != hi.depth_width || cp.depth_height != hi.depth_height) return done(sf::fail(SF_ERR_INVALID_ARG, "image dimensions do not match with calibration"));
  if (cp.color_width != hi.color_width || cp.color_height != hi.color_height)
    return done(sf::fail(SF_ERR_INVALID_ARG, "colour image dimensions do not match with calibration"));
  if (hi.color_compression != 0 && hi.color_compression != 2) return done(sf::fail(SF_ERR_UNSUPPORTED, "colour compression type %d cannot be rewritten", hi.color_compression));
  if ((rc = sf_calibrator_create(&cp, lut.data ? &lut : nullptr, hi.depth_shift, device, &cal)) != SF_OK) return done(rc);
  sf_sens_info ho = hi;
  {
    std::string name = std::string(hi.sensor_name) + " (calibrated)";
    std::snprintf(ho.sensor_name, sizeof(ho.sensor_name), "%s", name.c_str());
    std::memset(ho.color_extrinsic, 0, 64);
    std::memset(ho.depth_extrinsic, 0, 64);
    for (int i = 0; i < 4; i++) ho.color_extrinsic[5 * i] = ho.depth_extrinsic[5 * i] = 1.0f;
    std::memcpy(ho.color_intrinsic, cp.color_intrinsic, 64);
    std::memcpy(ho.depth_intrinsic, cp.color_intrinsic, 64);
    ho.depth_intrinsic[0] *= (float)hi.depth_width / (float)hi.color_width;              // :125-128
    ho.depth_intrinsic[5] *= (float)hi.depth_height / (float)hi.color_height;
    ho.depth_intrinsic[2] *= (float)(hi.depth_width - 1) / (float)(hi.color_width - 1);
    ho.depth_intrinsic[6] *= (float)(hi.depth_height - 1) / (float)(hi.color_height - 1);
  }
  if ((rc = sf_sens_create(&ho, &out)) != SF_OK) return done(rc);
  const uint64_t nframes = in->frames.size();
  const size_t npx = (size_t)hi.depth_width * hi.depth_height, cbytes = (size_t)hi.color_width * hi.color_height * 3;
  const int B = sf_calibrator_max_batch();
  int nthreads = threads > 0 ? threads : sf::usable_cpus();   // the cgroup quota, not the logical CPUs the container shows
  nthreads = std::max(1, std::min(nthreads, 64));
  // JPEG colour: the threads only entropy-decode; the coefficients go to the GPU, which reconstructs the picture into the stage's input
  // buffer (jpeg_gpu.hip, the bytes of the host decoder).  A frame whose coefficients do not fit, or whose layout the GPU path does not
  // take, is decoded on the host.
  const bool jpeg_in = hi.color_compression == 2 && std::getenv("SF_JPEG_HOST") == nullptr;
  const size_t pay_cap = jpeg_in ? calibrator_payload_capacity(cal) : 0;
  std::vector<std::vector<uint32_t>> pay((size_t)B);
  std::vector<uint32_t> pay_bytes((size_t)B, 0);
  if (jpeg_in) for (auto& v : pay) v.resize(pay_cap / 4);
  std::vector<uint16_t> d_in((size_t)B * npx), d_out((size_t)B * npx);
  std::vector<uint8_t> c_in((size_t)B * cbytes), c_out((size_t)B * cbytes);
  std::vector<std::vector<uint8_t>> jpg((size_t)B);
  std::vector<int> rcs((size_t)B);
  std::vector<std::string> errs((size_t)B);
  auto parallel = [&](int cnt, auto&& fn) {
    std::atomic<int> next{0};
    auto work = [&]() { for (;;) { const int k = next.fetch_add(1); if (k >= cnt) return; fn(k); } };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads && t < cnt; t++) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
  };
  // colour is a property of the FILE, not of where a frame falls in a batch: a scan without any colour blob is calibrated depth-only; in a
  // scan that has colour, a frame without a blob is an error, as it is for the reference (which decompresses colour per frame,
  // calibration.h:185-190, and throws on an empty blob, sensorData.h:607)
  bool rgb = false;
  for (uint64_t k = 0; k < nframes && !rgb; k++) rgb = in->frames[k].color_bytes != 0;
  for (uint64_t f0 = 0; f0 < nframes; f0 += (uint64_t)B) {
    const int cnt = (int)std::min<uint64_t>((uint64_t)B, nframes - f0);
    parallel(cnt, [&](int k) {
      rcs[k] = sens_decode_depth(in, f0 + k, &d_in[(size_t)k * npx]);
      pay_bytes[k] = 0;
      if (rcs[k] == SF_OK && rgb && jpeg_in && in->frames[f0 + k].color_bytes != 0) {
        const SensFrame& fr = in->frames[f0 + k];
        uint8_t* pp = reinterpret_cast<uint8_t*>(pay[k].data());
        if (jpeg_decode_coef(fr.color, fr.color_bytes, hi.color_width, hi.color_height, pp, pay_cap) == SF_OK)
          pay_bytes[k] = (uint32_t)sf_jpeg_payload_bytes(*reinterpret_cast<const SfJpegLayout*>(pp));
      }
      if (rcs[k] == SF_OK && rgb && pay_bytes[k] == 0) rcs[k] = sf_sens_decode_color(in, f0 + k, &c_in[(size_t)k * cbytes]);
      if (rcs[k] != SF_OK) errs[k] = sf_last_error();
    });
    for (int k = 0; k < cnt; k++)
      if (rcs[k] != SF_OK) return done(sf::fail(rcs[k], "frame %llu: %s", (unsigned long long)(f0 + k), errs[k].c_str()));
    const uint16_t* di[16]; uint16_t* dou[16]; const uint8_t* ri[16]; uint8_t* ro[16];
    for (int k = 0; k < cnt; k++) {
      di[k] = &d_in[(size_t)k * npx]; dou[k] = &d_out[(size_t)k * npx];
      ri[k] = &c_in[(size_t)k * cbytes]; ro[k] = &c_out[(size_t)k * cbytes];
    }
    if (rgb && jpeg_in) {
      const uint8_t* pl[16];
      for (int k = 0; k < cnt; k++) pl[k] = pay_bytes[k] ? reinterpret_cast<const uint8_t*>(pay[k].data()) : nullptr;
      if ((rc = calibrator_run_payload(cal, cnt, ri, pl, pay_bytes.data(), ro, di, dou)) != SF_OK) return done(rc);
    } else if ((rc = sf_calibrator_run(cal, cnt, rgb ? ri : nullptr, rgb ? ro : nullptr, di, dou)) != SF_OK) return done(rc);
    if (rgb && hi.color_compression == 2) {
      parallel(cnt, [&](int k) {
        jpg[k].clear();
        rcs[k] = jpeg_encode_rgb(ro[k], hi.color_width, hi.color_height, 90, 1, jpg[k]);
      });
      for (int k = 0; k < cnt; k++)
        if (rcs[k] != SF_OK) return done(sf::fail(rcs[k], "frame %llu: colour re-encode failed", (unsigned long long)(f0 + k)));
    }
    for (int k = 0; k < cnt; k++) {
      const SensFrame& fr = in->frames[f0 + k];
      const uint8_t* cptr = nullptr;
      uint64_t cb = 0;
      if (rgb) {
        if (hi.color_compression == 2) { cptr = jpg[k].data(); cb = jpg[k].size(); }
        else { cptr = ro[k]; cb = cbytes; }
      }
      if ((rc = sf_sens_add_frame(out, cptr, cb, dou[k], fr.pose, fr.ts_color, fr.ts_depth)) != SF_OK) return done(rc);
      st.frames++;
      st.frames_with_colour += rgb ? 1 : 0;
    }
  }
  out->imu = in->imu;
  if ((rc = sf_sens_save(out, out_sens)) != SF_OK) return done(rc);
  const bool same = std::strcmp(in_sens, out_sens) == 0;
  sf_sens_close(in);
  in = nullptr;
  if (!same) std::remove(in_sens);  // calibration.h:135
  st.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  st.threads = (uint32_t)nthreads;
  if (stats) *stats = st;
  return done(SF_OK);
}

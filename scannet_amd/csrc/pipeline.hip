// pipeline.hip -- .sens stream -> fuser: multi-threaded depth decode overlapped with H2D copies and the
// fusion kernels.  Replaces the per-frame loop of the external DepthSensing.exe around the reference's
// RGBDFrameCacheRead (SensReader/c++/src/sensorData.h:1717-1831: ONE decode thread, spin-waiting consumer):
// here a pool of decode threads fills a ring of pinned host buffers in frame order, the caller's thread
// issues the asynchronous copy + kernels for each frame as soon as it is decoded, and nothing spins.
// Frames whose camToWorld is -inf (tracking lost, sensorData.h:382) are skipped, as every reference
// consumer does (Alignment/src/alignment.h:26,54; Filter2dAnnotations.cpp:317).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "fuser_internal.h"
#include "sens.h"

int sf_fuser_run_batch(sf_fuser* f, const void* const* d_depth, const void* const* d_rgb, const float* const* poses, int n);  // fuser.hip

namespace {

struct Slot {
  uint16_t* h_depth = nullptr;
  uint8_t* h_rgb = nullptr;
  void* d_depth = nullptr;
  void* d_rgb = nullptr;
  hipEvent_t copied = nullptr;    // H2D of this slot finished (host buffer reusable)
  hipEvent_t consumed = nullptr;  // pre-pass of the frame in this slot finished (device buffer reusable)
  bool used = false;
  int decode_rc = SF_OK;
  std::string decode_err;
};

}  // namespace

SF_API int sf_fuse_run(sf_fuser* f, const sf_sens* s, uint64_t first, uint64_t last, int decode_threads, sf_run_stats* stats) {
  if (!f || !s) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint64_t nframes = s->frames.size();
  if (last == 0 || last > nframes) last = nframes;
  if (first > last) return sf::fail(SF_ERR_BOUNDS, "first frame %llu beyond last %llu", (unsigned long long)first, (unsigned long long)last);
  if ((int)s->info.depth_width != f->p.depth_width || (int)s->info.depth_height != f->p.depth_height)
    return sf::fail(SF_ERR_INVALID_ARG, "fuser was created for %dx%d depth frames, the .sens file holds %ux%u", f->p.depth_width, f->p.depth_height,
                    s->info.depth_width, s->info.depth_height);
  SF_HIP_CHECK(hipSetDevice(f->device));
  const auto t_start = std::chrono::steady_clock::now();
  const size_t npx = (size_t)f->p.depth_width * f->p.depth_height;
  // colour is fused when it is stored at depth resolution (raw or JPEG); other resolutions: geometry only
  // colour is fused when its frames match what the fuser was created for: depth resolution, or the colour resolution
  // given in sf_params (raw or JPEG); anything else: geometry only
  const bool same_res = s->info.color_width == s->info.depth_width && s->info.color_height == s->info.depth_height;
  const bool own_res = f->pk.cW > 0 && (int)s->info.color_width == f->pk.cW && (int)s->info.color_height == f->pk.cH;
  const bool use_rgb = ((same_res && f->pk.cW == 0) || own_res) && (s->info.color_compression == 0 || s->info.color_compression == 2);
  const size_t cpx = f->pk.cW > 0 ? (size_t)f->pk.cW * f->pk.cH : npx;
  int nthreads = decode_threads > 0 ? decode_threads : (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  const uint64_t total = last - first;
  const int B = f->batch;  // frames fused per pass over the voxel tiles
  const int R = (int)std::min<uint64_t>(std::max<uint64_t>(std::max<uint64_t>(4 * (uint64_t)nthreads, 16), 3 * (uint64_t)B), std::max<uint64_t>(total, 1));
  std::vector<Slot> ring((size_t)R);
  hipStream_t copy_stream = nullptr;
  auto cleanup = [&]() {
    for (Slot& sl : ring) {
      if (sl.h_depth) (void)hipHostFree(sl.h_depth);
      if (sl.h_rgb) (void)hipHostFree(sl.h_rgb);
      if (sl.d_depth) (void)hipFree(sl.d_depth);
      if (sl.d_rgb) (void)hipFree(sl.d_rgb);
      if (sl.copied) (void)hipEventDestroy(sl.copied);
      if (sl.consumed) (void)hipEventDestroy(sl.consumed);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  };
#define RUN_CHECK(call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) { cleanup(); return sf::fail(SF_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); } \
  } while (0)
  RUN_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  for (Slot& sl : ring) {
    RUN_CHECK(hipHostMalloc((void**)&sl.h_depth, npx * 2, hipHostMallocDefault));
    RUN_CHECK(hipMalloc(&sl.d_depth, npx * 2));
    if (use_rgb) {
      RUN_CHECK(hipHostMalloc((void**)&sl.h_rgb, cpx * 3, hipHostMallocDefault));
      RUN_CHECK(hipMalloc(&sl.d_rgb, cpx * 3));
    }
    RUN_CHECK(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    RUN_CHECK(hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));
  }

  // ---- decode pool: frame k (0-based within [first,last)) goes to slot k % R once frame k-R has been issued
  std::mutex mu;
  std::condition_variable cv_ready, cv_free;
  std::vector<uint8_t> ready((size_t)R, 0);  // slot holds a decoded frame
  uint64_t issued = 0;                       // frames the main thread has finished with (copy queued)
  std::atomic<uint64_t> next{0};
  std::atomic<bool> abort{false};
  std::atomic<uint64_t> decode_ns{0};
  auto worker = [&]() {
    for (;;) {
      const uint64_t k = next.fetch_add(1);
      if (k >= total || abort.load()) return;
      const int si = (int)(k % (uint64_t)R);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_free.wait(lk, [&] { return abort.load() || k < issued + (uint64_t)R; });
        if (abort.load()) return;
      }
      Slot& sl = ring[(size_t)si];
      const uint64_t frame = first + k;
      const auto t0 = std::chrono::steady_clock::now();
      int rc = SF_OK;
      const bool valid = s->frames[frame].pose[0] != -INFINITY;
      if (valid) {
        rc = sens_decode_depth(s, frame, sl.h_depth);
        if (rc == SF_OK && use_rgb && s->frames[frame].color_bytes) rc = sf_sens_decode_color(s, frame, sl.h_rgb);
      }
      decode_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
      {
        std::lock_guard<std::mutex> lk(mu);
        sl.decode_rc = rc;
        if (rc != SF_OK) sl.decode_err = sf_last_error();
        ready[(size_t)si] = 1;
      }
      cv_ready.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);

  int result = SF_OK;
  std::string err;
  uint64_t n_int = 0, n_skip = 0;
  // frames whose copies are queued but whose kernels are not: fused B at a time (one pass over the tiles per batch)
  int pend_slot[MAX_BATCH];
  const float* pend_pose[MAX_BATCH];
  int pend = 0;
  bool pend_rgb = false;
  auto flush = [&]() -> int {
    if (pend == 0) return SF_OK;
    hipStream_t in_stream = sf_input_stream(f, pend, pend_rgb, +1);  // the stream this batch's pre-pass runs on
    const void* dd[MAX_BATCH];
    const void* dr[MAX_BATCH];
    for (int q = 0; q < pend; q++) {
      Slot& ps = ring[(size_t)pend_slot[q]];
      dd[q] = ps.d_depth;
      dr[q] = pend_rgb ? ps.d_rgb : nullptr;
      if (hipStreamWaitEvent(in_stream, ps.copied, 0) != hipSuccess) return sf::fail(SF_ERR_DEVICE, "hipStreamWaitEvent failed");
    }
    const int rc = sf_fuser_run_batch(f, dd, pend_rgb ? dr : nullptr, pend_pose, pend);
    if (rc != SF_OK) return rc;
    for (int q = 0; q < pend; q++) {
      Slot& ps = ring[(size_t)pend_slot[q]];
      (void)hipEventRecord(ps.consumed, in_stream);  // device buffers of the batch are free once its pre-pass has run
      ps.used = true;
    }
    n_int += (uint64_t)pend;
    pend = 0;
    return SF_OK;
  };
  for (uint64_t k = 0; k < total && result == SF_OK; k++) {
    const int si = (int)(k % (uint64_t)R);
    Slot& sl = ring[(size_t)si];
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_ready.wait(lk, [&] { return ready[(size_t)si] != 0; });
      ready[(size_t)si] = 0;
    }
    const uint64_t frame = first + k;
    const float* pose = s->frames[frame].pose;
    if (sl.decode_rc != SF_OK) { result = sl.decode_rc; err = sl.decode_err; }
    else if (pose[0] == -INFINITY) { n_skip++; f->frames_skipped++; }
    else {
      const bool rgb = use_rgb && s->frames[frame].color_bytes;
      if (pend > 0 && rgb != pend_rgb) {  // a batch is all-colour or all-geometry
        const int rc = flush();
        if (rc != SF_OK) { result = rc; err = sf_last_error(); }
      }
      hipError_t e = hipSuccess;
      if (result == SF_OK) {
        if (sl.used) e = hipStreamWaitEvent(copy_stream, sl.consumed, 0);  // device buffer still read by an earlier pre-pass?
        if (e == hipSuccess) e = hipMemcpyAsync(sl.d_depth, sl.h_depth, npx * 2, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess && rgb) e = hipMemcpyAsync(sl.d_rgb, sl.h_rgb, cpx * 3, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(sl.copied, copy_stream);
        if (e != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("copy pipeline: ") + hipGetErrorString(e); }
      }
      if (result == SF_OK) {
        pend_slot[pend] = si;
        pend_pose[pend] = pose;
        pend_rgb = rgb;
        pend++;
        if (pend == B) {
          const int rc = flush();
          if (rc != SF_OK) { result = rc; err = sf_last_error(); }
        }
        // the pinned host buffer goes back to the decoders once its copy has landed
        if (result == SF_OK) (void)hipEventSynchronize(sl.copied);
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      issued = k + 1;
    }
    cv_free.notify_all();
  }
  if (result == SF_OK) {
    const int rc = flush();
    if (rc != SF_OK) { result = rc; err = sf_last_error(); }
  }
  if (result != SF_OK) {
    abort.store(true);
    { std::lock_guard<std::mutex> lk(mu); issued = total + (uint64_t)R; }
    cv_free.notify_all();
  }
  for (std::thread& t : pool) t.join();
  const hipError_t qe = sf_quiesce(f);
  (void)hipStreamSynchronize(copy_stream);
  cleanup();
  if (result != SF_OK) return sf::fail(result, "%s", err.c_str());
  if (qe != hipSuccess) return sf::fail(SF_ERR_DEVICE, "device error while fusing: %s", hipGetErrorString(qe));
  if (stats) {
    stats->frames_total = total;
    stats->frames_integrated = n_int;
    stats->frames_skipped = n_skip;
    stats->decode_threads = (uint32_t)nthreads;
    stats->color_fused = use_rgb ? 1u : 0u;
    stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    stats->seconds_decode_cpu = (double)decode_ns.load() * 1e-9;
  }
  return SF_OK;
}

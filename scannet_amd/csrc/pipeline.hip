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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "fuser_internal.h"
#include "jpeg_huff.h"
#include "jpeg_idct.h"
#include "sens.h"

int sf_fuser_run_batch(sf_fuser* f, const void* const* d_depth, const void* const* d_rgb, const float* const* poses, int n);  // fuser.hip
int sf_fuser_run_batch_ycc(sf_fuser* f, const void* const* d_depth, const void* const* d_planes, const void* const* d_layout, const float* const* poses, int n);  // fuser.hip
int jpeg_decode_coef(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);  // jpeg.cpp
void inflate_gpu_warm();     // inflate_gpu.hip, jpeg_gpu.hip, jpeg_huff_gpu.hip: load the file's code object now
void jpeg_gpu_warm();
void jpeg_huff_gpu_warm();
bool inflate_gpu_takes(const uint8_t* z, uint64_t n);  // inflate_gpu.hip
int inflate_gpu_batch(hipStream_t stream, int n, const uint32_t* const* d_words, const uint32_t* nbytes, uint8_t* const* d_out, uint16_t* const* d_plan, uint32_t expect,
                      const int32_t* tags, int32_t* d_status);  // inflate_gpu.hip
int jpeg_prepare_huff(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);  // jpeg.cpp
int jpeg_gpu_huffman(hipStream_t stream, int n, const uint8_t* const* d_prepared, uint8_t* const* d_payload, const uint32_t* max_entries, const int32_t* tags,
                     int32_t* d_status);  // jpeg_huff_gpu.hip
int jpeg_gpu_planes(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_planes, uint32_t max_blocks);   // jpeg_gpu.hip
int jpeg_gpu_reconstruct(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_rgb, uint8_t* const* d_planes, uint32_t max_blocks,
                         uint32_t max_width, uint32_t max_height);  // jpeg_gpu.hip

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of streams that share a queue
// run one after the other.  sf_fuse_run drives up to seven streams (the fuser's two, two for colour copies, three for the inflate kernels); on four
// queues every third batch's inflate sat in the integrate pass's queue (29 k -> 20 k frames/s in the loop).  The variable belongs to the PROCESS
// (it is read at its first HIP call): the library does not touch the environment -- the bin/ tools and bench.py export GPU_MAX_HW_QUEUES=16 in
// their own main() before the first HIP call, and a run that finds fewer queues than it has streams says so through
// sf_last_error() while returning SF_OK (sf_fuse_run_note).

// the default of the JPEG entropy decoding (see gpu_huffman in sf_fuse_run), decided by measurement (profiles/r05_e2e_rgbd.json: 1296x968 pictures of ~200 KB,
// device + 4 host threads 12.3 k frames/s before the five side streams, host decoding on 16 threads 10.0 k, on 4 threads 3.2 k): on the device whenever the
// batch has a side stream to decode on
#define SF_JPEG_DEVICE_HUFFMAN_DEFAULT(has_side_stream) (has_side_stream)

// -DSF_SIDE_PRIO=1 creates the side streams (inflate and JPEG kernels, colour copies) at the device's highest stream priority.  The idea: their kernels are
// small grids of LARGE workgroups (1024 lanes, 100+ registers per lane) that need a whole free CU, and at the default priority they might wait behind the
// integrate pass.  Measured (profiles/r06_e2e_rgbd_ab.txt): 16 267 against 16 182 frames/s on a 2 048-frame RGB-D scan -- nothing; the default stays 0.
#ifndef SF_SIDE_PRIO
#define SF_SIDE_PRIO 0
#endif

namespace {

hipError_t create_side_stream(hipStream_t* out) {
#if SF_SIDE_PRIO
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) return hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi);
#endif
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

thread_local uint64_t t_run_counts[4] = {0, 0, 0, 0};   // of this thread's last sf_fuse_run: depth frames inflated on the device / by the host threads, colour
thread_local char t_run_note[320] = "";   // sf_fuse_run_note(): a hint about the calling thread's last run that is not an error
                                                        // frames entropy-decoded on the device / by the host threads (sf_fuse_run_device_counts)

int hardware_queues_of_the_process() {   // what the runtime was (or will be) told; its default is 4
  const char* v = std::getenv("GPU_MAX_HW_QUEUES");
  const int n = v ? std::atoi(v) : 0;
  return n > 0 ? n : 4;
}

}  // namespace
void sf_run_resources_prepare_ex(int device, size_t pinned_bytes, size_t device_bytes, size_t plan_bytes, int side_streams, int copy_streams);
namespace {

// What sf_fuse_run sets up and does not need fresh: five streams (a hardware queue each: ~5 ms to create, and the runtime creates them one
// after the other whatever the threads do) and the pinned pool (~6 ms per 100 MB).  Kept per device for the life of the process and handed
// to one run at a time -- a dataset rebuild fuses 1513 scans in a process; a run that finds the set taken makes its own.
struct RunResources {
  int device = -1;
  bool taken = false;
  hipStream_t copy[2] = {nullptr, nullptr}, inflate[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t* h_pool = nullptr;
  size_t h_bytes = 0;
  uint8_t* d_pool = nullptr;   // the ring's device side and the inflate kernels' scratch: kept too -- the first DMA into freshly allocated device memory blocked
  size_t d_bytes = 0;          // the enqueueing thread 0.4 ms per call (36 ms of a first run's loop)
  uint8_t* d_plan[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t plan_bytes = 0;
  std::thread prep;   // sf_run_resources_prepare: fills inflate[0..2] and h_pool in the background; joined by whoever takes the set first
};
std::mutex g_res_mu;
std::vector<RunResources*> g_res;   // never freed: the streams and the pool die with the process
struct JoinAtExit {
  ~JoinAtExit() {
    std::lock_guard<std::mutex> lk(g_res_mu);
    for (RunResources* r : g_res)
      if (r->prep.joinable()) r->prep.join();
  }
} g_join_at_exit;
RunResources* acquire_resources(int device) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (RunResources* r : g_res)
    if (r->device == device && !r->taken) {
      if (r->prep.joinable()) r->prep.join();   // the preparation thread never takes g_res_mu
      r->taken = true;
      return r;
    }
  for (RunResources* r : g_res)
    if (r->device == device) return nullptr;   // taken: the caller works with resources of its own
  RunResources* r = new RunResources;
  r->device = device;
  r->taken = true;
  g_res.push_back(r);
  return r;
}
void release_resources(RunResources* r) {
  if (!r) return;
  std::lock_guard<std::mutex> lk(g_res_mu);
  r->taken = false;
}

}  // namespace

// fuser_internal.h: called by sf_fuser_create.  The first sf_fuse_run of a process used to spend 23 ms (depth only) creating its three side streams -- the
// runtime builds a hardware queue per stream, ~5 ms each -- and page-locking its ring before the first frame moved: 13 % of a 5 578-frame scan that is fused in
// 0.18 s, and one process per scan is the pipeline's contract (Server/scan_processor.py:138).  Now the FIRST fuser a process creates on a device starts that
// work on a thread of its own, beside its own allocations (4.3 GB of tiles to reserve and clear) and the caller's sf_sens_open; sf_fuse_run joins it.
// pinned_bytes = a guess of the ring's size (a run that needs more re-allocates, as before).
void sf_run_resources_prepare(int device, size_t pinned_bytes, size_t device_bytes, size_t plan_bytes) {
  sf_run_resources_prepare_ex(device, pinned_bytes, device_bytes, plan_bytes, 3, 0);
}

// The same with the number of side streams / copy streams the run will want, and GROWING a set that exists: a set prepared for depth-only runs (what
// sf_fuser_create asks for) is too small for a JPEG-colour scan -- 8 slots instead of 6, five side streams instead of three, two copy streams, 170 MB pinned
// and 2.7 GB of device memory instead of 87 / 205 MB -- and the first sf_fuse_run of such a scan paid for the difference inside its timed loop: 52 ms of set-up
// and ~110 ms of copy calls that blocked on fresh memory, of a scan that is fused in 0.3 s (profiles/r06_e2e_phase_clock.txt).  sf_fuse_run_prepare (the C ABI:
// the caller has the file open and knows) sizes the set from the file before the fuser is created; the work runs on a thread beside sf_fuser_create.
void sf_run_resources_prepare_ex(int device, size_t pinned_bytes, size_t device_bytes, size_t plan_bytes, int side_streams, int copy_streams) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  RunResources* r = nullptr;
  for (RunResources* q : g_res)
    if (q->device == device) {
      if (q->taken) return;   // a run is using the set
      r = q;
    }
  if (r) {
    if (r->prep.joinable()) r->prep.join();   // the preparation thread never takes g_res_mu
    int have_side = 0, have_copy = 0;
    for (hipStream_t x : r->inflate) have_side += x != nullptr;
    for (hipStream_t x : r->copy) have_copy += x != nullptr;
    if (r->h_bytes >= pinned_bytes && r->d_bytes >= device_bytes && (plan_bytes == 0 || r->plan_bytes >= plan_bytes) && have_side >= side_streams && have_copy >= copy_streams) return;
  } else {
    r = new RunResources;
    r->device = device;
    g_res.push_back(r);
  }
  side_streams = std::min(side_streams, 6);
  copy_streams = std::min(copy_streams, 2);
  try {
    r->prep = std::thread([r, device, pinned_bytes, device_bytes, plan_bytes, side_streams, copy_streams]() {
      if (hipSetDevice(device) != hipSuccess) return;
      if (side_streams > 0) inflate_gpu_warm();                       // the code objects of the kernels the side streams run
      if (side_streams > 3) { jpeg_gpu_warm(); jpeg_huff_gpu_warm(); }   // (five side streams: a JPEG-colour scan)
      for (int q = 0; q < side_streams; q++)
        if (!r->inflate[q] && create_side_stream(&r->inflate[q]) != hipSuccess) { r->inflate[q] = nullptr; break; }
      for (int q = 0; q < copy_streams; q++)
        if (!r->copy[q] && create_side_stream(&r->copy[q]) != hipSuccess) { r->copy[q] = nullptr; break; }
      if (pinned_bytes > r->h_bytes) {
        if (r->h_pool) { (void)hipHostFree(r->h_pool); r->h_pool = nullptr; r->h_bytes = 0; }
        if (hipHostMalloc((void**)&r->h_pool, pinned_bytes, hipHostMallocDefault) == hipSuccess) r->h_bytes = pinned_bytes;
        else r->h_pool = nullptr;
      }
      // the first DMA out of freshly page-locked memory pays for mapping it (measured: the first run's hipMemcpyAsync calls blocked 0.4 ms each, 37-48 ms
      // of a run): one pass of copies over the pool here, on this thread, pays it before the run
      bool fresh_device = false;
      if (device_bytes > r->d_bytes) {
        if (r->d_pool) { (void)hipFree(r->d_pool); r->d_pool = nullptr; r->d_bytes = 0; }
        if (hipMalloc((void**)&r->d_pool, device_bytes) == hipSuccess) { r->d_bytes = device_bytes; fresh_device = true; }
        else r->d_pool = nullptr;
      }
      if (r->h_pool && r->d_pool)   // one pass of copies over both pools: whatever the first transfer out of / into fresh memory pays is paid here
        for (size_t at = 0; at < r->h_bytes; at += r->d_bytes)
          if (hipMemcpy(r->d_pool, r->h_pool + at, std::min(r->d_bytes, r->h_bytes - at), hipMemcpyHostToDevice) != hipSuccess) break;
      if (r->d_pool && fresh_device) (void)hipMemset(r->d_pool, 0, r->d_bytes);
      if (plan_bytes != 0) {
        if (r->plan_bytes < plan_bytes) {   // scratch of another frame size: start over
          for (uint8_t*& q : r->d_plan) { if (q) (void)hipFree(q); q = nullptr; }
          r->plan_bytes = plan_bytes;
        }
        for (int q = 0; q < side_streams; q++) {
          if (r->d_plan[q]) continue;
          if (hipMalloc((void**)&r->d_plan[q], r->plan_bytes) != hipSuccess) { r->d_plan[q] = nullptr; break; }
          (void)hipMemset(r->d_plan[q], 0, r->plan_bytes);
        }
      }
    });
  } catch (...) {
    // no thread: the first run sets everything up itself, as before
  }
}

namespace {

// One ring slot = one batch of B frames: contiguous pinned host buffers, contiguous device buffers, two events.
struct BatchSlot {
  hipEvent_t copied = nullptr;    // H2D of this batch finished (its pinned buffers may be refilled)
  hipEvent_t copied_rgb = nullptr;  // the colour part of it, on the other copy stream
  hipEvent_t copied_rgb2 = nullptr; // ... its second half, when that has a stream of its own
  hipEvent_t inflated = nullptr;    // the frames that travelled compressed are pixels now (recorded on the inflate stream)
  // pre-pass of this batch finished (its device buffers may be overwritten): one event per input stream a sub-batch of the slot ran on --
  // the fuser orders its two streams among themselves, but the ring does not lean on that
  hipEvent_t consumed[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  std::atomic<int> decoded{0};    // frames of the current generation the pool has finished with
  std::atomic<int> failed{0};
  // per frame, what the pinned colour area holds: 0 = RGB; 1 = JPEG coefficients (the host entropy-decoded, the GPU reconstructs);
  // 2 = the entropy-coded segment, prepared (the GPU decodes AND reconstructs)
  uint8_t coef_mode[MAX_BATCH] = {0};
  uint32_t pay_used[MAX_BATCH] = {0};    // bytes of that payload
  bool packed_segs = false;   // every colour frame of the batch travelled as a prepared segment and the batch's segments went to the device as ONE piece (two halves):
                              // frame j's segment sits at d_rgb(slot, 0) + j * hcol_b -- the pinned stride -- instead of at the head of its own pixel area
};

}  // namespace

// Frames are handled in batches of B = sf_fuser_batch_frames(): batch g lives in ring slot g % NB.
//   decode pool : frame k is decoded as soon as batch (k / B) - NB has left its pinned buffers (counter `landed`, advanced by ONE
//                 thread that follows the copy events); the workers never enter the HIP runtime and share no lock -- progress
//                 counters are atomics polled with a short sleep (a mutex + condition variable woke 64 threads per frame and
//                 cost more than the decoding)
//   this thread : waits until a batch is fully decoded, queues ONE H2D copy per run of consecutive valid frames (up to
//                 B x 614 KB per call instead of B calls) on the copy stream, then the batch's kernels
SF_API int sf_fuse_run(sf_fuser* f, const sf_sens* s, uint64_t first, uint64_t last, int decode_threads, sf_run_stats* stats) {
  if (!f || !s) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  const uint64_t nframes = s->frames.size();
  if (last == 0 || last > nframes) last = nframes;
  if (first > last) return sf::fail(SF_ERR_BOUNDS, "first frame %llu beyond last %llu", (unsigned long long)first, (unsigned long long)last);
  if ((int)s->info.depth_width != f->in_W || (int)s->info.depth_height != f->in_H)
    return sf::fail(SF_ERR_INVALID_ARG, "fuser was created for %dx%d depth frames, the .sens file holds %ux%u", f->in_W, f->in_H,
                    s->info.depth_width, s->info.depth_height);
  SF_HIP_CHECK(hipSetDevice(f->device));
  const auto t_start = std::chrono::steady_clock::now();
  const bool timing = std::getenv("SF_RUN_TIMING") != nullptr;
  double t_wait_ready = 0, t_api = 0, t_flush = 0, t_launch_z = 0, t_memcpy = 0;
  auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const size_t npx = f->in_px;   // of an input frame (the pre-pass resamples to the integration size when the two differ)
  // colour is fused when its frames match what the fuser was created for: depth resolution, or the colour resolution
  // given in sf_params (raw or JPEG); anything else: geometry only
  const bool same_res = s->info.color_width == s->info.depth_width && s->info.color_height == s->info.depth_height;
  const bool own_res = f->pk.cW > 0 && (int)s->info.color_width == f->pk.cW && (int)s->info.color_height == f->pk.cH;
  const bool use_rgb = ((same_res && f->pk.cW == 0) || own_res) && (s->info.color_compression >= 0 && s->info.color_compression <= 2);   // raw, PNG (host decode), JPEG
  const size_t cpx = f->pk.cW > 0 ? (size_t)f->pk.cW * f->pk.cH : npx;
  // default pool size: inflating a depth frame takes ~0.13 ms, so 32 threads outrun the GPU (measured: 16 threads 28 k frames/s,
  // 64 threads 26 k); baseline-JPEG colour costs milliseconds per frame and takes up to 64 (128 measured slower: 5.0 k vs 8.1 k frames/s)
  const bool jpeg_colour = use_rgb && s->info.color_compression == 2;
  // zlib depth (the reference's writer: one final fixed-Huffman block per frame) is inflated on the GPU: a host thread only copies the compressed
  // frame into the pinned ring; streams the device does not take (dynamic / stored / several blocks, longer than the pixels) are inflated by
  // the host threads as before.  SF_INFLATE_HOST=1: always inflate on the host.
  const bool gpu_inflate = s->info.depth_compression == 1 && (npx * 2) % 4 == 0 && std::getenv("SF_INFLATE_HOST") == nullptr;
  const int hw = sf::usable_cpus();
  int nthreads = decode_threads > 0 ? decode_threads : std::min(hw, jpeg_colour ? 64 : 32);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  const uint64_t total = last - first;
  const int B = f->batch;  // frames fused per pass over the voxel tiles
  const uint64_t nbatches = (total + (uint64_t)B - 1) / (uint64_t)B;
  // enough batch slots for every decode thread to be busy while two batches sit between copy and pre-pass
  // 3 slots (one decoding, one in flight, one being read by the pre-pass) are enough: 4, 6 and 10 measured no faster (tools/gpu/h2d_bw.hip:
  // the link moves 57 GB/s from pinned memory on two streams; a colour run is bound by the fusion kernels and ~15 ms of set-up)
  // side streams: batch g is inflated (and its JPEG pictures entropy-decoded and reconstructed) on stream g % NZ, beside the fusion of the batches before
  // it.  A batch takes ~1.8 ms to inflate and 0.8 ms to fuse: three in flight; with JPEG colour the side work is ~4.9 ms per batch (k_jpeg_huff 2.4 ms
  // per 32 pictures of 200 KB on 32 CUs): five (profiles/r05_timeline_e2e_rgbd.txt)
  constexpr int MAX_NZ = 6;
  const int NZ = jpeg_colour ? 5 : 3;
  const int NB = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(gpu_inflate ? 3 + NZ : 3, ((uint64_t)nthreads + B - 1) / B + 2), std::max<uint64_t>(nbatches, 1)));
  std::vector<BatchSlot> ring((size_t)NB);
  // two streams = two SDMA engines: one alone moves ~20 GB/s.  With the GPU inflate the batch's copies ride on its inflate stream (and the next
  // one): copy, tokens, copies-kernel of batch g, then the copy of batch g + 3 -- a stream costs ~5 ms to create
  hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;
  hipStream_t inflate_stream[MAX_NZ] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t* d_plan[MAX_NZ] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // scratch of the inflate kernels (one u16 per output byte), one per stream
  int32_t* d_zstatus = nullptr;                                // 2 ints per ring slot and frame, written by the device's inflate only when a frame fails
  // ONE pinned host allocation and ONE device allocation for the whole ring
  uint8_t* h_pool = nullptr;
  uint8_t* d_pool = nullptr;
  RunResources* res = acquire_resources(f->device);   // nullptr: another run of this process holds the device's set
  const size_t depth_b = npx * 2, rgb_b = use_rgb ? cpx * 3 : 0;
  // JPEG colour: the host threads only entropy-decode; the coefficients travel in place of the pixels and the GPU reconstructs
  // (jpeg_gpu.hip).  The payload is sized from the first colour frame's layout (a scan's frames share it); a frame that does not fit,
  // or has a layout the GPU path does not take, is decoded on the host as before.  SF_JPEG_HOST=1: always decode on the host.
  // SF_JPEG_GPU_HUFFMAN=1: a host thread only parses the headers and strips the byte stuffing (jpeg_prepare_huff), the entropy-coded
  // segment travels and the GPU entropy-decodes too (jpeg_huff_gpu.hip; frames with restart intervals stay with the host threads).  Same
  // bytes; measured SLOWER with 16 host threads (6.1 k against 7.5 k frames/s at 1296x968: 1.25 ms per 16 pictures on 16 CUs) -- it is
  // for hosts with few cores, and opt-in until the kernel is spread over more CUs.
  size_t pay_b = 0, planes_b = 0;
  uint32_t pay_blocks = 0;
  if (jpeg_colour && std::getenv("SF_JPEG_HOST") == nullptr) {
    for (uint64_t k = first; k < last && pay_b == 0; k++) {
      const SensFrame& fr = s->frames[k];
      if (fr.pose[0] == -INFINITY || fr.color_bytes == 0) continue;
      const uint32_t cw_ = s->info.color_width, ch_ = s->info.color_height;
      const uint64_t padded = (uint64_t)((cw_ + 15) & ~15u) * ((ch_ + 15) & ~15u);
      std::vector<uint32_t> probe((sizeof(SfJpegLayout) + padded * 3 / 64 * 4 + padded * 3 * 4) / 4 + 64);
      if (jpeg_decode_coef(fr.color, fr.color_bytes, cw_, ch_, reinterpret_cast<uint8_t*>(probe.data()), probe.size() * 4) != SF_OK) break;
      const SfJpegLayout* L = reinterpret_cast<const SfJpegLayout*>(probe.data());
      pay_blocks = L->nblocks;
      // room for the table and as many entries as the pixels have bytes: a frame with more non-zero coefficients than that (finer than
      // anything a camera compresses to) is decoded on the host
      pay_b = (sizeof(SfJpegLayout) + 4 * (size_t)L->nblocks + rgb_b + 255) & ~(size_t)255;
      planes_b = (sf_jpeg_plane_bytes(*L) + 255) & ~(size_t)255;
    }
  }
  const bool gpu_jpeg = pay_b != 0;
  // SF_JPEG_RGB_IMAGE=1: the device writes every picture out as RGB (k_jpeg_rgb) and the pre-pass picks its pixels from that, as until round 5 (A/B measurements)
  const bool ycc_ok = gpu_jpeg && std::getenv("SF_JPEG_RGB_IMAGE") == nullptr;
  // where the entropy decoding runs: on the device when the batch has a side stream for it (the depth inflate's), else on the host threads;
  // SF_JPEG_GPU_HUFFMAN=1 / SF_JPEG_HOST_HUFFMAN=1 force one or the other
  const bool gpu_huffman = gpu_jpeg && std::getenv("SF_JPEG_HOST_HUFFMAN") == nullptr && (std::getenv("SF_JPEG_GPU_HUFFMAN") != nullptr || SF_JPEG_DEVICE_HUFFMAN_DEFAULT(gpu_inflate));
  const uint32_t pay_entries = gpu_jpeg ? (uint32_t)((pay_b - sizeof(SfJpegLayout) - 4 * (size_t)pay_blocks) / 4) : 0u;
  int32_t* d_jstatus = nullptr;   // 2 ints per ring slot and frame, written by the device's entropy decoder only when a picture fails
  // GPU inflate: the depth part of a pinned slot is PACKED -- per frame either the zlib stream from its third byte on (the device inflates it)
  // or, for a stream the device does not take, the pixels a host thread decoded; 64-byte aligned segments whose offsets are known before
  // anybody decodes (the sizes are in the file's frame table), so that the batch crosses PCIe in ONE copy (32 copies of ~330 KB cost the
  // enqueueing thread 0.6 ms per batch).  Otherwise: depth_b per frame.
  const size_t seg_px = (depth_b + 63) & ~(size_t)63;
  std::vector<uint8_t> zmode(gpu_inflate ? total : 0);      // per frame of the run: 1 = travels compressed
  std::vector<uint32_t> zoff(gpu_inflate ? total : 0);      // its segment's offset in the slot
  std::vector<uint32_t> zbytes(gpu_inflate ? nbatches : 0); // per batch: bytes to copy
  if (gpu_inflate)
    for (uint64_t g = 0; g < nbatches; g++) {
      size_t at = 0;
      for (uint64_t k = g * (uint64_t)B; k < std::min<uint64_t>(total, (g + 1) * (uint64_t)B); k++) {
        const SensFrame& fd = s->frames[first + k];
        zoff[k] = (uint32_t)at;
        if (fd.pose[0] == -INFINITY) continue;
        zmode[k] = fd.depth && fd.depth_bytes - 2 <= depth_b && inflate_gpu_takes(fd.depth, fd.depth_bytes) ? 1 : 0;
        at += zmode[k] ? ((size_t)fd.depth_bytes - 2 + 63) & ~(size_t)63 : seg_px;
      }
      zbytes[g] = (uint32_t)at;
    }
  size_t packed_max = 0;   // the largest batch's packed depth part: what a slot must hold (compressed frames: about half of the pixels)
  for (uint32_t z : zbytes) packed_max = std::max<size_t>(packed_max, z);
  const size_t slot_depth = ((gpu_inflate ? packed_max : depth_b * B) + 255) & ~(size_t)255, slot_planes = planes_b * B;
  const size_t dslot_depth = gpu_inflate ? (depth_b * B + 255) & ~(size_t)255 : slot_depth;   // on the device: the frames as pixels (where the inflate kernels write)
  // pinned slot: depth, then per frame ONE colour area that holds either pixels or coefficients (col_b = the larger of the two);
  // device slot: depth, pixels, coefficients, planes scratch
  const size_t col_b = std::max(rgb_b, pay_b), slot_col = (col_b * B + 255) & ~(size_t)255;
  // With the device's entropy decoder a picture travels as its prepared segment -- never more than its blob in the file plus the tables -- so the PINNED
  // slot holds that per frame instead of room for a decoded picture (1296x968: 0.26 MB instead of 3.8 MB per frame, 110 MB of page-locked memory for a run
  // instead of 833 MB: 45 ms of the first run of a process).  A picture the device does not take (restart intervals, ...) is decoded by its host thread
  // into a pageable buffer of its own and copied from there (coef_mode 3).
  size_t max_color_bytes = 0;
  if (gpu_huffman)
    for (uint64_t k = first; k < last; k++) max_color_bytes = std::max<size_t>(max_color_bytes, (size_t)s->frames[k].color_bytes);
  const bool small_col = gpu_huffman;
  const size_t hcol_b = small_col ? (max_color_bytes + sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) + 64 + 255) & ~(size_t)255 : col_b;
  const size_t hslot_col = (hcol_b * B + 255) & ~(size_t)255;
  const size_t slot_b = slot_depth + hslot_col;
  // the packed depth part on the device, with 256 readable bytes behind it (the lanes of k_inflate_tokens fetch 64 bytes at a time, two fetches ahead)
  const size_t slot_comp = gpu_inflate ? slot_depth + 256 : 0;
  const size_t dslot_b = dslot_depth + slot_col + (gpu_jpeg ? slot_col : 0) + slot_planes + slot_comp;   // every colour area strides by col_b: runs copy as one piece
  auto h_depth = [&](int sl, int j) { return (uint16_t*)(h_pool + (size_t)sl * slot_b + (size_t)j * depth_b); };
  auto d_depth = [&](int sl, int j) { return d_pool + (size_t)sl * dslot_b + (size_t)j * depth_b; };
  auto h_rgb = [&](int sl, int j) { return h_pool + (size_t)sl * slot_b + slot_depth + (size_t)j * hcol_b; };
  auto d_rgb = [&](int sl, int j) { return d_pool + (size_t)sl * dslot_b + dslot_depth + (size_t)j * col_b; };
  auto h_pay = h_rgb;
  auto d_pay = [&](int sl, int j) { return d_pool + (size_t)sl * dslot_b + dslot_depth + slot_col + (size_t)j * col_b; };
  auto d_planes = [&](int sl, int j) { return d_pool + (size_t)sl * dslot_b + dslot_depth + 2 * slot_col + (size_t)j * planes_b; };
  auto h_stage = [&](int sl) { return h_pool + (size_t)sl * slot_b; };                             // the packed depth part (GPU inflate)
  auto d_stage = [&](int sl) { return d_pool + (size_t)(sl + 1) * dslot_b - slot_comp; };            // ... on the device, behind everything else of the slot
  auto cleanup = [&]() {
    for (BatchSlot& sl : ring) {
      if (sl.copied) (void)hipEventDestroy(sl.copied);
      if (sl.copied_rgb) (void)hipEventDestroy(sl.copied_rgb);
      if (sl.copied_rgb2) (void)hipEventDestroy(sl.copied_rgb2);
      if (sl.inflated) (void)hipEventDestroy(sl.inflated);
      for (hipEvent_t ev : sl.consumed) if (ev) (void)hipEventDestroy(ev);
    }
    if (h_pool && !(res && res->h_pool == h_pool)) (void)hipHostFree(h_pool);
    if (d_pool && !(res && res->d_pool == d_pool)) (void)hipFree(d_pool);
    if (d_jstatus) (void)hipFree(d_jstatus);
    for (int q = 0; q < MAX_NZ; q++) if (d_plan[q] && !(res && res->d_plan[q] == d_plan[q])) (void)hipFree(d_plan[q]);
    if (d_zstatus) (void)hipFree(d_zstatus);
    if (!res) for (hipStream_t q : inflate_stream) if (q) (void)hipStreamDestroy(q);
    if (!res && copy_stream) (void)hipStreamDestroy(copy_stream);
    if (!res && copy_stream2) (void)hipStreamDestroy(copy_stream2);
    release_resources(res);
  };
  // Set-up.  Streams and the pinned pool come from the process-wide set when it is free (the first run on a device creates them: 23 ms for a
  // depth-only run, 55 ms with colour, of a scan that is fused in 0.25 s; creating the streams on threads of their own did not help, the
  // runtime makes its hardware queues one after the other).
  const double ts0 = timing ? now_s() : 0;
  {
    hipError_t e_ = hipSuccess;
    auto want_stream = [&](hipStream_t* cached, hipStream_t* out) {
      if (e_ != hipSuccess) return;
      if (cached && *cached) { *out = *cached; return; }
      e_ = create_side_stream(out);
      if (e_ == hipSuccess && cached) *cached = *out;
    };
    if (!gpu_inflate || use_rgb) {   // with the GPU inflate: for the colour part only (behind the inflate kernels of an earlier batch it arrived late)
      want_stream(res ? &res->copy[0] : nullptr, &copy_stream);
      want_stream(res ? &res->copy[1] : nullptr, &copy_stream2);
    }
    if (gpu_inflate)
      for (int q = 0; q < NZ; q++) want_stream(res ? &res->inflate[q] : nullptr, &inflate_stream[q]);
    const size_t h_need = (size_t)NB * slot_b;
    if (e_ == hipSuccess && res && res->h_bytes >= h_need) {
      h_pool = res->h_pool;
    } else if (e_ == hipSuccess) {
      if (res && res->h_pool) { (void)hipHostFree(res->h_pool); res->h_pool = nullptr; res->h_bytes = 0; }
      e_ = hipHostMalloc((void**)&h_pool, h_need, hipHostMallocDefault);
      if (e_ == hipSuccess && res) { res->h_pool = h_pool; res->h_bytes = h_need; }
    }
    const size_t d_need = (size_t)NB * dslot_b, plan_need = 2 * depth_b * (size_t)B;
    if (e_ == hipSuccess && res && res->d_bytes >= d_need) {
      d_pool = res->d_pool;
    } else if (e_ == hipSuccess) {
      if (res && res->d_pool) { (void)hipFree(res->d_pool); res->d_pool = nullptr; res->d_bytes = 0; }
      e_ = hipMalloc((void**)&d_pool, d_need);
      if (e_ == hipSuccess && res) { res->d_pool = d_pool; res->d_bytes = d_need; }
    }
    if (gpu_inflate) {
      if (res && res->plan_bytes < plan_need) {   // scratch of another frame size: start over
        for (uint8_t*& q : res->d_plan) { if (q) (void)hipFree(q); q = nullptr; }
        res->plan_bytes = plan_need;
      }
      // every cached entry holds res->plan_bytes (>= this run's need): an entry allocated NOW must have that size too, or a later run whose need
      // lies between the two would reuse it undersized (device out-of-bounds write of k_inflate_*)
      const size_t plan_alloc = res ? std::max(plan_need, res->plan_bytes) : plan_need;
      for (int q = 0; q < NZ && e_ == hipSuccess; q++) {
        if (res && res->d_plan[q]) { d_plan[q] = res->d_plan[q]; continue; }
        e_ = hipMalloc((void**)&d_plan[q], plan_alloc);
        if (e_ == hipSuccess && res) res->d_plan[q] = d_plan[q];
      }
      if (e_ == hipSuccess) e_ = hipMalloc((void**)&d_zstatus, (size_t)NB * B * 8);
      if (e_ == hipSuccess) e_ = hipMemset(d_zstatus, 0, (size_t)NB * B * 8);
    }
    if (gpu_huffman) {
      if (e_ == hipSuccess) e_ = hipMalloc((void**)&d_jstatus, (size_t)NB * B * 8);
      if (e_ == hipSuccess) e_ = hipMemset(d_jstatus, 0, (size_t)NB * B * 8);
    }
    for (BatchSlot& sl : ring) {
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming);
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.copied_rgb, hipEventDisableTiming);
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.copied_rgb2, hipEventDisableTiming);
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.inflated, hipEventDisableTiming);
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.consumed[0], hipEventDisableTiming);
      if (e_ == hipSuccess) e_ = hipEventCreateWithFlags(&sl.consumed[1], hipEventDisableTiming);
    }
    if (e_ != hipSuccess) { cleanup(); return sf::fail(SF_ERR_DEVICE, "sf_fuse_run set-up (streams, pinned and device pools) failed: %s", hipGetErrorString(e_)); }
  }

  const double t_setup_end = timing ? now_s() : 0;
  if (timing)
    std::fprintf(stderr, "sf_fuse_run set-up: frame table + layout %.1f ms; streams, pinned pool, device pool, events %.1f ms\n",
                 (ts0 - std::chrono::duration<double>(t_start.time_since_epoch()).count()) * 1e3, (t_setup_end - ts0) * 1e3);
  std::vector<std::vector<uint8_t>> fallback_rgb(small_col ? (size_t)NB * B : 0);   // coef_mode 3: pixels a host thread decoded, pageable, per slot and frame
  std::atomic<uint64_t> next{0}, landed{0}, issued{0};  // frame counter of the pool; batches whose copies completed / were queued
  std::atomic<bool> abort{false};
  std::atomic<uint64_t> decode_ns{0};
  std::mutex err_mu;
  std::string pool_err;
  int pool_rc = SF_OK;
  auto nap = [] { std::this_thread::sleep_for(std::chrono::microseconds(20)); };
  auto worker = [&]() {
    for (;;) {
      const uint64_t k = next.fetch_add(1);
      if (k >= total || abort.load(std::memory_order_relaxed)) return;
      const uint64_t g = k / (uint64_t)B;
      const int j = (int)(k % (uint64_t)B), sl = (int)(g % (uint64_t)NB);
      while (g >= landed.load(std::memory_order_acquire) + (uint64_t)NB) {  // batch g - NB still owns the pinned buffers
        if (abort.load(std::memory_order_relaxed)) return;
        nap();
      }
      const uint64_t frame = first + k;
      const auto t0 = std::chrono::steady_clock::now();
      int rc = SF_OK;
      if (s->frames[frame].pose[0] != -INFINITY) {
        const SensFrame& fd = s->frames[frame];
        if (gpu_inflate && zmode[k]) {
          uint8_t* dst = h_stage(sl) + zoff[k];
          const size_t nb = (size_t)fd.depth_bytes - 2;
          std::memcpy(dst, fd.depth + 2, nb);
          for (size_t q = nb; q & 63; q++) dst[q] = 0;   // the device reads whole words; the segment is whole 64 bytes
        } else if (gpu_inflate) {
          rc = sens_decode_depth(s, frame, reinterpret_cast<uint16_t*>(h_stage(sl) + zoff[k]));
        } else {
          rc = sens_decode_depth(s, frame, h_depth(sl, j));
        }
        if (rc == SF_OK && use_rgb && s->frames[frame].color_bytes) {
          int coef = 0;
          if (gpu_jpeg) {
            uint8_t* pay = h_pay(sl, j);
            const SensFrame& fr = s->frames[frame];
            if (gpu_huffman && jpeg_prepare_huff(fr.color, fr.color_bytes, s->info.color_width, s->info.color_height, pay, std::min(hcol_b, col_b)) == SF_OK &&   // the DEVICE area strides by col_b: a longer entropy segment takes the host path
                reinterpret_cast<const SfJpegLayout*>(pay)->nblocks == pay_blocks) {
              coef = 2;
              ring[(size_t)sl].pay_used[j] = (uint32_t)(sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) +
                                                        4 * (size_t)reinterpret_cast<const SfJpegHuffDesc*>(pay + sizeof(SfJpegLayout))->ecs_words);
            } else if (small_col) {
              // not a picture for the device: the host decoder's pixels, in a buffer of this frame's own (the pinned slot has no room for them)
              try {
                std::vector<uint8_t>& fb = fallback_rgb[(size_t)sl * B + (size_t)j];
                fb.resize(rgb_b);
                rc = sf_sens_decode_color(s, frame, fb.data());
                coef = 3;
              } catch (...) { rc = sf::fail(SF_ERR_IO, "out of memory decoding colour frame %llu", (unsigned long long)frame); coef = 3; }
            } else if (jpeg_decode_coef(fr.color, fr.color_bytes, s->info.color_width, s->info.color_height, pay, pay_b) == SF_OK &&
                       reinterpret_cast<const SfJpegLayout*>(pay)->nblocks == pay_blocks) {
              coef = 1;
              ring[(size_t)sl].pay_used[j] = (uint32_t)sf_jpeg_payload_bytes(*reinterpret_cast<const SfJpegLayout*>(pay));
            }
          }
          ring[(size_t)sl].coef_mode[j] = (uint8_t)coef;
          if (!coef) rc = sf_sens_decode_color(s, frame, h_rgb(sl, j));   // raw colour, or a JPEG the GPU path does not take (errors surface here)
        }
      }
      decode_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
      if (rc != SF_OK) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (pool_rc == SF_OK) { pool_rc = rc; pool_err = sf_last_error(); }
        ring[(size_t)sl].failed.fetch_add(1);
      }
      ring[(size_t)sl].decoded.fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
  std::thread reaper([&]() {  // the only other thread inside the HIP runtime: copies complete in order on the copy stream
    (void)hipSetDevice(f->device);
    for (uint64_t g = 0; g < nbatches; g++) {
      while (issued.load(std::memory_order_acquire) <= g) {
        if (abort.load(std::memory_order_relaxed)) return;
        nap();
      }
      (void)hipEventSynchronize(ring[(size_t)(g % (uint64_t)NB)].copied);
      landed.store(g + 1, std::memory_order_release);
    }
  });

  int result = SF_OK;
  std::string err;
  uint64_t n_int = 0, n_skip = 0;
  uint64_t n_dev_z = 0, n_host_z = 0, n_dev_j = 0, n_host_j = 0;
  for (uint64_t g = 0; g < nbatches && result == SF_OK; g++) {
    const int sl = (int)(g % (uint64_t)NB);
    BatchSlot& bs = ring[(size_t)sl];
    const int cnt = (int)std::min<uint64_t>((uint64_t)B, total - g * (uint64_t)B);
    {
      const double t0 = timing ? now_s() : 0;
      while (bs.decoded.load(std::memory_order_acquire) < cnt) nap();
      if (timing) t_wait_ready += now_s() - t0;
    }
    if (bs.failed.load() != 0) {
      std::lock_guard<std::mutex> lk(err_mu);
      result = pool_rc; err = pool_err;
      break;
    }
    bs.decoded.store(0, std::memory_order_relaxed);  // next generation of this slot starts only after `landed` passes g
    // ---- copies: one per run of consecutive valid frames
    const double t1 = timing ? now_s() : 0;
    hipError_t e = hipSuccess;
    // depth on one copy stream, colour on the other, batches alternating between them: two transfers are in flight at any time
    hipStream_t cs_depth = (g & 1) ? copy_stream2 : copy_stream, cs_rgb = (g & 1) ? copy_stream : copy_stream2;
    hipStream_t cs_rgb2 = cs_depth;   // the colour part is the larger one: its second half follows the depth on the other stream
    if (gpu_inflate) {                // ... or: depth in front of its stream's inflate kernels, the colour halves on the two copy streams
      cs_depth = inflate_stream[g % NZ];
      cs_rgb = use_rgb ? copy_stream : cs_depth;
      cs_rgb2 = use_rgb ? copy_stream2 : cs_depth;
    }
    for (int q = 0; q < 2 && e == hipSuccess; q++)
      if (bs.used[q]) {  // device buffers still read by this slot's previous pre-pass?
        e = hipStreamWaitEvent(cs_depth, bs.consumed[q], 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs_rgb, bs.consumed[q], 0);
        if (e == hipSuccess && cs_rgb2 != cs_depth && cs_rgb2 != cs_rgb) e = hipStreamWaitEvent(cs_rgb2, bs.consumed[q], 0);
      }
    bool valid[MAX_BATCH], rgbf[MAX_BATCH];
    for (int j = 0; j < cnt; j++) {
      const uint64_t frame = first + g * (uint64_t)B + (uint64_t)j;
      valid[j] = s->frames[frame].pose[0] != -INFINITY;
      rgbf[j] = valid[j] && use_rgb && s->frames[frame].color_bytes != 0;
      if (!valid[j]) { n_skip++; f->frames_skipped++; }
      else if (s->info.depth_compression == 1) { if (gpu_inflate && zmode[g * (uint64_t)B + (uint64_t)j]) n_dev_z++; else n_host_z++; }
      if (rgbf[j] && jpeg_colour) { if (bs.coef_mode[j] == 2) n_dev_j++; else n_host_j++; }
    }
    bool any_comp = false;
    const uint64_t k0 = g * (uint64_t)B;   // index of the batch's first frame in the run
    if (gpu_inflate) {   // the packed depth part in one piece
      const double tm = timing ? now_s() : 0;
      if (zbytes[g]) e = hipMemcpyAsync(d_stage(sl), h_stage(sl), zbytes[g], hipMemcpyHostToDevice, cs_depth);
      if (timing) t_memcpy += now_s() - tm;
      for (int j = 0; j < cnt; j++) any_comp = any_comp || (valid[j] && zmode[k0 + (uint64_t)j]);
    } else {
      for (int j = 0; j < cnt && e == hipSuccess;) {   // one copy per run of consecutive valid frames
        if (!valid[j]) { j++; continue; }
        int j1 = j;
        while (j1 < cnt && valid[j1]) j1++;
        e = hipMemcpyAsync(d_depth(sl, j), h_depth(sl, j), (size_t)(j1 - j) * depth_b, hipMemcpyHostToDevice, cs_depth);
        j = j1;
      }
    }
    bool any_rgb = false;
    for (int j = 0; j < cnt && e == hipSuccess;) {   // pixels: runs of frames decoded on the host
      if (!rgbf[j] || bs.coef_mode[j]) { j++; continue; }
      int j1 = j;
      while (j1 < cnt && rgbf[j1] && !bs.coef_mode[j1]) j1++;
      const int jm = j + (j1 - j + 1) / 2;
      e = hipMemcpyAsync(d_rgb(sl, j), h_rgb(sl, j), (size_t)(jm - j) * col_b, hipMemcpyHostToDevice, cs_rgb);
      if (e == hipSuccess && j1 > jm) e = hipMemcpyAsync(d_rgb(sl, jm), h_rgb(sl, jm), (size_t)(j1 - jm) * col_b, hipMemcpyHostToDevice, cs_rgb2);
      any_rgb = true;
      j = j1;
    }
    for (int j = 0; j < cnt && e == hipSuccess; j++) {   // pictures the host decoded into pageable buffers (the call returns once the bytes are staged)
      if (!rgbf[j] || bs.coef_mode[j] != 3) continue;
      e = hipMemcpyAsync(d_rgb(sl, j), fallback_rgb[(size_t)sl * B + (size_t)j].data(), rgb_b, hipMemcpyHostToDevice, cs_rgb);
      any_rgb = true;
    }
    // Every frame a prepared segment (the usual batch of a JPEG-colour scan): the pinned colour areas are one contiguous piece (stride hcol_b), and so they
    // travel -- two copies per batch, one per copy stream, instead of one ~200 KB copy per frame (32 calls of the runtime per batch: 0.4-0.5 ms of this
    // thread, the largest item of the loop).  They land packed at the head of the slot's pixel region; k_jpeg_huff of the WHOLE batch has read them before
    // the first k_jpeg_rgb writes a pixel there (same stream, in order).
    bs.packed_segs = false;
    if (small_col && cnt > 1 && (size_t)cnt * hcol_b <= slot_col) {
      bool all2 = true;
      for (int j = 0; j < cnt; j++) all2 = all2 && valid[j] && rgbf[j] && bs.coef_mode[j] == 2;
      if (all2 && e == hipSuccess) {
        const int jm = cnt / 2;
        const size_t tail = (size_t)(cnt - 1 - jm) * hcol_b + bs.pay_used[cnt - 1];   // the last frame's area only as far as it is used
        e = hipMemcpyAsync(d_rgb(sl, 0), h_pay(sl, 0), (size_t)jm * hcol_b, hipMemcpyHostToDevice, cs_rgb);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rgb(sl, 0) + (size_t)jm * hcol_b, h_pay(sl, jm), tail, hipMemcpyHostToDevice, cs_rgb2);
        bs.packed_segs = true;
        any_rgb = true;
      }
    }
    for (int j = 0, k = 0; j < cnt && e == hipSuccess && !bs.packed_segs; j++) {   // coefficients / entropy-coded segments: what each frame really holds, alternating streams
      if (!rgbf[j] || !bs.coef_mode[j] || bs.coef_mode[j] == 3) continue;
      // a prepared segment lands where the pixels will be written: it is dead once k_jpeg_huff has turned it into the coefficient payload
      e = hipMemcpyAsync(bs.coef_mode[j] == 2 ? d_rgb(sl, j) : d_pay(sl, j), h_pay(sl, j), bs.pay_used[j], hipMemcpyHostToDevice, (k++ & 1) ? cs_rgb2 : cs_rgb);
      any_rgb = true;
    }
    if (e == hipSuccess && any_rgb) {   // `copied` on the depth stream stands for both parts
      e = hipEventRecord(bs.copied_rgb, cs_rgb);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs_depth, bs.copied_rgb, 0);
      if (e == hipSuccess && cs_rgb2 != cs_depth && cs_rgb2 != cs_rgb) {
        e = hipEventRecord(bs.copied_rgb2, cs_rgb2);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs_depth, bs.copied_rgb2, 0);
      }
    }
    if (e == hipSuccess) e = hipEventRecord(bs.copied, cs_depth);
    if (e != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("copy pipeline: ") + hipGetErrorString(e); break; }
    issued.store(g + 1, std::memory_order_release);
    if (any_comp) {   // inflate behind the batch's copy: 1024 lanes per frame tokenise, a 256-lane workgroup per frame makes the copies (inflate_gpu.hip)
      hipStream_t zs = inflate_stream[g % NZ];
      uint8_t* zplan = d_plan[g % NZ];
      e = hipStreamWaitEvent(zs, bs.copied, 0);
      const uint32_t* zw[32];
      uint32_t zn[32];
      uint8_t* zo[32];
      uint16_t* zb[32];
      int32_t zt[32];
      int nz = 0, slot0 = 0;
      auto flush = [&]() {
        if (nz == 0 || e != hipSuccess || result != SF_OK) return;
        const double tz = timing ? now_s() : 0;
        const int rcz = inflate_gpu_batch(zs, nz, zw, zn, zo, zb, (uint32_t)depth_b, zt, d_zstatus + 2 * ((size_t)sl * B + (size_t)slot0));
        if (timing) t_launch_z += now_s() - tz;
        if (rcz != SF_OK) { result = rcz; err = sf_last_error(); }
        nz = 0;
      };
      for (int j = 0; j < cnt; j++) {
        if (!valid[j] || !zmode[k0 + (uint64_t)j]) continue;
        if (nz == 0) slot0 = j;
        zw[nz] = reinterpret_cast<const uint32_t*>(d_stage(sl) + zoff[k0 + (uint64_t)j]);
        zn[nz] = (uint32_t)(s->frames[first + k0 + (uint64_t)j].depth_bytes - 2); zo[nz] = d_depth(sl, j);
        zb[nz] = reinterpret_cast<uint16_t*>(zplan + 2 * depth_b * (size_t)j); zt[nz] = (int32_t)(first + g * (uint64_t)B + (uint64_t)j);
        if (++nz == 32) flush();
      }
      flush();
      if (e != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("inflate pipeline: ") + hipGetErrorString(e); }
      if (result != SF_OK) break;
    }
    // JPEG colour on the batch's side stream too (when there is one): entropy decoding of the pictures that travelled as segments and the
    // reconstruction of every picture that travelled as coefficients or segments -- beside the fusion of the batches before, three batches in flight,
    // instead of in front of this batch's pre-pass on the fuser's input stream (where a 32-picture batch cost 2 x 1.25 ms of a stream that also
    // carries allocation and compaction)
    bool side_jpeg = false, ycc_batch = false;
    if (gpu_jpeg && gpu_inflate && any_rgb) {
      hipStream_t zs = inflate_stream[g % NZ];
      if (!any_comp) e = hipStreamWaitEvent(zs, bs.copied, 0);
      const uint8_t* pp_[MAX_BATCH];
      uint8_t* rr_[MAX_BATCH];
      uint8_t* pl_[MAX_BATCH];
      int nj = 0;
      const uint8_t* seg[32];
      uint8_t* out[32];
      uint32_t cap[32];
      int32_t tag[32];
      int nh = 0, slot0 = 0;
      auto flush_h = [&]() {
        if (nh == 0 || result != SF_OK) return;
        const int rch = jpeg_gpu_huffman(zs, nh, seg, out, cap, tag, d_jstatus + 2 * ((size_t)sl * B + (size_t)slot0));
        if (rch != SF_OK) { result = rch; err = sf_last_error(); }
        nh = 0;
      };
      for (int q = 0; q < cnt; q++) {
        if (!(rgbf[q] && (bs.coef_mode[q] == 1 || bs.coef_mode[q] == 2))) continue;
        pp_[nj] = d_pay(sl, q); rr_[nj] = d_rgb(sl, q); pl_[nj] = d_planes(sl, q); nj++;
        if (bs.coef_mode[q] != 2) continue;
        if (nh == 0) slot0 = q;
        seg[nh] = bs.packed_segs ? d_rgb(sl, 0) + (size_t)q * hcol_b : d_rgb(sl, q);
        out[nh] = d_pay(sl, q); cap[nh] = pay_entries; tag[nh] = (int32_t)(first + g * (uint64_t)B + (uint64_t)q);
        if (++nh == 32) flush_h();
      }
      flush_h();
      // every colour frame of the batch a picture the device reconstructs: the component planes are all the pre-pass needs (it converts the one pixel per
      // depth pixel it looks up: k_prepass, YccPicture) -- no k_jpeg_rgb, no 3.8 MB RGB image per picture written and read back
      int n_rgbf = 0;
      for (int q = 0; q < cnt; q++) n_rgbf += rgbf[q] ? 1 : 0;
      ycc_batch = ycc_ok && nj > 0 && nj == n_rgbf;
      for (int q0 = 0; q0 < nj && result == SF_OK; q0 += 16) {   // jpeg_gpu.hip reconstructs at most 16 frames per launch
        const int rcj = ycc_batch ? jpeg_gpu_planes(zs, std::min(16, nj - q0), pp_ + q0, pl_ + q0, pay_blocks)
                                  : jpeg_gpu_reconstruct(zs, std::min(16, nj - q0), pp_ + q0, rr_ + q0, pl_ + q0, pay_blocks, s->info.color_width, s->info.color_height);
        if (rcj != SF_OK) { result = rcj; err = sf_last_error(); }
      }
      if (result != SF_OK) break;
      side_jpeg = nj > 0;
    }
    if ((any_comp || side_jpeg) && e == hipSuccess) e = hipEventRecord(bs.inflated, inflate_stream[g % NZ]);
    if (e != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("inflate / jpeg pipeline: ") + hipGetErrorString(e); break; }
    if (timing) t_api += now_s() - t1;
    // ---- kernels: the valid frames in order, a sub-batch is all-colour or all-geometry
    const double t2 = timing ? now_s() : 0;
    hipStream_t used_streams[2] = {nullptr, nullptr};   // the input streams this slot's sub-batches ran on
    for (int j = 0; j < cnt && result == SF_OK;) {
      if (!valid[j]) { j++; continue; }
      const void* dd[MAX_BATCH];
      const void* dr[MAX_BATCH];
      const void* dl[MAX_BATCH];
      const float* pp[MAX_BATCH];
      int m = 0;
      const bool rgb = rgbf[j];
      const int jfirst = j;
      while (j < cnt && m < B && (!valid[j] || rgbf[j] == rgb)) {
        if (valid[j]) {
          // pixels: inflated on the device into the slot's frame area, or (a stream the device does not take) as the host thread decoded them
          dd[m] = (gpu_inflate && !zmode[k0 + (uint64_t)j]) ? d_stage(sl) + zoff[k0 + (uint64_t)j] : d_depth(sl, j);
          dr[m] = rgb ? (ycc_batch ? d_planes(sl, j) : d_rgb(sl, j)) : nullptr; dl[m] = (rgb && ycc_batch) ? d_pay(sl, j) : nullptr; pp[m] = s->frames[first + g * (uint64_t)B + (uint64_t)j].pose; m++; }
        j++;
      }
      hipStream_t in_stream = sf_input_stream(f, m, rgb, +1);  // the stream this sub-batch's pre-pass runs on
      if (hipStreamWaitEvent(in_stream, bs.copied, 0) != hipSuccess || ((any_comp || side_jpeg) && hipStreamWaitEvent(in_stream, bs.inflated, 0) != hipSuccess)) {
        result = SF_ERR_DEVICE; err = "hipStreamWaitEvent failed"; break;
      }
      if (rgb && gpu_jpeg && !side_jpeg) {   // no side stream (depth not inflated on the device): IDCT + upsampling + colour conversion of this sub-batch's frames ahead of its pre-pass
        const uint8_t* pp_[MAX_BATCH];
        uint8_t* rr_[MAX_BATCH];
        uint8_t* pl_[MAX_BATCH];
        int nj = 0;
        for (int q = jfirst; q < j; q++)
          if (valid[q] && rgbf[q] && (bs.coef_mode[q] == 1 || bs.coef_mode[q] == 2)) { pp_[nj] = d_pay(sl, q); rr_[nj] = d_rgb(sl, q); pl_[nj] = d_planes(sl, q); nj++; }
        bool jpeg_failed = false;
        {   // entropy decoding of the frames that travelled as segments: one 1024-lane workgroup per picture, 16 pictures per launch
          const uint8_t* seg[16];
          uint8_t* out[16];
          uint32_t cap[16];
          int32_t tag[16];
          int nh = 0, slot0 = 0;
          auto flush = [&]() {
            if (nh == 0 || jpeg_failed) return;
            const int rch = jpeg_gpu_huffman(in_stream, nh, seg, out, cap, tag, d_jstatus + 2 * ((size_t)sl * B + (size_t)slot0));
            if (rch != SF_OK) { result = rch; err = sf_last_error(); jpeg_failed = true; }
            nh = 0;
          };
          for (int q = jfirst; q < j; q++) {
            if (!(valid[q] && rgbf[q] && bs.coef_mode[q] == 2)) continue;
            if (nh == 0) slot0 = q;
            seg[nh] = bs.packed_segs ? d_rgb(sl, 0) + (size_t)q * hcol_b : d_rgb(sl, q);
            out[nh] = d_pay(sl, q); cap[nh] = pay_entries; tag[nh] = (int32_t)(first + g * (uint64_t)B + (uint64_t)q);
            if (++nh == 16) flush();
          }
          flush();
        }
        for (int q0 = 0; q0 < nj && !jpeg_failed; q0 += 16) {   // jpeg_gpu.hip reconstructs at most 16 frames per launch
          const int rcj = jpeg_gpu_reconstruct(in_stream, std::min(16, nj - q0), pp_ + q0, rr_ + q0, pl_ + q0, pay_blocks, s->info.color_width, s->info.color_height);
          if (rcj != SF_OK) { result = rcj; err = sf_last_error(); jpeg_failed = true; break; }
        }
        if (jpeg_failed) break;
      }
      const int rc = (rgb && ycc_batch) ? sf_fuser_run_batch_ycc(f, dd, dr, dl, pp, m) : sf_fuser_run_batch(f, dd, rgb ? dr : nullptr, pp, m);
      if (rc != SF_OK) { result = rc; err = sf_last_error(); break; }
      n_int += (uint64_t)m;
      if (used_streams[0] == nullptr || used_streams[0] == in_stream) used_streams[0] = in_stream;
      else used_streams[1] = in_stream;
    }
    if (result == SF_OK) {
      for (int q = 0; q < 2; q++) {
        bs.used[q] = used_streams[q] != nullptr;   // false: nothing on that stream read the device buffers
        if (bs.used[q]) (void)hipEventRecord(bs.consumed[q], used_streams[q]);
      }
    }
    if (timing) t_flush += now_s() - t2;
  }
  if (result != SF_OK) abort.store(true);
  const double t_loop_end = timing ? now_s() : 0;
  for (std::thread& t : pool) t.join();
  if (result != SF_OK) issued.store(nbatches + 1);
  reaper.join();
  const hipError_t qe = sf_quiesce(f);
  if (timing)
    std::fprintf(stderr, "sf_fuse_run: setup %.3f s (streams, %.0f MB pinned, %.0f MB device), loop %.3f s (wait for decoded batches %.3f, copy enqueue %.3f, kernels enqueue %.3f), "
                         "join+drain %.3f s, %d batch slots x %d frames; inside copy enqueue: packed depth memcpy calls %.3f s, inflate launches %.3f s\n",
                 t_setup_end - std::chrono::duration<double>(t_start.time_since_epoch()).count(), (double)NB * slot_b / 1e6, (double)NB * dslot_b / 1e6, t_loop_end - t_setup_end,
                 t_wait_ready, t_api, t_flush, now_s() - t_loop_end, NB, B, t_memcpy, t_launch_z);
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  if (copy_stream2) (void)hipStreamSynchronize(copy_stream2);
  for (hipStream_t q : inflate_stream) if (q) (void)hipStreamSynchronize(q);
  if (result == SF_OK && qe == hipSuccess && d_zstatus) {   // a depth frame the device's inflate gave up on fails the run, as it would on the host
    std::vector<int32_t> st((size_t)NB * B * 2);
    const hipError_t re = hipMemcpy(st.data(), d_zstatus, st.size() * 4, hipMemcpyDeviceToHost);
    if (re != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("inflate: the device's frame status could not be read back: ") + hipGetErrorString(re); }
    for (size_t i = 0; i < st.size() && result == SF_OK; i += 2)
      if (st[i] != 0) {   // the frame itself was fused as "no measurement" (k_inflate_copy zero-fills what it gives up on)
        result = SF_ERR_FORMAT;
        err = "inflate: depth frame " + std::to_string(st[i + 1]) + ": corrupt stream, or it does not inflate to the frame's size (device status " + std::to_string(st[i]) + ")";
      }
  }
  if (result == SF_OK && qe == hipSuccess && d_jstatus) {   // a colour frame the device's entropy decoder gave up on fails the run, as it would on the host
    std::vector<int32_t> st((size_t)NB * B * 2);
    const hipError_t re = hipMemcpy(st.data(), d_jstatus, st.size() * 4, hipMemcpyDeviceToHost);
    if (re != hipSuccess) { result = SF_ERR_DEVICE; err = std::string("jpeg: the device's picture status could not be read back: ") + hipGetErrorString(re); }
    for (size_t i = 0; i < st.size() && result == SF_OK; i += 2)
      if (st[i] != 0) {
        result = SF_ERR_FORMAT;
        err = "jpeg: colour frame " + std::to_string(st[i + 1]) + ": corrupt or truncated entropy-coded segment (device status " + std::to_string(st[i]) + ")";
      }
  }
  cleanup();
  t_run_counts[0] = n_dev_z; t_run_counts[1] = n_host_z; t_run_counts[2] = n_dev_j; t_run_counts[3] = n_host_j;
  if (result != SF_OK) return sf::fail(result, "%s", err.c_str());
  if (qe != hipSuccess) return sf::fail(SF_ERR_DEVICE, "device error while fusing: %s", hipGetErrorString(qe));
  {   // a note, not an error: the run used more streams than the process has hardware queues (see the top of this file)
    const int streams_used = 2 + (copy_stream ? 2 : 0) + (gpu_inflate ? NZ : 0), queues = hardware_queues_of_the_process();
    t_run_note[0] = 0;
    if (streams_used > queues)   // through sf_fuse_run_note(), not sf_last_error(): a caller that reads a non-empty last error as a failure must not
      std::snprintf(t_run_note, sizeof(t_run_note), "sf_fuse_run drove %d streams over %d hardware queues (kernels of streams that share a queue run one after the other); "
                    "export GPU_MAX_HW_QUEUES=16 before the process's first HIP call", streams_used, queues);
  }
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

// scanfuse.h: the streams, the page-locked ring and the device ring a later sf_fuse_run of THIS file will want, made on a thread of their own from now on -- call it
// with the file open and BEFORE sf_fuser_create, whose own allocations (the volume: gigabytes to reserve and clear) then run beside it.  Sizes are upper
// bounds of what sf_fuse_run computes (a set that is large enough is taken as it is; one that is not is re-made by the run, as before): nothing here
// changes what a run does, only when the set-up is paid.
SF_API int sf_fuse_run_prepare(const sf_sens* s, const sf_params* p, int device) {
  if (!s || !p) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return sf::fail(SF_ERR_DEVICE, "no HIP device %d", device);
  const size_t npx = (size_t)s->info.depth_width * s->info.depth_height, depth_b = npx * 2;
  if (npx == 0 || s->frames.empty()) return SF_OK;
  const int B = MAX_BATCH;
  const bool same_res = s->info.color_width == s->info.depth_width && s->info.color_height == s->info.depth_height;
  const bool own_res = p->color_width > 0 && (uint32_t)p->color_width == s->info.color_width && (uint32_t)p->color_height == s->info.color_height;
  const bool use_rgb = ((same_res && p->color_width == 0) || own_res) && s->info.color_compression >= 0 && s->info.color_compression <= 2;
  const bool jpeg = use_rgb && s->info.color_compression == 2;
  const bool gpu_inflate = s->info.depth_compression == 1;
  const int NZ = jpeg ? 5 : 3, NB = gpu_inflate ? 3 + NZ : 3;
  size_t max_depth = 0, max_color = 0;
  for (const SensFrame& fr : s->frames) {
    max_depth = std::max<size_t>(max_depth, (size_t)fr.depth_bytes);
    max_color = std::max<size_t>(max_color, (size_t)fr.color_bytes);
  }
  const size_t seg = gpu_inflate ? ((std::min(max_depth, depth_b) + 63) & ~(size_t)63) : depth_b;   // a frame's share of the packed depth part, at most
  const size_t slot_depth = (seg * B + 255) & ~(size_t)255, dslot_depth = (depth_b * B + 255) & ~(size_t)255;
  const size_t cpx = use_rgb ? (size_t)s->info.color_width * s->info.color_height : 0, rgb_b = cpx * 3;
  size_t hcol_b = rgb_b, col_b = rgb_b, planes_b = 0;
  if (jpeg) {
    const size_t padded = (size_t)((s->info.color_width + 15) & ~15u) * ((s->info.color_height + 15) & ~15u);
    col_b = (sizeof(SfJpegLayout) + padded * 3 / 16 + rgb_b + 255) & ~(size_t)255;   // table of at most 3 padded / 64 blocks + as many entries as the pixels have bytes
    planes_b = (padded * 3 + 255) & ~(size_t)255;
    hcol_b = gpu_inflate ? (max_color + sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) + 64 + 255) & ~(size_t)255 : col_b;   // entropy decoding on the device: the prepared segment
  }
  const size_t slot_col = (col_b * B + 255) & ~(size_t)255, hslot_col = (hcol_b * B + 255) & ~(size_t)255;
  const size_t slot_comp = gpu_inflate ? slot_depth + 256 : 0;
  const size_t h_need = (size_t)NB * (slot_depth + (use_rgb ? hslot_col : 0));
  const size_t d_need = (size_t)NB * (dslot_depth + (use_rgb ? slot_col : 0) + (jpeg ? slot_col : 0) + planes_b * B + slot_comp);
  sf_run_resources_prepare_ex(device, h_need, d_need, gpu_inflate ? 2 * depth_b * (size_t)B : 0, gpu_inflate ? NZ : 0, (!gpu_inflate || use_rgb) ? 2 : 0);
  return SF_OK;
}

// scanfuse_internal.h: where the frames of this thread's last sf_fuse_run were decoded -- out[0] depth frames inflated on the device, out[1] zlib
// depth frames inflated by the host threads, out[2] JPEG colour frames entropy-decoded on the device, out[3] by the host threads.
SF_API const char* sf_fuse_run_note(void) { return t_run_note; }

SF_API int sf_fuse_run_device_counts(uint64_t out[4]) {
  if (!out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  for (int i = 0; i < 4; i++) out[i] = t_run_counts[i];
  return SF_OK;
}

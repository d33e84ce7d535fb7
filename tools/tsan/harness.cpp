// TEST INFRASTRUCTURE: ThreadSanitizer run of sf_fuse_run (scannet_amd/csrc/pipeline.hip) against the asynchronous fake HIP runtime.
// The device side is stubbed: a "pass" (sf_fuser_run_batch) is an operation on the fuser's input stream that reads every byte of the
// device buffers it was handed (checksum), optionally on a second stream pair like the real fuser; the device's inflate is an operation on its
// stream that inflates with the host inflater (two frames in three travel compressed).  The checksum over all frames must equal the one computed directly from the decoded file: the
// pipeline delivered every frame, intact, in order -- and TSan saw no race on the way.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fuser_internal.h"
#include "sens.h"

namespace sf { int usable_cpus(); }

static std::atomic<uint64_t> g_sum{0};
static uint64_t mix(uint64_t h, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
struct Pass { sf_fuser* f; std::vector<const void*> depth, rgb; std::vector<uint64_t> seq; size_t dbytes, cbytes; };
static void run_pass(void* a) {
  Pass* p = (Pass*)a;
  for (size_t j = 0; j < p->depth.size(); j++) {
    uint64_t h = mix(1469598103934665603ull ^ p->seq[j], (const uint8_t*)p->depth[j], p->dbytes);
    if (p->rgb[j]) h = mix(h, (const uint8_t*)p->rgb[j], p->cbytes);
    g_sum.fetch_add(h);
  }
  delete p;
}
hipError_t sf_quiesce(sf_fuser* f) { hipStreamSynchronize(f->front); return hipStreamSynchronize(f->stream); }
bool sf_single_stream_batch(const sf_fuser*, int n, bool, int) { return n == 1; }
hipStream_t sf_input_stream(const sf_fuser* f, int n, bool color, int sign) { return sf_single_stream_batch(f, n, color, sign) ? f->stream : f->front; }
int sf_fuser_run_batch(sf_fuser* f, const void* const* d_depth, const void* const* d_rgb, const float* const*, int n) {
  Pass* p = new Pass();
  p->f = f; p->dbytes = f->in_px * 2; p->cbytes = f->in_px * 3;
  for (int j = 0; j < n; j++) { p->depth.push_back(d_depth[j]); p->rgb.push_back(d_rgb ? d_rgb[j] : nullptr); p->seq.push_back(f->frames_integrated++); }
  hipStream_t s = sf_input_stream(f, n, d_rgb != nullptr, +1);
  // like run_batch: the pre-pass reads the inputs on the input stream; the integrate stream is ordered behind it
  fake_stream_enqueue(s, run_pass, p);
  if (s != f->stream) { hipEventRecord(f->ev_compact[0], s); hipStreamWaitEvent(f->stream, f->ev_compact[0], 0); }
  return SF_OK;
}
// The device's inflate, faked: an operation on the stream it is launched on that inflates every frame of the launch with the host inflater from the
// "device" copy of the compressed bytes into the frame buffer the pre-pass will read, and scribbles over the launch's plan scratch (shared by the
// launches of one stream: a second stream using it at the same time is a race TSan sees).  Two frames in three travel compressed, the third is
// inflated by a host thread -- both kinds in every batch, as a file with foreign streams gives.
void inflate_gpu_warm() {}
void jpeg_gpu_warm() {}
void jpeg_huff_gpu_warm() {}
bool inflate_gpu_takes(const uint8_t* z, uint64_t n) {
  return n >= 8 && (z[0] & 0x0F) == 8 && ((z[0] << 8 | z[1]) % 31) == 0 && !(z[1] & 0x20) && (z[2] & 7) == 3 && (z[n - 1] + z[n - 2]) % 3 != 0;
}
struct FakeInflate { std::vector<const uint32_t*> words; std::vector<uint32_t> nbytes; std::vector<uint8_t*> out; std::vector<uint16_t*> plan; uint32_t expect; int32_t* status; };
static std::atomic<uint64_t> g_inflated{0};
static void run_inflate(void* a) {
  FakeInflate* p = (FakeInflate*)a;
  g_inflated.fetch_add(p->out.size());
  for (size_t i = 0; i < p->out.size(); i++) {
    std::vector<uint8_t> z(2 + (size_t)p->nbytes[i]);
    z[0] = 0x78; z[1] = 0x01;
    std::memcpy(z.data() + 2, p->words[i], p->nbytes[i]);
    std::memset(p->plan[i], (int)i, 2 * (size_t)p->expect);
    uint64_t got = 0;
    if (sf_zlib_inflate(z.data(), z.size(), p->out[i], p->expect, &got) != SF_OK || got != p->expect) p->status[2 * i] = -4;
  }
  delete p;
}
int inflate_gpu_batch(hipStream_t stream, int n, const uint32_t* const* d_words, const uint32_t* nbytes, uint8_t* const* d_out, uint16_t* const* d_plan, uint32_t expect, const int32_t*,
                      int32_t* d_status) {
  FakeInflate* p = new FakeInflate();
  for (int i = 0; i < n; i++) { p->words.push_back(d_words[i]); p->nbytes.push_back(nbytes[i]); p->out.push_back(d_out[i]); p->plan.push_back(d_plan[i]); }
  p->expect = expect;
  p->status = d_status;
  fake_stream_enqueue(stream, run_inflate, p);
  return SF_OK;
}
int jpeg_gpu_huffman(hipStream_t, int, const uint8_t* const*, uint8_t* const*, const uint32_t*, const int32_t*, int32_t*) { return SF_OK; }
int jpeg_gpu_reconstruct(hipStream_t, int, const uint8_t* const*, uint8_t* const*, uint8_t* const*, uint32_t, uint32_t, uint32_t) { return SF_OK; }   // raw colour in this harness
int jpeg_gpu_planes(hipStream_t, int, const uint8_t* const*, uint8_t* const*, uint32_t) { return SF_OK; }
int sf_fuser_run_batch_ycc(sf_fuser*, const void* const*, const void* const*, const void* const*, const float* const*, int) { return SF_OK; }   // JPEG colour only

int main(int argc, char** argv) {
  const int W = 160, H = 120, N = argc > 1 ? atoi(argv[1]) : 300, threads = argc > 2 ? atoi(argv[2]) : 8;
  const std::string path = "/tmp/sf_tsan_harness.sens";
  sf_sens_info hi;
  std::memset(&hi, 0, sizeof(hi));
  hi.color_width = W; hi.color_height = H; hi.depth_width = W; hi.depth_height = H;
  hi.color_compression = 0; hi.depth_compression = 1; hi.depth_shift = 1000.0f;
  std::snprintf(hi.sensor_name, sizeof(hi.sensor_name), "tsan");
  sf_sens* w = nullptr;
  if (sf_sens_create(&hi, &w) != SF_OK) return 2;
  std::vector<uint16_t> d((size_t)W * H);
  std::vector<uint8_t> c((size_t)W * H * 3);
  uint64_t want = 0, seq = 0;
  for (int i = 0; i < N; i++) {
    for (size_t k = 0; k < d.size(); k++) d[k] = (uint16_t)(1000 + ((k * 7 + (size_t)i * 13) % 97));
    for (size_t k = 0; k < c.size(); k++) c[k] = (uint8_t)(k * 3 + (size_t)i);
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const bool lost = i % 37 == 5, nocol = i % 11 == 3;
    if (lost) for (float& v : pose) v = -__builtin_inff();   // tracking lost: skipped by the pipeline
    if (sf_sens_add_frame(w, nocol ? nullptr : c.data(), nocol ? 0 : c.size(), d.data(), pose, i, i) != SF_OK) return 3;
    if (!lost) {
      uint64_t h = mix(1469598103934665603ull ^ seq++, (const uint8_t*)d.data(), d.size() * 2);
      if (!nocol) h = mix(h, c.data(), c.size());
      want += h;
    }
  }
  if (sf_sens_save(w, path.c_str()) != SF_OK) return 4;
  sf_sens_close(w);
  sf_sens* s = nullptr;
  if (sf_sens_open(path.c_str(), &s) != SF_OK) { std::fprintf(stderr, "%s\n", sf_last_error()); return 5; }
  int rc_all = 0;
  for (int batch : {16, 1, 5}) {
    sf_fuser f;
    std::memset(&f.p, 0, sizeof(f.p));
    std::memset(&f.pk, 0, sizeof(f.pk));
    f.p.depth_width = W; f.p.depth_height = H; f.in_W = W; f.in_H = H; f.in_px = (size_t)W * H;
    f.pk.W = W; f.pk.H = H; f.batch = batch;
    hipStreamCreateWithFlags(&f.stream, 0); hipStreamCreateWithFlags(&f.front, 0);
    hipEventCreateWithFlags(&f.ev_compact[0], 0);
    g_sum = 0;
    g_inflated = 0;
    sf_run_stats st;
    const int rc = sf_fuse_run(&f, s, 0, 0, threads, &st);
    if (rc != SF_OK) { std::fprintf(stderr, "sf_fuse_run: %s\n", sf_last_error()); rc_all = 6; }
    const bool ok = g_sum.load() == want && st.frames_integrated == seq;
    std::printf("batch %2d: %llu frames fused (%llu of them inflated by the fake device), %llu skipped, checksum %s\n", batch, (unsigned long long)st.frames_integrated,
                (unsigned long long)g_inflated.load(), (unsigned long long)st.frames_skipped, ok ? "ok" : "MISMATCH");
    if (g_inflated.load() == 0 || g_inflated.load() == st.frames_integrated) rc_all = 8;   // both kinds of frames must have been there
    if (!ok) rc_all = 7;
    hipStreamDestroy(f.front); hipStreamDestroy(f.stream); hipEventDestroy(f.ev_compact[0]);
    f.stream = f.front = nullptr;
  }
  sf_sens_close(s);
  std::remove(path.c_str());
  return rc_all;
}

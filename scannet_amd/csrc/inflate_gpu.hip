// inflate_gpu.hip -- the depth frames' zlib inflate on gfx950, for the streams the reference's writer produces (one final fixed-Huffman block:
// stb_image_write.h:733-736): the lane programs of inflate_lanes.h, two kernels per batch of frames.
//
// Until round 4 the frame pipeline's host threads inflated every depth frame (0.8 ms per 640x480 frame and thread: 16 threads bound a scan at
// 19 k frames/s while the integrate pass alone runs at 37 k).  Now a host thread copies the compressed frame into the pinned ring (0.03 ms); the
// compressed bytes cross PCIe (less than the pixels) and are inflated here, straight into the buffer k_prepass reads.  Replaces
// RGBDFrame::decompressDepthAlloc_stb -> stbi_zlib_decode_malloc (SensReader/c++/src/sensorData.h:693-709, stb_image.h:3791-3846) for those
// streams; dynamic / stored / multi-block streams stay with the host inflater (zlib_codec.cpp).  Byte-identical to sf_zlib_inflate
// (tests/test_gpu_pipeline.py, tests/test_inflate_lanes.py).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "common.h"
#include "inflate_lanes.h"

namespace {

constexpr int IG_BATCH = 32;
constexpr int IG_LANES = 1024;

struct InflateBatch {
  const uint32_t* words[IG_BATCH];   // the deflate data (the zlib stream from its third byte on), 64-byte aligned, 256 readable bytes behind it, device
  uint32_t nbytes[IG_BATCH];         // bytes of it (incl. the Adler-32 trailer)
  uint8_t* out[IG_BATCH];            // the inflated frame, `expect` bytes, 4-byte aligned; nullptr: slot unused
  uint16_t* plan[IG_BATCH];          // scratch, one u16 per output byte (inflate_lanes.h: the plan), 8-byte aligned
  int32_t tag[IG_BATCH];             // what the caller wants to read back with a failure (a frame number)
  uint32_t expect;                   // bytes every frame must inflate to (a multiple of 4)
  int32_t* status;                   // written ONLY on failure: status[2 * slot] = IL_ST_* < 0, status[2 * slot + 1] = tag
  uint32_t skip;                     // read only in a -DSF_MEASURE_ABLATE build (sf_zlib_inflate_gpu_bench): 1 = no stage C, 2 = no stage A either
};

// exclusive prefix sum of one value per lane over the 1024 lanes of the workgroup; also the total
__device__ inline uint32_t block_exscan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  __syncthreads();   // s_wave may still be read by the previous scan
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < IG_LANES / 64; w++) {
    const uint32_t x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  total = tot;
  return base + incl - v;
}

// The stream as a lane reads it: a window of 128 bytes of its chunk in an LDS slot of its own (32 words indexed by word number, slots 33 words
// apart: no bank conflicts), advanced 64 bytes at a time by lane_pass below.  Word by word from memory every lane of the 1024 pulled a whole
// 128-byte line through the 32 KiB L1 for 4 bytes of it, and every token step of a wave waited for some lane's load; now a line is fetched
// twice and the fetch of the next 64 bytes is in flight while the lane decodes the current ones.
struct LaneWindow {
  uint32_t* slot;
  uint32_t nbits;
  const uint32_t* lit_;
  const uint32_t* dist_;
  __device__ uint32_t word(uint32_t w) const { return slot[w & 31u]; }
  __device__ uint32_t lit(uint32_t i) const { return lit_[i]; }
  __device__ uint32_t dist(uint32_t i) const { return dist_[i]; }
};
__device__ inline void slot_put(uint32_t* slot, uint32_t seg, const uint4& a, const uint4& b, const uint4& c, const uint4& d) {
  uint32_t* p = slot + ((seg & 1u) << 4);
  p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; p[4] = b.x; p[5] = b.y; p[6] = b.z; p[7] = b.w;
  p[8] = c.x; p[9] = c.y; p[10] = c.z; p[11] = c.w; p[12] = d.x; p[13] = d.y; p[14] = d.z; p[15] = d.w;
}
// One pass of a lane over its chunk from bit `start`: run(until) decodes the tokens that start in front of bit `until` and says whether the pass
// is complete.  Segment k = bits [512 k, 512 k + 512); a token that starts in segment k ends inside segment k + 1, so the window holds k and
// k + 1 while k + 2 is on its way.  Every lane of the wave calls this (active or not): the loop runs until the wave's last lane is through.
template <class Begin, class Run>
__device__ inline void lane_pass(const uint4* __restrict__ mem, uint32_t* slot, uint32_t start, bool active, Begin begin, Run run) {
  uint32_t seg = start >> 9;
  bool done = !active;
  if (active) {
    const uint4* m = mem + 4u * seg;
    const uint4 a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7];
    slot_put(slot, seg, a, b, c, d);
    slot_put(slot, seg + 1u, e, f, g, h);
    begin();
  }
  while (__ballot(!done) != 0ull) {
    uint4 a, b, c, d;
    if (!done) {
      const uint4* m = mem + 4u * (seg + 2u);
      a = m[0]; b = m[1]; c = m[2]; d = m[3];
      done = run((seg + 1u) << 9);
      if (!done) slot_put(slot, seg + 2u, a, b, c, d);
      seg++;
    }
  }
}

// the plan entries of a lane's chunk, four to a store where the alignment allows (the chunk's first and last entries share their 8 bytes with
// the neighbours' chunks: those go one by one)
struct PlanSink {
  uint16_t* plan;
  uint32_t o, head_end;   // next entry; the first multiple of 4 at or behind the chunk's first entry
  uint64_t acc;
  __device__ void put(uint16_t v) {
    if (o < head_end) {
      plan[o] = v;
    } else {
      const uint32_t k = o & 3u;
      acc = k == 0u ? (uint64_t)v : acc | ((uint64_t)v << (16u * k));
      if (k == 3u) *reinterpret_cast<uint64_t*>(plan + (o - 3u)) = acc;
    }
    o++;
  }
  __device__ void put_run(uint16_t v, uint32_t n) {   // n times the same entry, without a loop over the entries (matches are ~4 bytes long: an
                                                      // entry-by-entry head and tail cost more than the decoding)
    while (n != 0u && o < head_end) { put(v); n--; }   // the chunk's first entries only
    if (n == 0u) return;
    const uint64_t four = (uint64_t)v * 0x0001000100010001ull;
    const uint32_t k = o & 3u, room = 4u - k, take = n < room ? n : room;
    const uint64_t part = take == 4u ? four : (four & ((1ull << (16u * take)) - 1ull));
    acc = k == 0u ? part : acc | (part << (16u * k));
    o += take;
    n -= take;
    if ((o & 3u) == 0u) *reinterpret_cast<uint64_t*>(plan + (o - 4u)) = acc;
    for (; n >= 4u; n -= 4u, o += 4u) *reinterpret_cast<uint64_t*>(plan + o) = four;
    if (n != 0u) {
      acc = four & ((1ull << (16u * n)) - 1ull);
      o += n;
    }
  }
  __device__ void flush() {   // the entries of an unfinished packet
    const uint32_t k = o & 3u;
    if (o >= head_end)
      for (uint32_t i = 0; i < k; i++) plan[o - k + i] = (uint16_t)(acc >> (16u * i));
  }
};

__global__ __launch_bounds__(IG_LANES) void k_inflate_tokens(InflateBatch B) {
  __shared__ uint32_t s_lit[512], s_dist[32], s_end[IG_LANES], s_res[IG_LANES], s_start[IG_LANES], s_list[IG_LANES], s_wave[IG_LANES / 64];
  __shared__ uint32_t s_slots[IG_LANES * 33];
  // 154 KB of static LDS: this kernel needs gfx950's 160 KB per workgroup (the library is built for gfx950 alone; a 64 KB part would need 512 lanes)
  static_assert(sizeof(uint32_t) * (512 + 32 + 4 * IG_LANES + IG_LANES / 64 + IG_LANES * 33) <= 160 * 1024, "k_inflate_tokens: LDS budget of gfx950");
  const int f = blockIdx.x;
  if (B.out[f] == nullptr) return;
  for (uint32_t i = threadIdx.x; i < 512u; i += IG_LANES) s_lit[i] = il_lit_entry(i);
  if (threadIdx.x < 32u) s_dist[threadIdx.x] = il_dist_entry(threadIdx.x);
  __syncthreads();
  const uint32_t nbytes = B.nbytes[f];
  const uint4* __restrict__ mem = reinterpret_cast<const uint4*>(B.words[f]);
  uint32_t* slot = s_slots + 33u * threadIdx.x;
  LaneWindow S{slot, nbytes * 8u, s_lit, s_dist};
  uint32_t C, Bc;
  il_geometry(S.nbits, C, Bc);
  const uint32_t c = threadIdx.x;
  const bool mine = c < C;
  const uint32_t limit = (c + 1 == C) ? S.nbits : (c + 1) * Bc;
  // ---- stage A: the chunk starts to their fixed point (bit 3: behind BFINAL and BTYPE).  Round 0: every lane scans its own chunk from
  // IL_LEAD_BITS in front of it and counts from the first token that starts inside (il_guess_start): ~99 % of the lanes are on the true path by
  // then, the boundary they enter their chunk at IS the left neighbour's stop.  The chunks whose start turns out different rescan from the
  // neighbour's stop; those few are dealt to the first lanes -- left where they are, a dozen chunks kept most of the 16 waves busy for another
  // whole pass.
  uint32_t start = 3u, end = 0, outb = 0, flag = IL_FLAG_OK;
#ifdef SF_MEASURE_ABLATE
  const uint32_t skip = B.skip;
#else
  constexpr uint32_t skip = 0u;
#endif
  bool dirty = mine && skip < 2u;
  for (uint32_t round = 0; round < C + 2u && skip < 2u; round++) {
    uint32_t cc = c;   // the chunk this lane scans in this round
    bool work = dirty;
    uint32_t my_start = il_guess_start(c, Bc), my_own = c == 0 ? 3u : c * Bc;
    if (round >= 1u) {
      uint32_t ndirty;
      const uint32_t at = block_exscan(dirty ? 1u : 0u, s_wave, ndirty);
      if (dirty) { s_list[at] = c; s_start[c] = start; }
      __syncthreads();
      work = threadIdx.x < ndirty;
      if (work) { cc = s_list[threadIdx.x]; my_start = my_own = s_start[cc]; }
    }
    const uint32_t my_limit = (cc + 1 == C) ? S.nbits : (cc + 1) * Bc;
    ILScan sc;
    lane_pass(mem, slot, my_start, work, [&]() { il_scan_begin(S, sc, my_start, my_own); },
              [&](uint32_t until) {
                il_scan_run(S, sc, until < my_limit ? until : my_limit);
                return sc.off_end || sc.b.pos >= my_limit;
              });
    if (work) {
      s_end[cc] = sc.b.pos;
      s_res[cc] = (sc.out << 2) | sc.fl;
    }
    if (round == 0u && dirty) start = il_scan_first(sc);   // where the chunk's own tokens begin on the guessed path
    __syncthreads();
    if (dirty) { end = s_end[c]; outb = s_res[c] >> 2; flag = s_res[c] & 3u; }
    bool changed = false;
    if (mine && c > 0) {
      const uint32_t ns = s_end[c - 1];
      changed = ns != start;
      start = ns;
    }
    dirty = changed;
    if (!__syncthreads_or((int)changed)) break;
  }
  // ---- stage B: what lies behind the end-of-block code is the trailer, not tokens; where every chunk's bytes go
  uint32_t n_eob, total;
  const bool behind = block_exscan((mine && (flag & IL_FLAG_EOB)) ? 1u : 0u, s_wave, n_eob) != 0u;
  const bool live = mine && !behind;
  const uint32_t o = block_exscan(live ? outb : 0u, s_wave, total);
  int32_t status = IL_ST_OK;
  if (__syncthreads_or(live && (flag & IL_FLAG_ERR))) status = IL_ST_BAD_CODE;
  else if (n_eob == 0u) status = IL_ST_NO_EOB;
  else if (total != B.expect) status = IL_ST_SIZE;
  // ---- stage C: the plan -- per output byte, a literal or how far back its source lies
  {
    const bool writes = status == IL_ST_OK && live && outb != 0u && skip == 0u;
    PlanSink P{B.plan[f], o, (o + 3u) & ~3u, 0ull};
    ILWrite w;
    w.status = IL_ST_OK;
    const uint32_t o_end = o + outb;
    lane_pass(mem, slot, start, writes, [&]() { il_write_begin(S, w, start, o); },
              [&](uint32_t until) {
                il_write_run(S, w, o_end, until, P);
                return w.status != IL_ST_OK || w.o >= o_end;
              });
    if (writes) {
      P.flush();
      status = w.status;
    }
  }
  if (status != IL_ST_OK && (status == IL_ST_BAD_DISTANCE || threadIdx.x == 0)) {   // the others are the same in every lane
    B.status[2 * f] = status;
    B.status[2 * f + 1] = B.tag[f];
  }
}

struct GroupMem {
  uint8_t* ring_;     // IL_WINDOW bytes
  uint16_t* gref_;    // IL_GROUP entries
  uint8_t* gval_;
  __device__ uint8_t ring(uint32_t i) const { return ring_[i & (IL_WINDOW - 1u)]; }
  __device__ uint16_t& gref(uint32_t j) const { return gref_[j]; }
  __device__ uint8_t& gval(uint32_t j) const { return gval_[j]; }
};

// A barrier over the workgroup that orders its LDS traffic and nothing else: __syncthreads() also waits for the loads of the plan, which are
// issued blocks ahead precisely so that nobody waits for them (and for the group's own stores: 1.4 us per group when it did).
__device__ inline void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ inline bool lds_barrier_or(bool v, uint32_t* s_flag, uint32_t round) {   // any lane of the workgroup; s_flag: two words, used alternately
  if (v) s_flag[round & 1u] = round + 1u;
  lds_barrier();
  return s_flag[round & 1u] == round + 1u;
}

constexpr uint32_t IG_COPY_LANES = IL_GROUP / 4;   // 256
constexpr uint32_t IG_BLOCK = 8;                   // groups whose plan is fetched together, a block ahead of its use
__global__ __launch_bounds__(IG_COPY_LANES) void k_inflate_copy(InflateBatch B) {
  __shared__ uint32_t s_ring[IL_WINDOW / 4];
  __shared__ uint16_t s_gref[IL_GROUP];
  __shared__ uint8_t s_gval[IL_GROUP];
  __shared__ uint32_t s_flag[2];
  const int f = blockIdx.x;
  if (B.out[f] == nullptr) return;
  const uint32_t lane = threadIdx.x, total = B.expect;
  uint32_t* __restrict__ out32 = reinterpret_cast<uint32_t*>(B.out[f]);
  if (B.status[2 * f] != 0) {   // a frame k_inflate_tokens gave up on: depth 0 = "no measurement" everywhere, so that nothing of whatever the slot held
    for (uint32_t i = lane; i < total / 4u; i += IG_COPY_LANES) out32[i] = 0u;   // before is fused under this frame's pose while the failure travels to the host
    return;
  }
  const uint64_t* __restrict__ plan64 = reinterpret_cast<const uint64_t*>(B.plan[f]);
  const GroupMem M{reinterpret_cast<uint8_t*>(s_ring), s_gref, s_gval};
  if (lane < 2u) s_flag[lane] = 0u;
  const uint32_t quads = total / 4u;   // 8-byte plan packets = 4-byte output words
  uint64_t cur[IG_BLOCK], nxt[IG_BLOCK];
#pragma unroll
  for (uint32_t i = 0; i < IG_BLOCK; i++) cur[i] = IG_COPY_LANES * i + lane < quads ? plan64[IG_COPY_LANES * i + lane] : 0ull;
  // waited for HERE: left to the compiler, the wait for these lands inside the loop behind the next block's loads (one counter for all of
  // them) and every block would wait for the plan it has just asked for
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), the other counters untouched (gfx9 encoding)
  __syncthreads();
  bool bad = false;
  uint32_t round = 0;
  for (uint32_t pos0 = 0; pos0 < total; pos0 += IG_BLOCK * IL_GROUP) {
#pragma unroll
    for (uint32_t i = 0; i < IG_BLOCK; i++) {
      const uint32_t k = (pos0 + (IG_BLOCK + i) * IL_GROUP) / 4u + lane;
      nxt[i] = k < quads ? plan64[k] : 0ull;
    }
#pragma unroll
    for (uint32_t i = 0; i < IG_BLOCK; i++) {
      const uint32_t pos = pos0 + i * IL_GROUP;
      if (pos < total) {   // uniform over the workgroup
        const uint32_t n = total - pos < IL_GROUP ? total - pos : IL_GROUP;
        ILQuad q;
        bool open = il_quad_classify(M, pos, lane, n, cur[i], q, bad);
        il_quad_publish(M, lane, q);
        while (lds_barrier_or(open, s_flag, round++)) {
          if (open) open = il_quad_resolve(M, q);
          lds_barrier();            // everybody has read what was published
          il_quad_publish(M, lane, q);
        }
        if (4u * lane < n) {
          s_ring[(pos & (IL_WINDOW - 1u)) / 4u + lane] = q.v;
          out32[pos / 4u + lane] = q.v;
        }
        // the next group's ring reads follow the next barrier-or only after its classify: order them behind this group's ring writes
        lds_barrier();
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < IG_BLOCK; i++) cur[i] = nxt[i];
  }
  if (__syncthreads_or((int)bad)) {   // a match that reaches in front of the output: the frame is void (see above)
    for (uint32_t i = lane; i < total / 4u; i += IG_COPY_LANES) out32[i] = 0u;
    if (bad) {
      B.status[2 * f] = IL_ST_BAD_DISTANCE;
      B.status[2 * f + 1] = B.tag[f];
    }
  }
}

}  // namespace

// Is this zlib stream one the device inflates (deflate, no preset dictionary, ONE final block with the fixed code)?  The host threads of the
// frame pipeline ask before they decide where a depth frame is inflated.
// Size bound: the kernels count output bytes in 32 bits.  The densest token is a 258-byte match in 13 bits (8-bit length code 285 + 5-bit
// distance code, no extra bits): a stream of n bytes inflates to at most n * 8 / 13 * 258 < 159 n bytes, so below 2^24 bytes of deflate data no
// count -- per chunk, prefix sum or total -- can wrap (2^24 * 159 = 2.7e9 < 2^32).  A depth frame's stream is bounded by its pixels anyway.
constexpr uint64_t IG_MAX_STREAM_BYTES = 1ull << 24;
static_assert(IG_MAX_STREAM_BYTES * 8 / 13 * 258 < (1ull << 32), "32-bit output counts of the inflate kernels");
// The code object of this file is loaded by the runtime when one of its kernels is first used (milliseconds, inside a scan's first sf_fuse_run unless somebody asks
// earlier): the preparation thread of the frame pipeline asks (pipeline.hip, sf_run_resources_prepare_ex).
void inflate_gpu_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_inflate_tokens));
  (void)hipGetLastError();
}

bool inflate_gpu_takes(const uint8_t* z, uint64_t n) {
  return n >= 8 && (z[0] & 0x0F) == 8 && ((z[0] << 8 | z[1]) % 31) == 0 && !(z[1] & 0x20) && (z[2] & 7) == 3 && n - 2 < IG_MAX_STREAM_BYTES;
}

// Inflate up to 32 frames on `stream`: d_words[i] = the zlib stream from its third byte on (nbytes[i] bytes, 64-byte aligned, with 256 readable
// bytes behind it), d_out[i] = `expect` bytes, d_plan[i] = 2 * expect bytes of scratch (8-byte aligned); d_status: 2 * n ints the caller zeroed, written only for frames that fail:
// {IL_ST_* < 0, tags[i]}.
int inflate_gpu_batch(hipStream_t stream, int n, const uint32_t* const* d_words, const uint32_t* nbytes, uint8_t* const* d_out, uint16_t* const* d_plan, uint32_t expect,
                      const int32_t* tags, int32_t* d_status) {
  if (n < 1 || n > IG_BATCH || (expect & 3u) || !d_status) return sf::fail(SF_ERR_INVALID_ARG, "inflate_gpu_batch: %d frames of %u bytes", n, expect);
  InflateBatch b;
  std::memset(&b, 0, sizeof(b));
  for (int i = 0; i < n; i++)
    if (nbytes[i] >= IG_MAX_STREAM_BYTES) return sf::fail(SF_ERR_INVALID_ARG, "inflate_gpu_batch: stream %d has %u bytes (the device takes < 2^24: 32-bit output counts)", i, nbytes[i]);
  for (int i = 0; i < n; i++) {
    b.words[i] = d_words[i]; b.nbytes[i] = nbytes[i]; b.out[i] = d_out[i]; b.plan[i] = d_plan[i];
    b.tag[i] = tags ? tags[i] : i;
  }
  b.expect = expect;
  b.status = d_status;
  hipLaunchKernelGGL(k_inflate_tokens, dim3(n), dim3(IG_LANES), 0, stream, b);
  hipLaunchKernelGGL(k_inflate_copy, dim3(n), dim3(IG_COPY_LANES), 0, stream, b);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
}

// scanfuse_internal.h: the two kernels timed apart (HIP events) on `count` <= 32 resident streams, `repeats` times: microseconds per launch of
// k_inflate_tokens and of k_inflate_copy, for tools/gpu/inflate_bench.py.  The output is not returned (the parity tests check it).
SF_API int sf_zlib_inflate_gpu_bench(const void* const* srcs, const uint64_t* src_bytes, int count, uint64_t expect_bytes, int device, int repeats, int skip, double* us_tokens, double* us_copy) {
  if (!srcs || !src_bytes || count < 1 || count > IG_BATCH || repeats < 1 || !us_tokens || !us_copy || (expect_bytes & 3u)) return sf::fail(SF_ERR_INVALID_ARG, "sf_zlib_inflate_gpu_bench: bad argument");
#ifndef SF_MEASURE_ABLATE
  if (skip != 0) return sf::fail(SF_ERR_UNSUPPORTED, "sf_zlib_inflate_gpu_bench: the stage switches exist in a -DSF_MEASURE_ABLATE build only (SCANFUSE_BUILD_FLAGS)");
#endif
  SF_HIP_CHECK(hipSetDevice(device));
  InflateBatch b;
  std::memset(&b, 0, sizeof(b));
  std::vector<void*> owned;
  auto release = [&]() { for (void* p : owned) (void)hipFree(p); };
  hipError_t e = hipSuccess;
  for (int i = 0; i < count && e == hipSuccess; i++) {
    const uint8_t* z = static_cast<const uint8_t*>(srcs[i]);
    if (!inflate_gpu_takes(z, src_bytes[i])) { release(); return sf::fail(SF_ERR_UNSUPPORTED, "stream %d is not one final fixed-Huffman block", i); }
    const uint32_t nb = (uint32_t)(src_bytes[i] - 2);
    void *w = nullptr, *o = nullptr, *pl = nullptr;
    e = hipMalloc(&w, (size_t)nb + 384);
    if (e == hipSuccess) { owned.push_back(w); e = hipMemset(w, 0, (size_t)nb + 384); }
    if (e == hipSuccess) e = hipMemcpy(w, z + 2, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&o, expect_bytes);
    if (e == hipSuccess) { owned.push_back(o); e = hipMalloc(&pl, 2 * expect_bytes); }
    if (e == hipSuccess) owned.push_back(pl);
    b.words[i] = static_cast<const uint32_t*>(w); b.nbytes[i] = nb; b.out[i] = static_cast<uint8_t*>(o); b.plan[i] = static_cast<uint16_t*>(pl); b.tag[i] = i;
  }
  int32_t* d_status = nullptr;
  if (e == hipSuccess) e = hipMalloc((void**)&d_status, 8 * IG_BATCH);
  if (e == hipSuccess) { owned.push_back(d_status); e = hipMemset(d_status, 0, 8 * IG_BATCH); }
  b.expect = (uint32_t)expect_bytes;
  b.status = d_status;
  b.skip = (uint32_t)skip;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreate(&ev[i]);
  double t_tok = 0, t_cp = 0;
  for (int r = 0; r < repeats + 1 && e == hipSuccess; r++) {   // the first round warms up
    (void)hipEventRecord(ev[0], nullptr);
    hipLaunchKernelGGL(k_inflate_tokens, dim3(count), dim3(IG_LANES), 0, nullptr, b);
    (void)hipEventRecord(ev[1], nullptr);
    hipLaunchKernelGGL(k_inflate_copy, dim3(count), dim3(IG_COPY_LANES), 0, nullptr, b);
    (void)hipEventRecord(ev[2], nullptr);
    e = hipEventSynchronize(ev[2]);
    float a = 0, c = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&a, ev[0], ev[1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&c, ev[1], ev[2]);
    if (r > 0) { t_tok += a * 1e3; t_cp += c * 1e3; }
  }
  int32_t st[2 * IG_BATCH] = {0};
  if (e == hipSuccess) e = hipMemcpy(st, d_status, sizeof(st), hipMemcpyDeviceToHost);
  for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x);
  release();
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "sf_zlib_inflate_gpu_bench: %s", hipGetErrorString(e));
  for (int i = 0; i < count && skip == 0; i++) if (st[2 * i] != 0) return sf::fail(SF_ERR_FORMAT, "stream %d: device status %d", i, st[2 * i]);
  *us_tokens = t_tok / repeats;
  *us_copy = t_cp / repeats;
  return SF_OK;
}

// scanfuse_internal.h: one zlib stream through the device path, for the parity tests (the bytes must be sf_zlib_inflate's).  SF_ERR_UNSUPPORTED
// for streams the device leaves to the host (anything but one final fixed-Huffman block; expect_bytes not a multiple of 4), SF_ERR_FORMAT for
// corrupt ones and for streams that inflate to another size.
SF_API int sf_zlib_inflate_gpu(const void* src_, uint64_t n, uint64_t expect_bytes, int device, void* dst) {
  const uint8_t* src = static_cast<const uint8_t*>(src_);
  if (!src || !dst) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!inflate_gpu_takes(src, n) || (expect_bytes & 3u) || expect_bytes == 0 || expect_bytes > (1ull << 30))
    return sf::fail(SF_ERR_UNSUPPORTED, "zlib: the device inflates one final fixed-Huffman block to a multiple of 4 bytes; this stream goes to sf_zlib_inflate");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  const uint32_t nbytes = (uint32_t)(n - 2);
  uint32_t* d_words = nullptr;
  uint8_t* d_out = nullptr;
  uint16_t* d_plan = nullptr;
  int32_t* d_status = nullptr;
  auto release = [&]() { (void)hipFree(d_words); (void)hipFree(d_out); (void)hipFree(d_plan); (void)hipFree(d_status); };
  const size_t wbytes = ((size_t)nbytes + 3) / 4 * 4;
  hipError_t e = hipMalloc((void**)&d_words, wbytes + 320);
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, expect_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&d_plan, 2 * expect_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&d_status, 8);
  if (e == hipSuccess) e = hipMemset(d_status, 0, 8);
  if (e == hipSuccess) e = hipMemset(d_words, 0, wbytes + 320);
  if (e == hipSuccess) e = hipMemcpy(d_words, src + 2, nbytes, hipMemcpyHostToDevice);
  int rc = SF_OK;
  int32_t status[2] = {0, 0};
  if (e == hipSuccess) {
    const uint32_t* w = d_words;
    rc = inflate_gpu_batch(nullptr, 1, &w, &nbytes, &d_out, &d_plan, (uint32_t)expect_bytes, nullptr, d_status);
    if (rc == SF_OK) e = hipMemcpy(status, d_status, 8, hipMemcpyDeviceToHost);
    if (rc == SF_OK && e == hipSuccess && status[0] == 0) e = hipMemcpy(dst, d_out, expect_bytes, hipMemcpyDeviceToHost);
  }
  release();
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "sf_zlib_inflate_gpu: %s", hipGetErrorString(e));
  if (rc == SF_OK && status[0] != 0)
    return sf::fail(SF_ERR_FORMAT, "zlib: the device reports %s (status %d)",
                    status[0] == IL_ST_SIZE ? "a stream that does not inflate to the expected size" : status[0] == IL_ST_BAD_DISTANCE ? "a match that reaches in front of the output"
                    : status[0] == IL_ST_NO_EOB ? "a block without an end-of-block code" : "an invalid code", status[0]);
  return rc;
}

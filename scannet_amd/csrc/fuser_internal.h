// fuser_internal.h -- device-side layout and the sf_fuser handle, shared by fuser.hip and mc.hip
#pragma once
#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

#include "common.h"

constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr uint64_t KEY_TOMB = ~0ull - 1ull;
constexpr int MAX_DDA_ITERS = 1024;
constexpr int MAX_PROBES = 4096;
constexpr int MAX_BATCH = 32;      // most frames one pass over the voxel tiles can fuse: the frame mask of a block is one 32-bit word (temporal blocking)
constexpr int DEFAULT_BATCH = 32;  // what a fuser starts with (sf_fuser_tune "batch"): 16 -> 32 frames per pass is +3.3 % frames/s (half as many launches and gaps; profiles/r03_small_experiments.txt)

struct HashEntry {
  uint64_t key;
  int32_t ptr;
  uint32_t birth;  // sequence number of the first frame that asked for this block (0xFFFFFFFF until then)
};
static_assert(sizeof(HashEntry) == 16, "hash entry is 16 bytes");

struct ParamsK {
  int W, H;
  float fx, fy, mx, my;
  float depth_shift, dmin, dmax;
  float voxel, tbase, tscale, maxd;
  int wsample, wmax;
  uint32_t num_buckets, bucket_size, total_slots, num_blocks;
  // partition of one large scan over several GPUs: this fuser only allocates blocks it owns; slab_axis < 0 = no partition.
  //   slab_thick == 0: one contiguous slab, coordinate on slab_axis in [slab_lo, slab_hi)
  //   slab_thick  > 0: stripes of slab_thick block layers dealt round-robin from slab_lo: owner = floor((c - slab_lo) / thick) mod world
  //                    (every frame's blocks spread over all GPUs; a slab per GPU leaves all but one idle while the camera is elsewhere)
  int slab_axis, slab_lo, slab_hi, slab_thick, slab_world, slab_rank;
  // colour frames at their own resolution (cW == 0: same as depth)
  int cW, cH;
  float cfx, cfy, cmx, cmy;
  // input depth frames at inW x inH resampled (nearest) to W x H by the pre-pass; inW == 0: the frames are W x H already
  int inW, inH;
  float rsx, rsy;   // (inW - 1) / (W - 1), (inH - 1) / (H - 1)
  // upstream-conformance switches (scanfuse.h sf_params; 0 everywhere = SURVEY App. C)
  int frustum_mode, colour_round, colour_first, weight_mode;
};

__host__ __device__ inline bool slab_owns_coord(const ParamsK& P, int c) {
  if (P.slab_thick <= 0) return c >= P.slab_lo && c < P.slab_hi;
  const int d = c - P.slab_lo;
  const int q = d >= 0 ? d / P.slab_thick : -((P.slab_thick - 1 - d) / P.slab_thick);   // floor division
  int m = q % P.slab_world;
  if (m < 0) m += P.slab_world;
  return m == P.slab_rank;
}
__host__ __device__ inline bool slab_owns(const ParamsK& P, int bx, int by, int bz) {
  if (P.slab_axis < 0) return true;
  return slab_owns_coord(P, P.slab_axis == 0 ? bx : (P.slab_axis == 1 ? by : bz));
}
// lowest layer of one of this fuser's slabs / stripes: owned, the block below it on the partition axis is somebody else's
__host__ __device__ inline bool slab_boundary(const ParamsK& P, int bx, int by, int bz) {
  if (P.slab_axis < 0) return false;
  const int c = P.slab_axis == 0 ? bx : (P.slab_axis == 1 ? by : bz);
  return slab_owns_coord(P, c) && !slab_owns_coord(P, c - 1);
}
// what this fuser needs from the others before meshing: their boundary blocks that sit right above one of its own layers
__host__ __device__ inline bool slab_wants_ghost(const ParamsK& P, int bx, int by, int bz) {
  if (P.slab_axis < 0) return false;
  const int c = P.slab_axis == 0 ? bx : (P.slab_axis == 1 ? by : bz);
  return !slab_owns_coord(P, c) && slab_owns_coord(P, c - 1);
}

struct FrameK {
  float T[12];
  float Ti[12];
  float xa[2], xc[2], xr[2];
  float ya[2], yc[2], yr[2];
  float radius, zfar;
};

// Per-batch kernel arguments (passed by value: kernarg segment, read with scalar loads).
struct BatchIn {  // k_prepass
  const uint16_t* depth[MAX_BATCH];
  const uint8_t* rgb[MAX_BATCH];   // RGB8 pixels -- or, where lay[j] is set, the component planes k_jpeg_idct left of a JPEG picture
  const uint8_t* lay[MAX_BATCH];   // nullptr, or the picture's SfJpegLayout (device): the pre-pass upsamples and converts the pixels it looks up itself
};
struct BatchFrames {  // k_alloc, k_compactify
  int n;
  uint32_t seq0;  // sequence number of frame 0 of the batch; frame j has seq0 + j
  FrameK f[MAX_BATCH];
};
struct BatchTi {  // k_integrate: world -> camera rows 0..2 of every frame of the batch
  float Ti[MAX_BATCH][12];
};
// Kernel-argument segments: HIP's portability note names 4 KB (CUDA's limit); the AMD runtime sizes the kernarg segment from the kernel's own
// metadata, and k_alloc_ray / k_compactify take BatchFrames (4 872 B) + ParamsK + pointers = ~5.2 KB by value on gfx950 under ROCm 7.2 (every
// -m gpu test launches them).  The bound below is what this code relies on having been tested; a toolchain with a smaller limit fails the launch
// loudly (hipErrorInvalidValue -> SF_ERR_DEVICE), it does not truncate.
static_assert(sizeof(BatchFrames) == 8 + MAX_BATCH * sizeof(FrameK) && sizeof(BatchFrames) + 512 <= 6144, "BatchFrames kernarg grew: re-test the launch or move FrameK to a device buffer");
static_assert(sizeof(BatchIn) <= 768 && sizeof(BatchTi) <= 1536, "kernarg structs");

enum Counter {
  C_HEAP_FREE = 0,
  C_HIGH_WATER = 2,
  C_ALLOC_FAIL = 3,
  C_SLOTS_USED = 4,
  C_LAST_BLOCKS = 5,
  C_GC_FREED = 7,
  C_IMPORTED = 8,
  // 64-bit compaction counters (8-byte aligned, own cache line): low word = entries in the compact list,
  // high word = blocks the LAST frame of the batch updates
  C_COMPACT = 16,
  C_COMPACT_B = 18,  // second batch slot (batches alternate between two sets of per-batch buffers)
  C_EXPORT = 20,
  C_TOTAL_LO = 32,   // 64-bit sum of N_blk over all frames lives in counters[32..33] (own cache line)
  C_TILES_LO = 34,   // 64-bit sum over the passes of the tiles each pass touched (the pass's list length): counters[34..35], same line
  // the allocation kernels' two statistics, one atomic per wave each -- in the line of the compaction's sums (another kernel's, later in the stream), NOT in the
  // first line: an atomic there, returning or not, takes its turn with the heap pops every claiming wave waits for (the probe count in word 10 cost a 20-frame
  // call into an empty volume 2 %: k_alloc_ray 121 -> 131 us per launch)
  C_ALLOC_DIRECT = 40,  // blocks an allocation workgroup could not queue in LDS and took to the global table one by one (the slow path)
  C_ALLOC_PROBED = 41,  // look-ups of the allocation kernels that went to the hash table (not answered by the presence cache)
  C_COUNT = 48
};

__host__ __device__ inline uint64_t pack_key(int x, int y, int z) {
  return (((uint64_t)x & 0x1FFFFFull) << 42) | (((uint64_t)y & 0x1FFFFFull) << 21) | ((uint64_t)z & 0x1FFFFFull);
}
__host__ __device__ inline void unpack_key(uint64_t k, int& x, int& y, int& z) {
  x = ((int)((k >> 42) & 0x1FFFFF) << 11) >> 11;
  y = ((int)((k >> 21) & 0x1FFFFF) << 11) >> 11;
  z = ((int)(k & 0x1FFFFF) << 11) >> 11;
}

__device__ inline uint32_t hash_bucket(int x, int y, int z, uint32_t num_buckets) {
  const uint32_t h = ((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u);
  return h % num_buckets;
}
// Where a block's probe sequence starts in the table (open addressing over all num_buckets x bucket_size slots, linear from here): every block hashed on its
// own.  (Round 6 tried a brick-local start -- the 64 blocks of a 4x4x4 brick at 64 consecutive slots, so that the ~1 300 blocks a pixel tile names per frame at
// 1 mm fall on a few hundred cache lines: linear probing over clustered starts makes long runs wherever two bricks meet, and every look-up got slower:
// one frame per launch 10.2 k -> 8.8 k frames/s at 4 mm, 324 -> 260 at 1 mm.  profiles/r06_alloc_1mm.txt)
__device__ inline uint32_t hash_home(const ParamsK& P, int x, int y, int z) { return hash_bucket(x, y, z, P.num_buckets) * P.bucket_size; }

// ---------------------------------------------------------------------------------------------------
// Presence cache of the allocation kernels ("bricks", round 6).  A frame names every block of its truncation band -- 1.55 M at 1 mm voxels -- and all but the
// ~1 % the camera's motion adds exist already; finding that out cost one probe of the hash table per block and frame: random 16-byte reads of a 671 MB table,
// 1.7 of k_alloc<6>'s 2.1 ms (profiles/r06_alloc_1mm.txt).  The cache answers "this block is in the table, and was before this batch began" from a structure
// whose footprint per frame is ~100 x smaller: one 16-byte entry {tag, 64-bit mask} per 4x4x4-block BRICK, direct-mapped, the 8 x-consecutive bricks of a
// row in one 128-byte line (a word of the cube window's occupancy bitmap = 32 x-consecutive blocks = that row: one or two lines answer the whole word).
//   * it only ever says "present" for a block some lane FOUND in the table with a birth frame before its own batch (hash_find_or_claim): such a block needs
//     neither a slot nor a birth update from any later frame, so skipping the probe changes nothing -- the allocated set and every birth frame stay the
//     same, bit for bit (tests/test_gpu_tsdf.py runs every allocation test with the cache on and off);
//   * within a launch a tag goes 0 -> brick once (CAS) and never changes, mask bits are only ever set, and only behind a matching tag: a reader that sees a
//     stale line misses and probes the table as before; conflicts (another brick owns the entry) are simply not cached;
//   * whatever removes entries from the table (garbage collection, reset) clears the cache in stream order.
// ---------------------------------------------------------------------------------------------------
struct BrickCache {
  unsigned long long* e;   // {tag, mask} pairs; nullptr: no cache
  uint32_t line_mask;      // lines - 1 (a line = 8 entries)
};
#ifdef __HIPCC__   // (the host-only sanitizer builds of tools/tsan include this header with g++)
constexpr unsigned long long BRICK_TAG = 1ull << 63;   // pack_key uses 63 bits: a live tag is never 0, the cleared state
__device__ inline unsigned long long brick_tag(int X, int Y, int Z) { return pack_key(X, Y, Z) | BRICK_TAG; }
__device__ inline uint32_t brick_index(uint32_t line_mask, int X, int Y, int Z) {
  uint32_t h = ((uint32_t)(X >> 3) * 73856093u) ^ ((uint32_t)Y * 19349669u) ^ ((uint32_t)Z * 83492791u);
  h ^= h >> 15;
  return ((h & line_mask) << 3) | ((uint32_t)X & 7u);
}
__device__ inline uint32_t brick_bit(int bx, int by, int bz) { return (uint32_t)((bx & 3) | ((by & 3) << 2) | ((bz & 3) << 4)); }
// "block (bx, by, bz) is in the table and was born before this batch"
__device__ inline bool brick_known(const BrickCache& c, int bx, int by, int bz) {
  const int X = bx >> 2, Y = by >> 2, Z = bz >> 2;
  const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(c.e + 2 * (size_t)brick_index(c.line_mask, X, Y, Z));
  return v.x == brick_tag(X, Y, Z) && ((v.y >> brick_bit(bx, by, bz)) & 1ull) != 0ull;
}
// the same for the 32 x-consecutive blocks bx0 .. bx0 + 31 (bx0 a multiple of 4) of row (by, bz): bit i = block bx0 + i is known
__device__ inline uint32_t brick_known_row(const BrickCache& c, int bx0, int by, int bz) {
  const int X0 = bx0 >> 2, Y = by >> 2, Z = bz >> 2;
  const uint32_t off = (uint32_t)(((by & 3) << 2) | ((bz & 3) << 4));
  uint32_t known = 0u;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(c.e + 2 * (size_t)brick_index(c.line_mask, X0 + q, Y, Z));
    const uint32_t nib = v.x == brick_tag(X0 + q, Y, Z) ? (uint32_t)(v.y >> off) & 0xFu : 0u;
    known |= nib << (4 * q);
  }
  return known;
}
// a lane found the block in the table, born before this batch: remember it
__device__ inline void brick_note(const BrickCache& c, int bx, int by, int bz) {
  const int X = bx >> 2, Y = by >> 2, Z = bz >> 2;
  unsigned long long* e = c.e + 2 * (size_t)brick_index(c.line_mask, X, Y, Z);
  const unsigned long long tag = brick_tag(X, Y, Z);
  unsigned long long t = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t == 0ull) {
    t = atomicCAS(e, 0ull, tag);
    if (t == 0ull) t = tag;
  }
  if (t == tag) atomicOr(e + 1, 1ull << brick_bit(bx, by, bz));   // no return value wanted: fire and forget
}

#endif  // __HIPCC__

// lookup only: heap slot of block (x,y,z) or -1
__device__ inline int hash_lookup(const HashEntry* __restrict__ table, const ParamsK& P, int x, int y, int z) {
  const uint64_t key = pack_key(x, y, z);
  uint32_t slot = hash_home(P, x, y, z);
  for (int probe = 0; probe < MAX_PROBES; ++probe) {
    const uint64_t k = table[slot].key;
    if (k == key) return table[slot].ptr;
    if (k == KEY_EMPTY) return -1;
    slot++;
    if (slot == P.total_slots) slot = 0;
  }
  return -1;
}

#ifdef __HIPCC__
// Packed fp32 helpers and the two hand-expanded, correctly rounded divisions of k_integrate.
#ifdef SF_PACKED_PAIRS   // rounds 1-4: v_pk_fma / v_pk_mul / v_pk_add_f32 on the lane's voxel pair (build with -DSF_PACKED_PAIRS and without -fno-slp-vectorize)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline v2f splat(float x) { return (v2f){x, x}; }
// one v_pk_add_f32 (the compiler splits a packed add whose two results go separate ways into two v_add_f32)
__device__ inline v2f pk_add(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
// The default since round 5: the same pairs as two plain fp32 operations each.  tools/gpu/valu_peak.hip (profiles/r05_valu_issue_table.txt): on gfx950 a
// v_pk_*_f32 holds the SIMD for ~4.2 cycles and issues beside nothing; a plain v_fma / v_mul / v_add_f32 holds it ~2.2 cycles and issues beside the
// conversions, compares and selects of another wave -- packing buys no throughput on this part, and the splats cost moves.  Same arithmetic, same bits;
// the pass 830 -> 808 us, 9 registers fewer (5 waves per SIMD instead of 4).  fuser.hip is built with -fno-slp-vectorize so that the compiler does not
// pack the pairs again.
struct v2f {
  float x, y;
  __device__ float& operator[](int i) { return i ? y : x; }
  __device__ const float& operator[](int i) const { return i ? y : x; }
};
__device__ inline v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ inline v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ inline v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ inline v2f operator-(v2f a) { return v2f{-a.x, -a.y}; }
__device__ inline v2f pk_fma(v2f a, v2f b, v2f c) { return v2f{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
__device__ inline v2f splat(float x) { return v2f{x, x}; }
__device__ inline v2f pk_add(v2f a, v2f b) { return a + b; }
#endif
// RN(1 / b) for normal-range b: v_rcp_f32 seed (1 ulp) + two Newton steps
__device__ inline v2f recip_rn(v2f b) {
  v2f r = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
  const v2f one = splat(1.0f);
  r = pk_fma(pk_fma(-b, r, one), r, r);
  return pk_fma(pk_fma(-b, r, one), r, r);
}
// RN(1 / b) and RN(a / b) for normal-range scalars: the reciprocal as above; the quotient is the core of the correctly rounded sequence the
// compiler itself emits for a / b (q0 = a * y, two residual corrections) without its v_div_scale / v_div_fixup range handling -- given
// y = RN(1 / b) it costs 5 instructions where the full expansion costs ten and a quarter-rate v_rcp_f32.  k_alloc divides twelve times per
// ray (world -> block through / voxel, the DDA's tMax / tDelta through / direction); device self-test: sf_selftest_division.
__device__ inline float recip_rn(float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = fmaf(fmaf(-b, r, 1.0f), r, r);
  return fmaf(fmaf(-b, r, 1.0f), r, r);
}
__device__ inline float div_rn(float a, float b, float y) {
  const float q0 = a * y;
  const float q1 = fmaf(fmaf(-b, q0, a), y, q0);
  return fmaf(fmaf(-b, q1, a), y, q1);
}
// RN(n / m) given r = RN(1 / m): one Markstein correction (|n| >= 2^-100, m a small integer)
__device__ inline v2f quot_rn(v2f n, v2f m, v2f r) {
  const v2f q0 = n * r;
  return pk_fma(pk_fma(-m, q0, n), r, q0);
}
#endif

struct sf_fuser {
  sf_params p;
  ParamsK pk;
  int in_W = 0, in_H = 0;   // size of an INPUT depth frame as given to sf_fuser_create (p.depth_width / height hold the integration size pk.W x pk.H)
  size_t in_px = 0;
  int device = 0;
  hipStream_t stream = nullptr;  // integrate / deintegrate and everything synchronous
  hipStream_t front = nullptr;   // pre-pass, allocation, compaction of the NEXT frame (overlaps integrate)
  hipStream_t front_lo = nullptr;   // the same at the device's LOWEST priority: the front chain beside the persistent kernel of one frame per launch (sf_input_stream)
  hipStream_t last_front = nullptr; // which of the two the pass before used (a change of stream is ordered by ev_front_switch)
  hipEvent_t ev_front_switch = nullptr;
  int front_prio = -1;              // tune "front_prio"
  bool front_lo_lowest = true;      // tune "front_lo_lowest": front_lo at the device's lowest priority (default) or at the main stream's
  hipEvent_t ev_compact[2] = {nullptr, nullptr};   // front: frame slot ready for integrate
  hipEvent_t ev_fused[2] = {nullptr, nullptr};     // stream: frame slot consumed
  hipEvent_t ev_input = nullptr;                   // front: the caller's staging work queued so far (single-stream batches wait for it)
  int slot = 0;
  bool serial_tail = false;  // the most recent batches ran on `stream` alone (front has not been ordered behind them yet)
  bool overlap = true;  // sf_fuser_tune("overlap", 0) runs everything on one stream
  float* depthf2[2] = {nullptr, nullptr};      // MAX_BATCH x W*H per batch slot
  bool head_pass = false;   // set around the first run_batch of a multi-pass sf_fuser_integrate_batch_device call
  int alloc_group_head = 4; // tune "alloc_group_head": frames per allocation workgroup in that pass (0 = as every pass).  A 20-frame call: 30.8 k -> 31.8 k frames/s
  bool tail_pass = false;   // set around the last run_batch of a sf_fuser_integrate_batch_device call
  int tail_wide = 1;        // tune "tail_wide": that pass runs the 8-waves-per-SIMD variant of k_integrate
  bool xrow = true;         // tune "xrow": passes of several frames run k_integrate in the x-row lane layout (a lane = one x-row of the block: fuse_project_xr)
  uint2* color2[2] = {nullptr, nullptr};       // MAX_BATCH x W*H {depth bits, rgb} texels per batch slot (RGB-D batches)
  int32_t* compact2[2] = {nullptr, nullptr};   // heap slots of the blocks some frame of the batch sees
  uint32_t* cmask2[2] = {nullptr, nullptr};    // per compact entry: bit j = frame j of the batch updates this block
  int32_t* block_entry = nullptr;              // directory: table index of the entry of the block in heap slot i
  uint8_t* block_flags = nullptr;              // directory: bit 0 = ghost (imported copy of a neighbour slab's block: read by meshing, never fused or meshed)
  uint32_t frame_seq = 1;                      // sequence number of the next frame
  int batch = DEFAULT_BATCH;                   // frames per pass (sf_fuser_tune "batch", 1..MAX_BATCH)
  HashEntry* table = nullptr;
  unsigned long long* bricks = nullptr;        // presence cache of the allocation kernels: brick_lines x 8 entries of {tag, mask} (BrickCache above)
  uint32_t brick_lines = 0;
  bool brick_on = true;                        // tune "brick_cache" 0: every look-up probes the table (rounds 1-5)
  int32_t* heap = nullptr;
  uint64_t* block_keys = nullptr;
  uint4* voxels = nullptr;
  int32_t* compact = nullptr;  // alias of compact2[0], used by the synchronous paths (export, GC)
  int32_t* counters = nullptr;
  float* ray_kx = nullptr;   // (x - mx) / fx per column, (y - my) / fy per row of the integration image (colour look-up of the pre-pass)
  float* ray_ky = nullptr;
  // host-buffer entry points (sf_fuser_integrate / deintegrate: a live stream hands one pageable frame per call): a ring of HOST_RING slots,
  // each a page-locked host copy + a device copy of one frame.  The call copies the caller's frame into the page-locked slot (so the caller's
  // buffer is free when the call returns), queues the H2D and the frame's kernels, and returns; a slot is reused HOST_RING frames later, behind
  // the events below -- no stream is drained per frame.
  static constexpr int HOST_RING = 3;
  void* staging_depth[HOST_RING] = {nullptr, nullptr, nullptr};  // device copies of host-supplied frames
  void* staging_rgb[HOST_RING] = {nullptr, nullptr, nullptr};
  void* pinned_depth[HOST_RING] = {nullptr, nullptr, nullptr};   // page-locked host copies
  void* pinned_rgb[HOST_RING] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_h2d[HOST_RING] = {nullptr, nullptr, nullptr};       // the slot's H2D copy has read the page-locked buffer
  hipEvent_t ev_consumed[HOST_RING] = {nullptr, nullptr, nullptr};  // the kernels that read the slot's device copy are queued behind this
  int host_slot = 0;
  uint64_t host_frames = 0;
  int32_t* host_mirror = nullptr;  // pinned, device-visible: N_blk of the most recent integrate
  int num_cus = 256;
  bool alloc_win64 = false;  // 64^3-block LDS window when a ray segment spans more than ~20 blocks
  bool prepass_fuse = true;  // one colourless frame per pass: k_alloc_ray converts the depth itself, no k_prepass launch (tune "prepass_fuse")
  int alloc_ablate = 0;      // measurement only (tune "alloc_ablate"): parts of k_alloc_ray switched off, the volume is WRONG with any bit set
  int alloc_wgs = 0;         // > 0: allocation workgroups per CU capped (LDS padding) so that the integrate kernel beside them keeps its waves (tune "alloc_wgs")
  bool alloc_ray = false;    // k_alloc_ray (occupancy bitmap in ray space: 16 x 16 blocks across the pixel tile's pencil of rays, 256 slabs along it) instead of the cube window
  bool xcd_walk = true;  // k_integrate: each XCD walks one contiguous eighth of the list (tune "xcd_walk" 0: plain grid-stride)
  int pipe_mode = 1;    // 1: colourless one-frame launches run k_integrate_pipe (tune "pipe" 0: k_integrate)
  int pipe_wgs = 3;     // persistent workgroups per CU of k_integrate_pipe (48 KiB of LDS each)
  int front_cus = 0;    // > 0: the front stream owns that many CUs, the main stream the others (tune "front_cus")
  int nt_mode = -1;     // k_integrate_pipe tile traffic non-temporal: -1 = when the previous pass's tiles exceed 512 MiB, 0 never, 1 always (tune "nt")
  bool pipe_beside = false;  // the latched decision for the next pass (see sf_single_stream_batch)
  int pipe_overlap = -1;  // the next frame's pre-pass / allocation / compaction on the front stream beside k_integrate_pipe: -1 = when the previous pass's tiles exceed 512 MiB, 0 never, 1 always (tune "pipe_overlap")
  int ramp = 8;         // > 0: the first pass of an integrate_batch call takes only that many frames (tune "ramp"; a 20-frame call: 29.3 k frames/s at 0, 30.0 k at 4, 30.7 k at 8)
  bool ramp_geo = true; // the passes behind the first double (ramp, 2 ramp, ... batch) instead of jumping to the batch size (tune "ramp_geo")
  int alloc_group_win64 = 1;   // the same for k_alloc<6> (voxels below ~1.6 mm: the 64^3-block window), tune "alloc_group_win64"
  int alloc_group = 16; // consecutive frames of a batch one k_alloc workgroup walks (tune "alloc_group"; 4 -> 16: 35.0 -> 35.8 k frames/s, the blocks a pixel tile queues are looked up in the table once per batch)
  int compact_grid = 1024;  // 1024 directory entries per workgroup, grid-stride beyond
  uint64_t frames_integrated = 0, frames_skipped = 0;
  bool profile = false;
  void* mc_bounce[2] = {nullptr, nullptr};            // page-locked bounce buffers of the mesh download (mc.hip), allocated on first use
  hipEvent_t mc_bounce_ev[2] = {nullptr, nullptr};
  double mc_timing[12] = {0};   // phases of the most recent sf_fuser_extract_mesh (mc.hip; sf_fuser_mc_timing)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
};


hipError_t sf_quiesce(sf_fuser* f);                 // drain both streams
void sf_run_resources_prepare(int device, size_t pinned_bytes, size_t device_bytes, size_t plan_bytes);   // pipeline.hip: the side streams and the pinned ring of sf_fuse_run, created in the background
bool sf_single_stream_batch(const sf_fuser* f, int n, bool color, int sign);   // run_batch keeps this batch on f->stream alone
hipStream_t sf_input_stream(const sf_fuser* f, int n, bool color, int sign);   // where the batch's frames must be staged
int sf_compact_live(sf_fuser* f, int32_t* n_out, int include_ghosts = 1);   // live heap slots -> f->compact, synchronous

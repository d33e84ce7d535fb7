// fuser.hip -- voxel-hash TSDF fusion for gfx950 (MI355X): depth pre-pass, sparse block allocation,
// frustum compaction, integrate / deintegrate, garbage collection, block export.
//
// Replaces the scene-representation part of the external DepthSensing.exe / FriedLiver.exe that the
// reference pipeline shells out to (Server/scan_processor.py:126,138); the arithmetic is the specification
// in DESIGN.md section 3, reproduced operation for operation so that the voxels are
// bit-identical to the CPU checker's.  Built with -ffp-contract=off: every fmaf() below is a deliberate
// fused multiply-add of the spec, nothing else is contracted.
//
// Data layout in HBM:
//   table      HashEntry[num_buckets * bucket_size]   16 B: {u64 key (3 x 21-bit block coords), i32 heap block, u32 birth frame}
//   block_keys u64[num_sdf_blocks]                    directory: key of the block living in heap slot i, or EMPTY
//   heap       i32[num_sdf_blocks] + free counter     free list of heap slots (wave-aggregated pops)
//   voxels     8 B x 512 x num_sdf_blocks             {f32 sdf; u8 r,g,b,weight}, one 4 KiB tile per block, index z*64+y*8+x
//   depthf     f32[B][W*H], texel {f32 depth, rgb8} [B][W*H]   pre-pass outputs of the B <= 32 frames of a batch (L2 / Infinity-Cache resident); the
//                                                             texel plane only for RGB-D batches: ONE 8-byte gather per voxel and frame fetches both
//   compact    i32[num_sdf_blocks] + u32 mask         heap slots of the blocks some frame of the batch sees; bit j = frame j updates it
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fuser_internal.h"
#include "jpeg_idct.h"

namespace {

// bounding sphere of the block against the four side planes and the z range of the frustum (frustum_mode 0), or -- frustum_mode 1,
// DESIGN 6b -- VoxelHashing's isSDFBlockInCameraFrustumApprox: the block centre projected, normalised device coordinates x 0.95 inside
// [-1, 1]^2 x [0, 1] with z normalised by the SENSOR depth range.  Every operation individually rounded, true divisions: oracle/tsdf_oracle.c
// block_in_frustum runs the same sequence.
__device__ inline bool block_in_frustum(const ParamsK& P, const FrameK& F, int bx, int by, int bz) {
  const float cx = ((float)(8 * bx) + 3.5f) * P.voxel;
  const float cy = ((float)(8 * by) + 3.5f) * P.voxel;
  const float cz = ((float)(8 * bz) + 3.5f) * P.voxel;
  const float px = fmaf(F.Ti[0], cx, fmaf(F.Ti[1], cy, fmaf(F.Ti[2], cz, F.Ti[3])));
  const float py = fmaf(F.Ti[4], cx, fmaf(F.Ti[5], cy, fmaf(F.Ti[6], cz, F.Ti[7])));
  const float pz = fmaf(F.Ti[8], cx, fmaf(F.Ti[9], cy, fmaf(F.Ti[10], cz, F.Ti[11])));
  if (P.frustum_mode == 1) {   // kernarg scalar: a uniform branch
    const float zn = ((pz - P.dmin) / (P.dmax - P.dmin)) * 0.95f;
    if (!(zn >= 0.0f && zn <= 1.0f) || !(pz > 0.0f)) return false;   // also every NaN
    const float u = (px * P.fx) / pz + P.mx;
    const float v = (py * P.fy) / pz + P.my;
    const float wm1 = (float)(P.W - 1), hm1 = (float)(P.H - 1);
    const float nx = ((2.0f * u - wm1) / wm1) * 0.95f;
    const float ny = ((hm1 - 2.0f * v) / hm1) * 0.95f;
    return nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f;
  }
  bool in = pz > -F.radius;
  in = in && (pz < F.zfar + F.radius);
  in = in && (fmaf(F.xa[0], px, F.xc[0] * pz) >= -F.xr[0]);
  in = in && (fmaf(F.xa[1], px, F.xc[1] * pz) >= -F.xr[1]);
  in = in && (fmaf(F.ya[0], py, F.yc[0] * pz) >= -F.yr[0]);
  in = in && (fmaf(F.ya[1], py, F.yc[1] * pz) >= -F.yr[1]);
  return in;
}

// rv = RN(1 / voxel): the division itself through div_rn (fuser_internal.h), bit for bit w / voxel
__device__ inline int world_to_block(float w, float voxel, float rv) {
  const float q = div_rn(w, voxel, rv);
  const int vi = (int)(q >= 0.0f ? q + 0.5f : q - 0.5f);
  return vi >> 3;
}

// min(a, b) as ONE v_min_f32: fminf() makes clang canonicalise both operands first (v_max_f32 x, x, x each -- three instructions per voxel
// where the spec's min needs one; 16 of the 267 VALU instructions of a lane's frame).  The operands here are never NaN (depths come from
// 16-bit integers), and on equal or infinite operands v_min_f32 and fminf agree.
__device__ inline float min_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// One pixel of a JPEG picture from its component planes (what k_jpeg_idct of jpeg_gpu.hip leaves): chroma upsampling and the fixed-point YCbCr -> RGB of
// jpeg_idct.h, the integer functions the host decoder and k_jpeg_rgb are built from -- the same bytes, for the pixels the pre-pass looks up only.
struct YccPicture {
  int ncomp, sx[3], sy[3], cw[3], ch[3], bw[3];
  const uint8_t* plane[3];
  __device__ YccPicture(const SfJpegLayout* __restrict__ L, const uint8_t* planes) {
    const int W = L->width, H = L->height;
    ncomp = L->ncomp;
    const uint8_t* q = planes;
    for (int c = 0; c < 3; c++) {
      const int cc = c < ncomp ? c : 0;
      sx[c] = L->hmax > L->h[cc] ? 2 : 1; sy[c] = L->vmax > L->v[cc] ? 2 : 1;
      cw[c] = (W + sx[c] - 1) >> (sx[c] - 1); ch[c] = (H * L->v[cc] + L->vmax - 1) >> (L->vmax - 1);
      bw[c] = L->bw[cc];
      plane[c] = q;
      if (c < ncomp) q += (size_t)L->bw[cc] * L->bh[cc];
    }
  }
  __device__ uint32_t pixel(int x, int y) const {   // r | g << 8 | b << 16
    uint8_t o[3];
    if (ncomp == 1) { o[0] = o[1] = o[2] = plane[0][(size_t)y * bw[0] + x]; }
    else {
      int v[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        __builtin_assume(sx[c] >= 1 && sx[c] <= 2 && sy[c] >= 1 && sy[c] <= 2);
        v[c] = sf_jpeg_upsample(plane[c], bw[c], cw[c], ch[c], sx[c], sy[c], x, y);
      }
      sf_jpeg_ycc_to_rgb(v[0], v[1], v[2], o);
    }
    return (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16);
  }
};

// ---------------------------------------------------------------------------------------------------
// K1: depth pre-pass.  u16 -> metres (sensorData.h:968-977: d = depth / depthShift, 0 invalid), range
// gate (zParametersScanNet.txt:34-35) -> -inf; optional rgb -> packed u32.  8 pixels per lane; blockIdx.y = frame
// of the batch (every frame of a batch is converted by ONE launch).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepass(BatchIn in, float* __restrict__ depthf_all, uint2* __restrict__ texel_all, int n,
                                                 float shift, float dmin, float dmax, int32_t* counters, int compact_counter, ParamsK P,
                                                 const float* __restrict__ ray_kx, const float* __restrict__ ray_ky) {
  const int j = blockIdx.y;  // frame of the batch
  const uint16_t* __restrict__ depth = in.depth[j];
  const uint8_t* __restrict__ rgb = in.rgb[j];
  float* __restrict__ depthf = depthf_all + (size_t)j * n;
  // RGB-D: the frame's pixels once more as 8-byte texels {depth as the float's bits, rgb in the low three bytes}: the integrate kernel gathers a
  // voxel's depth AND colour with one request (round 4: two 4-byte gathers per voxel and frame kept the CU's texture-address unit busy 82 % of a pass)
  uint2* __restrict__ texel = texel_all + (size_t)j * n;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (blockIdx.x == 0 && j == 0 && threadIdx.x == 0) {
    atomicExch(reinterpret_cast<unsigned long long*>(&counters[compact_counter]), 0ull);
  }
  const bool ycc = rgb != nullptr && in.lay[j] != nullptr;   // uniform
  if (i0 >= n && !ycc) return;   // (the planes' look-ups below are dealt out across the whole workgroup)
  uint16_t u[8];
  if (P.inW > 0) {
    // s_integrationWidth / Height: nearest resample of the inW x inH input (scanfuse.h sf_params::integration_width)
    for (int k = 0; k < 8; k++) {
      const int i = i0 + k;
      if (i >= n) { u[k] = 0; continue; }
      const unsigned xi = (unsigned)((float)(i % P.W) * P.rsx + 0.5f), yi = (unsigned)((float)(i / P.W) * P.rsy + 0.5f);
      u[k] = (xi < (unsigned)P.inW && yi < (unsigned)P.inH) ? depth[(size_t)yi * P.inW + xi] : (uint16_t)0;
    }
  } else if (i0 + 8 <= n) {
    const uint4 raw = *reinterpret_cast<const uint4*>(depth + i0);
    u[0] = raw.x & 0xFFFF; u[1] = raw.x >> 16; u[2] = raw.y & 0xFFFF; u[3] = raw.y >> 16;
    u[4] = raw.z & 0xFFFF; u[5] = raw.z >> 16; u[6] = raw.w & 0xFFFF; u[7] = raw.w >> 16;
  } else {
    for (int k = 0; k < 8; k++) u[k] = (i0 + k < n) ? depth[i0 + k] : (uint16_t)0;
  }
  float d[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float v = (float)u[k] / shift;
    if (u[k] == 0 || v < dmin || v > dmax) v = -INFINITY;
    d[k] = v;
  }
  if (i0 + 8 <= n) {
    *reinterpret_cast<float4*>(depthf + i0) = make_float4(d[0], d[1], d[2], d[3]);
    *reinterpret_cast<float4*>(depthf + i0 + 4) = make_float4(d[4], d[5], d[6], d[7]);
  } else {
    for (int k = 0; k < 8 && i0 + k < n; k++) depthf[i0 + k] = d[k];
  }
  if (ycc) {
    // a JPEG picture as component planes: the pixel under each depth pixel (its own, or -- colour at its own resolution -- the one under the depth pixel's ray,
    // the look-up below) is upsampled and converted here.  Consecutive LANES take consecutive pixels for this part (the depths change hands through LDS): a
    // wave's look-ups then fall on one or two rows of each plane and its texel stores are whole 512-byte runs; with the lane's own eight consecutive pixels
    // every byte load of a wave touched 64 different cache lines (k_prepass 200 -> 440 us per 32-frame batch beside the fusion).
    __shared__ float s_d[2048];
#pragma unroll
    for (int k = 0; k < 8; k++) s_d[threadIdx.x * 8 + k] = d[k];
    __syncthreads();
    const YccPicture pic(reinterpret_cast<const SfJpegLayout*>(in.lay[j]), rgb);
    const int wg0 = blockIdx.x * 2048;
#pragma unroll 2
    for (int k = 0; k < 8; k++) {
      const int p = wg0 + k * 256 + (int)threadIdx.x;
      if (p >= n) break;
      const int y = p / P.W, x = p - y * P.W;
      uint32_t c = 0u;
      if (P.cW == 0) c = pic.pixel(x, y);
      else {
        const float u = fmaf(ray_kx[x], P.cfx, P.cmx) + 0.5f;
        const float v = fmaf(ray_ky[y], P.cfy, P.cmy) + 0.5f;
        if (u >= 0.0f && u < (float)P.cW && v >= 0.0f && v < (float)P.cH) c = pic.pixel((int)u, (int)v);
      }
      texel[p] = make_uint2(__float_as_uint(s_d[k * 256 + (int)threadIdx.x]), c);
    }
  } else if (rgb) {
    if (P.cW == 0) {
      // colour at depth resolution: the lane's 8 pixels are 24 contiguous bytes = three 8-byte loads (24 * lane is 8-byte aligned when the
      // image base is), repacked to one dword per pixel
      if (i0 + 8 <= n && ((uintptr_t)rgb & 7) == 0) {
        const uint2* q = reinterpret_cast<const uint2*>(rgb + 3 * (size_t)i0);
        const uint2 a = q[0], b = q[1], c = q[2];
        const uint32_t w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
        uint32_t px[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int bit = 24 * k, lo = bit >> 5, sh = bit & 31;   // three bytes starting at bit 24 k of the 192-bit run
          const uint64_t two = (uint64_t)w[lo] | ((uint64_t)(lo + 1 < 6 ? w[lo + 1] : 0u) << 32);
          px[k] = (uint32_t)(two >> sh) & 0xFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 8; k += 2)
          *reinterpret_cast<uint4*>(texel + i0 + k) = make_uint4(__float_as_uint(d[k]), px[k], __float_as_uint(d[k + 1]), px[k + 1]);
      } else {
        for (int k = 0; k < 8 && i0 + k < n; k++) {
          const uint8_t* c = rgb + 3 * (size_t)(i0 + k);
          texel[i0 + k] = make_uint2(__float_as_uint(d[k]), (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16));
        }
      }
    } else {
      // colour image at its own resolution: the colour pixel under the depth pixel's ray (nearest), black outside.  The ray slopes
      // (x - mx) / fx and (y - my) / fy depend on the column / row only: they come from the tables k_ray_tables filled once with the
      // same IEEE divisions (round-1 code divided twice per pixel: 58 us per 16-frame batch against 8 us without colour).
      for (int k = 0; k < 8 && i0 + k < n; k++) {
        const int x = (i0 + k) % P.W, y = (i0 + k) / P.W;
        const float u = fmaf(ray_kx[x], P.cfx, P.cmx) + 0.5f;
        const float v = fmaf(ray_ky[y], P.cfy, P.cmy) + 0.5f;
        if (!(u >= 0.0f && u < (float)P.cW && v >= 0.0f && v < (float)P.cH)) { texel[i0 + k] = make_uint2(__float_as_uint(d[k]), 0u); continue; }
        const uint8_t* c = rgb + 3 * ((size_t)(int)v * (size_t)P.cW + (size_t)(int)u);
        texel[i0 + k] = make_uint2(__float_as_uint(d[k]), (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16));
      }
    }
  }
}

// (x - mx) / fx per column and (y - my) / fy per row of the integration image: the ray slopes the colour look-up of k_prepass multiplies
__global__ void k_ray_tables(float* kx, float* ky, ParamsK P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.W) kx[i] = ((float)i - P.mx) / P.fx;
  if (i < P.H) ky[i] = ((float)i - P.my) / P.fy;
}

// ---------------------------------------------------------------------------------------------------
// K2: allocation.  One lane per depth pixel, one 256-thread workgroup per 16x16 pixel tile and per GROUP of
// consecutive frames of the batch (blockIdx.z): the same pixel tile of neighbouring frames looks at almost the
// same blocks, so the workgroup walks its frames in order and only the blocks a frame adds go any further.
//   phase 1, per frame (no global memory traffic except the 1 KiB of depth):
//            every lane walks its 3-D DDA over the blocks of [d - t, d + t] and sets ONE BIT per visited block in
//            an LDS occupancy bitmap of a WIN^3-block window anchored at the tile's first ray (non-returning
//            ds_or: no latency on the lane, duplicates across the 256 rays and across steps collapse for free);
//            then the bitmap words are scanned: bits not yet queued by an earlier frame of the group are
//            frustum-tested for THIS frame and queued as (key, frame).  Rays that leave the window (tiles
//            straddling a depth discontinuity) go through a small LDS hash set instead.
//   phase 2, once per workgroup: the queued keys are probed in the global hash table by all lanes in parallel
//            (one memory round trip instead of one per DDA step); an EMPTY slot is claimed with a lock-free
//            64-bit CAS, the entry's birth frame becomes the minimum over everybody who asked for the block, and
//            the freshly claimed slots of a wave receive their heap blocks through ONE wave-aggregated pop
//            (ballot + prefix popcount).
// The allocated SET and every block's birth frame are deterministic (no insertion ever gives up, so no fix-point
// iteration as upstream); which heap slot a block lands in is not (neither is it upstream).
// ---------------------------------------------------------------------------------------------------
constexpr int ALLOC_SET = 256;        // LDS hash-set slots per workgroup (2 KiB): blocks outside the window
constexpr int ALLOC_LIST = 512;       // queue of (key, frame) for phase 2 (4 KiB + 0.5 KiB); 14.5 KiB LDS per workgroup in all => 8 workgroups per CU
constexpr int ALLOC_SET_PROBES = 32;

struct HashRefs {
  HashEntry* table;
  int32_t* heap;
  uint64_t* block_keys;
  int32_t* block_entry;
  uint8_t* block_flags;
  int32_t* counters;
  BrickCache bricks;   // presence cache (fuser_internal.h); bricks.e == nullptr: none
  uint32_t seq0;       // sequence number of the batch's first frame: a block born before it is older than every frame that asks now
};

// A block is "born" in the first frame that asks for it: frames of one batch are allocated by ONE launch, so the
// entry keeps the minimum sequence number over everybody who found or claimed it (the frames before its birth
// must not update the block -- sequentially it did not exist yet).
__device__ inline uint32_t note_birth(HashEntry* e, uint32_t seq) {
  const uint32_t b = __hip_atomic_load(&e->birth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b > seq) atomicMin(&e->birth, seq);
  return b;
}

// find-or-claim `key`; returns the claimed entry (needs a heap block) or nullptr (already present / table full)
__device__ inline HashEntry* hash_find_or_claim(const HashRefs& h, const ParamsK& P, uint64_t key, int bx, int by, int bz, uint32_t seq, int probe0 = 0) {
  uint32_t slot = hash_home(P, bx, by, bz) + (uint32_t)probe0;   // (probe0 > 0: the caller has looked at the first probe0 slots itself)
  if (slot >= P.total_slots) slot -= P.total_slots;
  for (int probe = probe0; probe < MAX_PROBES; ++probe) {
    HashEntry* e = h.table + slot;
    const uint64_t k = __hip_atomic_load(&e->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) {
      // found, and born before this batch: no later frame has anything to do for this block -- the presence cache may say so from now on
      if (note_birth(e, seq) < h.seq0 && h.bricks.e != nullptr) brick_note(h.bricks, bx, by, bz);
      return nullptr;
    }
    if (k == KEY_EMPTY) {
      const uint64_t old = atomicCAS((unsigned long long*)&e->key, (unsigned long long)KEY_EMPTY, (unsigned long long)key);
      if (old == KEY_EMPTY) { atomicMin(&e->birth, seq); return e; }   // (ours: no need to look at the birth frame first -- one round trip less in a chain of four)
      if (old == key) { note_birth(e, seq); return nullptr; }
    }
    slot++;
    if (slot == P.total_slots) slot = 0;
  }
  atomicAdd(&h.counters[C_ALLOC_FAIL], 1);
  return nullptr;
}

// hands heap position `at` to the claimed entry; returns the block's index + 1 for the caller's high-water mark (0: heap exhausted)
__device__ inline int give_block_quiet(const HashRefs& h, HashEntry* e, uint64_t key, int at) {
  if (at >= 0) {
    const int idx = h.heap[at];
    e->ptr = idx;
    h.block_keys[idx] = key;
    h.block_entry[idx] = (int32_t)(e - h.table);
    h.block_flags[idx] = 0;
    return idx + 1;
  }
  // heap exhausted: the entry stays claimed without a block; undo the pop
  atomicAdd(&h.counters[C_HEAP_FREE], 1);
  atomicAdd(&h.counters[C_ALLOC_FAIL], 1);
  return 0;
}
// (no look at the mark first: a load of the word every workgroup's atomics land on waits in their queue like one of them, and the wave waits for IT -- measured
// on a 20-frame call into an empty volume, where every block is new: k_alloc_ray 114 -> 151 us per launch; the atomic without a return value costs the wave nothing)
__device__ inline void raise_high_water(const HashRefs& h, int hw) {
  if (hw > 0) atomicMax(&h.counters[C_HIGH_WATER], hw);
}
__device__ inline void give_block(const HashRefs& h, HashEntry* e, uint64_t key, int at) { raise_high_water(h, give_block_quiet(h, e, key, at)); }

// -DSF_ALLOC_TIMING (measurement build, tools/gpu/alloc_1mm_probe.py): where a workgroup of k_alloc spends its time -- thread 0 adds the 100 MHz clock between
// the barriers that separate the phases into g_alloc_t (sf_alloc_timing_read): [0] zeroing + ray set-up, [1] anchoring, [2] walk, [3] scan, [4] drain, [5] whole
// workgroup, [8] the longest workgroup, [9] rounds walked, [10] workgroups
#ifdef SF_ALLOC_TIMING
__device__ unsigned long long g_alloc_t[16];
__device__ unsigned int g_alloc_log[1 + 256 * 12];   // workgroups that took more than 250 us: [0] how many, then 12 words each (sf_alloc_timing_log)
#define AT_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_alloc_t[i], now_ - at_prev_); at_ph_[i] += (unsigned)(now_ - at_prev_); at_prev_ = now_; } } while (0)
#else
#define AT_MARK(i) do { } while (0)
#endif

template <int WIN_LOG2, bool MULTI>
__global__ __launch_bounds__(256) void k_alloc(const float* __restrict__ depthf_all, HashEntry* table, int32_t* heap,
                                               uint64_t* block_keys, int32_t* block_entry, uint8_t* block_flags, int32_t* counters, ParamsK P,
                                               BatchFrames B, int group_frames, BrickCache bricks) {
  constexpr int WIN = 1 << WIN_LOG2;                // window edge in blocks
  // the queue of a workgroup: at 1 mm voxels (WIN 64) a pixel tile's rays visit ~1 300 blocks per frame -- with the 512 entries that serve 4 mm ALL of them
  // overflowed into the one-by-one path (sf_fuser_alloc_direct_count: 1.3 M blocks per frame, k_alloc<6> 2.2 ms: tools/gpu/alloc_1mm_probe.py)
#ifndef SF_ALLOC6_LIST
#define SF_ALLOC6_LIST 4096
#endif
  constexpr int LIST = WIN_LOG2 >= 6 ? SF_ALLOC6_LIST : ALLOC_LIST;
  // (8 192 entries and a 2 048-slot set take the direct path from 1.3 M to 8 k blocks per frame and the kernel nowhere: its time is the table probes themselves,
  // profiles/r06_alloc_1mm.txt; 4 096 entries keep two workgroups per CU)
  constexpr int SET_LOG2 = 8, SET = 1 << SET_LOG2;
  constexpr int WIN_WORDS = (WIN * WIN * WIN) / 32; // occupancy bitmap words: 4 KiB (WIN 32) / 32 KiB (WIN 64)
  __shared__ uint32_t s_frame[WIN_WORDS];           // blocks the current frame's rays visit
  __shared__ uint32_t s_done[MULTI ? WIN_WORDS : 1];// blocks an earlier frame of the group has already queued
  __shared__ unsigned long long s_keys[SET];  // the same for blocks outside the window
  __shared__ unsigned long long s_list[LIST]; // queue for phase 2
  __shared__ uint8_t s_birth[LIST];           // ... and the frame (index in the batch) that queued the key
  __shared__ int s_count;
  __shared__ int s_chooser;
  __shared__ int s_anchored;
  __shared__ int s_anchor[3];
  __shared__ int s_box[6];
  __shared__ int s_claimed, s_pop_base;   // drain(): entries claimed by the workgroup in this call, and where its blocks start in the heap
  // the current frame's constants for the frustum tests of the scan (and of rays outside the window), two frames' worth so that a frame's copy never lands
  // under the previous frame's readers.  Read as B.f[j] they come through the scalar unit from the kernarg segment, a few words per load, each load a round
  // trip the wave waits for: at 1 mm voxels a tile names ~1 300 blocks per frame and a wave of k_alloc<6> spent its life -- 610 scalar loads, three quarters
  // of its cycles waiting (profiles/r06_pmc_alloc_1mm.txt) -- in that chain
  __shared__ uint32_t s_fk[2][sizeof(FrameK) / 4];
#ifdef SF_ALLOC_TIMING
  unsigned long long at_prev_ = wall_clock64();
  const unsigned long long at_start_ = at_prev_;
  unsigned at_ph_[5] = {0, 0, 0, 0, 0}, at_rounds_ = 0, at_queued_ = 0;
#endif
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_count = 0; s_chooser = 256; s_anchored = 0; }
  if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? INT_MAX : INT_MIN;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const HashRefs h{table, heap, block_keys, block_entry, block_flags, counters, bricks, B.seq0};
  for (int i = threadIdx.x; i < SET; i += 256) s_keys[i] = KEY_EMPTY;
  if (MULTI)
    for (int i = threadIdx.x; i < WIN_WORDS; i += 256) s_done[i] = 0u;
  const size_t npx = (size_t)P.W * P.H;
  const int j_begin = blockIdx.z * group_frames;
  const int j_end = min(B.n, j_begin + group_frames);

  // a block the workgroup cannot queue (queue full: pathological tile) goes straight to the global table
  int n_direct = 0;   // sf_fuser_alloc_direct_count (added up once per wave at the end: one atomic per call on a single word halved the 1 mm front chain)
  int n_probed = 0;   // look-ups that went to the hash table (the presence cache did not answer): sf_fuser_alloc_probe_count
  auto direct = [&](uint64_t key, int bx, int by, int bz, uint32_t seq) {
    n_direct++;
    if (h.bricks.e != nullptr && brick_known(h.bricks, bx, by, bz)) return;
    n_probed++;
    HashEntry* e = hash_find_or_claim(h, P, key, bx, by, bz, seq);
    if (e) {
      atomicAdd(&counters[C_SLOTS_USED], 1);
      give_block(h, e, key, atomicSub(&counters[C_HEAP_FREE], 1) - 1);
    }
  };

  // ---- phase 2: queued keys -> global hash, all lanes in parallel (callers put a barrier between the last queue write and this; every thread calls it)
  // The heap is popped ONCE per workgroup and call: the entries the lanes claimed are first packed into LDS (their table slots, 4 bytes each, over the keys
  // already read), then one atomic on the heap's free count serves them all.  Popped per wave and iteration -- and the high-water mark raised per lane --
  // the three words every workgroup of the launch shares were what a tile of a newly seen surface waited for: at 1 mm voxels ~1 900 new blocks, 8 iterations,
  // 25 us each; such workgroups (3 % of them) took 200 - 800 us where the mean is 59, and the longest one IS the kernel (profiles/r06_alloc_1mm.txt).
  auto drain = [&]() {
#ifdef SF_ABLATE_ALLOC_PHASE2   // measurement only (the volume is WRONG): no table probes
    const int n_unique = 0;
#else
    const int n_unique = min(s_count, LIST);
#endif
    uint32_t* const s_ent = reinterpret_cast<uint32_t*>(s_list);
    if (threadIdx.x == 0) s_claimed = 0;
    // DU keys per lane and iteration, their first probes side by side: the table is 16-byte entries scattered over hundreds of megabytes, a look-up is a chain
    // of round trips (the key, the compare-and-swap, the birth frame), and a chain at a time kept a tile of ~4 000 new blocks 13 us per 256 keys
    constexpr int DU = 4;
    for (int i0 = 0; i0 < n_unique; i0 += 256 * DU) {
      uint64_t key[DU], k0[DU];
      uint32_t seq[DU];
      HashEntry* e0[DU];
      HashEntry* claimed[DU];
      bool live[DU], won[DU];
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int i = i0 + u * 256 + (int)threadIdx.x;
        key[u] = i < n_unique ? s_list[i] : KEY_EMPTY;
        const uint32_t bi = i < n_unique ? s_birth[i] : 0u;   // bit 7: queued by a ray outside the window -- the presence cache has not been asked about this block yet
        seq[u] = B.seq0 + (bi & 0x7Fu);
        live[u] = key[u] != KEY_EMPTY;
        claimed[u] = nullptr;
        int bx, by, bz;
        unpack_key(key[u], bx, by, bz);
        if (live[u] && (bi & 0x80u) != 0u && h.bricks.e != nullptr && brick_known(h.bricks, bx, by, bz)) live[u] = false;   // (the scan queues only what the cache does not know)
        e0[u] = h.table + hash_home(P, bx, by, bz);
        k0[u] = live[u] ? __hip_atomic_load(&e0[u]->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (live[u]) n_probed++;
      }
      __syncthreads();   // the keys of this iteration are in registers: the packed entries (never more than the keys read so far, half their size) may grow over them
#pragma unroll
      for (int u = 0; u < DU; u++)   // an empty home slot: try to take it
        won[u] = live[u] && k0[u] == KEY_EMPTY && atomicCAS((unsigned long long*)&e0[u]->key, (unsigned long long)KEY_EMPTY, (unsigned long long)key[u]) == KEY_EMPTY;
#pragma unroll
      for (int u = 0; u < DU; u++) {
        if (!live[u]) continue;
        if (won[u]) { atomicMin(&e0[u]->birth, seq[u]); claimed[u] = e0[u]; }   // taken
        else {   // somebody else's, ours already, or lost the race for it: the general walk from the home slot (one entry it has seen before, rarely)
          int bx, by, bz;
          unpack_key(key[u], bx, by, bz);
          claimed[u] = hash_find_or_claim(h, P, key[u], bx, by, bz, seq[u]);
        }
      }
      uint64_t cm[DU];
      int n_wave = 0;
#pragma unroll
      for (int u = 0; u < DU; u++) { cm[u] = __ballot(claimed[u] != nullptr); n_wave += __popcll((unsigned long long)cm[u]); }
      if (n_wave != 0) {
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(&s_claimed, n_wave);
        wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
        for (int u = 0; u < DU; u++) {
          if (claimed[u] != nullptr) s_ent[wbase + __popcll((unsigned long long)(cm[u] & ((1ull << lane) - 1ull)))] = (uint32_t)(claimed[u] - h.table);
          wbase += __popcll((unsigned long long)cm[u]);
        }
      }
    }
    __syncthreads();
    const int n_claimed = s_claimed;
    if (n_claimed == 0) return;   // (uniform)
    if (threadIdx.x == 0) {
      s_pop_base = atomicSub(&counters[C_HEAP_FREE], n_claimed);
      atomicAdd(&counters[C_SLOTS_USED], n_claimed);
    }
    __syncthreads();
    const int base = s_pop_base;
    int hw = 0;
    for (int i0 = 0; i0 < n_claimed; i0 += 256 * DU) {
      HashEntry* e[DU];
      uint64_t key[DU];
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int i = i0 + u * 256 + (int)threadIdx.x;
        e[u] = i < n_claimed ? h.table + s_ent[i] : nullptr;
        key[u] = e[u] != nullptr ? __hip_atomic_load(&e[u]->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < DU; u++)
        if (e[u] != nullptr) hw = max(hw, give_block_quiet(h, e[u], key[u], base - 1 - (i0 + u * 256 + (int)threadIdx.x)));
    }
    for (int o = 32; o > 0; o >>= 1) hw = max(hw, __shfl_xor(hw, o));
    if (lane == 0) raise_high_water(h, hw);
  };

#ifndef SF_ALLOC6_ROUNDS
#define SF_ALLOC6_ROUNDS 8
#endif
  constexpr int ROUNDS = (WIN_LOG2 >= 6 && !MULTI) ? SF_ALLOC6_ROUNDS : 1;   // windows a frame's rays may be walked in before the slow path (below)
  const bool in_image = x < P.W && y < P.H;
  const float kx = ((float)x - P.mx) / P.fx, ky = ((float)y - P.my) / P.fy;  // the pixel's ray direction is the same for every frame
  const float rvoxel = 1.0f / P.voxel;                                        // RN(1 / voxel) for world_to_block
  float d_next = in_image && j_begin < j_end ? depthf_all[(size_t)j_begin * npx + (size_t)(y * P.W + x)] : -INFINITY;
  for (int j = j_begin; j < j_end; ++j) {
    const FrameK& F = B.f[j];  // uniform index: scalar loads from the kernarg segment
    const float d_cur = d_next;
    // the next frame's depth is requested now and lands while this frame's rays are walked
    d_next = in_image && j + 1 < j_end ? depthf_all[(size_t)(j + 1) * npx + (size_t)(y * P.W + x)] : -INFINITY;
    for (int i = threadIdx.x; i < WIN_WORDS; i += 256) s_frame[i] = 0u;
    if (threadIdx.x < sizeof(FrameK) / 4) s_fk[j & 1][threadIdx.x] = reinterpret_cast<const uint32_t*>(&B.f[j])[threadIdx.x];   // per-lane words: vector loads
    const FrameK& FL = *reinterpret_cast<const FrameK*>(s_fk[j & 1]);   // valid behind the barrier below

    // ---- ray set-up
    bool active = false;
    int a_cx = 0, a_cy = 0, a_cz = 0, a_sx = 0, a_sy = 0, a_sz = 0, a_ex = 0, a_ey = 0, a_ez = 0;
    float a_tmx = INFINITY, a_tmy = INFINITY, a_tmz = INFINITY, a_tdx = INFINITY, a_tdy = INFINITY, a_tdz = INFINITY;
    if (in_image) {
      const float d = d_cur;
      if (d != -INFINITY && d < P.maxd) {
        const float t = fmaf(P.tscale, d, P.tbase);
        const float lo = min_f32(P.maxd, d - t);
        const float hi = min_f32(P.maxd, d + t);
        if (lo < hi) {
          float p0[3], p1[3];
          {
            const float ax = kx * lo, ay = ky * lo, az = lo;
#pragma unroll
            for (int r = 0; r < 3; r++) p0[r] = fmaf(F.T[4 * r], ax, fmaf(F.T[4 * r + 1], ay, fmaf(F.T[4 * r + 2], az, F.T[4 * r + 3])));
          }
          {
            const float ax = kx * hi, ay = ky * hi, az = hi;
#pragma unroll
            for (int r = 0; r < 3; r++) p1[r] = fmaf(F.T[4 * r], ax, fmaf(F.T[4 * r + 1], ay, fmaf(F.T[4 * r + 2], az, F.T[4 * r + 3])));
          }
          const float bsize = 8.0f * P.voxel;
          int cur[3], stp[3], bnd[3];
          float tm[3], td[3];
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float dir = p1[c] - p0[c];
            cur[c] = world_to_block(p0[c], P.voxel, rvoxel);
            const int e = world_to_block(p1[c], P.voxel, rvoxel);
            stp[c] = dir > 0.0f ? 1 : (dir < 0.0f ? -1 : 0);
            bnd[c] = e + stp[c];
            if (stp[c] == 0) { tm[c] = INFINITY; td[c] = INFINITY; }
            else {
              const int nb = cur[c] + (stp[c] > 0 ? 1 : 0);
              const float plane = ((float)(8 * nb) - 0.5f) * P.voxel;
              const float rdir = recip_rn(dir);   // one reciprocal for both quotients
              tm[c] = div_rn(plane - p0[c], dir, rdir);
              td[c] = div_rn((float)stp[c] * bsize, dir, rdir);
            }
          }
          a_cx = cur[0]; a_cy = cur[1]; a_cz = cur[2];
          a_sx = stp[0]; a_sy = stp[1]; a_sz = stp[2]; a_ex = bnd[0]; a_ey = bnd[1]; a_ez = bnd[2];
          a_tmx = tm[0]; a_tmy = tm[1]; a_tmz = tm[2]; a_tdx = td[0]; a_tdy = td[1]; a_tdz = td[2];
          active = true;
        }
      }
    }
    // ROUNDS > 1 (the 64^3 window, one frame per workgroup): rays that leave the window are not taken through the slow path at once -- the window is laid
    // out again around THEM and they walk again, up to ROUNDS times.  A pixel tile on a depth discontinuity has two clusters of rays metres apart; one window
    // holds one of them, and the other's ~5 000 block visits went one by one through a 256-slot LDS set and then the global table (profiles/r06_alloc_1mm.txt:
    // ~35 such tiles per frame set the kernel's duration).  A block two rounds name is queued twice and found the second time: the set is the same.
    bool pending = active;
#pragma unroll 1
    for (int round = 0; round < ROUNDS; ++round) {
      if (round > 0) {   // behind drain()'s barrier: nobody reads the previous round's window any more
        if (threadIdx.x == 0) { s_chooser = 256; s_anchored = 0; }
        if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? INT_MAX : INT_MIN;
        for (int i = threadIdx.x; i < WIN_WORDS; i += 256) s_frame[i] = 0u;
      }
      __syncthreads();  // s_frame zeroed, previous frame's scan finished
      AT_MARK(0);
      // The tile's rays stay inside a small region of block space: the first active lane of the first frame that has
      // one anchors the WIN^3 window there for the whole group.
      if (s_anchored == 0) {
        if (pending) atomicMin(&s_chooser, (int)threadIdx.x);
        if (WIN_LOG2 >= 6) {
          // the box around every ray segment of the tile (first and last block per axis): where it fits, the window is centred on it.  Anchored on the first
          // active ray alone (WIN / 4 blocks behind its start), a tile whose other rays start 16 blocks nearer -- 13 cm at 1 mm voxels -- loses those rays
          int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
          if (pending) {
            const int ex = a_ex - a_sx, ey = a_ey - a_sy, ez = a_ez - a_sz;   // the last block of the walk
            lo[0] = min(a_cx, ex); hi[0] = max(a_cx, ex);
            lo[1] = min(a_cy, ey); hi[1] = max(a_cy, ey);
            lo[2] = min(a_cz, ez); hi[2] = max(a_cz, ez);
          }
#pragma unroll
          for (int c = 0; c < 3; c++) {
            for (int o = 32; o > 0; o >>= 1) { lo[c] = min(lo[c], __shfl_xor(lo[c], o)); hi[c] = max(hi[c], __shfl_xor(hi[c], o)); }
            if (lane == 0 && lo[c] <= hi[c]) { atomicMin(&s_box[c], lo[c]); atomicMax(&s_box[3 + c], hi[c]); }
          }
        }
        __syncthreads();
        if ((int)threadIdx.x == s_chooser) {
          // (a multiple of 4 in x: a word of the bitmap is then 8 whole bricks of the presence cache; where the window lies never changes WHAT is allocated)
          int an[3] = {a_cx - (a_sx >= 0 ? WIN / 4 : 3 * WIN / 4), a_cy - (a_sy >= 0 ? WIN / 4 : 3 * WIN / 4), a_cz - (a_sz >= 0 ? WIN / 4 : 3 * WIN / 4)};
          if (WIN_LOG2 >= 6) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const int ext = s_box[3 + c] - s_box[c] + 1;
              if (ext <= WIN) an[c] = s_box[c] - (WIN - ext) / 2;   // else: clusters of rays more than a window apart -- the first ray's stays, the rest is the next round's
            }
          }
          s_anchor[0] = an[0] & ~3;
          s_anchor[1] = an[1];
          s_anchor[2] = an[2];
          s_anchored = 1;
        }
        __syncthreads();
      }
      if (ROUNDS > 1 && s_chooser == 256) break;   // uniform: no ray (left) to walk
      const int anx = s_anchor[0], any_ = s_anchor[1], anz = s_anchor[2];
      AT_MARK(1);
#ifdef SF_ALLOC_TIMING
      if (threadIdx.x == 0) { atomicAdd(&g_alloc_t[9], 1ull); at_rounds_++; }
#endif

      // ---- DDA: one LDS bit per visited block
      bool left_window = false;
#ifdef SF_ABLATE_ALLOC_WALK   // measurement only (the volume is WRONG): no DDA walk
      if (false) {
#else
      if (pending) {
#endif
        int c_x = a_cx, c_y = a_cy, c_z = a_cz;   // (the ray's start stays: it may walk again)
        float tmx = a_tmx, tmy = a_tmy, tmz = a_tmz;
        uint64_t last_key = KEY_EMPTY;
        for (int it = 0; it < MAX_DDA_ITERS; ++it) {
          const uint32_t ux = (uint32_t)(c_x - anx), uy = (uint32_t)(c_y - any_), uz = (uint32_t)(c_z - anz);
          const bool inwin = (ux | uy | uz) < (uint32_t)WIN;
          const uint32_t bit = inwin ? ((uz << (2 * WIN_LOG2)) | (uy << WIN_LOG2) | ux) : 0xFFFFFFFFu;
          // The 8x8 pixel patch of a wave mostly sits in ONE block: 64 ds_or to the same LDS word serialise.  Drop
          // the lane when its left neighbour (DPP row_shr:1, free) sets the same bit; a disabled or out-of-row
          // neighbour reads as "different" (old value, bound_ctrl off), so run heads always write.
          const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)bit, 0x111, 0xF, 0xF, false);
          if (inwin) {
            if (left != bit) atomicOr(&s_frame[bit >> 5], 1u << (bit & 31));
          } else if (round + 1 < ROUNDS) {
            left_window = true;   // walks again in the next round's window
          } else {
            const uint64_t key = pack_key(c_x, c_y, c_z);
            if (key != last_key) {
              last_key = key;
              if (slab_owns(P, c_x, c_y, c_z) && block_in_frustum(P, FL, c_x, c_y, c_z)) {
                uint32_t sl = ((uint32_t)(key ^ (key >> 21) ^ (key >> 42)) * 2654435761u) >> (32 - SET_LOG2);
                bool placed = false;
                for (int pr = 0; pr < ALLOC_SET_PROBES; ++pr) {
                  const unsigned long long old = atomicCAS(&s_keys[sl], (unsigned long long)KEY_EMPTY, (unsigned long long)key);
                  if (old == key) { placed = true; break; }  // queued by an earlier step / ray / frame
                  if (old == KEY_EMPTY) {
                    const int pos = atomicAdd(&s_count, 1);
                    if (pos < LIST) { s_list[pos] = key; s_birth[pos] = (uint8_t)(j | 0x80); placed = true; }
                    break;  // queue full: direct path below
                  }
                  sl = (sl + 1) & (SET - 1);
                }
                if (!placed) direct(key, c_x, c_y, c_z, B.seq0 + (uint32_t)j);
              }
            }
          }
          bool done;
          if (tmx < tmy && tmx < tmz) { c_x += a_sx; done = (c_x == a_ex); tmx += a_tdx; }
          else if (tmz < tmy) { c_z += a_sz; done = (c_z == a_ez); tmz += a_tdz; }
          else { c_y += a_sy; done = (c_y == a_ey); tmy += a_tdy; }
          if (done) break;
        }
      }
      pending = left_window;
      __syncthreads();
      AT_MARK(2);
      // ---- scan: blocks this frame visits that no earlier frame of the group queued -> frustum test -> queue
#ifndef SF_ABLATE_ALLOC_SCAN   // measurement only (the volume is WRONG): no scan
      // A lane takes a WORD (32 x-consecutive blocks) as far as whole words go -- read it, take out what an earlier frame of the group queued and what the presence
      // cache knows -- and a BLOCK from there on: the wave then walks the words that have bits left two at a time, lane b of each half-wave testing block b.
      // (One thread per word all the way -- a loop over the word's bits around the frustum test -- kept a wave as long as the fullest of its 64 words: a tile that looks at
      // a surface for the first time has ~5 500 blocks in ~400 words, a sixth of the lanes busy, and took 200 us here where the mean is 17; the longest workgroup IS
      // the kernel at one frame per launch.  profiles/r06_alloc_1mm.txt)
      for (int base = wave * 64; base < WIN_WORDS; base += 256) {
        const int w = base + lane;
        const uint32_t seen = MULTI ? (s_frame[w] & ~s_done[w]) : s_frame[w];
        uint32_t bits = seen;
        if (bits != 0u && bricks.e != nullptr) {
          // the word's 32 blocks are 8 whole bricks: what the presence cache knows of them is in the table already and older than this batch -- nothing to test,
          // queue or probe for those (and nothing for a later frame of the group either)
          const uint32_t bit0 = (uint32_t)w << 5;
          bits &= ~brick_known_row(bricks, anx + (int)(bit0 & (WIN - 1)), any_ + (int)((bit0 >> WIN_LOG2) & (WIN - 1)), anz + (int)(bit0 >> (2 * WIN_LOG2)));
        }
        uint32_t queued = seen & ~bits;
        uint64_t todo = __ballot(bits != 0u);
        while (todo != 0ull) {   // (uniform)
          const int l0 = __ffsll((unsigned long long)todo) - 1;
          todo &= todo - 1ull;
          const int l1 = todo != 0ull ? __ffsll((unsigned long long)todo) - 1 : l0;
          const bool second = todo != 0ull;
          todo &= todo - 1ull;   // (0 stays 0)
          const int src = lane < 32 ? l0 : l1;
          const uint32_t wbits = (uint32_t)__shfl((int)bits, src);
          const int b = lane & 31;
          const uint32_t wbit0 = (uint32_t)(base + src) << 5;
          const int bx = anx + (int)(wbit0 & (WIN - 1)) + b, by = any_ + (int)((wbit0 >> WIN_LOG2) & (WIN - 1)), bz = anz + (int)(wbit0 >> (2 * WIN_LOG2));
          const bool mine = ((wbits >> b) & 1u) != 0u && (lane < 32 || second);
          const bool foreign = mine && !slab_owns(P, bx, by, bz);   // another GPU's block: never ours, stop looking at it
          const bool pass = mine && !foreign && block_in_frustum(P, FL, bx, by, bz);   // (outside this frame's frustum: a later frame may still want it)
          const uint64_t pm = __ballot(pass);
          if (pm != 0ull) {
            int pos = 0;
            if (lane == 0) pos = atomicAdd(&s_count, __popcll((unsigned long long)pm));
            pos = __builtin_amdgcn_readfirstlane(pos) + __popcll((unsigned long long)(pm & ((1ull << lane) - 1ull)));
            if (pass) {
              if (pos < LIST) { s_list[pos] = pack_key(bx, by, bz); s_birth[pos] = (uint8_t)j; }
              else direct(pack_key(bx, by, bz), bx, by, bz, B.seq0 + (uint32_t)j);
            }
          }
          if (MULTI) {
            const uint64_t qm = __ballot(pass || foreign);
            if (lane == l0) queued |= (uint32_t)qm;
            if (second && lane == l1) queued |= (uint32_t)(qm >> 32);
          }
        }
        if (MULTI && queued) s_done[w] |= queued;  // word w is only ever touched by this lane
      }
#endif
      if (ROUNDS > 1) {   // the queue is emptied between rounds
        __syncthreads();
        AT_MARK(3);
#ifdef SF_ALLOC_TIMING
        if (threadIdx.x == 0) at_queued_ += (unsigned)s_count;
#endif
        drain();
        __syncthreads();
        AT_MARK(4);
        if (threadIdx.x == 0) s_count = 0;
      }
    }
  }
  __syncthreads();
  AT_MARK(3);
  drain();
  for (int o = 32; o > 0; o >>= 1) { n_direct += __shfl_xor(n_direct, o); n_probed += __shfl_xor(n_probed, o); }
  if (lane == 0 && n_direct) atomicAdd(&counters[C_ALLOC_DIRECT], n_direct);
  if (lane == 0 && n_probed) atomicAdd(&counters[C_ALLOC_PROBED], n_probed);
#ifdef SF_ALLOC_TIMING
  __syncthreads();
  AT_MARK(4);
  if (threadIdx.x == 0) {
    atomicAdd(&g_alloc_t[5], at_prev_ - at_start_);
    atomicMax(&g_alloc_t[8], at_prev_ - at_start_);
    atomicAdd(&g_alloc_t[10], 1ull);
  }
  if (threadIdx.x == 0 && at_prev_ - at_start_ > 25000ull) {
    const unsigned slot = atomicAdd(&g_alloc_log[0], 1u);
    if (slot < 256u) {
      unsigned* r = &g_alloc_log[1 + 12 * slot];
      r[0] = blockIdx.x | (blockIdx.y << 16); r[1] = at_rounds_; r[2] = at_ph_[0]; r[3] = at_ph_[1]; r[4] = at_ph_[2]; r[5] = at_ph_[3]; r[6] = at_ph_[4];
      r[7] = (unsigned)(at_prev_ - at_start_); r[8] = (unsigned)n_direct; r[9] = (unsigned)n_probed; r[10] = at_queued_ + (unsigned)s_count; r[11] = 0;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// K2r: the same allocation with the occupancy bitmap laid out in RAY SPACE (the default whenever the geometry fits, see alloc_ray in
// sf_fuser_create).  A 16x16 pixel tile looks down a thin pencil of rays: a few blocks wide but as deep as the scene -- and where the tile
// straddles a depth discontinuity (every furniture edge of a real room) its rays sit in two clusters metres apart.  The cube window of
// k_alloc (32^3 blocks = 1 m at 4 mm voxels, anchored at the first ray) covers one cluster; the other fell through to an LDS hash set and,
// when that filled, to one global-table probe PER DDA STEP: measured on the furnished room, 115 us -> 450 us (up to 1.2 ms) per batch.
// Here the window follows the pencil: block (c_a, c_u, c_v) -- a = the axis the tile's centre ray mostly runs along, u, v the other two --
// maps to   k  = +-(c_a - k0)                        slab index along the ray, 0 at the camera, RW_DEPTH = 256 slabs (8 m at 4 mm)
//           du = c_u - (ou + ((su k + fu) >> 12))    lateral offset from the centre ray's block in slab k, RW_LAT = 16 wide
// (dv likewise), bit = k * 256 + dv * 16 + du.  The map is a bijection onto the window for any integers k0, su, ou, ... -- how well the
// centre line is placed only decides how many rays stay inside --, so it is fixed once per workgroup from the group's FIRST frame and the
// "already queued by an earlier frame" bitmap stays valid across the frames of the group.  One slab = 256 bits = 8 words = one thread of the
// workgroup: the scan is two 16-byte LDS reads per thread and frame, and only the ~10 threads whose slab is occupied do anything more.
// Everything else (ray set-up, DDA, frustum test, queue, table probe, heap pop) is k_alloc's, statement for statement: the allocated SET
// and every birth frame are the same (tests/test_gpu_tsdf.py runs both kernels against the oracle).
// ---------------------------------------------------------------------------------------------------
constexpr int RW_LAT_LOG2 = 4, RW_LAT = 1 << RW_LAT_LOG2, RW_DEPTH = 256;
constexpr int RW_WORDS = RW_DEPTH * RW_LAT * RW_LAT / 32;   // 2048 words = 8 KiB per bitmap

// One ray's walk over the blocks of [d - t, d + t] in WINDOW coordinates, for the window axis AXIS (compile time): slab k along the pencil and
// the lateral block coordinates relative to the window origin (ru, rv), so that a step costs an add on one of them instead of the whole map.
// The three-way branch of the reference walk is evaluated as lane masks -- the same comparisons in the same order: x if strictly smallest, else
// z if smaller than y, else y -- so no lane waits for the branches the others take.  This walk only sets bits; it returns true when the ray left
// the window (rare: the map follows the camera), and the caller walks such a ray AGAIN for the blocks outside.
struct RayWalk {
  int cx, cy, cz, sx, sy, sz, ex, ey, ez;      // first block, step and one-past-the-last block per axis
  float tmx, tmy, tmz, tdx, tdy, tdz;          // parameter of the next block face / per block, per axis
};
struct WindowMap {
  int k0, sgn, su, ou, fu, sv, ov, fv;
};
template <int AXIS>
__device__ inline bool ray_walk_bits(const RayWalk& r, const WindowMap& w, uint32_t* s_frame, bool no_atomics) {
  // (a, u, v) = (AXIS, AXIS + 1, AXIS + 2) mod 3
  const int c_a = AXIS == 0 ? r.cx : (AXIS == 1 ? r.cy : r.cz), c_u = AXIS == 0 ? r.cy : (AXIS == 1 ? r.cz : r.cx), c_v = AXIS == 0 ? r.cz : (AXIS == 1 ? r.cx : r.cy);
  const int s_a = AXIS == 0 ? r.sx : (AXIS == 1 ? r.sy : r.sz), s_u = AXIS == 0 ? r.sy : (AXIS == 1 ? r.sz : r.sx), s_v = AXIS == 0 ? r.sz : (AXIS == 1 ? r.sx : r.sy);
  const int e_a = AXIS == 0 ? r.ex : (AXIS == 1 ? r.ey : r.ez), e_u = AXIS == 0 ? r.ey : (AXIS == 1 ? r.ez : r.ex), e_v = AXIS == 0 ? r.ez : (AXIS == 1 ? r.ex : r.ey);
  int k = w.sgn > 0 ? c_a - w.k0 : w.k0 - c_a;
  const int k_end = w.sgn > 0 ? e_a - w.k0 : w.k0 - e_a;
  const int dk = w.sgn > 0 ? s_a : -s_a;
  int ru = c_u - w.ou, rv = c_v - w.ov;
  const int ru_end = e_u - w.ou, rv_end = e_v - w.ov;
  float tmx = r.tmx, tmy = r.tmy, tmz = r.tmz;
  bool left_window = false;
  for (int it = 0; it < MAX_DDA_ITERS; ++it) {
    // (a slab index far outside the window only has to fail the range test: the 24-bit product may be anything there)
    const uint32_t du = (uint32_t)(ru - ((__mul24(w.su, k) + w.fu) >> 12));
    const uint32_t dv = (uint32_t)(rv - ((__mul24(w.sv, k) + w.fv) >> 12));
    const bool inwin = (uint32_t)k < (uint32_t)RW_DEPTH && (du | dv) < (uint32_t)RW_LAT;
    const uint32_t bit = inwin ? (((uint32_t)k << (2 * RW_LAT_LOG2)) | (dv << RW_LAT_LOG2) | du) : 0xFFFFFFFFu;
    // lanes whose left neighbour (DPP row_shr:1) sets the same bit stay silent: 64 same-address ds_or serialise (see k_alloc)
    const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)bit, 0x111, 0xF, 0xF, false);
    if (inwin && left != bit && !no_atomics) atomicOr(&s_frame[bit >> 5], 1u << (bit & 31));
    left_window = left_window || !inwin;
    const bool go_x = tmx < tmy && tmx < tmz;
    const bool go_z = !go_x && tmz < tmy;
    const bool go_y = !go_x && !go_z;
    tmx += go_x ? r.tdx : 0.0f;   // x + 0 = x: the axes not taken keep their value bit for bit
    tmy += go_y ? r.tdy : 0.0f;
    tmz += go_z ? r.tdz : 0.0f;
    const bool go_a = AXIS == 0 ? go_x : (AXIS == 1 ? go_y : go_z);
    const bool go_u = AXIS == 0 ? go_y : (AXIS == 1 ? go_z : go_x);
    k += go_a ? dk : 0;
    ru += go_u ? s_u : 0;
    rv += (!go_a && !go_u) ? s_v : 0;
    // "the coordinate that moved reached its end" as two selects and ONE compare: written as a nested conditional of three compares the compiler built it out of
    // nested exec-mask regions (three s_and_saveexec / s_cbranch_execz pairs per step of the hot walk)
    const int moved = go_a ? k : (go_u ? ru : rv);
    const int moved_end = go_a ? k_end : (go_u ? ru_end : rv_end);
    if (moved == moved_end) break;
  }
  return left_window;
}

template <bool MULTI>
#ifndef SF_ALLOC_WAVES_MIN
#define SF_ALLOC_WAVES_MIN 1   // waves per SIMD k_alloc_ray is register-budgeted for (8: <= 64 registers: a wave of it fits any slot a 64-register wave of k_integrate frees)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SF_ALLOC_WAVES_MIN, 8))) void k_alloc_ray(const float* __restrict__ depthf_all, HashEntry* table, int32_t* heap,
                                                   uint64_t* block_keys, int32_t* block_entry, uint8_t* block_flags, int32_t* counters, ParamsK P,
                                                   BatchFrames B, int group_frames, int ablate, const uint16_t* __restrict__ fuse_depth16,
                                                   float* depthf_out, int compact_counter) {
  // ablate (tune "alloc_ablate", measurements only -- the volume is wrong with any bit set): 1 no LDS atomics, 2 no scan, 4 no DDA walk, 8 no barriers
  // fuse_depth16 != nullptr (one frame per pass, no colour, no resampling: a live stream): the kernel is ALSO the depth pre-pass -- every lane
  // converts its own pixel (k_prepass's arithmetic), stores it for the integrate kernel's gathers and walks it; one launch and one
  // dependency hop less in a chain of four that is the whole frame time
  __shared__ uint4 s_frame4[RW_WORDS / 4];              // blocks the current frame's rays visit (slab-major)
  __shared__ uint4 s_done4[MULTI ? RW_WORDS / 4 : 1];   // blocks an earlier frame of the group has already queued
  __shared__ unsigned long long s_keys[ALLOC_SET];      // the same for blocks outside the window
  __shared__ unsigned long long s_list[ALLOC_LIST];     // queue for phase 2
  __shared__ uint8_t s_birth[ALLOC_LIST];               // ... and the frame (index in the batch) that queued the key
  __shared__ int s_count;
  uint32_t* const s_frame = reinterpret_cast<uint32_t*>(s_frame4);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_count = 0;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  // No presence cache here (fuser_internal.h BrickCache: the cube window's kernels use it): the "already queued" bitmap of the ray-space window leaves this kernel
  // few look-ups to save -- 34.7 k against 34.8 k frames/s on the long stream with the cache on / off -- and its code, even switched off at run time, cost a 20-frame
  // call into an empty volume 2 % (35.1 k -> 34.3 k, five libraries on one box: tools/gpu/r06_zr.sh).  The cache stays right: it only ever holds blocks the cube
  // kernels FOUND in the table, and whatever takes blocks out of the table clears it.
  const HashRefs h{table, heap, block_keys, block_entry, block_flags, counters, BrickCache{nullptr, 0u}, B.seq0};
  for (int i = threadIdx.x; i < ALLOC_SET; i += 256) s_keys[i] = KEY_EMPTY;
  for (int i = threadIdx.x; i < RW_WORDS / 4; i += 256) s_frame4[i] = make_uint4(0, 0, 0, 0);
  if (MULTI)
    for (int i = threadIdx.x; i < RW_WORDS / 4; i += 256) s_done4[i] = make_uint4(0, 0, 0, 0);
  const size_t npx = (size_t)P.W * P.H;
  const int j_begin = blockIdx.z * group_frames;
  const int j_end = min(B.n, j_begin + group_frames);

  int n_direct = 0;   // sf_fuser_alloc_direct_count (added up once per wave at the end: one atomic per call on a single word halved the 1 mm front chain)
  auto direct = [&](uint64_t key, int bx, int by, int bz, uint32_t seq) {
    n_direct++;
    HashEntry* e = hash_find_or_claim(h, P, key, bx, by, bz, seq);
    if (e) {
      atomicAdd(&counters[C_SLOTS_USED], 1);
      give_block(h, e, key, atomicSub(&counters[C_HEAP_FREE], 1) - 1);
    }
  };

  // ---- the window map (uniform: every thread computes the same numbers), laid along the MEAN of the tile's centre rays in frames ja and jb --
  // the first and the last frame it will serve: a camera that turns during the pass sweeps the pencil sideways (0.2 degrees per frame in the
  // bench walk's corners = 7 blocks at 4 m over 16 frames), and rays that leave the window take the slow path (an LDS hash set, then one
  // global-table probe per step: the workgroups of such tiles ran 3x longer than the rest and set the kernel's duration)
  int w_axis = 0, w_k0 = 0, w_sgn = 1, w_su = 0, w_ou = 0, w_fu = 0, w_sv = 0, w_ov = 0, w_fv = 0;
  const float bs = 8.0f * P.voxel;
  const float kxc = (((float)(blockIdx.x * 16) + 7.5f) - P.mx) / P.fx, kyc = (((float)(blockIdx.y * 16) + 7.5f) - P.my) / P.fy;
  auto centre_ray = [&](int j, float (&dir)[3], float (&org)[3]) {
    const FrameK& Fj = B.f[min(max(j, 0), MAX_BATCH - 1)];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      dir[r] = Fj.T[4 * r] * kxc + Fj.T[4 * r + 1] * kyc + Fj.T[4 * r + 2];   // camera-space z component 1: dir * z = the point at depth z
      org[r] = Fj.T[4 * r + 3] / bs;                                         // camera centre in block units
    }
  };
  auto anchor = [&](int ja, int jb) {
    float da_[3], oa_[3], db_[3], ob_[3], dir[3], org[3];
    centre_ray(ja, da_, oa_);
    centre_ray(jb, db_, ob_);
#pragma unroll
    for (int r = 0; r < 3; r++) { dir[r] = 0.5f * (da_[r] + db_[r]); org[r] = 0.5f * (oa_[r] + ob_[r]); }
    const float ax = fabsf(dir[0]), ay = fabsf(dir[1]), az = fabsf(dir[2]);
    int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const float da = axis == 0 ? dir[0] : (axis == 1 ? dir[1] : dir[2]);
    const float oa = axis == 0 ? org[0] : (axis == 1 ? org[1] : org[2]);
    const float du_ = axis == 0 ? dir[1] : (axis == 1 ? dir[2] : dir[0]);   // u = (a + 1) % 3, v = (a + 2) % 3
    const float dv_ = axis == 0 ? dir[2] : (axis == 1 ? dir[0] : dir[1]);
    const float ou_ = axis == 0 ? org[1] : (axis == 1 ? org[2] : org[0]);
    const float ov_ = axis == 0 ? org[2] : (axis == 1 ? org[0] : org[1]);
    int sgn = da < 0.0f ? -1 : 1;
    const int cb = (int)floorf(oa);
    int k0 = cb - sgn;                                 // slab 1 holds the camera, slab 0 is one block of margin behind it
    const float inv = da != 0.0f ? 1.0f / da : 0.0f;
    const float slu = du_ * inv * (float)sgn, slv = dv_ * inv * (float)sgn;   // lateral blocks per slab, |.| <= 1
    // lateral position of the centre line at the middle of slab 0 (block units), minus half the window
    const float a0 = ((float)k0 + 0.5f) - oa;
    const float iu = ou_ + du_ * inv * a0 - (float)(RW_LAT / 2), iv = ov_ + dv_ * inv * a0 - (float)(RW_LAT / 2);
    const float fiu = floorf(iu), fiv = floorf(iv);
    w_axis = __builtin_amdgcn_readfirstlane(axis); w_k0 = __builtin_amdgcn_readfirstlane(k0); w_sgn = __builtin_amdgcn_readfirstlane(sgn);
    w_su = __builtin_amdgcn_readfirstlane((int)rintf(slu * 4096.0f)); w_ou = __builtin_amdgcn_readfirstlane((int)fiu);
    w_fu = __builtin_amdgcn_readfirstlane((int)((iu - fiu) * 4096.0f));
    w_sv = __builtin_amdgcn_readfirstlane((int)rintf(slv * 4096.0f)); w_ov = __builtin_amdgcn_readfirstlane((int)fiv);
    w_fv = __builtin_amdgcn_readfirstlane((int)((iv - fiv) * 4096.0f));
  };
  // How many consecutive frames one map can serve: the tile's centre point at the integration distance moves D blocks between the group's
  // first and last frame; anchored on the mean, a map holds a sweep of ~9 blocks (window +-8, half a tile's width and the block rounding
  // off).  A faster camera gets a fresh map -- and a cleared "already queued" bitmap, which only costs repeated look-ups -- every n_map frames.
  int n_map = max(1, j_end - j_begin);
  if (MULTI && j_end - j_begin > 1) {
    float d0[3], o0[3], d1[3], o1[3];
    centre_ray(j_begin, d0, o0);
    centre_ray(j_end - 1, d1, o1);
    const float far = P.maxd / bs;
    float D = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; r++) D = fmaxf(D, fabsf((o1[r] + d1[r] * far) - (o0[r] + d0[r] * far)));
    if (D > 9.0f) n_map = max(1, (int)((float)(j_end - j_begin) * 9.0f / D));
    n_map = __builtin_amdgcn_readfirstlane(n_map);
  }
  int next_map = j_begin;
  const bool in_image = x < P.W && y < P.H;
  const float kx = ((float)x - P.mx) / P.fx, ky = ((float)y - P.my) / P.fy;  // the pixel's ray direction is the same for every frame
  const float rvoxel = 1.0f / P.voxel;                                        // RN(1 / voxel) for world_to_block
  float d_next;
  if (fuse_depth16 != nullptr) {   // uniform
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicExch(reinterpret_cast<unsigned long long*>(&counters[compact_counter]), 0ull);
    d_next = -INFINITY;
    if (in_image) {
      const uint16_t u = fuse_depth16[(size_t)(y * P.W + x)];
      float v = (float)u / P.depth_shift;
      if (u == 0 || v < P.dmin || v > P.dmax) v = -INFINITY;
      depthf_out[(size_t)(y * P.W + x)] = v;
      d_next = v;
    }
  } else {
    d_next = in_image && j_begin < j_end ? depthf_all[(size_t)j_begin * npx + (size_t)(y * P.W + x)] : -INFINITY;
  }
  __syncthreads();   // bitmaps zeroed
  for (int j = j_begin; j < j_end; ++j) {
    const FrameK& F = B.f[j];  // uniform index: scalar loads from the kernarg segment
    const float d_cur = d_next;
    d_next = in_image && j + 1 < j_end ? depthf_all[(size_t)(j + 1) * npx + (size_t)(y * P.W + x)] : -INFINITY;
    if (j == next_map) {   // uniform
      anchor(j, min(j + n_map, j_end) - 1);
      next_map = j + n_map;
      if (MULTI && j != j_begin)   // the queued-blocks bitmap was laid out by the old map (read again only behind the next barrier)
        for (int i = threadIdx.x; i < RW_WORDS / 4; i += 256) s_done4[i] = make_uint4(0, 0, 0, 0);
    }

    // ---- ray set-up (k_alloc's, statement for statement)
    bool active = false;
    int a_cx = 0, a_cy = 0, a_cz = 0, a_sx = 0, a_sy = 0, a_sz = 0, a_ex = 0, a_ey = 0, a_ez = 0;
    float a_tmx = INFINITY, a_tmy = INFINITY, a_tmz = INFINITY, a_tdx = INFINITY, a_tdy = INFINITY, a_tdz = INFINITY;
    if (in_image) {
      const float d = d_cur;
      if (d != -INFINITY && d < P.maxd) {
        const float t = fmaf(P.tscale, d, P.tbase);
        const float lo = min_f32(P.maxd, d - t);
        const float hi = min_f32(P.maxd, d + t);
        if (lo < hi) {
          float p0[3], p1[3];
          {
            const float ax = kx * lo, ay = ky * lo, az = lo;
#pragma unroll
            for (int r = 0; r < 3; r++) p0[r] = fmaf(F.T[4 * r], ax, fmaf(F.T[4 * r + 1], ay, fmaf(F.T[4 * r + 2], az, F.T[4 * r + 3])));
          }
          {
            const float ax = kx * hi, ay = ky * hi, az = hi;
#pragma unroll
            for (int r = 0; r < 3; r++) p1[r] = fmaf(F.T[4 * r], ax, fmaf(F.T[4 * r + 1], ay, fmaf(F.T[4 * r + 2], az, F.T[4 * r + 3])));
          }
          const float bsize = 8.0f * P.voxel;
          int cur[3], stp[3], bnd[3];
          float tm[3], td[3];
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float dir = p1[c] - p0[c];
            cur[c] = world_to_block(p0[c], P.voxel, rvoxel);
            const int e = world_to_block(p1[c], P.voxel, rvoxel);
            stp[c] = dir > 0.0f ? 1 : (dir < 0.0f ? -1 : 0);
            bnd[c] = e + stp[c];
            if (stp[c] == 0) { tm[c] = INFINITY; td[c] = INFINITY; }
            else {
              const int nb = cur[c] + (stp[c] > 0 ? 1 : 0);
              const float plane = ((float)(8 * nb) - 0.5f) * P.voxel;
              const float rdir = recip_rn(dir);   // one reciprocal for both quotients
              tm[c] = div_rn(plane - p0[c], dir, rdir);
              td[c] = div_rn((float)stp[c] * bsize, dir, rdir);
            }
          }
          a_cx = cur[0]; a_cy = cur[1]; a_cz = cur[2];
          a_sx = stp[0]; a_sy = stp[1]; a_sz = stp[2]; a_ex = bnd[0]; a_ey = bnd[1]; a_ez = bnd[2];
          a_tmx = tm[0]; a_tmy = tm[1]; a_tmz = tm[2]; a_tdx = td[0]; a_tdy = td[1]; a_tdz = td[2];
          active = true;
        }
      }
    }

    // ---- DDA: one LDS bit per visited block.  The walk runs in WINDOW coordinates: slab k along the pencil and the lateral block
    // coordinates relative to the window origin (ru, rv), so that a step costs an add on one of them instead of the whole map -- and the
    // three-way branch of the reference walk is evaluated as three lane masks (the same comparisons in the same order: x if strictly
    // smallest, else z if smaller than y, else y), so no lane waits for the branches the others take.
    if (active && !(ablate & 4)) {
      // The axis the window runs along is the same for every lane of the workgroup (a scalar): the hot walk is compiled THREE times, once per
      // axis, and chosen by a scalar branch -- inside each copy "the window axis" is a compile-time name for one of x / y / z, so a step's
      // "which coordinate moves" is the very lane mask its comparison produced.  (With w_axis as a run-time select on the three masks the
      // compiler materialised them into registers and picked among them with vector selects: 11 of the 48 vector instructions of a step.)
      const RayWalk rw{a_cx, a_cy, a_cz, a_sx, a_sy, a_sz, a_ex, a_ey, a_ez, a_tmx, a_tmy, a_tmz, a_tdx, a_tdy, a_tdz};
      const WindowMap wm{w_k0, w_sgn, w_su, w_ou, w_fu, w_sv, w_ov, w_fv};
      bool left_window;
      if (w_axis == 0) left_window = ray_walk_bits<0>(rw, wm, s_frame, (ablate & 1) != 0);
      else if (w_axis == 1) left_window = ray_walk_bits<1>(rw, wm, s_frame, (ablate & 1) != 0);
      else left_window = ray_walk_bits<2>(rw, wm, s_frame, (ablate & 1) != 0);
      if (left_window) {   // the same walk once more (one copy, the axis a run-time value), this time for the blocks OUTSIDE the window: LDS hash set, then the global table
        const int c_a = w_axis == 0 ? a_cx : (w_axis == 1 ? a_cy : a_cz), c_u = w_axis == 0 ? a_cy : (w_axis == 1 ? a_cz : a_cx), c_v = w_axis == 0 ? a_cz : (w_axis == 1 ? a_cx : a_cy);
        const int s_a = w_axis == 0 ? a_sx : (w_axis == 1 ? a_sy : a_sz), s_u = w_axis == 0 ? a_sy : (w_axis == 1 ? a_sz : a_sx), s_v = w_axis == 0 ? a_sz : (w_axis == 1 ? a_sx : a_sy);
        const int e_a = w_axis == 0 ? a_ex : (w_axis == 1 ? a_ey : a_ez), e_u = w_axis == 0 ? a_ey : (w_axis == 1 ? a_ez : a_ex), e_v = w_axis == 0 ? a_ez : (w_axis == 1 ? a_ex : a_ey);
        int k = w_sgn > 0 ? c_a - w_k0 : w_k0 - c_a;
        const int k_end = w_sgn > 0 ? e_a - w_k0 : w_k0 - e_a;
        const int dk = w_sgn > 0 ? s_a : -s_a;
        int ru = c_u - w_ou, rv = c_v - w_ov;
        const int ru_end = e_u - w_ou, rv_end = e_v - w_ov;
        uint64_t last_key = KEY_EMPTY;
#pragma unroll 1
        for (int it = 0; it < MAX_DDA_ITERS; ++it) {
          const uint32_t du = (uint32_t)(ru - ((__mul24(w_su, k) + w_fu) >> 12));
          const uint32_t dv = (uint32_t)(rv - ((__mul24(w_sv, k) + w_fv) >> 12));
          const bool inwin = (uint32_t)k < (uint32_t)RW_DEPTH && (du | dv) < (uint32_t)RW_LAT;
          if (!inwin) {
            const int ca = w_sgn > 0 ? w_k0 + k : w_k0 - k, cu = ru + w_ou, cv = rv + w_ov;
            const int cx = w_axis == 0 ? ca : (w_axis == 1 ? cv : cu), cy = w_axis == 0 ? cu : (w_axis == 1 ? ca : cv), cz = w_axis == 0 ? cv : (w_axis == 1 ? cu : ca);
            const uint64_t key = pack_key(cx, cy, cz);
            if (key != last_key) {
              last_key = key;
              if (slab_owns(P, cx, cy, cz) && block_in_frustum(P, F, cx, cy, cz)) {
                uint32_t sl = ((uint32_t)(key ^ (key >> 21) ^ (key >> 42)) * 2654435761u) >> 24;  // 8 bits
                bool placed = false;
#pragma unroll 1
                for (int pr = 0; pr < ALLOC_SET_PROBES; ++pr) {
                  const unsigned long long old = atomicCAS(&s_keys[sl], (unsigned long long)KEY_EMPTY, (unsigned long long)key);
                  if (old == key) { placed = true; break; }  // queued by an earlier step / ray / frame
                  if (old == KEY_EMPTY) {
                    const int pos = atomicAdd(&s_count, 1);
                    if (pos < ALLOC_LIST) { s_list[pos] = key; s_birth[pos] = (uint8_t)j; placed = true; }
                    break;  // queue full: direct path below
                  }
                  sl = (sl + 1) & (ALLOC_SET - 1);
                }
                if (!placed) direct(key, cx, cy, cz, B.seq0 + (uint32_t)j);
              }
            }
          }
          const bool go_x = a_tmx < a_tmy && a_tmx < a_tmz;
          const bool go_z = !go_x && a_tmz < a_tmy;
          const bool go_y = !go_x && !go_z;
          a_tmx += go_x ? a_tdx : 0.0f;
          a_tmy += go_y ? a_tdy : 0.0f;
          a_tmz += go_z ? a_tdz : 0.0f;
          const bool go_a = w_axis == 0 ? go_x : (w_axis == 1 ? go_y : go_z);
          const bool go_u = w_axis == 0 ? go_y : (w_axis == 1 ? go_z : go_x);
          k += go_a ? dk : 0;
          ru += go_u ? s_u : 0;
          rv += (!go_a && !go_u) ? s_v : 0;
          const bool done = go_a ? k == k_end : (go_u ? ru == ru_end : rv == rv_end);
          if (done) break;
        }
      }
    }
    if (!(ablate & 8)) __syncthreads();
    // ---- scan: thread t owns slab t (8 words): blocks this frame visits that no earlier frame of the group queued -> frustum test -> queue
    if (!(ablate & 2)) {
      const int k = (int)threadIdx.x;
      const uint4 f0 = s_frame4[2 * k], f1 = s_frame4[2 * k + 1];
      bool occupied = (f0.x | f0.y | f0.z | f0.w | f1.x | f1.y | f1.z | f1.w) != 0u;   // ~10 threads of the workgroup
      if (MULTI && occupied) {
        // the usual case inside a pass: everything this frame visits in the slab was queued by an earlier frame -- two more reads say so, and
        // the slab is cleared for the next frame without walking its words
        const uint4 d0 = s_done4[2 * k], d1 = s_done4[2 * k + 1];
        if (((f0.x & ~d0.x) | (f0.y & ~d0.y) | (f0.z & ~d0.z) | (f0.w & ~d0.w) | (f1.x & ~d1.x) | (f1.y & ~d1.y) | (f1.z & ~d1.z) | (f1.w & ~d1.w)) == 0u) {
          s_frame4[2 * k] = make_uint4(0, 0, 0, 0);
          s_frame4[2 * k + 1] = make_uint4(0, 0, 0, 0);
          occupied = false;
        }
      }
      if (occupied) {
        uint32_t* const s_done = reinterpret_cast<uint32_t*>(s_done4);
        const int ca = w_sgn > 0 ? w_k0 + k : w_k0 - k;
        const int cu0 = w_ou + ((w_su * k + w_fu) >> 12), cv0 = w_ov + ((w_sv * k + w_fv) >> 12);
#pragma unroll 1
        for (int w = 0; w < 8; w++) {   // the words come from LDS again: a register array indexed by w would live in scratch
          const uint32_t fw = s_frame[8 * k + w];
          if (fw == 0u) continue;
          s_frame[8 * k + w] = 0u;      // ready for the next frame (nobody else touches this slab before the next barrier)
          const uint32_t dw = MULTI ? s_done[8 * k + w] : 0u;
          uint32_t bits = fw & ~dw, queued = 0u;
          while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const int idx = w * 32 + b;
            const int cu = cu0 + (idx & (RW_LAT - 1)), cv = cv0 + (idx >> RW_LAT_LOG2);
            const int bx = w_axis == 0 ? ca : (w_axis == 1 ? cv : cu);
            const int by = w_axis == 0 ? cu : (w_axis == 1 ? ca : cv);
            const int bz = w_axis == 0 ? cv : (w_axis == 1 ? cu : ca);
            if (!slab_owns(P, bx, by, bz)) { queued |= 1u << b; continue; }  // another GPU's block: never ours, stop looking at it
            if (!block_in_frustum(P, F, bx, by, bz)) continue;  // a later frame may still want it
            queued |= 1u << b;
            const int pos = atomicAdd(&s_count, 1);
            if (pos < ALLOC_LIST) { s_list[pos] = pack_key(bx, by, bz); s_birth[pos] = (uint8_t)j; }
            else direct(pack_key(bx, by, bz), bx, by, bz, B.seq0 + (uint32_t)j);
          }
          if (MULTI && queued) s_done[8 * k + w] = dw | queued;
        }
      }
    }
    if (!(ablate & 8)) __syncthreads();   // slabs re-zeroed before the next frame's rays set bits
  }

  // ---- phase 2: queued keys -> global hash, all lanes in parallel (k_alloc's)
  const int n_unique = min(s_count, ALLOC_LIST);
  for (int i0 = 0; i0 < n_unique; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const uint64_t key = i < n_unique ? s_list[i] : KEY_EMPTY;
    HashEntry* claimed = nullptr;
    if (key != KEY_EMPTY) {
      int bx, by, bz;
      unpack_key(key, bx, by, bz);
      claimed = hash_find_or_claim(h, P, key, bx, by, bz, B.seq0 + (uint32_t)s_birth[i]);
    }
    const uint64_t cm = __ballot(claimed != nullptr);
    if (cm != 0ull) {
      const int n = __popcll((unsigned long long)cm);
      const int first = __ffsll((unsigned long long)cm) - 1;
      int base = 0;
      if (lane == first) {
        base = atomicSub(&counters[C_HEAP_FREE], n);
        atomicAdd(&counters[C_SLOTS_USED], n);
      }
      base = __shfl(base, first);
      int hw = 0;
      if (claimed != nullptr) {
        const int rank = __popcll((unsigned long long)(cm & ((1ull << lane) - 1ull)));
        hw = give_block_quiet(h, claimed, key, base - 1 - rank);
      }
      for (int o = 32; o > 0; o >>= 1) hw = max(hw, __shfl_xor(hw, o));   // the high-water mark once per wave, not per lane
      if (lane == first) raise_high_water(h, hw);
    }
  }
  for (int o = 32; o > 0; o >>= 1) n_direct += __shfl_xor(n_direct, o);
  if (lane == 0 && n_direct) atomicAdd(&counters[C_ALLOC_DIRECT], n_direct);
}

// ---------------------------------------------------------------------------------------------------
// K3: compactify.  Scans the block directory (8 B per heap slot up to the high-water mark -- not the
// 16 B x buckets x 10 hash table upstream scans) and appends the slots of the blocks that at least one
// frame of the batch updates, together with the bit mask of those frames: bit j is set iff the block is in
// frame j's frustum AND was born no later than frame j.  1024 directory entries per workgroup, ballot
// prefix sums inside the waves, one LDS exchange and TWO global atomics per workgroup (list position +
// last-frame count in one 64-bit word, the N_blk total in another cache line; a single counter word
// saturates at ~88 atomics/us on this chip).  all_live = 1 lists every live block (export), 2 every live block this
// fuser owns (GC, meshing); ghost copies of a neighbour slab's blocks are never fused.
// ---------------------------------------------------------------------------------------------------
// The frustum tests of a batch are (directory entry) x (frame) independent tests of ~25 instructions.  Until round 5 every lane ran the B.n tests of its four
// entries one after the other with the frame's constants re-read from the kernarg segment per test: 62-105 us per pass at 2 % of the vector ALUs' issue rate
// (1 600 waves in flight, each a serial chain of 128 scalar-load round trips).  Now, for batches of more than FEW frames, a lane IS a (entry, frame) pair:
// lane = sub * FPL + q holds frame q's constants in registers for the life of the workgroup (FPL = 8 / 16 / 32 frames per lane group), the workgroup's 1024
// block coordinates wait in LDS, and one step tests 64 / FPL entries against all frames at once -- the ballot of the step IS the entries' frame masks.
// Same function (block_in_frustum), same operands: the masks are the ones the serial loop produced.
constexpr int COMPACT_FEW = 4;   // up to this many frames per pass the serial loop stays (a live stream's one frame per pass would leave 31 of 32 lanes idle)

// ONE by-value argument, so that the frames' constants sit at a known offset of the kernarg segment: the workgroup copies them into LDS with one round of
// vector loads (all in flight together).  Read as `B.f[q]` they arrive through the scalar unit, a few cache lines per frame, each a separate round trip
// the wave waits for: the chain of ~80 such loads per workgroup, not the tests, was what the kernel's 56-62 us consisted of (0.9 M wave instructions, 2 %
// of the issue rate; profiles/r06_compactify.txt).
struct CompactArgs {
  const uint64_t* block_keys;
  const int32_t* block_entry;
  const uint8_t* block_flags;
  const HashEntry* table;
  int32_t* compact;
  uint32_t* cmask;
  int32_t* counters;
  int counter_id, all_live;
  ParamsK P;
  BatchFrames B;
};
constexpr int FRAMEK_WORDS = (int)(sizeof(FrameK) / 4);

// One thread per directory entry, COMPACT_THREADS entries per workgroup: a wave tests its 64 entries against all frames in 32 steps of ~300 dependent cycles.
// (Four entries per thread -- 128 steps per wave -- left the kernel at the length of that one chain: 52-62 us for 1.4 M wave instructions.  1024 threads per
// workgroup took the chain to 20 us ALONE but 170 us beside the integrate pass: a workgroup of 16 waves of 90 registers needs a whole CU to itself, and the
// integrate kernel's waves hold 480 of a SIMD's 512 registers -- a workgroup of 4 waves finds a home as soon as one wave per SIMD retires:
// profiles/r06_compactify.txt.)
constexpr int COMPACT_THREADS = 256;
constexpr int COMPACT_WAVES = COMPACT_THREADS / 64;

__global__ __launch_bounds__(COMPACT_THREADS) void k_compactify(CompactArgs A) {
  const uint64_t* __restrict__ block_keys = A.block_keys;
  const int32_t* __restrict__ block_entry = A.block_entry;
  const uint8_t* __restrict__ block_flags = A.block_flags;
  const HashEntry* __restrict__ table = A.table;
  int32_t* __restrict__ compact = A.compact;
  uint32_t* __restrict__ cmask = A.cmask;
  int32_t* counters = A.counters;
  const int counter_id = A.counter_id, all_live = A.all_live;
  const ParamsK& P = A.P;
  const BatchFrames& B = A.B;
  __shared__ int s_wtot[COMPACT_WAVES], s_wlast[COMPACT_WAVES], s_wpop[COMPACT_WAVES];
  __shared__ int s_base;
  __shared__ int4 s_c[COMPACT_THREADS];        // (bx, by, bz, listed?) of the workgroup's directory entries
  __shared__ uint32_t s_m[COMPACT_THREADS];    // their frame masks
  __shared__ uint32_t s_fk[MAX_BATCH * FRAMEK_WORDS];   // the batch's FrameK array, copied from the kernarg segment
  const int hw = counters[C_HIGH_WATER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t last_bit = 1u << (B.n - 1);
  const bool wide = !all_live && B.n > COMPACT_FEW;   // uniform
  // frames per lane group: the smallest of 8 / 16 / 32 that holds the batch
  const int fshift = B.n <= 8 ? 3 : (B.n <= 16 ? 4 : 5);
  const int q = lane & ((1 << fshift) - 1), sub = lane >> fshift, epi = 64 >> fshift;
  FrameK F;
  if (wide && (int)(blockIdx.x * COMPACT_THREADS) < hw) {
    // the frames' constants: kernarg segment -> LDS by vector loads (per-lane addresses: every load of the workgroup is in flight at once), then frame q's
    // into this lane's registers for the life of the workgroup
    typedef __attribute__((address_space(4))) const uint32_t* karg_t;
    typedef __attribute__((address_space(4))) const char* kbyte_t;
    const karg_t kp = (karg_t)((kbyte_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CompactArgs, B) + offsetof(BatchFrames, f));
    for (int i = threadIdx.x; i < B.n * FRAMEK_WORDS; i += COMPACT_THREADS) s_fk[i] = kp[i];
    __syncthreads();
    uint32_t* fw = reinterpret_cast<uint32_t*>(&F);
    const uint32_t* mine = s_fk + (q < B.n ? q : 0) * FRAMEK_WORDS;
#pragma unroll
    for (int i = 0; i < FRAMEK_WORDS; i++) fw[i] = mine[i];
  }
  for (int base = blockIdx.x * COMPACT_THREADS; base < hw; base += gridDim.x * COMPACT_THREADS) {
    const int i = base + (int)threadIdx.x;   // this thread's directory entry
    uint32_t m = 0u;
    if (wide) {
      int4 c = make_int4(0, 0, 0, 0);
      if (i < hw) {
        const uint64_t k = block_keys[i];
        if (k != KEY_EMPTY && !(block_flags[i] & 1)) {   // ghosts are never fused
          unpack_key(k, c.x, c.y, c.z);
          c.w = 1;
        }
      }
      s_c[threadIdx.x] = c;
      __syncthreads();
      const uint64_t gmask = fshift == 5 ? 0xFFFFFFFFull : ((1ull << (1 << fshift)) - 1ull);
#pragma unroll 2
      for (int e0 = wave * 64; e0 < wave * 64 + 64; e0 += epi) {   // this wave's 64 entries, 64 / FPL of them per step, against all frames at once
        const int4 cc = s_c[e0 + sub];
        const bool in = cc.w != 0 && q < B.n && block_in_frustum(P, F, cc.x, cc.y, cc.z);
        const uint64_t bal = __ballot(in);
        if (q == 0) s_m[e0 + sub] = (uint32_t)((bal >> (sub << fshift)) & gmask);
      }
      __syncthreads();
      m = s_m[threadIdx.x];
      if (m != 0u) {
        const uint32_t birth = table[block_entry[i]].birth;
        if (birth > B.seq0) {
          const uint32_t d = birth - B.seq0;
          m = d >= 32u ? 0u : (m & ~((1u << d) - 1u));
        }
      }
    }   // (passes of up to COMPACT_FEW frames and the list of every live block: k_compactify_few)
    const uint64_t bal = __ballot(m != 0u);
    const int rank = __popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
    const int wtotal = __popcll((unsigned long long)bal);
    const int wlast = __popcll((unsigned long long)__ballot((m & last_bit) != 0u));
    int pop = __popc(m);
    for (int o = 32; o > 0; o >>= 1) pop += __shfl_xor(pop, o);
    if (lane == 0) { s_wtot[wave] = wtotal; s_wlast[wave] = wlast; s_wpop[wave] = pop; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int total = 0, tlast = 0;
      for (int w = 0; w < COMPACT_WAVES; w++) { total += s_wtot[w]; tlast += s_wlast[w]; }
      s_base = 0;
      if (total) {
        const unsigned long long add = (unsigned long long)(uint32_t)total | ((unsigned long long)(uint32_t)tlast << 32);
        s_base = (int)(uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(&counters[counter_id]), add);
      }
    } else if (threadIdx.x == 64 && !all_live) {   // the two statistics: another wave's lane, so that nobody waits for them behind the returning atomic
      int total = 0, tpop = 0;
      for (int w = 0; w < COMPACT_WAVES; w++) { total += s_wtot[w]; tpop += s_wpop[w]; }
      if (total) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_TOTAL_LO]), (unsigned long long)tpop);
        atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_TILES_LO]), (unsigned long long)total);
      }
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wtot[w];
    if (m != 0u) {
      compact[off + rank] = i;
      cmask[off + rank] = m;
    }
    __syncthreads();
  }
}

// The same list for FEW frames per pass (up to COMPACT_FEW: a live stream's one frame per launch) or for every live block (all_live): one thread per entry and a
// loop over the frames -- no lane groups, no LDS staging of the frames' constants -- and a kernel of its own so that neither sets the other's register budget.
__global__ __launch_bounds__(COMPACT_THREADS) void k_compactify_few(CompactArgs A) {
  const uint64_t* __restrict__ block_keys = A.block_keys;
  const int32_t* __restrict__ block_entry = A.block_entry;
  const uint8_t* __restrict__ block_flags = A.block_flags;
  const HashEntry* __restrict__ table = A.table;
  int32_t* __restrict__ compact = A.compact;
  uint32_t* __restrict__ cmask = A.cmask;
  int32_t* counters = A.counters;
  const int counter_id = A.counter_id, all_live = A.all_live;
  const ParamsK& P = A.P;
  const BatchFrames& B = A.B;
  __shared__ int s_wlast[COMPACT_WAVES], s_wpop[COMPACT_WAVES];
  __shared__ int s_base;
  const int hw = counters[C_HIGH_WATER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t last_bit = 1u << (B.n - 1);
  // Few frames per pass and a long directory (a live stream at small voxels: 1.6 M entries at 1 mm): EU entries per thread and ONE place in the list asked for
  // per workgroup and 2 048 entries.  Asked for per 256 entries, the 6 100 returning atomics on the one list counter (and twice as many on the two statistics
  // words) WERE the kernel: 150 us for a pass over 15 MB of keys, ~25 ns per atomic (tools/gpu/period_summary.py, profiles/r06_alloc_1mm.txt).  The list keeps
  // its order: ascending directory index within a workgroup's stretch.
  constexpr int EU = 8;
  __shared__ int s_ut[EU][COMPACT_WAVES];
  for (int base = blockIdx.x * COMPACT_THREADS * EU; base < hw; base += gridDim.x * COMPACT_THREADS * EU) {
    uint64_t k[EU];
    uint8_t fl[EU];
    uint32_t m[EU];
    int rank[EU];
#pragma unroll
    for (int u = 0; u < EU; u++) {   // every load of the stretch in flight together
      const int i = base + u * COMPACT_THREADS + (int)threadIdx.x;
      k[u] = i < hw ? block_keys[i] : KEY_EMPTY;
      fl[u] = i < hw ? block_flags[i] : (uint8_t)0;
    }
    int wlast = 0, pop = 0;
#pragma unroll
    for (int u = 0; u < EU; u++) {
      m[u] = 0u;
      if (k[u] != KEY_EMPTY && !(all_live != 1 && (fl[u] & 1))) {  // ghosts are listed by all_live == 1 only
        if (all_live) m[u] = 1u;
        else {
          int bx, by, bz;
          unpack_key(k[u], bx, by, bz);
          for (int qq = 0; qq < B.n; qq++)
            if (block_in_frustum(P, B.f[qq], bx, by, bz)) m[u] |= 1u << qq;
          if (m[u] != 0u && B.n > 1) {
            const uint32_t birth = table[block_entry[base + u * COMPACT_THREADS + (int)threadIdx.x]].birth;
            if (birth > B.seq0) {
              const uint32_t d = birth - B.seq0;
              m[u] = d >= 32u ? 0u : (m[u] & ~((1u << d) - 1u));
            }
          }
        }
      }
      const uint64_t bal = __ballot(m[u] != 0u);
      rank[u] = __popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
      if (lane == 0) s_ut[u][wave] = __popcll((unsigned long long)bal);
      wlast += __popcll((unsigned long long)__ballot((m[u] & last_bit) != 0u));
      pop += __popc(m[u]);
    }
    for (int o = 32; o > 0; o >>= 1) pop += __shfl_xor(pop, o);
    if (lane == 0) { s_wlast[wave] = wlast; s_wpop[wave] = pop; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int total = 0, tlast = 0;
      for (int w = 0; w < COMPACT_WAVES; w++) {
        tlast += s_wlast[w];
        for (int u = 0; u < EU; u++) total += s_ut[u][w];
      }
      s_base = 0;
      if (total) {
        const unsigned long long add = (unsigned long long)(uint32_t)total | ((unsigned long long)(uint32_t)tlast << 32);
        s_base = (int)(uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(&counters[counter_id]), add);
      }
    } else if (threadIdx.x == 64 && !all_live) {   // the two statistics: another wave's lane, so that nobody waits for them behind the returning atomic
      int total = 0, tpop = 0;
      for (int w = 0; w < COMPACT_WAVES; w++) {
        tpop += s_wpop[w];
        for (int u = 0; u < EU; u++) total += s_ut[u][w];
      }
      if (total) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_TOTAL_LO]), (unsigned long long)tpop);
        atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_TILES_LO]), (unsigned long long)total);
      }
    }
    __syncthreads();
    int off = s_base;
#pragma unroll
    for (int u = 0; u < EU; u++) {
      int mine = off;
      for (int w = 0; w < COMPACT_WAVES; w++) {
        const int t = s_ut[u][w];
        if (w < wave) mine += t;
        off += t;
      }
      if (m[u] != 0u) {
        compact[mine + rank[u]] = base + u * COMPACT_THREADS + (int)threadIdx.x;
        cmask[mine + rank[u]] = m[u];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// K4: integrate / deintegrate.  One wave per 8^3 block: the 4 KiB tile is read with four 16 B-per-lane
// loads (two x-adjacent voxels per load; in the x-row layout of multi-frame passes the lane's own 64 contiguous bytes), updated in
// registers by EVERY frame of the batch that sees the block (temporal blocking: HBM traffic per frame
// falls by the batch size, the kernel turns from HBM-bound at B = 1 to VALU/L2-gather-bound) and written
// back with the same pattern.  There is no reuse inside a tile, so it is not
// staged through LDS; the depth image (1.2 MB f32) is gathered through L1/L2.
// pair layout: lane l, load j: uint4 q = 64 j + l -> voxels 2q, 2q+1 -> x = (2l)&7 (+1), y = (l>>2)&7, z = 2j + (l>>5);
// x-row layout (XR): lane l, load j: uint4 q = 4 l + j -> x = 2j (+1), y = l & 7, z = l >> 3.
// ---------------------------------------------------------------------------------------------------
// One frame into one tile held in registers (8 voxels per lane).  Once a batch amortises the HBM traffic the kernel sits on its instruction mix
// (SQ_INSTS_VALU x 2 clk / SIMD clk = 0.57 of the guide's issue peak, the texture addresser busy 0.56 of the time, HBM at 9 %: bench.py `roofline`,
// profiles/r06_integrate_xrow_ab.txt), so the update is written for instruction count:
//   * two straight-line phases: phase A projects all eight voxels and issues the eight depth gathers together at
//     clamped addresses, phase B applies the update under a select (a per-voxel early-out chain serialises eight
//     L2 round trips and costs a scalar branch pair per test);
//   * the lane's voxel pairs are written as v2f pairs (fuser_internal.h) and compiled as two plain fp32 operations each: on gfx950 a v_pk_*_f32 holds the
//     SIMD as long as two plain ones and issues beside nothing (rounds 1-4 shipped the packed form; -DSF_PACKED_PAIRS still builds it);
//   * the two IEEE divisions of DESIGN.md 3.5 are expanded by hand.  1/pcz: v_rcp_f32 seed + two Newton steps --
//     the arithmetic core of the compiler's own correctly rounded expansion without the div_scale / div_fixup
//     range handling (pcz is a camera-space depth in metres; exhaustive check over all mantissas and seed errors up
//     to 2 ulp: tools/check_division.c).  (old*w + sdf*wn) / (w + wn): the divisor is a small integer, its
//     correctly rounded reciprocal comes from an LDS table and ONE Markstein correction q1 = fma(fma(-m, q0, n), r, q0)
//     yields the correctly rounded quotient (same tool: 1.4e9 cases incl. near-halfway); numerators below 2^-100
//     take the plain division so that underflow cannot bite.
// Every value stored is bit-identical to oracle/tsdf_oracle.c fuse_block.
// (v2f, pk_fma, splat, recip_rn, quot_rn live in fuser_internal.h: the device self-test in calib.hip runs the same code)

constexpr int RTAB = 512;  // LDS table of correctly rounded 1/m, m = weight + weight_sample < 512

// A wave-uniform value in a VECTOR register -- an experiment kept behind -DSF_VREG_CONSTANTS.  On gfx950 v_fma / v_mul / v_add_f32, v_add_u32, v_and_b32,
// v_mov_b32 ... issue every ~2.2 cycles per SIMD when all their sources are vector registers or inline constants and every ~4.3 cycles as soon as one
// source is a scalar register (tools/gpu/valu_peak.hip, profiles/r05_valu_issue_table.txt), and the compiler feeds the per-frame matrix and the camera
// constants to every fma straight from the scalar registers they were loaded into.  Copying them into vector registers once per frame (18 v_mov, opaque
// to the compiler) made the pass SLOWER (packed 830 -> 870 us, plain pairs 808 -> 837 us: profiles/r05_integrate_ab.txt).  The launch already runs at the
// no-overlap price of its instruction mix (frac 1.03); why 88 fp32 instructions fewer in the 4.3-cycle class do not show up in it is not
// understood -- the 18 extra moves per frame and what they do to the schedule cost more than the cheaper fma's gave back.
__device__ inline float vreg(float x) {
#ifdef SF_VREG_CONSTANTS
  float r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
  return r;
#else
  return x;
#endif
}
// the per-frame / per-kernel constants the projection and the update multiply with, in vector registers (fuse_project, fuse_update)
struct FrameV {
  float ti[12];
  float fx, fy, mx, my, tscale, tbase;
};
__device__ inline FrameV frame_constants(const ParamsK& P, const float* __restrict__ Ti) {
  FrameV F;
#pragma unroll
  for (int k = 0; k < 12; k++) F.ti[k] = vreg(Ti[k]);
  F.fx = vreg(P.fx); F.fy = vreg(P.fy); F.mx = vreg(P.mx); F.my = vreg(P.my);
  F.tscale = vreg(P.tscale); F.tbase = vreg(P.tbase);
  return F;
}

__device__ inline int cvt_i32(float x) {  // v_cvt_i32_f32: truncates, saturates, NaN -> 0 (a C cast of NaN / inf would be undefined)
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// a + b saturating at 2^32 - 1: v_add_u32 with the VOP3 clamp bit (b in a scalar register: VOP3 takes no literal on gfx9)
__device__ inline uint32_t add_sat_u32(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b));
  return r;
}

// weight_mode 1 (VoxelHashing): the weight of an observation falls with its depth; (uchar) of the float, at most 255
__device__ inline int depth_weight(const ParamsK& P, float d) {
  const float z01 = (d - P.dmin) / (P.dmax - P.dmin);
  const float wf = fmaxf(((float)P.wsample * 1.5f) * (1.0f - z01), 1.0f);
  return min(cvt_i32(wf), 255);   // saturating conversion: a masked lane's garbage depth cannot trap
}

// TAB: the weighted-mean division goes through the LDS reciprocal table (integrate with 1 <= weight_sample <= 256).
// Rows [J0, J0 + NJ) of the tile (a row = the 64 x 2 voxels one 16 B load per lane covers).  NJ = 4 gives the most
// independent work per issue slot, NJ = 2 called twice halves the live registers (single-frame, occupancy-bound variant).
// Phase A of one frame on rows [J0, J0 + NJ): camera-space z of the lane's voxel pairs, the pixel each voxel projects to
// (0 when it projects outside) and whether it projects inside.
// CLAMP: pixels that project outside read pixel 0 (callers that gather with plain global loads); without it the index of an outside
// voxel is whatever the saturating conversion gave (callers that gather through a bounds-checked buffer resource and mask by `ok`).
template <int J0, int NJ, bool CLAMP>
__device__ inline void fuse_project(const ParamsK& P, const FrameV& FV, v2f wx, float wy, const float (&wz)[4], v2f (&pz)[NJ],
                                    uint32_t (&pix)[2 * NJ], bool (&ok)[2 * NJ]) {
  const float* Ti = FV.ti;
  const uint32_t uw = (uint32_t)P.W, uh = (uint32_t)P.H;
  // Row constants first, two rows or two components per packed instruction:
  //      a{x,y}_j = fma(Ti[1|5], wy, fma(Ti[2|6], wz_j, Ti[3|7])),  az_j = fma(Ti[9], wy, fma(Ti[10], wz_j, Ti[11]))
  v2f axy[NJ], azz[NJ / 2];
#pragma unroll
  for (int j = 0; j < NJ; j++)
    axy[j] = pk_fma((v2f){Ti[1], Ti[5]}, splat(wy), pk_fma((v2f){Ti[2], Ti[6]}, splat(wz[J0 + j]), (v2f){Ti[3], Ti[7]}));
#pragma unroll
  for (int jj = 0; jj < NJ / 2; jj++)
    azz[jj] = pk_fma(splat(Ti[9]), splat(wy), pk_fma(splat(Ti[10]), (v2f){wz[J0 + 2 * jj], wz[J0 + 2 * jj + 1]}, splat(Ti[11])));
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const v2f pcx = pk_fma(splat(Ti[0]), wx, splat(axy[j].x));
    const v2f pcy = pk_fma(splat(Ti[4]), wx, splat(axy[j].y));
    const v2f pcz = pk_fma(splat(Ti[8]), wx, splat(azz[j >> 1][j & 1]));
    const v2f rz = recip_rn(pcz);
    const v2f uf = pk_add(pk_fma(pcx * splat(FV.fx), rz, splat(FV.mx)), splat(0.5f));
    const v2f vf = pk_add(pk_fma(pcy * splat(FV.fy), rz, splat(FV.my)), splat(0.5f));
    pz[j] = pcz;
#pragma unroll
    for (int hx = 0; hx < 2; hx++) {
      // pixel = (int)(u + 0.5f), THEN "skip if outside the image".  v_cvt_i32_f32 truncates towards zero like the C cast
      // ((-1, 0) -> pixel 0) and saturates, so "0 <= pixel < W" is ONE unsigned compare of the converted value
      const uint32_t px = (uint32_t)cvt_i32(uf[hx]), py = (uint32_t)cvt_i32(vf[hx]);
      const bool in = (pcz[hx] > 0.0f) && (px < uw) && (py < uh);
      const uint32_t p = __umul24(py, uw) + px;   // v_mad_u32_u24: exact for every inside pixel (py < H, W < 2^24), garbage outside
      ok[2 * j + hx] = in;
      pix[2 * j + hx] = CLAMP ? (in ? p : 0u) : p;
    }
  }
}

// The same projection for the X-ROW layout (XR): a lane holds the eight voxels of ONE x-row of the block -- y = lane & 7, z = lane >> 3, register pair j =
// voxels x = 2j, 2j + 1 -- instead of two x-neighbours in each of four z-layers.  Two things follow:
//   * the inner two fma of every camera-space coordinate, fma(Ti[1], wy, fma(Ti[2], wz, Ti[3])), depend on (y, z) only: ONE set per lane and frame instead of one
//     per z-row (6 fma instead of 24; the nesting -- hence every bit -- is the specification's);
//   * one gather instruction now reads the voxels of one x-plane of the block, 8 y x 8 z: 16 consecutive lanes (the unit the L1 coalesces) are 8 y x 2 z at one
//     x -- two image rows' worth of pixels for a level camera -- where the pair layout spread 4 x by 4 y over four or five rows.  The L1 tag pipeline was the
//     busiest unit of the pass (0.79-0.86 look-ups per CU and clock, 33.6 per gather instruction: profiles/r06_*).
template <int J0, int NJ>
__device__ inline void fuse_project_xr(const ParamsK& P, const FrameV& FV, const v2f (&wxp)[4], float wy, float wz, v2f (&pz)[NJ],
                                       uint32_t (&pix)[2 * NJ], bool (&ok)[2 * NJ]) {
  const float* Ti = FV.ti;
  const uint32_t uw = (uint32_t)P.W, uh = (uint32_t)P.H;
  const float ax = fmaf(Ti[1], wy, fmaf(Ti[2], wz, Ti[3]));
  const float ay = fmaf(Ti[5], wy, fmaf(Ti[6], wz, Ti[7]));
  const float az = fmaf(Ti[9], wy, fmaf(Ti[10], wz, Ti[11]));
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const v2f pcx = pk_fma(splat(Ti[0]), wxp[J0 + j], splat(ax));
    const v2f pcy = pk_fma(splat(Ti[4]), wxp[J0 + j], splat(ay));
    const v2f pcz = pk_fma(splat(Ti[8]), wxp[J0 + j], splat(az));
    const v2f rz = recip_rn(pcz);
    const v2f uf = pk_add(pk_fma(pcx * splat(FV.fx), rz, splat(FV.mx)), splat(0.5f));
    const v2f vf = pk_add(pk_fma(pcy * splat(FV.fy), rz, splat(FV.my)), splat(0.5f));
    pz[j] = pcz;
#pragma unroll
    for (int hx = 0; hx < 2; hx++) {
      const uint32_t px = (uint32_t)cvt_i32(uf[hx]), py = (uint32_t)cvt_i32(vf[hx]);
      const bool in = (pcz[hx] > 0.0f) && (px < uw) && (py < uh);
      ok[2 * j + hx] = in;
      pix[2 * j + hx] = __umul24(py, uw) + px;
    }
  }
}

// Phase B: the update of DESIGN.md 3.5 from the gathered depths (colours) into the tile registers.
// WM (weight mode): 0 = any weight_sample / weight_max, 1 = weight_sample == 1, 2 = weight_sample == 1 and weight_max == 255 (the shipped
// parameters after the uchar clamp): the weight byte then increments with saturation as ONE add-with-carry on the {rgb, weight} word;
// 3 = the observation's weight depends on its depth (sf_params::weight_mode 1), otherwise as 0.
// dirty[j]: lane mask (a scalar register pair) of the lanes whose row j changed -- kept on the scalar unit across the frames of a batch.
// COLOR: 0 = geometry only, 1 = colour (every switch a wave-uniform mask), 2 = colour with colour_first == 0 compiled in.
// ROWS: dirty[] holds one lane mask per row (one frame per launch: the HBM-bound schedule writes back only the rows some lane changed); without
// it dirty[0] is a wave-uniform "some frame touched this tile" flag and the caller writes the whole tile back -- a ballot of an i1 that is not
// itself a compare costs a v_cndmask + v_cmp per row (8 of the 241 VALU instructions of a lane's frame), and a pass of 32 frames is VALU-bound
// with HBM at 8 % of its peak.
template <int SIGN, int COLOR, bool TAB, int WM, int J0, int NJ, bool ROWS = true>
__device__ inline void fuse_update(const ParamsK& P, const FrameV& FV, const v2f (&rcp_m)[NJ], const float (&d)[2 * NJ], const uint32_t (&c)[2 * NJ], const v2f (&pz)[NJ],
                                   const bool (&ok)[2 * NJ], uint4 (&v)[4], uint64_t (&dirty)[4]) {
  constexpr bool WS1 = WM == 1 || WM == 2;
  // ---- phase B1: which voxels does this frame update?  Then a wave-uniform early-out: 10-25 % of the (block, frame) pairs the frustum
  // test lets through update nothing (blocks behind the surface, beyond the integration distance, over invalid depth, in the sliver
  // between the image border and the conservative sphere test) -- everything below (weighted mean, weights, selects: ~40 % of the
  // instructions of a frame) is skipped for them.  Measured on the configs[1] stream with the CPU checker: tools/waste.py.
  const float wn = (float)P.wsample;
  const uint32_t round_mask = P.colour_round ? 0x010101u : 0u;             // scalar registers
  const uint32_t first_mask = P.colour_first ? 0x00FFFFFFu : 0xFF000000u;
  constexpr bool wdep = WM == 3;   // depth-dependent observation weight (sf_params::weight_mode 1): its own instantiation, the generic path pays nothing for it
  const uint32_t maxd_bits = __float_as_uint(P.maxd);
  v2f q[NJ], sdfc[NJ];
  v2f wnv[NJ];          // weight of this observation per voxel (a splat unless wdep)
  int wni[2 * NJ];
  uint32_t ncw[2 * NJ];
  bool upd[2 * NJ];
  bool sat[2 * NJ];
  bool any_upd = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const v2f dk = {d[2 * j], d[2 * j + 1]};
    v2f sdf = dk - pz[j];
    const v2f t = pk_fma(splat(FV.tscale), dk, splat(FV.tbase));
#pragma unroll
    for (int hx = 0; hx < 2; hx++) {
      // valid depth (-inf has the sign bit set, valid depths are positive) below the integration distance, not behind the band
      upd[2 * j + hx] = ok[2 * j + hx] && (__float_as_uint(dk[hx]) < maxd_bits) && (sdf[hx] > -t[hx]);
      sdf[hx] = min_f32(sdf[hx], t[hx]);
      sat[2 * j + hx] = false;
      any_upd = any_upd || upd[2 * j + hx];
      wni[2 * j + hx] = wdep ? depth_weight(P, dk[hx]) : P.wsample;
    }
    wnv[j] = wdep ? (v2f){(float)wni[2 * j], (float)wni[2 * j + 1]} : splat(wn);
    sdfc[j] = sdf;
  }
  if (!__any((int)any_upd)) return;
  // ---- phase B2: new values into temporaries (the tile itself stays untouched until the end)
  bool slow = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const v2f sdf = sdfc[j];
    const uint32_t cwj[2] = {v[J0 + j].y, v[J0 + j].w};
    const v2f wo = {(float)(cwj[0] >> 24), (float)(cwj[1] >> 24)};
    const v2f old = {__uint_as_float(v[J0 + j].x), __uint_as_float(v[J0 + j].z)};
    if (SIGN > 0) {
      const v2f n = pk_fma(old, wo, WS1 ? sdf : sdf * wnv[j]);  // x * 1.0f == x bit for bit
      const v2f m = wo + wnv[j];
      if (TAB) {
        q[j] = quot_rn(n, m, rcp_m[j]);
        slow = slow || (fabsf(n.x) < 0x1p-100f) || (fabsf(n.y) < 0x1p-100f);
      } else {
        q[j] = (v2f){n.x / m.x, n.y / m.y};
      }
#pragma unroll
      for (int hx = 0; hx < 2; hx++) {
        const uint32_t cw = cwj[hx];
        const uint32_t w = cw >> 24;
        uint32_t rgb = cw;   // bytes 0..2 = the accumulated colour (byte 3, the weight, is masked out where the word is assembled)
        if (COLOR) {
          // (a + b) / 2 per channel (integer division) is ONE instruction on this ISA: v_lerp_u8 D = per byte (S0 + S1 + S2[bit 0 of the
          // byte]) >> 1 -- the sum is formed in 9 bits, nothing crosses a byte.  colour_round 1 (combineVoxel upstream, DESIGN 6b:
          // (uchar)(0.5f a + 0.5f b + 0.5f) = (a + b + 1) >> 1) is the same instruction with bit 0 of every colour byte of S2 set.  The weight byte
          // of the result is garbage and never used.  colour_first 1: "first observation" is a black accumulated colour instead of a zero weight --
          // a wave-uniform mask on the word, not a branch.  (Round 3 spent 12 VALU instructions per voxel on this blend: xor / and / shift / add3.)
          const uint32_t ck = c[2 * j + hx];
          const uint32_t avg = __builtin_amdgcn_lerp(cw, ck, round_mask);
          // COLOR 2 (colour_first == 0, the shipped semantics): "no observation yet" = the weight byte is zero = the word is below 2^24 -- one compare
          // against a literal instead of a mask and a compare
          const bool first = COLOR == 2 ? cw < 0x01000000u : (cw & first_mask) == 0u;
          rgb = first ? ck : avg;
        }
        if (WM == 2) {
          if (COLOR) {
            // weight byte + 1 saturating at 255: an unsigned add with the clamp bit on the whole word saturates to 0xFFFFFFFF exactly when the
            // weight was 255; only byte 3 of the sum is kept
            ncw[2 * j + hx] = (rgb & 0x00FFFFFFu) | (add_sat_u32(cw, 0x01000000u) & 0xFF000000u);
          } else {
            // without colour "keep the word at 255" is "do not touch the word": the carry of the add folds into the final select
            uint32_t inc;
            const bool full = __builtin_add_overflow(cw, 0x01000000u, &inc);
            ncw[2 * j + hx] = inc;
            sat[2 * j + hx] = full;
          }
        } else {
          uint32_t nw = w + (uint32_t)wni[2 * j + hx];
          if (nw > (uint32_t)P.wmax) nw = (uint32_t)P.wmax;
          ncw[2 * j + hx] = (rgb & 0x00FFFFFFu) | (nw << 24);
        }
      }
    } else {
      const v2f n = pk_fma(old, wo, -(sdf * wnv[j]));
      const v2f m = wo - wnv[j];
      q[j] = (v2f){n.x / m.x, n.y / m.y};  // discarded when the weight drops to <= 0 (then m <= 0)
#pragma unroll
      for (int hx = 0; hx < 2; hx++) {
        const int nw = (int)(cwj[hx] >> 24) - wni[2 * j + hx];
        if (nw <= 0) { q[j][hx] = __uint_as_float(0u); ncw[2 * j + hx] = 0u; }
        else ncw[2 * j + hx] = (cwj[hx] & 0xFFFFFFu) | ((uint32_t)nw << 24);
      }
    }
  }
  if (SIGN > 0 && TAB && __builtin_expect(__any((int)slow), 0)) {
    // some numerator of the wave is in the underflow range (practically: never): plain IEEE division for this tile
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const v2f wo = {(float)(v[J0 + j].y >> 24), (float)(v[J0 + j].w >> 24)};
      const v2f old = {__uint_as_float(v[J0 + j].x), __uint_as_float(v[J0 + j].z)};
      const v2f n = pk_fma(old, wo, sdfc[j] * wnv[j]);
      const v2f m = wo + wnv[j];
      q[j] = (v2f){n.x / m.x, n.y / m.y};
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    v[J0 + j].x = upd[2 * j] ? __float_as_uint(q[j].x) : v[J0 + j].x;
    v[J0 + j].y = (upd[2 * j] && !sat[2 * j]) ? ncw[2 * j] : v[J0 + j].y;
    v[J0 + j].z = upd[2 * j + 1] ? __float_as_uint(q[j].y) : v[J0 + j].z;
    v[J0 + j].w = (upd[2 * j + 1] && !sat[2 * j + 1]) ? ncw[2 * j + 1] : v[J0 + j].w;
    if (ROWS) dirty[J0 + j] |= __ballot(upd[2 * j] || upd[2 * j + 1]);
  }
  if (!ROWS) dirty[0] = ~0ull;   // reached only when some lane of the wave updates a voxel (the early-out above)
}


// The depth (colour) image of one frame as a buffer resource: gathers address it as SGPR descriptor + 32-bit VGPR byte offset (one
// v_mul_u32_u24 + one v_lshl_add_u32 per voxel instead of a 64-bit multiply-add, a select and a 64-bit shift-add), and an offset past
// the image -- a voxel that projects outside, whose index is garbage -- reads 0 instead of faulting; such voxels are masked by `ok`.
__device__ inline __amdgpu_buffer_rsrc_t image_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword data format (gfx9)
}

template <int SIGN, int COLOR, bool TAB, int WM, int J0, int NJ, bool ROWS, bool XR = false>
__device__ inline void fuse_rows(const ParamsK& P, const float* __restrict__ Ti, const float* __restrict__ depthf,
                                 const uint2* __restrict__ texel, const float* rtab, v2f wx, float wy, const float (&wz)[4], const v2f (&wxp)[4],
                                 uint4 (&v)[4], uint64_t (&dirty)[4]) {
  v2f pz[NJ], rcp_m[NJ];
  float d[2 * NJ];
  uint32_t c[2 * NJ];
  bool ok[2 * NJ];
  uint32_t pix[2 * NJ];
  // the weights are known before anything else: start the eight table reads now, they are consumed in phase B
  if (TAB) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      // weight_sample == 1 in the shipped parameters (WM >= 1): a constant index offset folds into the LDS instruction's immediate
      const uint32_t ws = (WM == 1 || WM == 2) ? 1u : (uint32_t)P.wsample;
      rcp_m[j] = (v2f){rtab[(v[J0 + j].y >> 24) + ws], rtab[(v[J0 + j].w >> 24) + ws]};
    }
  }
  // ---- phase A: project; then the gathers, all issued together
  const FrameV FV = frame_constants(P, Ti);
  if (XR) fuse_project_xr<J0, NJ>(P, FV, wxp, wy, wz[0], pz, pix, ok);   // wz[0]: the lane's one z
  else fuse_project<J0, NJ, false>(P, FV, wx, wy, wz, pz, pix, ok);
#ifdef SF_ABLATE_GATHER   // measurement only (wrong voxels): every lane gathers ONE texel per row -- what do the gathers' cache look-ups cost a pass?
#pragma unroll
  for (int k = 0; k < 2 * NJ; k++) pix[k] = (uint32_t)(SF_ABLATE_GATHER == 1 ? 0 : (pix[k] & ~63u));
#endif
  const uint32_t img_bytes = (uint32_t)(P.W * P.H) * 4u;
  if (COLOR) {
    // RGB-D: depth and colour of a pixel sit side by side in the pre-pass's texel plane -- one 8-byte gather per voxel (two 4-byte gathers into
    // two planes were 16 requests per lane and frame; the texture-address unit, not the vector ALU, was the busier one)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rt = image_rsrc(texel, 2u * img_bytes);
#pragma unroll
    for (int k = 0; k < 2 * NJ; k++) {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rt, pix[k] << 3, 0, 0);
      d[k] = __uint_as_float(t.x);
      c[k] = t.y;
    }
  } else {
    const __amdgpu_buffer_rsrc_t rd = image_rsrc(depthf, img_bytes);
#pragma unroll
    for (int k = 0; k < 2 * NJ; k++) d[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, pix[k] << 2, 0, 0));
  }
  fuse_update<SIGN, COLOR, TAB, WM, J0, NJ, ROWS>(P, FV, rcp_m, d, c, pz, ok, v, dirty);
}

// 4 waves per SIMD (<= 128 VGPRs).  Tried for the one-frame-per-launch case: 5 waves / 96 VGPRs with the tile in two
// half passes -- the spills cost more than the occupancy buys (183 us vs 112 us per launch).
#ifndef SF_INT_WAVES
#define SF_INT_WAVES 5   // workgroups (of 4 waves) per CU the register budget of k_integrate is set for (plain pairs: 91 registers)
#endif
#ifndef SF_INT_NJ
#define SF_INT_NJ 4      // rows of the tile fused together per frame: 4 = the whole tile at once, 2 / 1 = in halves / quarters (fewer live registers)
#endif
// NJ = 2 (the tile in halves: 63 registers, 8 waves per SIMD) looks 15 % faster in the two-stream schedule (696 against 814 us per launch) only because its
// waves take every register of the SIMDs and the allocation kernel on the other stream starves (372 -> 818 us): the pass as a whole is slower
// (profiles/r05_integrate_ab.txt).  Alone the two variants are within a few per cent.  NJ = 2 runs the LAST pass of a sf_fuser_integrate_batch_device call
// -- nothing is queued behind that pass, no front chain runs beside it: +0.8 % on a 20-frame call, measured --, every other pass NJ = 4 at 5 waves.  Same
// voxels either way (tests/test_gpu_tsdf.py::test_batched_pass_equals_frame_by_frame runs both).
template <int SIGN, int COLOR, bool TAB, int WM, bool ROWS, int NJ = SF_INT_NJ, bool XR = false>
__global__ __launch_bounds__(256, NJ == 2 ? (XR ? 7 : 8) : SF_INT_WAVES) void k_integrate(uint4* __restrict__ voxels, const uint64_t* __restrict__ block_keys,
                                                   const int32_t* __restrict__ compact, const uint32_t* __restrict__ cmask,
                                                   const float* __restrict__ depthf_all, const uint2* __restrict__ texel_all,
                                                   int32_t* counters, int32_t* host_mirror, int compact_counter, int xcd_walk, ParamsK P,
                                                   BatchTi B) {
  __shared__ float s_rtab[RTAB];  // correctly rounded 1/m for the weighted-mean division (fuse_tile)
#ifdef SF_INT_PAD_VGPR
  // occupancy experiment: name a high register so that the kernel's allocation is SF_INT_PAD_VGPR + 1 registers whatever it uses (fewer waves per SIMD,
  // slots of the size the allocation kernel's waves need)
#define SF_STR2(x) #x
#define SF_STR(x) SF_STR2(x)
  asm volatile("" ::: "v" SF_STR(SF_INT_PAD_VGPR));
#endif
  if (TAB) {
    for (int i = threadIdx.x; i < RTAB; i += 256) s_rtab[i] = 1.0f / (float)(i > 0 ? i : 1);
    __syncthreads();
  }
  const int n = counters[compact_counter];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicExch(&counters[C_LAST_BLOCKS], counters[compact_counter + 1]);  // counters share cache lines with words the front stream updates atomically
    if (host_mirror) *host_mirror = n;
  }
  // pair layout: lane l, load j = uint4 64 j + l = voxels x = (2l) & 7 (+1), y = (l >> 2) & 7, z = 2j + (l >> 5);
  // x-row layout (XR): lane l, load j = uint4 4 l + j = voxels x = 2j (+1), y = l & 7, z = l >> 3 (the lane's 64 contiguous bytes of the tile)
  const int lx = XR ? 0 : (2 * lane) & 7;
  const int ly = XR ? lane & 7 : (lane >> 2) & 7;
  const int lzb = XR ? lane >> 3 : lane >> 5;
  const size_t npx = (size_t)P.W * P.H;
  // XCD-aware walk of the list: workgroup b runs on XCD b % 8 (observed placement; a speed hint only, any placement is
  // correct).  Each XCD takes ONE contiguous eighth of the list -- neighbouring list entries are neighbouring blocks
  // that gather neighbouring depth pixels, so an XCD's 4 MiB L2 holds its own part of the batch's depth images instead
  // of all eight L2s each cycling through all 16 x 1.2 MB.  xcd_walk == 0: plain grid-stride order.
  const int wg_total = (n + 3) >> 2;                         // workgroups' worth of list entries
  const int chunk = xcd_walk ? (wg_total + 7) >> 3 : wg_total;
  const int lanes = xcd_walk ? 8 : 1;                        // interleaved sub-grids
  const int sub = xcd_walk ? (int)(blockIdx.x & 7) : 0;
  const int per_sub = max(1, (int)gridDim.x / lanes);
  for (int loc = xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; loc < chunk; loc += per_sub) {
    const int i = ((sub * chunk + loc) << 2) + wave;
    if (i >= n) continue;
    const int slot = compact[i];
    uint32_t frames = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmask[i]);  // wave-uniform: the frame loop runs on the scalar unit
    int bx, by, bz;
    unpack_key(block_keys[slot], bx, by, bz);
    uint4* vb = voxels + (size_t)slot * 256;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = vb[XR ? lane * 4 + j : j * 64 + lane];
    const v2f wx = {(float)(8 * bx + lx) * P.voxel, (float)(8 * bx + lx + 1) * P.voxel};
    const float wy = (float)(8 * by + ly) * P.voxel;
    float wz[4];
    v2f wxp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      wz[j] = (float)(8 * bz + (XR ? 0 : 2 * j) + lzb) * P.voxel;
      wxp[j] = v2f{(float)(8 * bx + 2 * j) * P.voxel, (float)(8 * bx + 2 * j + 1) * P.voxel};
    }
    uint64_t dirty[4] = {0ull, 0ull, 0ull, 0ull};
    // temporal blocking: the tile stays in registers while every frame of the batch that sees the block is fused
    // into it, in frame order (the same sequence of updates per voxel as frame-by-frame integration)
    while (frames != 0u) {
      const int q = __builtin_ctz(frames);
      frames &= frames - 1u;
      const float* Ti = B.Ti[q];
      const float* __restrict__ depthf = depthf_all + (size_t)q * npx;
      const uint2* __restrict__ texel = texel_all + (size_t)q * npx;
#pragma unroll
      for (int j0 = 0; j0 < 4; j0 += NJ) {
        if (j0 == 0) fuse_rows<SIGN, COLOR, TAB, WM, 0, NJ, ROWS, XR>(P, Ti, depthf, texel, s_rtab, wx, wy, wz, wxp, v, dirty);
        if (j0 == 1) fuse_rows<SIGN, COLOR, TAB, WM, 1 % (5 - NJ), NJ, ROWS, XR>(P, Ti, depthf, texel, s_rtab, wx, wy, wz, wxp, v, dirty);
        if (j0 == 2) fuse_rows<SIGN, COLOR, TAB, WM, 2 % (5 - NJ), NJ, ROWS, XR>(P, Ti, depthf, texel, s_rtab, wx, wy, wz, wxp, v, dirty);
        if (j0 == 3) fuse_rows<SIGN, COLOR, TAB, WM, 3 % (5 - NJ), NJ, ROWS, XR>(P, Ti, depthf, texel, s_rtab, wx, wy, wz, wxp, v, dirty);
      }
    }
    if (ROWS) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((dirty[j] >> lane) & 1ull) vb[XR ? lane * 4 + j : j * 64 + lane] = v[j];
    } else if (dirty[0] != 0ull) {   // wave-uniform: some frame of the pass changed a voxel of this tile -- the whole tile goes back, four 1 KiB stores
#pragma unroll
      for (int j = 0; j < 4; j++) vb[XR ? lane * 4 + j : j * 64 + lane] = v[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K4p: the same update for the HBM-bound regime (ONE frame per launch: a live stream, SF_BATCH=1), software-pipelined.
// In k_integrate every wave is a serial chain  tile load -> project -> 8 depth gathers -> update -> store  and a SIMD holds
// four chains; measured the chains, not HBM, bound it.  Here a persistent wave walks its share of the list
// and keeps three things in flight for LATER tiles while it updates tile k in registers:
//   * tile k+2 and k+3 travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, four 1 KiB requests per tile, no VGPRs) into
//     a two-slot ring per wave;
//   * the eight depth gathers of tile k+1 (projected one turn early) land in LDS as well (global_load_lds_dword: per-lane
//     source address, lane-linear destination), two 2 KiB slots per wave;
// so nothing asynchronous ever targets a VGPR and every wait is a hand-counted s_waitcnt vmcnt(N) (vector-memory operations
// return in order: "at most N outstanding" = everything but the N youngest has landed).  Per turn k the issue order is
//   [tile k+1's stores of the previous turn: S(k-1)]  G(k+1) x8  D(k+3) x4   and the two waits are
//   top : tile k+1 (requested two turns ago) has landed      -- younger: S(k-2)? G(k) 8, D(k+2) 4, S(k-1)  => vmcnt(12)
//   mid : the gathers of tile k (issued last turn) have landed -- younger: D(k+2) 4, S(k-1), G(k+1) 8, D(k+3) 4 => vmcnt(16)
// (stores only make the true count larger, i.e. the waits conservative).  hipcc never sees these loads (it would wait
// vmcnt(0) at every use while an LDS-DMA is in flight); it only sees ordinary ds_reads after the waits.
// LDS per wave: 2 x 4 KiB tiles + 2 x 2 KiB gathers = 12 KiB => 3 workgroups (12 waves, 144 KiB) per CU, and 16 KiB left for a workgroup of
// the next frame's allocation (10.6 KiB for one frame per launch) to run beside it.  Geometry only, no colour:
// the colour variant stays on k_integrate.  Arithmetic = fuse_project / fuse_update, bit-identical to k_integrate.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NT: tile loads and stores carry the non-temporal hint -- for passes whose tile set is many times the 256 MiB Infinity Cache (1 mm voxels:
// 5-7 GB per frame), where keeping streamed tiles on-die only evicts the depth image and the list; below that size the cache hits of
// consecutive frames are worth more (measured), so run_batch picks the variant from the previous pass's list length.
template <bool TAB, int WM, bool NT>
__global__ __launch_bounds__(256, 3) void k_integrate_pipe(uint4* __restrict__ voxels, const uint64_t* __restrict__ block_keys,
                                                        const int32_t* __restrict__ compact, const float* __restrict__ depthf, int32_t* counters,
                                                        int32_t* host_mirror, int compact_counter, ParamsK P, BatchTi B) {
  __shared__ uint4 s_tile[4][2][256];   // per wave: two 4 KiB tile slots
  __shared__ float s_gath[4][2][512];   // per wave: two slots of 8 gathers x 64 lanes
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int stride = (int)gridDim.x * 4;
  const int i0 = (int)blockIdx.x * 4 + wave;
  const int n = counters[compact_counter];
  uint4* const ring = &s_tile[wave][0][0];
  float* const gath = &s_gath[wave][0][0];
  const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)ring);
  const uint32_t gath_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)gath);
  auto slot_of = [&](int i) { return i < n ? __builtin_amdgcn_readfirstlane(compact[i]) : 0; };
  // LDS-DMA of one tile (4 x 1 KiB) into ring slot `ts`; the immediate offset applies to the global AND the LDS address
  auto dma_tile = [&](int slot, int ts) {
    const uint4* src = voxels + (size_t)slot * 256 + lane;
    uint32_t keep;
    if (NT)
      asm volatile(
          "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %[src], off nt\n\tglobal_load_lds_dwordx4 %[src], off offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %[src], off offset:2048 nt\n\tglobal_load_lds_dwordx4 %[src], off offset:3072 nt\n\t"
          "s_mov_b32 m0, %[keep]"
          : [keep] "=&s"(keep)
          : [src] "v"(src), [lds] "s"(ring_lds + (uint32_t)ts * 4096u)
          : "memory");
    else
      asm volatile(
          "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %[src], off\n\tglobal_load_lds_dwordx4 %[src], off offset:1024\n\t"
          "global_load_lds_dwordx4 %[src], off offset:2048\n\tglobal_load_lds_dwordx4 %[src], off offset:3072\n\t"
          "s_mov_b32 m0, %[keep]"
          : [keep] "=&s"(keep)
          : [src] "v"(src), [lds] "s"(ring_lds + (uint32_t)ts * 4096u)
          : "memory");
  };
  // the 8 gathers of one tile into gather slot `gs` (request j -> bytes [256 j, 256 j + 256) of the slot)
  auto gather8 = [&](const uint32_t (&pix)[8], int gs) {
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = pix[k] << 2;
    uint32_t keep;
    const uint32_t base = gath_lds + (uint32_t)gs * 2048u;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[b]\n\ts_nop 0\n\tglobal_load_lds_dword %[o0], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %[o1], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x200\n\ts_nop 0\n\tglobal_load_lds_dword %[o2], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x300\n\ts_nop 0\n\tglobal_load_lds_dword %[o3], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x400\n\ts_nop 0\n\tglobal_load_lds_dword %[o4], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x500\n\ts_nop 0\n\tglobal_load_lds_dword %[o5], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x600\n\ts_nop 0\n\tglobal_load_lds_dword %[o6], %[d]\n\t"
        "s_add_u32 m0, %[b], 0x700\n\ts_nop 0\n\tglobal_load_lds_dword %[o7], %[d]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [o4] "v"(o[4]), [o5] "v"(o[5]), [o6] "v"(o[6]), [o7] "v"(o[7]),
          [d] "s"(depthf), [b] "s"(base)
        : "memory", "scc");
  };
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicExch(&counters[C_LAST_BLOCKS], counters[compact_counter + 1]);
    if (host_mirror) *host_mirror = n;
  }
  if (i0 >= n) return;
  const int lx = (2 * lane) & 7;
  const int ly = (lane >> 2) & 7;
  const int lzb = lane >> 5;
  const float* Ti = B.Ti[0];
  const FrameV FV = frame_constants(P, Ti);   // one frame per launch: the constants are the kernel's
  auto project = [&](uint64_t key, v2f (&pz)[4], uint32_t (&pix)[8], uint32_t& okmask) {
    int bx, by, bz;
    unpack_key(key, bx, by, bz);
    const v2f wx = {(float)(8 * bx + lx) * P.voxel, (float)(8 * bx + lx + 1) * P.voxel};
    const float wy = (float)(8 * by + ly) * P.voxel;
    float wz[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wz[j] = (float)(8 * bz + 2 * j + lzb) * P.voxel;
    bool ok[8];
    fuse_project<0, 4, true>(P, FV, wx, wy, wz, pz, pix, ok);
    okmask = 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) okmask |= ok[k] ? (1u << k) : 0u;
  };
  // ---- prologue: tiles 0 and 1 requested, tile 0 read and projected, its gathers and tile 2 requested.
  // List entries and block keys are wave-uniform scalar loads fetched ahead of their use (slot of tile k+4 and key of tile
  // k+2 during turn k), so that no dependent scalar round trip ever opens a turn.
  int i = i0;
  int slot = slot_of(i), slot1 = slot_of(i + stride), slot2 = slot_of(i + 2 * stride), slot3 = slot_of(i + 3 * stride);
  dma_tile(slot, 0);
  if (i + stride < n) dma_tile(slot1, 1);
  const uint64_t key0 = block_keys[slot];
  uint64_t key1 = i + stride < n ? block_keys[slot1] : 0ull;
  if (i + stride < n) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint4 v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = ring[0 * 256 + j * 64 + lane];
  v2f pz[4];
  uint32_t okmask;
  {
    uint32_t pix[8];
    project(key0, pz, pix, okmask);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tile 0 is in registers: its ring slot may be overwritten
    gather8(pix, 0);
    if (i + 2 * stride < n) dma_tile(slot2, 0);
  }
  int par = 0;  // parity of the current turn: tile k sits in gather slot par, tile k+1 in ring slot par ^ 1
  for (;;) {
    const int i1 = i + stride, i4 = i + 4 * stride;
    const bool has1 = i1 < n, has2 = i + 2 * stride < n, has3 = i + 3 * stride < n;  // wave-uniform
    uint4 vn[4];
    v2f pzn[4];
    uint32_t okn = 0u;
    int slot4 = 0;
    uint64_t key2 = 0ull;
    if (has1) {
      // top: tile k+1 has landed (younger than it: at least G(k) 8 + D(k+2) 4 when tile k+2 exists, else only G(k) 8)
      if (has2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; j++) vn[j] = ring[(par ^ 1) * 256 + j * 64 + lane];
      uint32_t pixn[8];
      project(key1, pzn, pixn, okn);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tile k+1 is in registers before its slot is handed to tile k+3
      gather8(pixn, par ^ 1);
      if (has3) dma_tile(slot3, par ^ 1);
      // scalar prefetch for later turns (after the lgkmcnt wait above, so that it is not waited for here)
      if (i4 < n) slot4 = __builtin_amdgcn_readfirstlane(compact[i4]);
      if (has2) key2 = block_keys[slot2];
      // mid: the gathers of tile k have landed (younger: D(k+2) 4 if any, G(k+1) 8, D(k+3) 4 if any)
      if (has3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (has2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = gath[par * 512 + k * 64 + lane];
    // RN(1 / (weight + sample)) by v_rcp_f32 + two Newton steps (recip_rn: correctly rounded for every normal divisor, the same bits
    // as k_integrate's LDS table) -- this kernel has VALU slots to spare and its LDS decides who may run beside it: 48 KiB per
    // workgroup x 3 leaves 16 KiB per CU, room for one workgroup of the NEXT frame's allocation / compaction on the front stream
    v2f rcp_m[4];
    if (TAB) {
#pragma unroll
      for (int j = 0; j < 4; j++) rcp_m[j] = recip_rn((v2f){(float)((v[j].y >> 24) + (uint32_t)P.wsample), (float)((v[j].w >> 24) + (uint32_t)P.wsample)});
    }
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ok[k] = (okmask >> k) & 1u;
    uint32_t cdummy[8];
    uint64_t dirty[4] = {0ull, 0ull, 0ull, 0ull};
    fuse_update<1, 0, TAB, WM, 0, 4>(P, FV, rcp_m, d, cdummy, pz, ok, v, dirty);  // consumes d: the gather slot is free again
    uint4* vb = voxels + (size_t)slot * 256;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if ((dirty[j] >> lane) & 1ull) {
        if (NT) __builtin_nontemporal_store((u32x4){v[j].x, v[j].y, v[j].z, v[j].w}, reinterpret_cast<u32x4*>(&vb[j * 64 + lane]));
        else vb[j * 64 + lane] = v[j];
      }
    if (!has1) break;
    i = i1;
    slot = slot1; slot1 = slot2; slot2 = slot3; slot3 = slot4;
    key1 = key2;
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = vn[j]; pz[j] = pzn[j]; }
    okmask = okn;
    par ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------
// Measurement aid: the memory traffic of k_integrate WITHOUT its arithmetic -- every tile of the compact list is read
// with the same four 1 KiB loads per wave and (mode 0) written back unchanged, same grid, same list walk.  Its duration
// is the ceiling the access pattern itself (scattered 4 KiB read-modify-write) allows on this HBM; bench.py reports
// the one-frame-per-launch kernel against it (sf_fuser_calib_tile_rmw).  The volume is left bit-identical.
// ---------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256, 4) void k_tile_rmw(uint4* __restrict__ voxels, const int32_t* __restrict__ compact,
                                                  const int32_t* __restrict__ counters, int compact_counter, int xcd_walk, int read_only,
                                                  uint32_t* sink) {
  const int n = counters[compact_counter];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wg_total = (n + 3) >> 2;
  const int chunk = xcd_walk ? (wg_total + 7) >> 3 : wg_total;
  const int lanes = xcd_walk ? 8 : 1;
  const int sub = xcd_walk ? (int)(blockIdx.x & 7) : 0;
  const int per_sub = max(1, (int)gridDim.x / lanes);
  uint32_t acc = 0;
  for (int loc = xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; loc < chunk; loc += per_sub) {
    const int i = ((sub * chunk + loc) << 2) + wave;
    if (i >= n) continue;
    uint4* vb = voxels + (size_t)compact[i] * 256;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (NT) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&vb[j * 64 + lane])); v[j] = make_uint4(t.x, t.y, t.z, t.w); }
      else v[j] = vb[j * 64 + lane];
    }
    if (read_only) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        asm volatile("" : "+v"(v[j].x));  // opaque to the optimiser: the store below stays
        if (NT) __builtin_nontemporal_store((u32x4){v[j].x, v[j].y, v[j].z, v[j].w}, reinterpret_cast<u32x4*>(&vb[j * 64 + lane]));
        else vb[j * 64 + lane] = v[j];
      }
    }
  }
  if (read_only && acc == 0x9E3779B9u) *sink = acc;
}

// The same traffic taken apart (sf_fuser_calib_tile_rmw_ex): WHICH tiles -- the pass's list (scattered over the pool) or tiles 0 .. n - 1 of the pool
// (one contiguous span of the same size) -- and HOW a wave turns from reading to writing -- tile by tile, or G tiles read and then G tiles written.
// If the contiguous copy runs no faster than the scattered one, the 4 KiB granularity is not what holds the pattern below the read-only rate; if the
// batched turnaround does not either, it is HBM's read / write mix itself.
template <bool NT, int G>
__global__ __launch_bounds__(256, 4) void k_tile_rmw_ex(uint4* __restrict__ voxels, const int32_t* __restrict__ compact, const int32_t* __restrict__ counters,
                                                     int compact_counter, int xcd_walk, int read_only, int contiguous, uint32_t* sink) {
  const int n = counters[compact_counter];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int groups = (n + G - 1) / G;                 // wave-sized units of work: G tiles each
  const int wg_total = (groups + 3) >> 2;
  const int chunk = xcd_walk ? (wg_total + 7) >> 3 : wg_total;
  const int lanes = xcd_walk ? 8 : 1;
  const int sub = xcd_walk ? (int)(blockIdx.x & 7) : 0;
  const int per_sub = max(1, (int)gridDim.x / lanes);
  uint32_t acc = 0;
  for (int loc = xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; loc < chunk; loc += per_sub) {
    const int u = ((sub * chunk + loc) << 2) + wave;
    if (u >= groups) continue;
    uint4 v[G][4];
    uint4* vb[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int i = min(u * G + g, n - 1);            // the last group repeats its last tile: written back unchanged twice
      vb[g] = voxels + (size_t)(contiguous ? i : compact[i]) * 256;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (NT) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&vb[g][j * 64 + lane])); v[g][j] = make_uint4(t.x, t.y, t.z, t.w); }
        else v[g][j] = vb[g][j * 64 + lane];
      }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (read_only) { acc ^= v[g][j].x ^ v[g][j].y ^ v[g][j].z ^ v[g][j].w; continue; }
        asm volatile("" : "+v"(v[g][j].x));  // opaque to the optimiser: the store below stays
        if (NT) __builtin_nontemporal_store((u32x4){v[g][j].x, v[g][j].y, v[g][j].z, v[g][j].w}, reinterpret_cast<u32x4*>(&vb[g][j * 64 + lane]));
        else vb[g][j * 64 + lane] = v[g][j];
      }
    }
  }
  if (read_only && acc == 0x9E3779B9u) *sink = acc;
}

// ---------------------------------------------------------------------------------------------------
// Garbage collection: one 256-thread workgroup per live block; min |sdf| over observed
// voxels and max weight reduced through wave shuffles + LDS; freed blocks are zeroed, unlinked
// (tombstone) and pushed back on the heap.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gc(uint4* voxels, uint64_t* block_keys, const int32_t* __restrict__ live,
                                            HashEntry* table, int32_t* heap, int32_t* counters, float thr, ParamsK P) {
  __shared__ float s_min[4];
  __shared__ uint32_t s_max[4];
  const int n = counters[C_EXPORT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = live[i];
    uint4* vb = voxels + (size_t)slot * 256;
    const uint4 v = vb[threadIdx.x];
    float mn = INFINITY;
    uint32_t mw = 0;
    const uint32_t w0 = v.y >> 24, w1 = v.w >> 24;
    if (w0 > 0) mn = fminf(mn, fabsf(__uint_as_float(v.x)));
    if (w1 > 0) mn = fminf(mn, fabsf(__uint_as_float(v.z)));
    mw = max(w0, w1);
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, o));
      mw = max(mw, (uint32_t)__shfl_xor((int)mw, o));
    }
    if (lane == 0) { s_min[wave] = mn; s_max[wave] = mw; }
    __syncthreads();
    mn = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    mw = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    __syncthreads();
    if (mw == 0 || mn >= thr) {
      vb[threadIdx.x] = make_uint4(0, 0, 0, 0);
      if (threadIdx.x == 0) {
        const uint64_t key = block_keys[slot];
        int bx, by, bz;
        unpack_key(key, bx, by, bz);
        uint32_t s = hash_home(P, bx, by, bz);
        for (int probe = 0; probe < MAX_PROBES; ++probe) {
          if (table[s].key == key) { table[s].key = KEY_TOMB; table[s].ptr = -1; break; }
          if (table[s].key == KEY_EMPTY) break;
          s++;
          if (s == P.total_slots) s = 0;
        }
        block_keys[slot] = KEY_EMPTY;
        const int at = atomicAdd(&counters[C_HEAP_FREE], 1);
        heap[at] = slot;
        atomicAdd(&counters[C_GC_FREED], 1);
      }
    }
  }
}

// After a collection that freed blocks: the hash table rebuilt from the directory.  Lock-free open addressing cannot reuse a tombstone
// safely while other lanes insert the same key (one claims the tombstone, another has already walked past it and claims an empty slot
// further on), and tombstones that are never reused only lengthen every probe chain over a long scan (round-1 finding).  Collection is
// synchronous, so it simply leaves no tombstone behind: table cleared, every live block (ghosts included) re-inserted at its home
// position, block_entry re-pointed.  A surviving block existed before the next batch, so its birth stamp restarts at 0.
__global__ __launch_bounds__(256) void k_rehash(HashEntry* table, const uint64_t* __restrict__ block_keys, int32_t* block_entry, int32_t* counters, ParamsK P) {
  const int hw = counters[C_HIGH_WATER];
  for (int slot = blockIdx.x * 256 + threadIdx.x; slot < hw; slot += gridDim.x * 256) {
    const uint64_t key = block_keys[slot];
    if (key == KEY_EMPTY) continue;
    int bx, by, bz;
    unpack_key(key, bx, by, bz);
    uint32_t at = hash_home(P, bx, by, bz);
    for (int probe = 0; probe < MAX_PROBES; ++probe) {
      if (atomicCAS((unsigned long long*)&table[at].key, (unsigned long long)KEY_EMPTY, (unsigned long long)key) == KEY_EMPTY) {
        table[at].ptr = slot;
        table[at].birth = 0u;
        block_entry[slot] = (int32_t)at;
        atomicAdd(&counters[C_SLOTS_USED], 1);
        break;
      }
      at++;
      if (at == P.total_slots) at = 0;
      // no slot within MAX_PROBES (the rebuild inserts in directory order, a key can land further from home than it was): the block stays in
      // the directory but cannot be looked up -- counted, sf_fuser_garbage_collect reports SF_ERR_CAPACITY
      if (probe == MAX_PROBES - 1) atomicAdd(&counters[C_ALLOC_FAIL], 1);
    }
  }
}

__global__ void k_init_heap(int32_t* heap, uint64_t* block_keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { heap[i] = n - 1 - i; block_keys[i] = KEY_EMPTY; }
}

__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ voxels, const uint64_t* __restrict__ block_keys,
                                                const int32_t* __restrict__ live, int n, int32_t* coords, uint4* out) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = live[i];
    out[(size_t)i * 256 + threadIdx.x] = voxels[(size_t)slot * 256 + threadIdx.x];
    if (threadIdx.x == 0) {
      int bx, by, bz;
      unpack_key(block_keys[slot], bx, by, bz);
      coords[3 * i] = bx; coords[3 * i + 1] = by; coords[3 * i + 2] = bz;
    }
  }
}

// Filtered export: the live blocks whose coordinate on `axis` lies in [lo, hi) (axis == -1: all; axis == -2: the boundary layers
// of this fuser's slab / stripes), appended in no particular order.  One workgroup per candidate block.
__global__ __launch_bounds__(256) void k_gather_where(const uint4* __restrict__ voxels, const uint64_t* __restrict__ block_keys,
                                                      const int32_t* __restrict__ live, int n, int axis, int lo, int hi, int capacity,
                                                      int32_t* counter, int32_t* coords, uint4* out, ParamsK P) {
  __shared__ int s_pos;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = live[i];
    int bx, by, bz;
    unpack_key(block_keys[slot], bx, by, bz);
    const int c = axis == 0 ? bx : (axis == 1 ? by : bz);
    if (axis >= 0 && (c < lo || c >= hi)) continue;  // uniform per workgroup
    if (axis == -2 && !slab_boundary(P, bx, by, bz)) continue;  // the boundary layers of this fuser's slab / stripes
    if (threadIdx.x == 0) s_pos = atomicAdd(counter, 1);
    __syncthreads();
    const int pos = s_pos;
    if (pos < capacity && out != nullptr) {
      out[(size_t)pos * 256 + threadIdx.x] = voxels[(size_t)slot * 256 + threadIdx.x];
      if (threadIdx.x == 0) { coords[3 * pos] = bx; coords[3 * pos + 1] = by; coords[3 * pos + 2] = bz; }
    }
    __syncthreads();
  }
}

// Import: one workgroup per block; lane 0 finds or creates the entry (+ heap pop), all lanes copy the 4 KiB tile.  only_wanted: of an
// all-gathered payload keep just the blocks this fuser needs as ghosts (slab_wants_ghost), counted in C_IMPORTED.
__global__ __launch_bounds__(256) void k_import(const int32_t* __restrict__ coords, const uint4* __restrict__ src, int n, int ghost, int only_wanted,
                                                uint4* voxels, HashEntry* table, int32_t* heap, uint64_t* block_keys, int32_t* block_entry,
                                                uint8_t* block_flags, int32_t* counters, ParamsK P) {
  __shared__ int s_slot;
  const HashRefs h{table, heap, block_keys, block_entry, block_flags, counters};
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    if (threadIdx.x == 0) {
      const int bx = coords[3 * i], by = coords[3 * i + 1], bz = coords[3 * i + 2];
      int slot = -1;
      if (!only_wanted || slab_wants_ghost(P, bx, by, bz)) {
        const uint64_t key = pack_key(bx, by, bz);
        HashEntry* e = hash_find_or_claim(h, P, key, bx, by, bz, 0u);
        if (e) {
          atomicAdd(&counters[C_SLOTS_USED], 1);
          give_block(h, e, key, atomicSub(&counters[C_HEAP_FREE], 1) - 1);
          slot = e->ptr >= 0 && block_keys[e->ptr] == key ? e->ptr : -1;  // -1: heap exhausted
        } else {
          slot = hash_lookup(table, P, bx, by, bz);  // already present (re-import): overwrite
        }
        if (slot >= 0) { block_flags[slot] = ghost ? 1 : 0; atomicAdd(&counters[C_IMPORTED], 1); }
      }
      s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot >= 0) voxels[(size_t)slot * 256 + threadIdx.x] = src[(size_t)i * 256 + threadIdx.x];
    __syncthreads();
  }
}

}  // namespace

// ======================================================================================================
// host side
// ======================================================================================================
hipError_t sf_quiesce(sf_fuser* f) {
  hipError_t e = hipSuccess;
  if (f->front) e = hipStreamSynchronize(f->front);
  if (f->front_lo) { const hipError_t e1 = hipStreamSynchronize(f->front_lo); if (e == hipSuccess) e = e1; }
  const hipError_t e2 = hipStreamSynchronize(f->stream);
  return e != hipSuccess ? e : e2;
}

namespace {

// per-frame constants in double, rounded once to float.
bool frame_setup(const sf_params& p, const float* pose, FrameK& f) {
  if (pose[0] == -INFINITY) return false;
  for (int i = 0; i < 12; i++) f.T[i] = pose[i];
  const double a00 = pose[0], a01 = pose[1], a02 = pose[2], t0 = pose[3];
  const double a10 = pose[4], a11 = pose[5], a12 = pose[6], t1 = pose[7];
  const double a20 = pose[8], a21 = pose[9], a22 = pose[10], t2 = pose[11];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double inv[9] = {c00 / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det,
                         c01 / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det,
                         c02 / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det};
  for (int r = 0; r < 3; r++) {
    const double ti = -(inv[3 * r] * t0 + inv[3 * r + 1] * t1 + inv[3 * r + 2] * t2);
    f.Ti[4 * r] = (float)inv[3 * r];
    f.Ti[4 * r + 1] = (float)inv[3 * r + 1];
    f.Ti[4 * r + 2] = (float)inv[3 * r + 2];
    f.Ti[4 * r + 3] = (float)ti;
  }
  const double rad = 4.0 * std::sqrt(3.0) * (double)p.voxel_size;
  f.radius = (float)rad;
  const double fx = p.fx, fy = p.fy, mx = p.mx, my = p.my;
  // lower planes at u = -1.5: the voxel rule casts (int)(u + 0.5f) before it tests the pixel, and the cast truncates (-1, 0) to pixel 0
  // (DESIGN.md 3.2 / 3.5); the block test must keep every block the voxel rule would touch
  const double xl = mx + 1.5, xh = ((double)p.depth_width - 0.5) - mx;
  const double yl = my + 1.5, yh = ((double)p.depth_height - 0.5) - my;
  f.xa[0] = (float)fx;  f.xc[0] = (float)xl; f.xr[0] = (float)(rad * std::sqrt(fx * fx + xl * xl));
  f.xa[1] = (float)-fx; f.xc[1] = (float)xh; f.xr[1] = (float)(rad * std::sqrt(fx * fx + xh * xh));
  f.ya[0] = (float)fy;  f.yc[0] = (float)yl; f.yr[0] = (float)(rad * std::sqrt(fy * fy + yl * yl));
  f.ya[1] = (float)-fy; f.yc[1] = (float)yh; f.yr[1] = (float)(rad * std::sqrt(fy * fy + yh * yh));
  f.zfar = p.max_integration_dist + std::fmaf(p.trunc_scale, p.max_integration_dist, p.trunc_base);
  return true;
}

// One batch: n <= f->batch frames with valid poses, all with or all without colour.  The front stream prepares
// batch slot `sl` (pre-pass, allocation, compaction for all n frames, three launches) while the back stream is
// still fusing the previous batch out of the other slot.  Allocation only touches new hash entries / heap slots,
// integrate only the tiles of its own compact list, so the two never write the same data; slot reuse is ordered
// by ev_fused[sl].
}  // namespace

// One frame per launch without colour runs the persistent k_integrate_pipe, which fills every CU: kernels of the next frame
// on the front stream would only get CUs by starving some of its waves (measured: 113 us overlapped vs 90 us alone, and no
// more frames/s), so such a batch goes down ONE stream, pre-pass to integrate.
static bool pipe_batch(const sf_fuser* f, int n, bool color, int sign) {
  const bool tab_ok = f->p.weight_sample >= 1 && f->p.weight_sample <= RTAB - 256 && f->p.weight_mode == 0;
  return sign > 0 && n == 1 && !color && tab_ok && f->pipe_mode != 0;
}
// Whether the next frame's pre-pass / allocation / compaction should run on the front stream beside the persistent kernel.  Measured on
// MI355X (profiles/r02): at 4 mm (48-67 k tiles per frame, front kernels 36 us) running them beside k_integrate_pipe stretches it by more
// than it hides (7.8 k frames/s serial, 7.5 k overlapped, also with the two streams on disjoint CU masks); at 1 mm (1.7 M tiles, front
// kernels 1 ms) it hides 0.5 ms per frame (310 vs 269 frames/s).  So: overlap once the previous pass's tile set is beyond 512 MiB.
static bool big_pass(const sf_fuser* f) { return (uint64_t)(uint32_t)*f->host_mirror * 4096ull > (512ull << 20); }
// The decision is latched at the end of every pass (run_batch) so that a caller's staging (sf_input_stream) and the pass that follows see the
// same answer: host_mirror is written by the device while they run.
bool sf_single_stream_batch(const sf_fuser* f, int n, bool color, int sign) { return pipe_batch(f, n, color, sign) && !f->pipe_beside; }
// the stream the pre-pass of such a batch reads its frames on: where callers must have staged them
hipStream_t sf_input_stream(const sf_fuser* f, int n, bool color, int sign) {
  if (!f->overlap || sf_single_stream_batch(f, n, color, sign)) return f->stream;
  // The front stream has the device's highest priority: in a pass of several frames its short kernels must slip in between the workgroups of the integrate
  // kernel, or the next pass waits for them.  Beside the PERSISTENT kernel of one frame per launch (1 mm voxels: the tile set is far beyond the cache) that
  // priority is what the integrate kernel pays for: the allocation's 72 KiB workgroups, dispatched first, take the LDS its third workgroup per CU needs until
  // they are through -- 0.54 of peak HBM shipped and 322-335 frames/s, against 0.60-0.61 / 360 with the front chain at the MAIN stream's priority and 0.62 / 369
  // at the device's lowest (tools/gpu/r06_zk.sh, r06_zv.sh: two or three runs each, nothing else changed).  So such a frame's front chain goes down a second
  // front stream, `front_lo` (front_prio -1, the default; 1 / 0: always the high-priority / always the second one), at the device's lowest priority.  One thing to
  // know about that: the first stream of a priority class makes the runtime open that class's hardware queues for the life of the PROCESS, and a later sf_fuse_run
  // in the same process -- seven to nine busy streams -- then runs 12 % slower (depth-only end to end 34.2 k -> 30.0 k frames/s; bench.py therefore runs its 1 mm
  // leg last).  A process that does both sets front_lo_lowest 0: the second stream at the main stream's priority, 0.60-0.61 here instead of 0.62.
  const bool lo = f->front_lo != nullptr && (f->front_prio == 0 || (f->front_prio < 0 && pipe_batch(f, n, color, sign)));
  return lo ? f->front_lo : f->front;
}

namespace {

int run_batch(sf_fuser* f, const void* const* d_depth, const void* const* d_rgb, const float* const* poses, int n, int sign, const void* const* d_lay = nullptr) {
  BatchIn in;
  BatchFrames bf;
  BatchTi bt;
  std::memset(&in, 0, sizeof(in));
  std::memset(&bf, 0, sizeof(bf));
  std::memset(&bt, 0, sizeof(bt));
  bf.n = n;
  bf.seq0 = f->frame_seq;
  const bool col = d_rgb != nullptr && d_rgb[0] != nullptr;
  for (int j = 0; j < n; j++) {
    if (!frame_setup(f->p, poses[j], bf.f[j])) return sf::fail(SF_ERR_INVALID_ARG, "run_batch: invalid pose in batch");
    std::memcpy(bt.Ti[j], bf.f[j].Ti, sizeof(bt.Ti[j]));
    in.depth[j] = (const uint16_t*)d_depth[j];
    in.rgb[j] = col ? (const uint8_t*)d_rgb[j] : nullptr;
    in.lay[j] = (col && d_lay) ? (const uint8_t*)d_lay[j] : nullptr;
  }
  f->frame_seq += (uint32_t)n;
  const int npx = f->p.depth_width * f->p.depth_height;
  const int sl = f->slot;
  f->slot ^= 1;
  const int cc = sl ? (int)C_COMPACT_B : (int)C_COMPACT;
  hipStream_t s = f->stream;
  // One frame per launch without colour runs the persistent k_integrate_pipe, which fills every CU: kernels of the next frame
  // on the front stream would only get CUs by starving some of its waves (measured: 113 us overlapped vs 90 us alone, and
  // no more frames/s), so for such a frame everything goes down ONE stream.
  const bool tab_ok = f->p.weight_sample >= 1 && f->p.weight_sample <= RTAB - 256 && f->p.weight_mode == 0;   // the table is indexed by weight + sample
  const bool ws1 = f->p.weight_sample == 1 && f->p.weight_mode == 0;   // every observation weighs exactly 1
  const bool pipe = pipe_batch(f, n, col, sign);
  hipStream_t sa = sf_input_stream(f, n, col, sign);  // callers stage the batch's frames on this stream too
  if (f->overlap && sa != s) {
    if (f->last_front != nullptr && f->last_front != sa) {   // the other front stream served the pass before: this pass's front chain starts behind that one's
      (void)hipEventRecord(f->ev_front_switch, f->last_front);
      (void)hipStreamWaitEvent(sa, f->ev_front_switch, 0);
    }
    f->last_front = sa;
    if (f->serial_tail) {  // single-stream batches came before: the front stream starts behind everything they queued
      (void)hipEventRecord(f->ev_input, s);
      (void)hipStreamWaitEvent(sa, f->ev_input, 0);
      f->serial_tail = false;
    }
    (void)hipStreamWaitEvent(sa, f->ev_fused[sl], 0);
  }
  // one colourless frame at the integration size through the ray-space allocation kernel: that kernel converts the depth itself
  const bool fuse_pre = f->prepass_fuse && f->alloc_ray && n == 1 && sign > 0 && !col && f->pk.inW == 0;
  if (!fuse_pre)
    hipLaunchKernelGGL(k_prepass, dim3((npx / 8 + 255) / 256 + 1, n), dim3(256), 0, sa, in, f->depthf2[sl], f->color2[sl], npx, f->p.depth_shift,
                       f->p.depth_min, f->p.depth_max, f->counters, cc, f->pk, f->ray_kx, f->ray_ky);
  if (sign > 0) {
    // WIN 64 (32 KiB bitmap) has no room for the second bitmap: one frame per workgroup there
    // frames one allocation workgroup walks.  The FIRST pass of a batch call has nothing to run beside: its front chain is pure latency in front of the first
    // integrate launch (a 20-frame call: k_alloc_ray 133 us of a 650 us region at 8 frames per workgroup), so it is cut into more, shorter workgroups
    // (tune "alloc_group_head"); every other pass hides its allocation behind the previous integrate launch and takes the cheaper, longer ones
    const int group = (f->head_pass && f->alloc_group_head > 0) ? std::min(f->alloc_group, f->alloc_group_head) : f->alloc_group;
    // WIN 64 (32 KiB bitmap): a second bitmap of that size makes the workgroup 102 KiB -- one per CU; tune "alloc_group_win64" (default 1: one frame per workgroup)
    const int gf = (f->alloc_win64 && !f->alloc_ray) ? std::min(f->alloc_group_win64, n) : std::min(group, n);
    const dim3 ag((f->p.depth_width + 15) / 16, (f->p.depth_height + 15) / 16, (n + gf - 1) / gf);
    const BrickCache bc{f->brick_on ? f->bricks : nullptr, f->brick_lines - 1u};
#define LAUNCH_ALLOC(WL, MU) \
  hipLaunchKernelGGL((k_alloc<WL, MU>), ag, dim3(256), 0, sa, f->depthf2[sl], f->table, f->heap, f->block_keys, f->block_entry, f->block_flags, f->counters, f->pk, bf, gf, bc)
    // alloc_wgs > 0: at most that many allocation workgroups per CU, by asking for LDS the kernel does not use (160 KiB per CU).  The kernel
    // is latency-bound (barriers, LDS atomics: 35 % VALU utilisation) and, unthrottled, parks 4-5 waves of 72 VGPRs on every SIMD for ~200 us of
    // each pass -- registers the integrate kernel next to it needs for ITS waves (timeline: profiles/r03_timeline_*.txt)
    const unsigned alloc_pad = (f->alloc_wgs > 0 && n > 1) ? (unsigned)std::max(0, (160 * 1024) / (f->alloc_wgs + 1) + 1024 - 23048) : 0u;
#define LAUNCH_ALLOC_RAY(MU) \
  hipLaunchKernelGGL((k_alloc_ray<MU>), ag, dim3(256), alloc_pad, sa, f->depthf2[sl], f->table, f->heap, f->block_keys, f->block_entry, f->block_flags, f->counters, f->pk, bf, gf, f->alloc_ablate, \
                     fuse_pre ? in.depth[0] : (const uint16_t*)nullptr, f->depthf2[sl], cc)
    if (f->alloc_ray) { if (gf == 1) LAUNCH_ALLOC_RAY(false); else LAUNCH_ALLOC_RAY(true); }
    else if (f->alloc_win64) { if (gf == 1) LAUNCH_ALLOC(6, false); else LAUNCH_ALLOC(6, true); }
    else if (gf == 1) LAUNCH_ALLOC(5, false);
    else LAUNCH_ALLOC(5, true);
#undef LAUNCH_ALLOC
#undef LAUNCH_ALLOC_RAY
  }
  if (n > COMPACT_FEW)
    hipLaunchKernelGGL(k_compactify, dim3(f->compact_grid * (1024 / COMPACT_THREADS)), dim3(COMPACT_THREADS), 0, sa, (CompactArgs{f->block_keys, f->block_entry, f->block_flags, f->table, f->compact2[sl],
                       f->cmask2[sl], f->counters, cc, 0, f->pk, bf}));
  else
    hipLaunchKernelGGL(k_compactify_few, dim3(f->compact_grid * (1024 / COMPACT_THREADS)), dim3(COMPACT_THREADS), 0, sa, (CompactArgs{f->block_keys, f->block_entry, f->block_flags, f->table, f->compact2[sl],
                       f->cmask2[sl], f->counters, cc, 0, f->pk, bf}));
  if (f->overlap && sa != s) {
    (void)hipEventRecord(f->ev_compact[sl], sa);
    (void)hipStreamWaitEvent(s, f->ev_compact[sl], 0);
  }
  // grid: enough workgroups (4 blocks each) for the last list length the device reported, +25 %; the kernel's
  // grid-stride loop covers any excess, surplus workgroups exit at once.
  const int last = *f->host_mirror;
  int est = last + last / 4 + 4096;
  int grid = (est + 3) / 4;
  const int grid_max = f->num_cus * 64;
  if (grid > grid_max) grid = grid_max;
  grid = (grid + 7) & ~7;  // whole sub-grids for the XCD-aware walk
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (f->profile) {
    if (f->events_used == f->events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return sf::fail(SF_ERR_DEVICE, "hipEventCreate failed");
      f->events.emplace_back(a, b);
    }
    e0 = f->events[f->events_used].first;
    e1 = f->events[f->events_used].second;
    f->events_used++;
    (void)hipEventRecord(e0, s);
  }
#define LAUNCH_INT_R(SG, CL, TB, W1, RW)                                                                                                       \
  hipLaunchKernelGGL((k_integrate<SG, CL, TB, W1, RW>), dim3(grid), dim3(256), 0, s, f->voxels, f->block_keys, f->compact2[sl], f->cmask2[sl], \
                     f->depthf2[sl], f->color2[sl], f->counters, f->host_mirror, cc, f->xcd_walk ? 1 : 0, f->pk, bt)
  // per-row write-back masks for one frame per launch (HBM-bound) and for deintegration; a pass of several frames (VALU-bound) writes touched tiles whole
#define LAUNCH_INT(SG, CL, TB, W1)                                    \
  do {                                                                \
    if (SG < 0 || n == 1) LAUNCH_INT_R(SG, CL, TB, W1, true);         \
    else LAUNCH_INT_R(1, CL, TB, W1, false);                          \
  } while (0)
  const bool tab = tab_ok;  // the LDS reciprocal table covers weight + sample < 512
  // one frame per launch without colour: the software-pipelined kernel (SF_PIPE=0: always k_integrate)
  if (pipe) {
    const dim3 pg((unsigned)((f->num_cus - f->front_cus) * f->pipe_wgs));   // persistent: exactly what the main stream's CUs hold
    // non-temporal tile traffic once the previous pass's tile set was beyond twice the Infinity Cache (tune "nt": 0 never, 1 always)
    const bool nt = f->nt_mode == 1 || (f->nt_mode < 0 && big_pass(f));
#define LAUNCH_PIPE(WMODE)                                                                                                                     \
  do {                                                                                                                                         \
    if (nt) hipLaunchKernelGGL((k_integrate_pipe<true, WMODE, true>), pg, dim3(256), 0, s, f->voxels, f->block_keys, f->compact2[sl],          \
                               f->depthf2[sl], f->counters, f->host_mirror, cc, f->pk, bt);                                                    \
    else hipLaunchKernelGGL((k_integrate_pipe<true, WMODE, false>), pg, dim3(256), 0, s, f->voxels, f->block_keys, f->compact2[sl],            \
                            f->depthf2[sl], f->counters, f->host_mirror, cc, f->pk, bt);                                                       \
  } while (0)
    if (ws1 && f->pk.wmax == 255) LAUNCH_PIPE(2);
    else if (ws1) LAUNCH_PIPE(1);
    else LAUNCH_PIPE(0);
#undef LAUNCH_PIPE
  } else if (sign > 0) {
    if (ws1 && f->pk.wmax == 255) {   // the shipped setting
      const bool wide = f->tail_pass && f->tail_wide && n > 1;   // the last pass of a batch call: no front chain beside it, the 8-wave variant (k_integrate)
#define LAUNCH_INT_WIDE(CL) hipLaunchKernelGGL((k_integrate<1, CL, true, 2, false, 2>), dim3(grid), dim3(256), 0, s, f->voxels, f->block_keys, f->compact2[sl], f->cmask2[sl], \
                                               f->depthf2[sl], f->color2[sl], f->counters, f->host_mirror, cc, f->xcd_walk ? 1 : 0, f->pk, bt)
      // the x-row lane layout (fuse_project_xr) for every pass of several frames: same voxels, the gathers of one instruction on two image rows instead of five
#define LAUNCH_INT_XR(CL, NJV) hipLaunchKernelGGL((k_integrate<1, CL, true, 2, false, NJV, true>), dim3(grid), dim3(256), 0, s, f->voxels, f->block_keys, f->compact2[sl], f->cmask2[sl], \
                                                  f->depthf2[sl], f->color2[sl], f->counters, f->host_mirror, cc, f->xcd_walk ? 1 : 0, f->pk, bt)
      const bool xr = f->xrow && n > 1;
      if (col && !f->p.colour_first) { if (xr && wide) LAUNCH_INT_XR(2, 2); else if (xr) LAUNCH_INT_XR(2, SF_INT_NJ); else if (wide) LAUNCH_INT_WIDE(2); else LAUNCH_INT(1, 2, true, 2); }
      else if (col) LAUNCH_INT(1, 1, true, 2);
      else { if (xr && wide) LAUNCH_INT_XR(0, 2); else if (xr) LAUNCH_INT_XR(0, SF_INT_NJ); else if (wide) LAUNCH_INT_WIDE(0); else LAUNCH_INT(1, 0, true, 2); }
#undef LAUNCH_INT_XR
#undef LAUNCH_INT_WIDE
    }
    else if (ws1) { if (col) LAUNCH_INT(1, 1, true, 1); else LAUNCH_INT(1, 0, true, 1); }
    else if (tab)                { if (col) LAUNCH_INT(1, 1, true, 0); else LAUNCH_INT(1, 0, true, 0); }
    else if (f->p.weight_mode == 1) { if (col) LAUNCH_INT(1, 1, false, 3); else LAUNCH_INT(1, 0, false, 3); }
    else                         { if (col) LAUNCH_INT(1, 1, false, 0); else LAUNCH_INT(1, 0, false, 0); }
  } else if (f->p.weight_mode == 1) { if (col) LAUNCH_INT(-1, 1, false, 3); else LAUNCH_INT(-1, 0, false, 3); }
  else                           { if (col) LAUNCH_INT(-1, 1, false, 0); else LAUNCH_INT(-1, 0, false, 0); }
#undef LAUNCH_INT
#undef LAUNCH_INT_R
  if (f->profile) (void)hipEventRecord(e1, s);
  if (f->overlap && sa != s) (void)hipEventRecord(f->ev_fused[sl], s);
  if (f->overlap && sa == s) f->serial_tail = true;  // no cross-stream traffic at all while single-stream batches follow each other
  f->pipe_beside = f->pipe_overlap == 1 || (f->pipe_overlap < 0 && big_pass(f));
  if ((f->pipe_beside || f->front_prio == 0) && f->front_lo == nullptr && f->front_cus == 0 && f->overlap) {   // the second front stream, on first need (sf_input_stream)
    int prio_lo = 0, prio_hi = 0;
    hipError_t e_ = hipSuccess;
    if (f->front_lo_lowest) {
      e_ = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      if (e_ == hipSuccess) e_ = hipStreamCreateWithPriority(&f->front_lo, hipStreamNonBlocking, prio_lo);
    } else {
      e_ = hipStreamCreateWithFlags(&f->front_lo, hipStreamNonBlocking);   // the main stream's priority
    }
    if (e_ != hipSuccess) {
      f->front_lo = nullptr;   // (the high-priority one serves)
      (void)hipGetLastError();
    }
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return sf::fail(SF_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(err));
  f->frames_integrated += (uint64_t)n;
  return SF_OK;
}

int run_frame(sf_fuser* f, const void* d_depth, const void* d_rgb, const float* pose, int sign) {
  if (pose[0] == -INFINITY) {
    f->frames_skipped++;
    return sf::fail(SF_ERR_SKIPPED, "frame skipped: camToWorld is -inf (tracking lost)");
  }
  const void* dd[1] = {d_depth};
  const void* dr[1] = {d_rgb};
  const float* pp[1] = {pose};
  return run_batch(f, dd, dr, pp, 1, sign);
}

}  // namespace

SF_API int sf_device_count(int* count) {
  if (!count) return sf::fail(SF_ERR_INVALID_ARG, "count is NULL");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return sf::fail(SF_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return SF_OK;
}

SF_API int sf_fuser_create(const sf_params* p, int device, sf_fuser** out) {
  if (!p || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (p->depth_width <= 0 || p->depth_height <= 0 || !(p->voxel_size > 0) || p->hash_num_buckets == 0 || p->hash_bucket_size == 0 ||
      p->num_sdf_blocks == 0 || !(p->fx > 0) || !(p->fy > 0) || !(p->depth_shift > 0))
    return sf::fail(SF_ERR_INVALID_ARG, "invalid reconstruction parameters");
  if (p->color_width > 0 && p->color_height > 0 && !(p->cfx > 0 && p->cfy > 0))
    return sf::fail(SF_ERR_INVALID_ARG, "colour resolution given without colour intrinsics");
  if ((p->integration_width > 0) != (p->integration_height > 0) || p->integration_width == 1 || p->integration_height == 1 ||
      (p->integration_width > 0 && (p->depth_width < 2 || p->depth_height < 2)))
    return sf::fail(SF_ERR_INVALID_ARG, "integration size %d x %d", p->integration_width, p->integration_height);
  if ((p->frustum_mode | p->colour_round | p->colour_first | p->weight_mode | p->weight_wrap) & ~1)
    return sf::fail(SF_ERR_INVALID_ARG, "frustum_mode / colour_round / colour_first / weight_mode / weight_wrap must be 0 or 1");
  if (p->frustum_mode == 1 && !(p->depth_max > p->depth_min))
    return sf::fail(SF_ERR_INVALID_ARG, "frustum_mode 1 normalises z by the sensor depth range: depth_max must exceed depth_min");
  if ((uint64_t)p->hash_num_buckets * p->hash_bucket_size > 0x7FFFFFFFull || p->num_sdf_blocks > 0x3FFFFFFFu)
    return sf::fail(SF_ERR_INVALID_ARG, "hash table / heap too large for 32-bit slot indices");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return sf::fail(SF_ERR_DEVICE, "no HIP device: libscanfuse has no CPU fallback, the fuser needs an MI355X");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  // (until round 6 the first fuser of a process started sf_fuse_run's side streams and rings on a background thread here -- 90 MB pinned and
  // 330 MB of device memory also for callers that never run a file.  A caller that will -- bin/depthsensing, bench.py, tools/e2e_bench.py -- asks with
  // sf_fuse_run_prepare(file, params, device) before this call; without it the first sf_fuse_run makes its set itself.)
  sf_fuser* f = new sf_fuser();
  // every failure below leaves through sf_fuser_destroy (streams, events, device and pinned allocations made so far)
#define SF_CREATE_CHECK(call)                                                                               \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      sf_fuser_destroy(f);                                                                                  \
      return sf::fail(SF_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_));                        \
    }                                                                                                       \
  } while (0)
  f->p = *p;
  if (f->p.weight_max > 255 && !f->p.weight_wrap) f->p.weight_max = 255;  // uchar weight saturates (SURVEY App. C decision)
  if (f->p.weight_max > 0x00FFFFFF) f->p.weight_max = 0x00FFFFFF;   // weight_wrap: "no limit" for an 8-bit sum; the clamp compare stays exact in 32 bits
  if (f->p.weight_max < 1) f->p.weight_max = 1;
  f->device = device;
  {
    // longest ray segment 2 * trunc(max distance) in blocks decides the LDS window size of k_alloc
    const float seg = 2.0f * (p->trunc_base + p->trunc_scale * p->max_integration_dist) / (8.0f * p->voxel_size);
    f->alloc_win64 = seg > 20.0f;
    // the ray-space window (k_alloc_ray) holds the pencil of a 16x16 pixel tile when 16 blocks span its width plus a few blocks of camera
    // motion inside a batch, and 256 slabs its depth: half a tile at the integration distance within 4 blocks, the longest ray within 250
    const float bsz = 8.0f * p->voxel_size;
    const float half_tile = 8.0f * p->max_integration_dist / std::min(p->fx, p->fy);
    const float reach = (p->max_integration_dist + p->trunc_base + p->trunc_scale * p->max_integration_dist) * 1.25f;
    f->alloc_ray = half_tile / bsz <= 4.0f && reach / bsz <= (float)(RW_DEPTH - 6);
  }
  ParamsK& k = f->pk;
  k.W = p->depth_width; k.H = p->depth_height; k.fx = p->fx; k.fy = p->fy; k.mx = p->mx; k.my = p->my;
  k.inW = k.inH = 0; k.rsx = k.rsy = 1.0f;
  const bool resample = p->integration_width > 0 && p->integration_height > 0 &&
                        (p->integration_width != p->depth_width || p->integration_height != p->depth_height);
  if (resample) {
    // everything behind the pre-pass works at the integration size with the intrinsics that follow the resample
    k.inW = p->depth_width; k.inH = p->depth_height;
    k.W = p->integration_width; k.H = p->integration_height;
    k.rsx = (float)(k.inW - 1) / (float)(k.W - 1);
    k.rsy = (float)(k.inH - 1) / (float)(k.H - 1);
    k.fx = p->fx * ((float)k.W / (float)k.inW); k.fy = p->fy * ((float)k.H / (float)k.inH);
    k.mx = p->mx * ((float)(k.W - 1) / (float)(k.inW - 1)); k.my = p->my * ((float)(k.H - 1) / (float)(k.inH - 1));
    f->p.depth_width = k.W; f->p.depth_height = k.H; f->p.fx = k.fx; f->p.fy = k.fy; f->p.mx = k.mx; f->p.my = k.my;   // frame_setup sees the integration camera
  }
  f->in_W = p->depth_width; f->in_H = p->depth_height;
  f->in_px = (size_t)p->depth_width * p->depth_height;
  k.depth_shift = p->depth_shift; k.dmin = p->depth_min; k.dmax = p->depth_max; k.voxel = p->voxel_size;
  k.tbase = p->trunc_base; k.tscale = p->trunc_scale; k.maxd = p->max_integration_dist;
  k.wsample = p->weight_sample; k.wmax = f->p.weight_max;
  k.num_buckets = p->hash_num_buckets; k.bucket_size = p->hash_bucket_size;
  k.total_slots = p->hash_num_buckets * p->hash_bucket_size; k.num_blocks = p->num_sdf_blocks;
  k.slab_axis = -1; k.slab_lo = 0; k.slab_hi = 0; k.slab_thick = 0; k.slab_world = 1; k.slab_rank = 0;
  k.cW = p->color_width > 0 && p->color_height > 0 ? p->color_width : 0;
  k.cH = k.cW ? p->color_height : 0;
  k.cfx = p->cfx; k.cfy = p->cfy; k.cmx = p->cmx; k.cmy = p->cmy;
  k.frustum_mode = p->frustum_mode; k.colour_round = p->colour_round; k.colour_first = p->colour_first; k.weight_mode = p->weight_mode;
  if (resample && k.cW == 0) {   // colour frames at the INPUT depth resolution: "their own resolution" as far as the integration camera is concerned
    k.cW = p->depth_width; k.cH = p->depth_height;
    k.cfx = p->fx; k.cfy = p->fy; k.cmx = p->mx; k.cmy = p->my;
  }
  hipDeviceProp_t prop;
  SF_CREATE_CHECK(hipGetDeviceProperties(&prop, device));
  f->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  f->compact_grid = (int)((k.num_blocks + 1023) / 1024);
  if (f->compact_grid > f->num_cus * 8) f->compact_grid = f->num_cus * 8;
  const size_t npx = (size_t)k.W * k.H;
#define SF_ALLOC(ptr, bytes)                                                                      \
  do {                                                                                            \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                           \
    if (e_ != hipSuccess) {                                                                       \
      sf_fuser_destroy(f);                                                                        \
      return sf::fail(SF_ERR_DEVICE, "hipMalloc(%zu bytes) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
    }                                                                                             \
  } while (0)
  SF_CREATE_CHECK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
  {
    // the front stream runs short latency-bound kernels that must slip in between the workgroups of the
    // bandwidth-bound integrate kernel: give it the highest priority the device offers
    int prio_lo = 0, prio_hi = 0;
    SF_CREATE_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    SF_CREATE_CHECK(hipStreamCreateWithPriority(&f->front, hipStreamNonBlocking, prio_hi));
    // (front_lo, the front chain's stream beside the persistent kernel, is made by the first pass that latches that schedule: a stream nobody uses still takes its turn
    // in the mapping of streams to hardware queues, and sf_fuse_run's seven to nine streams are short of those already)
    SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_front_switch, hipEventDisableTiming));
  }
  for (int q = 0; q < 2; q++) {
    SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_compact[q], hipEventDisableTiming));
    SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_fused[q], hipEventDisableTiming));
  }
  SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_input, hipEventDisableTiming));
  SF_ALLOC(f->table, (size_t)k.total_slots * sizeof(HashEntry));
  {
    // one entry per 8 heap blocks, 8 entries per line: 2 MB at 2^20 blocks (4 mm), 64 MB at 2^25 (1 mm) -- what a frame touches of it is ~100 x less than
    // the table entries it would probe
    uint32_t lines = 1024;
    while ((uint64_t)lines * 64ull < (uint64_t)k.num_blocks && lines < (1u << 24)) lines <<= 1;
    f->brick_lines = lines;
    SF_ALLOC(f->bricks, (size_t)lines * 8 * 16);
  }
  SF_ALLOC(f->heap, (size_t)k.num_blocks * 4);
  SF_ALLOC(f->block_keys, (size_t)k.num_blocks * 8);
  SF_ALLOC(f->block_entry, (size_t)k.num_blocks * 4);
  SF_ALLOC(f->block_flags, (size_t)k.num_blocks);
  SF_ALLOC(f->voxels, (size_t)k.num_blocks * 4096);
  for (int q = 0; q < 2; q++) {
    SF_ALLOC(f->depthf2[q], npx * 4 * MAX_BATCH);
    SF_ALLOC(f->color2[q], npx * 8 * MAX_BATCH);   // {depth, rgb} texels of an RGB-D batch
    SF_ALLOC(f->compact2[q], (size_t)k.num_blocks * 4);
    SF_ALLOC(f->cmask2[q], (size_t)k.num_blocks * 4);
  }
  f->compact = f->compact2[0];
  SF_ALLOC(f->counters, C_COUNT * 4);
  SF_ALLOC(f->ray_kx, (size_t)k.W * 4);
  SF_ALLOC(f->ray_ky, (size_t)k.H * 4);
  const size_t rgb_bytes = (k.cW ? (size_t)k.cW * k.cH : npx) * 3;
  for (int q = 0; q < sf_fuser::HOST_RING; q++) {
    SF_ALLOC(f->staging_depth[q], f->in_px * 2);
    SF_ALLOC(f->staging_rgb[q], rgb_bytes);
  }
#undef SF_ALLOC
  for (int q = 0; q < sf_fuser::HOST_RING; q++) {
    SF_CREATE_CHECK(hipHostMalloc(&f->pinned_depth[q], f->in_px * 2, hipHostMallocDefault));
    SF_CREATE_CHECK(hipHostMalloc(&f->pinned_rgb[q], rgb_bytes, hipHostMallocDefault));
    SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_h2d[q], hipEventDisableTiming));
    SF_CREATE_CHECK(hipEventCreateWithFlags(&f->ev_consumed[q], hipEventDisableTiming));
  }
  SF_CREATE_CHECK(hipHostMalloc((void**)&f->host_mirror, 64, hipHostMallocMapped));
  *f->host_mirror = 0;
  SF_CREATE_CHECK(hipMemsetAsync(f->table, 0xFF, (size_t)k.total_slots * sizeof(HashEntry), f->stream));
  SF_CREATE_CHECK(hipMemsetAsync(f->bricks, 0, (size_t)f->brick_lines * 128, f->stream));
  SF_CREATE_CHECK(hipMemsetAsync(f->voxels, 0, (size_t)k.num_blocks * 4096, f->stream));
  SF_CREATE_CHECK(hipMemsetAsync(f->block_flags, 0, (size_t)k.num_blocks, f->stream));
  SF_CREATE_CHECK(hipMemsetAsync(f->counters, 0, C_COUNT * 4, f->stream));
  hipLaunchKernelGGL(k_init_heap, dim3((k.num_blocks + 255) / 256), dim3(256), 0, f->stream, f->heap, f->block_keys, (int)k.num_blocks);
  hipLaunchKernelGGL(k_ray_tables, dim3((std::max(k.W, k.H) + 255) / 256), dim3(256), 0, f->stream, f->ray_kx, f->ray_ky, k);
  const int32_t free0 = (int32_t)k.num_blocks;
  SF_CREATE_CHECK(hipMemcpyAsync(&f->counters[C_HEAP_FREE], &free0, 4, hipMemcpyHostToDevice, f->stream));
  SF_CREATE_CHECK(sf_quiesce(f));
#undef SF_CREATE_CHECK
  *out = f;
  return SF_OK;
}

SF_API void sf_fuser_destroy(sf_fuser* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->front) (void)hipStreamSynchronize(f->front);   // every stream drained before ANYTHING they read or write goes (the host-frame ring's
  if (f->front_lo) (void)hipStreamSynchronize(f->front_lo);
  if (f->stream) (void)hipStreamSynchronize(f->stream); // page-locked slots are read by copies queued on either of them)
  for (auto& e : f->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  (void)hipFree(f->table); (void)hipFree(f->bricks); (void)hipFree(f->heap); (void)hipFree(f->block_keys); (void)hipFree(f->block_entry); (void)hipFree(f->block_flags); (void)hipFree(f->voxels);
  for (int q = 0; q < 2; q++) { (void)hipFree(f->depthf2[q]); (void)hipFree(f->color2[q]); (void)hipFree(f->compact2[q]); (void)hipFree(f->cmask2[q]); }
  (void)hipFree(f->counters); (void)hipFree(f->ray_kx); (void)hipFree(f->ray_ky);
  for (int q = 0; q < 2; q++) { if (f->ev_compact[q]) (void)hipEventDestroy(f->ev_compact[q]); if (f->ev_fused[q]) (void)hipEventDestroy(f->ev_fused[q]); }
  if (f->ev_input) (void)hipEventDestroy(f->ev_input);
  if (f->front) { (void)hipStreamSynchronize(f->front); (void)hipStreamDestroy(f->front); }
  if (f->front_lo) { (void)hipStreamSynchronize(f->front_lo); (void)hipStreamDestroy(f->front_lo); }
  if (f->ev_front_switch) (void)hipEventDestroy(f->ev_front_switch);
  for (int q = 0; q < sf_fuser::HOST_RING; q++) {
    (void)hipFree(f->staging_depth[q]); (void)hipFree(f->staging_rgb[q]);
    if (f->pinned_depth[q]) (void)hipHostFree(f->pinned_depth[q]);
    if (f->pinned_rgb[q]) (void)hipHostFree(f->pinned_rgb[q]);
    if (f->ev_h2d[q]) (void)hipEventDestroy(f->ev_h2d[q]);
    if (f->ev_consumed[q]) (void)hipEventDestroy(f->ev_consumed[q]);
  }
  if (f->host_mirror) (void)hipHostFree(f->host_mirror);
  for (int q = 0; q < 2; q++) { if (f->mc_bounce[q]) (void)hipHostFree(f->mc_bounce[q]); if (f->mc_bounce_ev[q]) (void)hipEventDestroy(f->mc_bounce_ev[q]); }
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
}

// Back to the state sf_fuser_create leaves: empty table, full heap, zeroed tiles (only the slots ever handed out are touched: a 4 mm room is
// ~0.5 GB of the 4.3 GB reserved), counters and frame numbering restarted; parameters, streams and tuning stay.  What the reference's tools do
// between two scans is to exit and start again (Server/scan_processor.py:137-141 runs one process per scan); a long-lived caller -- the scan
// queue of scannet_amd/shard.py, bench.py's repetitions -- keeps the 4+ GB allocation instead.
SF_API int sf_fuser_reset(sf_fuser* f) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  int32_t hw = 0;
  SF_HIP_CHECK(hipMemcpy(&hw, &f->counters[C_HIGH_WATER], 4, hipMemcpyDeviceToHost));
  const ParamsK& k = f->pk;
  if (hw < 0 || (uint32_t)hw > k.num_blocks) hw = (int32_t)k.num_blocks;
  SF_HIP_CHECK(hipMemsetAsync(f->table, 0xFF, (size_t)k.total_slots * sizeof(HashEntry), f->stream));
  SF_HIP_CHECK(hipMemsetAsync(f->bricks, 0, (size_t)f->brick_lines * 128, f->stream));
  if (hw > 0) {
    SF_HIP_CHECK(hipMemsetAsync(f->voxels, 0, (size_t)hw * 4096, f->stream));
    SF_HIP_CHECK(hipMemsetAsync(f->block_flags, 0, (size_t)hw, f->stream));
  }
  SF_HIP_CHECK(hipMemsetAsync(f->counters, 0, C_COUNT * 4, f->stream));
  hipLaunchKernelGGL(k_init_heap, dim3((k.num_blocks + 255) / 256), dim3(256), 0, f->stream, f->heap, f->block_keys, (int)k.num_blocks);
  const int32_t free0 = (int32_t)k.num_blocks;
  SF_HIP_CHECK(hipMemcpyAsync(&f->counters[C_HEAP_FREE], &free0, 4, hipMemcpyHostToDevice, f->stream));
  SF_HIP_CHECK(sf_quiesce(f));
  *f->host_mirror = 0;
  f->frame_seq = 1;
  f->slot = 0;
  f->serial_tail = false;
  f->pipe_beside = f->pipe_overlap == 1;
  f->frames_integrated = f->frames_skipped = 0;
  f->events_used = 0;
  return SF_OK;
}

static int fuse_host(sf_fuser* f, const uint16_t* depth, const uint8_t* rgb, const float* pose, int sign) {
  if (!f || !depth || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (pose[0] == -INFINITY) { f->frames_skipped++; return sf::fail(SF_ERR_SKIPPED, "frame skipped: camToWorld is -inf (tracking lost)"); }
  SF_HIP_CHECK(hipSetDevice(f->device));
  const size_t npx = (size_t)f->pk.W * f->pk.H;
  const size_t rgb_bytes = (f->pk.cW ? (size_t)f->pk.cW * f->pk.cH : npx) * 3;
  // The caller's buffers are ordinary (pageable) memory and are its own again the moment this call returns -- a live stream decodes the next
  // frame into the same buffer, a binding may free it.  So the frame is copied into a page-locked slot of the ring HERE, on the caller's
  // thread (614 KB: ~50 us), and everything behind that -- H2D, pre-pass, allocation, compaction, integrate -- is queued and left running.
  // Round 2 drained both streams and waited for the H2D in every call: the GPU idled while the host copied and the host idled while the GPU
  // fused (one frame per launch: 7.8 k frames/s with resident frames, less through this entry point).
  const int q = f->host_slot;
  f->host_slot = (q + 1) % sf_fuser::HOST_RING;
  if (f->host_frames >= (uint64_t)sf_fuser::HOST_RING) SF_HIP_CHECK(hipEventSynchronize(f->ev_h2d[q]));   // the copy of HOST_RING frames ago has read the slot
  std::memcpy(f->pinned_depth[q], depth, f->in_px * 2);
  if (rgb) std::memcpy(f->pinned_rgb[q], rgb, rgb_bytes);
  hipStream_t in_stream = sf_input_stream(f, 1, rgb != nullptr, sign);  // the stream the pre-pass reads the frame on
  if (f->host_frames >= (uint64_t)sf_fuser::HOST_RING) SF_HIP_CHECK(hipStreamWaitEvent(in_stream, f->ev_consumed[q], 0));   // the kernels that read the device copy
  SF_HIP_CHECK(hipMemcpyAsync(f->staging_depth[q], f->pinned_depth[q], f->in_px * 2, hipMemcpyHostToDevice, in_stream));
  if (rgb) SF_HIP_CHECK(hipMemcpyAsync(f->staging_rgb[q], f->pinned_rgb[q], rgb_bytes, hipMemcpyHostToDevice, in_stream));
  SF_HIP_CHECK(hipEventRecord(f->ev_h2d[q], in_stream));
  const int rc = run_frame(f, f->staging_depth[q], rgb ? f->staging_rgb[q] : nullptr, pose, sign);
  // the frame's pre-pass (the only reader of the device copy) precedes its integrate launch in stream order: an event behind that launch covers it
  SF_HIP_CHECK(hipEventRecord(f->ev_consumed[q], f->stream));
  f->host_frames++;
  return rc;
}

SF_API int sf_fuser_integrate(sf_fuser* f, const uint16_t* depth, const uint8_t* rgb, const float pose[16]) {
  return fuse_host(f, depth, rgb, pose, +1);
}
SF_API int sf_fuser_deintegrate(sf_fuser* f, const uint16_t* depth, const uint8_t* rgb, const float pose[16]) {
  return fuse_host(f, depth, rgb, pose, -1);
}
SF_API int sf_fuser_integrate_device(sf_fuser* f, const void* d_depth, const void* d_rgb, const float pose[16]) {
  if (!f || !d_depth || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  return run_frame(f, d_depth, d_rgb, pose, +1);
}
SF_API int sf_fuser_deintegrate_device(sf_fuser* f, const void* d_depth, const void* d_rgb, const float pose[16]) {
  if (!f || !d_depth || !pose) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  return run_frame(f, d_depth, d_rgb, pose, -1);
}
static int integrate_batch_device(sf_fuser* f, const void* d_depth, uint64_t frame_stride_bytes, const void* d_rgb, uint64_t rgb_stride_bytes, const float* poses,
                                  uint64_t n) {
  if (!f || !d_depth || !poses) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  const void* dd[MAX_BATCH];
  const void* dr[MAX_BATCH];
  const float* pp[MAX_BATCH];
  int m = 0;
  int pass = 0;
  for (uint64_t i = 0; i <= n; i++) {
    if (i < n) {
      if (poses[16 * i] == -INFINITY) { f->frames_skipped++; continue; }  // tracking lost: skip (sensorData.h:382)
      dd[m] = (const uint8_t*)d_depth + i * frame_stride_bytes;
      dr[m] = d_rgb ? (const uint8_t*)d_rgb + i * rgb_stride_bytes : nullptr;
      pp[m] = poses + 16 * i;
      m++;
    }
    // the first pass of a call has nothing to hide its pre-pass / allocation / compaction behind: a short one (f->ramp frames) gets the
    // integrate stream busy sooner, and (ramp_geo) the passes behind it double -- ramp, 2 ramp, 4 ramp ... up to the batch size -- so that the
    // front chain of pass k + 1 still fits under the integrate launch of pass k (same voxels under any batching)
    // A call that would fit ONE pass (ramp < n <= batch) is fused as two halves: the second half's front chain hides behind the first half's integrate launch
    // (20 frames: 8 + 12 gives 31.4 k frames/s, 10 + 10 32.5 k, 6 + 12 + 2 29.6 k, 12 + 8 31.2 k).
    int want = f->batch;
    if (f->ramp > 0 && f->ramp < f->batch && n > (uint64_t)f->ramp) {
      if (n <= (uint64_t)f->batch && f->ramp_geo) want = (int)((n + 1) / 2);
      else if (pass == 0) want = f->ramp;
      else if (f->ramp_geo && pass < 6) want = std::min(f->batch, f->ramp << pass);
    }
    if (m == want || (i == n && m > 0)) {
      f->tail_pass = i == n;   // nothing of this call follows: its integrate launch has the chip to itself
      f->head_pass = pass == 0 && n > (uint64_t)m;   // nothing of this call runs beside its front chain (a one-pass call: neither head nor hidden)
      const int rc = run_batch(f, dd, d_rgb ? dr : nullptr, pp, m, +1);
      f->tail_pass = false;
      f->head_pass = false;
      if (rc != SF_OK) return rc;
      m = 0;
      pass++;
    }
  }
  return SF_OK;
}
SF_API int sf_fuser_integrate_batch_device(sf_fuser* f, const void* d_depth, uint64_t frame_stride_bytes, const float* poses, uint64_t n) {
  return integrate_batch_device(f, d_depth, frame_stride_bytes, nullptr, 0, poses, n);
}
SF_API int sf_fuser_integrate_batch_device_rgb(sf_fuser* f, const void* d_depth, uint64_t frame_stride_bytes, const void* d_rgb, uint64_t rgb_stride_bytes,
                                               const float* poses, uint64_t n) {
  if (!d_rgb) return sf::fail(SF_ERR_INVALID_ARG, "NULL colour frames (sf_fuser_integrate_batch_device is the geometry-only entry point)");
  return integrate_batch_device(f, d_depth, frame_stride_bytes, d_rgb, rgb_stride_bytes, poses, n);
}

SF_API int sf_fuser_batch_frames(const sf_fuser* f) { return f ? f->batch : 0; }

// scanfuse_internal.h: scheduling switches (bench.py, tests); every setting leaves the voxels bit-identical
SF_API int sf_fuser_tune(sf_fuser* f, const char* key, int value) {
  if (!f || !key) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  const std::string k(key);
  auto in = [&](int lo, int hi) { return value >= lo && value <= hi; };
  if (k == "batch" && in(1, MAX_BATCH)) f->batch = value;
  else if (k == "overlap" && in(0, 1)) f->overlap = value != 0;
  else if (k == "xcd_walk" && in(0, 1)) f->xcd_walk = value != 0;
  else if (k == "pipe" && in(0, 1)) f->pipe_mode = value;
  else if (k == "pipe_wgs" && in(1, 3)) f->pipe_wgs = value;
  else if (k == "pipe_overlap" && in(-1, 1)) { f->pipe_overlap = value; f->pipe_beside = value == 1; }
  else if (k == "nt" && in(-1, 1)) f->nt_mode = value;
  else if (k == "front_cus" && in(0, 128)) {
    // the two streams on disjoint sets of CUs (hipExtStreamCreateWithCUMask): `value` CUs, spread evenly over the chip, run the pre-pass /
    // allocation / compaction of the NEXT pass while the rest runs integrate -- the short latency-bound kernels no longer queue behind (or
    // squeeze in between) the workgroups of the bandwidth-bound one.  0: both streams on every CU.
    const int ncu = f->num_cus;
    if (value >= ncu) return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_tune: front_cus = %d on a device with %d CUs (the main stream needs at least one)", value, ncu);
    std::vector<uint32_t> front_mask((size_t)(ncu + 31) / 32, 0u), main_mask((size_t)(ncu + 31) / 32, 0u);
    const int every = value > 0 ? ncu / value : 0;   // >= 1 because value < ncu
    for (int c = 0; c < ncu; c++) {
      const bool to_front = value > 0 && (c % every) == 0 && (c / every) < value;
      (to_front ? front_mask : main_mask)[(size_t)c / 32] |= 1u << (c % 32);
    }
    // the new pair first: a failure leaves the fuser on its old streams instead of on none
    hipStream_t ns = nullptr, nf = nullptr;
    hipError_t e;
    if (value == 0) {
      int prio_lo = 0, prio_hi = 0;
      e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&nf, hipStreamNonBlocking, prio_hi);
    } else {
      e = hipExtStreamCreateWithCUMask(&ns, (uint32_t)main_mask.size(), main_mask.data());
      if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&nf, (uint32_t)front_mask.size(), front_mask.data());
    }
    if (e != hipSuccess) {
      if (ns) (void)hipStreamDestroy(ns);
      if (nf) (void)hipStreamDestroy(nf);
      return sf::fail(SF_ERR_DEVICE, "sf_fuser_tune: front_cus = %d: %s", value, hipGetErrorString(e));
    }
    (void)hipStreamDestroy(f->stream);
    (void)hipStreamDestroy(f->front);
    if (f->front_lo) { (void)hipStreamDestroy(f->front_lo); f->front_lo = nullptr; }
    f->stream = ns;
    f->front = nf;
    f->last_front = nullptr;
    // (a CU-masked front stream is the only front stream: the mask, not a priority, keeps it out of the integrate kernel's way; back at 0 the second
    // one is made again on first need)
    f->front_cus = value;
  }
  else if (k == "alloc_group" && in(1, MAX_BATCH)) f->alloc_group = value;
  else if (k == "alloc_group_head" && in(0, MAX_BATCH)) f->alloc_group_head = value;
  else if (k == "alloc_group_win64" && in(1, MAX_BATCH)) f->alloc_group_win64 = value;
  else if (k == "alloc_wgs" && in(0, 8)) f->alloc_wgs = value;
  else if (k == "prepass_fuse" && in(0, 1)) f->prepass_fuse = value != 0;
#ifdef SF_MEASURE_ABLATE
  else if (k == "alloc_ablate" && in(0, 15)) f->alloc_ablate = value;   // parts of k_alloc_ray switched off: the volume is WRONG with any bit set
#else
  else if (k == "alloc_ablate")
    return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_tune: alloc_ablate switches parts of the allocation off (the volume is wrong under it): only in a library built with -DSF_MEASURE_ABLATE");
#endif
  // which of the two front streams a pass's pre-pass / allocation / compaction goes down (sf_input_stream): -1 the second one (front_lo) beside the persistent kernel of
  // one frame per launch, the high-priority one otherwise (default); 1 always the high-priority one (rounds 2-5); 0 always the second one
  else if (k == "front_prio" && in(-1, 1)) f->front_prio = value;
  else if (k == "front_lo_lowest" && in(0, 1)) f->front_lo_lowest = value != 0;   // before the stream's first need: 1 = the device's lowest priority, 0 = the main stream's
  else if (k == "alloc_ray" && in(0, 1)) f->alloc_ray = value != 0;   // 1: the ray-space window whatever the geometry (rays outside it take the slow path), 0: the cube window
  else if (k == "ramp" && in(0, MAX_BATCH)) f->ramp = value;
  else if (k == "ramp_geo" && in(0, 1)) f->ramp_geo = value != 0;
  else if (k == "tail_wide" && in(0, 1)) f->tail_wide = value;
  else if (k == "xrow" && in(0, 1)) f->xrow = value != 0;
  else if (k == "brick_cache" && in(0, 1)) {   // 0: every look-up of the allocation kernels probes the table
    f->brick_on = value != 0;
    SF_HIP_CHECK(hipMemsetAsync(f->bricks, 0, (size_t)f->brick_lines * 128, f->stream));
    SF_HIP_CHECK(sf_quiesce(f));
  }
  else return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_tune: unknown key or value out of range: %s = %d", key, value);
  return SF_OK;
}

// Internal (pipeline.hip): fuse n <= MAX_BATCH device-resident frames with valid poses in one pass.
int sf_fuser_run_batch(sf_fuser* f, const void* const* d_depth, const void* const* d_rgb, const float* const* poses, int n) {
  if (n < 1 || n > f->batch) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (limit %d)", n, f->batch);
  return run_batch(f, d_depth, d_rgb, poses, n, +1);
}

// ... the colour frames as JPEG component planes (d_planes[j]: what jpeg_gpu_planes left; d_layout[j]: the picture's SfJpegLayout on the device)
int sf_fuser_run_batch_ycc(sf_fuser* f, const void* const* d_depth, const void* const* d_planes, const void* const* d_layout, const float* const* poses, int n) {
  if (n < 1 || n > f->batch) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (limit %d)", n, f->batch);
  return run_batch(f, d_depth, d_planes, poses, n, +1, d_layout);
}

SF_API int sf_fuser_sync(sf_fuser* f) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  return SF_OK;
}
SF_API void* sf_fuser_stream(sf_fuser* f) { return f ? (void*)f->stream : nullptr; }

SF_API int sf_fuser_stats(sf_fuser* f, sf_stats* out) {
  if (!f || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  int32_t c[C_COUNT];
  SF_HIP_CHECK(hipMemcpyAsync(c, f->counters, sizeof(c), hipMemcpyDeviceToHost, f->stream));
  SF_HIP_CHECK(sf_quiesce(f));
  std::memset(out, 0, sizeof(*out));
  out->frames_integrated = f->frames_integrated;
  out->frames_skipped = f->frames_skipped;
  out->heap_free = (uint32_t)c[C_HEAP_FREE];
  out->blocks_allocated = f->pk.num_blocks - (uint32_t)c[C_HEAP_FREE];
  out->last_frame_blocks = (uint32_t)c[C_LAST_BLOCKS];
  out->alloc_failures = (uint32_t)c[C_ALLOC_FAIL];
  uint64_t tot;
  std::memcpy(&tot, &c[C_TOTAL_LO], 8);
  out->total_frame_blocks = tot;
  std::memcpy(&tot, &c[C_TILES_LO], 8);
  out->total_pass_tiles = tot;
  out->hash_slots_used = (uint32_t)c[C_SLOTS_USED];
  out->high_water = (uint32_t)c[C_HIGH_WATER];
  return SF_OK;
}

SF_API int sf_fuser_alloc_direct_count(sf_fuser* f, uint64_t* out) {
  if (!f || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  int32_t c = 0;
  SF_HIP_CHECK(hipMemcpy(&c, &f->counters[C_ALLOC_DIRECT], 4, hipMemcpyDeviceToHost));
  *out = (uint64_t)(uint32_t)c;
  return SF_OK;
}

SF_API int sf_fuser_alloc_probe_count(sf_fuser* f, uint64_t* out) {
  if (!f || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  int32_t c = 0;
  SF_HIP_CHECK(hipMemcpy(&c, &f->counters[C_ALLOC_PROBED], 4, hipMemcpyDeviceToHost));
  *out = (uint64_t)(uint32_t)c;
  return SF_OK;
}

#ifdef SF_ALLOC_TIMING
// measurement build only: the 16 words of g_alloc_t, cleared behind the read
SF_API int sf_alloc_timing_read(sf_fuser* f, uint64_t* out16) {
  if (!f || !out16) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  SF_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_alloc_t), 16 * 8));
  unsigned long long z[16] = {};
  SF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_alloc_t), z, 16 * 8));
  return SF_OK;
}
// ... and the log of its slow workgroups: 1 + 256 * 12 words, cleared behind the read
SF_API int sf_alloc_timing_log(sf_fuser* f, uint32_t* out) {
  if (!f || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  SF_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_alloc_log), (1 + 256 * 12) * 4));
  static unsigned z[1 + 256 * 12];
  SF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_alloc_log), z, sizeof(z)));
  return SF_OK;
}
#endif

SF_API int sf_fuser_profile_enable(sf_fuser* f, int on) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  f->profile = on != 0;
  return SF_OK;
}
SF_API int sf_fuser_profile_read(sf_fuser* f, double* integrate_ms, uint64_t* launches, uint64_t* blocks) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  double ms = 0;
  for (size_t i = 0; i < f->events_used; i++) {
    float t = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&t, f->events[i].first, f->events[i].second));
    ms += t;
  }
  if (integrate_ms) *integrate_ms = ms;
  if (launches) *launches = f->events_used;
  if (blocks) {
    int32_t c[2];
    SF_HIP_CHECK(hipMemcpy(c, &f->counters[C_TOTAL_LO], 8, hipMemcpyDeviceToHost));
    std::memcpy(blocks, c, 8);
  }
  f->events_used = 0;
  return SF_OK;
}

SF_API int sf_fuser_calib_tile_rmw(sf_fuser* f, int read_only, int iters, double* avg_us, uint32_t* tiles) {
  if (!f || iters < 1) return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_calib_tile_rmw: bad argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  const int sl = f->slot ^ 1;  // the list of the most recent pass
  const int cc = sl ? (int)C_COMPACT_B : (int)C_COMPACT;
  int32_t n = 0;
  SF_HIP_CHECK(hipMemcpy(&n, &f->counters[cc], 4, hipMemcpyDeviceToHost));
  int grid = (n + n / 4 + 4096 + 3) / 4;  // the sizing rule of run_batch
  if (grid > f->num_cus * 64) grid = f->num_cus * 64;
  grid = (grid + 7) & ~7;
  uint32_t* sink = nullptr;
  SF_HIP_CHECK(hipMalloc((void**)&sink, 4));
  hipEvent_t e0, e1;
  SF_HIP_CHECK(hipEventCreate(&e0));
  SF_HIP_CHECK(hipEventCreate(&e1));
  double total_ms = 0;
  for (int it = 0; it < iters + 1; it++) {  // first launch untimed
    SF_HIP_CHECK(hipEventRecord(e0, f->stream));
    // the same cache policy the integrate kernel would pick for this tile set (non-temporal beyond 512 MiB)
    if (f->nt_mode == 1 || (f->nt_mode < 0 && (uint64_t)(uint32_t)n * 4096ull > (512ull << 20)))
      hipLaunchKernelGGL(k_tile_rmw<true>, dim3(grid), dim3(256), 0, f->stream, f->voxels, f->compact2[sl], f->counters, cc, f->xcd_walk ? 1 : 0,
                         read_only ? 1 : 0, sink);
    else
      hipLaunchKernelGGL(k_tile_rmw<false>, dim3(grid), dim3(256), 0, f->stream, f->voxels, f->compact2[sl], f->counters, cc, f->xcd_walk ? 1 : 0,
                         read_only ? 1 : 0, sink);
    SF_HIP_CHECK(hipEventRecord(e1, f->stream));
    SF_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0) total_ms += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (avg_us) *avg_us = total_ms * 1e3 / iters;
  if (tiles) *tiles = (uint32_t)n;
  return SF_OK;
}

// scanfuse_internal.h: the pattern ceiling taken apart.  mode bit 0: read only; bit 1: contiguous tiles 0 .. n - 1 instead of the pass's list; bits 2-3:
// tiles per turnaround 1 / 2 / 4 (0, 1, 2).  Every tile is written back as it was read: the volume is unchanged whatever it holds.
SF_API int sf_fuser_calib_tile_rmw_ex(sf_fuser* f, int mode, int iters, double* avg_us, uint32_t* tiles) {
  if (!f || iters < 1 || mode < 0 || (mode >> 2) > 2) return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_calib_tile_rmw_ex: bad argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  const int sl = f->slot ^ 1;  // the list of the most recent pass
  const int cc = sl ? (int)C_COMPACT_B : (int)C_COMPACT;
  int32_t n = 0;
  SF_HIP_CHECK(hipMemcpy(&n, &f->counters[cc], 4, hipMemcpyDeviceToHost));
  if (n < 1 || (uint32_t)n > (uint32_t)f->p.num_sdf_blocks) return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_calib_tile_rmw_ex: no pass to repeat");
  const int read_only = mode & 1, contiguous = (mode >> 1) & 1, G = 1 << (mode >> 2);
  const int units = (n + G - 1) / G;
  int grid = (units + units / 4 + 4096 + 3) / 4;
  if (grid > f->num_cus * 64) grid = f->num_cus * 64;
  grid = (grid + 7) & ~7;
  uint32_t* sink = nullptr;
  SF_HIP_CHECK(hipMalloc((void**)&sink, 4));
  hipEvent_t e0, e1;
  SF_HIP_CHECK(hipEventCreate(&e0));
  SF_HIP_CHECK(hipEventCreate(&e1));
  const bool nt = f->nt_mode == 1 || (f->nt_mode < 0 && (uint64_t)(uint32_t)n * 4096ull > (512ull << 20));
  double total_ms = 0;
  for (int it = 0; it < iters + 1; it++) {  // first launch untimed
    SF_HIP_CHECK(hipEventRecord(e0, f->stream));
#define LAUNCH_RMW(NTV, GV) hipLaunchKernelGGL((k_tile_rmw_ex<NTV, GV>), dim3(grid), dim3(256), 0, f->stream, f->voxels, f->compact2[sl], f->counters, cc, f->xcd_walk ? 1 : 0, read_only, contiguous, sink)
    if (nt) { if (G == 1) LAUNCH_RMW(true, 1); else if (G == 2) LAUNCH_RMW(true, 2); else LAUNCH_RMW(true, 4); }
    else { if (G == 1) LAUNCH_RMW(false, 1); else if (G == 2) LAUNCH_RMW(false, 2); else LAUNCH_RMW(false, 4); }
#undef LAUNCH_RMW
    SF_HIP_CHECK(hipEventRecord(e1, f->stream));
    SF_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0) total_ms += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (avg_us) *avg_us = total_ms * 1e3 / iters;
  if (tiles) *tiles = (uint32_t)n;
  return SF_OK;
}

int sf_compact_live(sf_fuser* f, int32_t* n_out, int include_ghosts) {
  SF_HIP_CHECK(sf_quiesce(f));
  BatchFrames dummy;
  std::memset(&dummy, 0, sizeof(dummy));
  dummy.n = 1;
  SF_HIP_CHECK(hipMemsetAsync(&f->counters[C_EXPORT], 0, 8, f->stream));
  hipLaunchKernelGGL(k_compactify_few, dim3(f->compact_grid * (1024 / COMPACT_THREADS)), dim3(COMPACT_THREADS), 0, f->stream, (CompactArgs{f->block_keys, f->block_entry, f->block_flags, f->table, f->compact,
                     f->cmask2[0], f->counters, (int)C_EXPORT, include_ghosts ? 1 : 2, f->pk, dummy}));
  SF_HIP_CHECK(hipMemcpyAsync(n_out, &f->counters[C_EXPORT], 4, hipMemcpyDeviceToHost, f->stream));
  SF_HIP_CHECK(sf_quiesce(f));
  return SF_OK;
}

SF_API int sf_fuser_garbage_collect(sf_fuser* f, uint32_t* freed) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  SF_HIP_CHECK(hipSetDevice(f->device));
  int32_t n = 0;
  const int rc = sf_compact_live(f, &n, 0);
  if (rc != SF_OK) return rc;
  SF_HIP_CHECK(hipMemsetAsync(&f->counters[C_GC_FREED], 0, 4, f->stream));
  const float thr = std::fmaf(f->p.trunc_scale, f->p.depth_max, f->p.trunc_base);
  if (n > 0)
    hipLaunchKernelGGL(k_gc, dim3(n < f->num_cus * 8 ? n : f->num_cus * 8), dim3(256), 0, f->stream, f->voxels, f->block_keys, f->compact,
                       f->table, f->heap, f->counters, thr, f->pk);
  int32_t fr = 0;
  SF_HIP_CHECK(hipMemcpyAsync(&fr, &f->counters[C_GC_FREED], 4, hipMemcpyDeviceToHost, f->stream));
  SF_HIP_CHECK(sf_quiesce(f));
  int32_t fail0 = 0, fail1 = 0;
  if (fr > 0) {   // leave no tombstone behind: rebuild the table from the directory
    SF_HIP_CHECK(hipMemcpyAsync(&fail0, &f->counters[C_ALLOC_FAIL], 4, hipMemcpyDeviceToHost, f->stream));
    SF_HIP_CHECK(hipMemsetAsync(f->table, 0xFF, (size_t)f->pk.total_slots * sizeof(HashEntry), f->stream));
    SF_HIP_CHECK(hipMemsetAsync(f->bricks, 0, (size_t)f->brick_lines * 128, f->stream));   // blocks left the table: the presence cache starts again
    SF_HIP_CHECK(hipMemsetAsync(&f->counters[C_SLOTS_USED], 0, 4, f->stream));
    hipLaunchKernelGGL(k_rehash, dim3(f->compact_grid), dim3(256), 0, f->stream, f->table, f->block_keys, f->block_entry, f->counters, f->pk);
    SF_HIP_CHECK(hipMemcpyAsync(&fail1, &f->counters[C_ALLOC_FAIL], 4, hipMemcpyDeviceToHost, f->stream));
    SF_HIP_CHECK(sf_quiesce(f));
  }
  if (freed) *freed = (uint32_t)fr;
  if (fail1 != fail0) return sf::fail(SF_ERR_CAPACITY, "garbage collection: %d surviving blocks found no hash slot within %d probes when the table was rebuilt", fail1 - fail0, MAX_PROBES);
  return SF_OK;
}

SF_API int sf_fuser_export_blocks(sf_fuser* f, int32_t* coords, void* voxels, uint64_t capacity, uint64_t* n_out) {
  if (!f || !n_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(f->device));
  int32_t n = 0;
  const int rc = sf_compact_live(f, &n);
  if (rc != SF_OK) return rc;
  *n_out = (uint64_t)n;
  if (!coords && !voxels) return SF_OK;
  if (!coords || !voxels) return sf::fail(SF_ERR_INVALID_ARG, "coords and voxels must both be given");
  if (capacity < (uint64_t)n) return sf::fail(SF_ERR_BOUNDS, "capacity %llu < %d live blocks", (unsigned long long)capacity, n);
  if (n == 0) return SF_OK;
  int32_t* d_coords = nullptr;
  uint4* d_vox = nullptr;
  SF_HIP_CHECK(hipMalloc((void**)&d_coords, (size_t)n * 12));
  if (hipMalloc((void**)&d_vox, (size_t)n * 4096) != hipSuccess) { (void)hipFree(d_coords); return sf::fail(SF_ERR_DEVICE, "hipMalloc export buffer failed"); }
  hipLaunchKernelGGL(k_gather, dim3(n < 65535 ? n : 65535), dim3(256), 0, f->stream, f->voxels, f->block_keys, f->compact, n, d_coords, d_vox);
  hipError_t e1 = hipMemcpyAsync(coords, d_coords, (size_t)n * 12, hipMemcpyDeviceToHost, f->stream);
  hipError_t e2 = hipMemcpyAsync(voxels, d_vox, (size_t)n * 4096, hipMemcpyDeviceToHost, f->stream);
  hipError_t e3 = sf_quiesce(f);
  (void)hipFree(d_coords);
  (void)hipFree(d_vox);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return sf::fail(SF_ERR_DEVICE, "export copy failed");
  return SF_OK;
}

SF_API int sf_device_malloc(int device, uint64_t bytes, void** out) {
  if (!out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SF_HIP_CHECK(hipSetDevice(device));
  SF_HIP_CHECK(hipMalloc(out, bytes));
  return SF_OK;
}
SF_API int sf_device_free(void* p) { SF_HIP_CHECK(hipFree(p)); return SF_OK; }
SF_API int sf_device_upload(void* dst, const void* src, uint64_t bytes) { SF_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return SF_OK; }
SF_API int sf_device_download(void* dst, const void* src, uint64_t bytes) { SF_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return SF_OK; }

// ------------------------------------------------------------------------------------------------------
// One large scan over several GPUs (BASELINE configs[4]): slab ownership, boundary layer export / import
// ------------------------------------------------------------------------------------------------------
SF_API int sf_fuser_set_slab(sf_fuser* f, int axis, int32_t lo_block, int32_t hi_block) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  if (axis > 2) return sf::fail(SF_ERR_INVALID_ARG, "axis must be 0, 1, 2 or negative (no partition)");
  if (axis >= 0 && !(lo_block < hi_block)) return sf::fail(SF_ERR_INVALID_ARG, "empty slab [%d, %d)", lo_block, hi_block);
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  f->pk.slab_axis = axis < 0 ? -1 : axis;
  f->pk.slab_lo = lo_block;
  f->pk.slab_hi = hi_block;
  f->pk.slab_thick = 0; f->pk.slab_world = 1; f->pk.slab_rank = 0;
  return SF_OK;
}

SF_API int sf_fuser_set_stripes(sf_fuser* f, int axis, int32_t origin_block, int32_t thickness_blocks, int world, int rank) {
  if (!f) return sf::fail(SF_ERR_INVALID_ARG, "NULL fuser");
  if (axis < 0 || axis > 2 || thickness_blocks < 1 || world < 1 || rank < 0 || rank >= world)
    return sf::fail(SF_ERR_INVALID_ARG, "sf_fuser_set_stripes: axis %d, thickness %d, rank %d of %d", axis, thickness_blocks, rank, world);
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  f->pk.slab_axis = axis;
  f->pk.slab_lo = origin_block;
  f->pk.slab_hi = 0;
  f->pk.slab_thick = thickness_blocks; f->pk.slab_world = world; f->pk.slab_rank = rank;
  return SF_OK;
}

SF_API int sf_fuser_export_boundary(sf_fuser* f, int32_t* coords, void* voxels, uint64_t capacity, uint64_t* n_out, int dst_on_device) {
  return sf_fuser_export_blocks_where(f, -2, 0, 0, 0, coords, voxels, capacity, n_out, dst_on_device);
}

SF_API int sf_fuser_export_blocks_where(sf_fuser* f, int axis, int32_t lo, int32_t hi, int include_ghosts, int32_t* coords, void* voxels,
                                        uint64_t capacity, uint64_t* n_out, int dst_on_device) {
  if (!f || !n_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if ((coords == nullptr) != (voxels == nullptr)) return sf::fail(SF_ERR_INVALID_ARG, "coords and voxels must both be given (or both NULL to count)");
  SF_HIP_CHECK(hipSetDevice(f->device));
  int32_t n_live = 0;
  const int rc = sf_compact_live(f, &n_live, include_ghosts);
  if (rc != SF_OK) return rc;
  *n_out = 0;
  if (n_live == 0) return SF_OK;
  SF_HIP_CHECK(hipMemsetAsync(&f->counters[C_GC_FREED], 0, 4, f->stream));  // scratch counter (GC is synchronous, never concurrent)
  int32_t* d_coords = nullptr;
  uint4* d_vox = nullptr;
  const bool want = coords != nullptr;
  const int cap = (int)std::min<uint64_t>(capacity, 0x7FFFFFFFull);
  if (want && !dst_on_device && cap > 0) {
    SF_HIP_CHECK(hipMalloc((void**)&d_coords, (size_t)cap * 12));
    if (hipMalloc((void**)&d_vox, (size_t)cap * 4096) != hipSuccess) { (void)hipFree(d_coords); return sf::fail(SF_ERR_DEVICE, "hipMalloc export buffer failed"); }
  } else if (want) {
    d_coords = coords;
    d_vox = (uint4*)voxels;
  }
  hipLaunchKernelGGL(k_gather_where, dim3(n_live < 65535 ? n_live : 65535), dim3(256), 0, f->stream, f->voxels, f->block_keys, f->compact, n_live, axis, lo, hi,
                     want ? cap : 0, &f->counters[C_GC_FREED], d_coords, want && cap > 0 ? d_vox : nullptr, f->pk);
  int32_t n = 0;
  hipError_t e = hipMemcpyAsync(&n, &f->counters[C_GC_FREED], 4, hipMemcpyDeviceToHost, f->stream);
  if (e == hipSuccess) e = sf_quiesce(f);
  if (e == hipSuccess && want && !dst_on_device && cap > 0) {
    const size_t m = (size_t)std::min(n, cap);
    e = hipMemcpy(coords, d_coords, m * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(voxels, d_vox, m * 4096, hipMemcpyDeviceToHost);
  }
  if (want && !dst_on_device && cap > 0) { (void)hipFree(d_coords); (void)hipFree(d_vox); }
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "export failed: %s", hipGetErrorString(e));
  *n_out = (uint64_t)n;
  if (want && (uint64_t)n > capacity) return sf::fail(SF_ERR_BOUNDS, "capacity %llu < %d matching blocks", (unsigned long long)capacity, n);
  return SF_OK;
}

static int import_blocks(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int ghost, int src_on_device, int only_wanted, uint64_t* imported);

SF_API int sf_fuser_import_blocks(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int ghost, int src_on_device) {
  return import_blocks(f, coords, voxels, n, ghost, src_on_device, 0, nullptr);
}
SF_API int sf_fuser_import_ghosts(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int src_on_device, uint64_t* imported) {
  return import_blocks(f, coords, voxels, n, 1, src_on_device, 1, imported);
}

static int import_blocks(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int ghost, int src_on_device, int only_wanted, uint64_t* imported) {
  if (imported) *imported = 0;
  if (!f || (n && (!coords || !voxels))) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (n == 0) return SF_OK;
  if (n > 0x7FFFFFFFull) return sf::fail(SF_ERR_INVALID_ARG, "too many blocks");
  SF_HIP_CHECK(hipSetDevice(f->device));
  SF_HIP_CHECK(sf_quiesce(f));
  const int32_t* d_coords = coords;
  const uint4* d_vox = (const uint4*)voxels;
  int32_t* tmp_c = nullptr;
  uint4* tmp_v = nullptr;
  if (!src_on_device) {
    SF_HIP_CHECK(hipMalloc((void**)&tmp_c, n * 12));
    if (hipMalloc((void**)&tmp_v, n * 4096) != hipSuccess) { (void)hipFree(tmp_c); return sf::fail(SF_ERR_DEVICE, "hipMalloc import buffer failed"); }
    hipError_t e = hipMemcpy(tmp_c, coords, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tmp_v, voxels, n * 4096, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(tmp_c); (void)hipFree(tmp_v); return sf::fail(SF_ERR_DEVICE, "import copy failed: %s", hipGetErrorString(e)); }
    d_coords = tmp_c;
    d_vox = tmp_v;
  }
  int32_t fail0 = 0, fail1 = 0, took = 0;
  (void)hipMemcpy(&fail0, &f->counters[C_ALLOC_FAIL], 4, hipMemcpyDeviceToHost);
  (void)hipMemsetAsync(&f->counters[C_IMPORTED], 0, 4, f->stream);
  hipLaunchKernelGGL(k_import, dim3(n < 65535 ? (unsigned)n : 65535u), dim3(256), 0, f->stream, d_coords, d_vox, (int)n, ghost, only_wanted, f->voxels, f->table,
                     f->heap, f->block_keys, f->block_entry, f->block_flags, f->counters, f->pk);
  hipError_t e = hipMemcpyAsync(&fail1, &f->counters[C_ALLOC_FAIL], 4, hipMemcpyDeviceToHost, f->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&took, &f->counters[C_IMPORTED], 4, hipMemcpyDeviceToHost, f->stream);
  if (e == hipSuccess) e = sf_quiesce(f);
  if (imported) *imported = (uint64_t)took;
  if (tmp_c) { (void)hipFree(tmp_c); (void)hipFree(tmp_v); }
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "import failed: %s", hipGetErrorString(e));
  if (fail1 != fail0) return sf::fail(SF_ERR_CAPACITY, "%d imported blocks did not fit (heap or hash table exhausted)", fail1 - fail0);
  return SF_OK;
}

// jpeg_huff_gpu.hip -- baseline-JPEG entropy decoding on gfx950: the lane programs of jpeg_huff.h, one 1024-lane workgroup per picture.
//
// Until round 4 the frame pipeline's host threads Huffman-decoded every colour frame (0.5 ms per 1296x968 picture and thread: with the depth
// inflate beside it the 16 decode threads of the GPU box bound a real ScanNet scan -- JPEG colour at 1296x968 over zlib depth -- at 8.6 k frames/s).
// Now a host thread only parses the headers and copies the entropy-coded segment with its byte stuffing removed (jpeg.cpp: jpeg_prepare_huff,
// 0.06 ms); the segment travels over PCIe instead of the coefficients (a fifth of their bytes) and is decoded here into the payload k_jpeg_idct
// reads (jpeg_gpu.hip).  Replaces stbi__jpeg_decode_block / stbi__jpeg_huff_decode of the reference's decoder (stb_image.h:1521-1760) as
// RGBDFrame::decompressColorAlloc_stb reaches it (SensReader/c++/src/sensorData.h:600-616); what the device does not take (restart intervals,
// sampling factors above 2) stays with the host decoder.  Results are the host decoder's coefficients, hence its bytes (tests/test_gpu_pipeline.py).
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"
#include "jpeg_huff.h"

int jpeg_prepare_huff(const uint8_t* data, uint64_t n, uint32_t expect_w, uint32_t expect_h, uint8_t* payload, uint64_t payload_capacity);   // jpeg.cpp
int jpeg_gpu_reconstruct(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_rgb, uint8_t* const* d_planes, uint32_t max_blocks,
                         uint32_t max_width, uint32_t max_height);                                                                              // jpeg_gpu.hip

namespace {

constexpr int JH_BATCH = 32;   // pictures per launch: a whole batch of the frame pipeline, one CU each
constexpr int JH_LANES = 1024;

struct JpegHuffBatch {
  const uint8_t* prepared[JH_BATCH];   // SfJpegLayout + SfJpegHuffDesc + the unstuffed entropy-coded segment (device)
  uint8_t* payload[JH_BATCH];          // out: SfJpegLayout + block table + entries (device); nullptr: slot unused
  uint32_t max_entries[JH_BATCH];      // capacity of the entry array
  int32_t tag[JH_BATCH];               // what the caller wants to read back with a failure (a frame number)
  int32_t* status;                     // nullable; written ONLY on failure: status[2 * slot] = code < 0 (corrupt / truncated / does not fit: the
                                       // picture is left black), status[2 * slot + 1] = tag
};

// exclusive prefix sum of one value per lane over the 1024 lanes of the workgroup (wave shuffles + one LDS exchange); also the total
__device__ inline int block_exscan(int v, int* s_wave, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  __syncthreads();   // s_wave may still be read by the previous scan
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < JH_LANES / 64; w++) {
    const int x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  total = tot;
  return base + incl - v;
}

struct JHChunkCounts {
  uint32_t blocks, entries;
  int32_t dc0, dc1, dc2;
};

struct DeviceWriter {
  const SfJpegLayout& L;
  const SfJpegHuffGeom& G;
  uint32_t* table;
  uint32_t* entries;
  uint32_t* final_entries;   // the lane that completes the picture's last block leaves the number of entries here
  uint32_t ordinal, e;
  int pred0, pred1, pred2;   // the DC predictors, one register each (as an array indexed by the component they were 12 of the kernel's 72 bytes of private memory)
  const uint8_t* zz;         // the zig-zag table in LDS (jh_zigzag's own table is a constant in global memory: a load per coefficient)
  __device__ void entry(bool isdc, int ci, int k, int v) {   // a DC entry carries the value itself: predictor + difference, at position 0
    pred0 += (isdc && ci == 0) ? v : 0;
    pred1 += (isdc && ci == 1) ? v : 0;
    pred2 += (isdc && ci == 2) ? v : 0;
    const int p = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
    entries[e++] = isdc ? (uint32_t)(uint16_t)(int16_t)p : ((uint32_t)zz[k & 63] << 16) | (uint32_t)(uint16_t)(int16_t)v;
  }
  __device__ bool block_done(int, int bi, uint32_t cnt) {
    if (ordinal >= G.total_blocks) return false;
    table[jh_block_index(L, G, ordinal, bi)] = ((e - cnt) << 7) | cnt;
    ordinal++;
    if (ordinal == G.total_blocks) *final_entries = e;
    return ordinal < G.total_blocks;
  }
};

__global__ __launch_bounds__(JH_LANES) void k_jpeg_huff(JpegHuffBatch B) {
  __shared__ SfJpegHuffTable s_dc[3], s_ac[3];
  __shared__ SfJpegHuffGeom s_geom;
  __shared__ SfJpegLayout s_layout;
  __shared__ JHState s_end[JH_LANES], s_start[JH_LANES];   // per CHUNK: where its decode starts (the left neighbour's end of the round before) and ends
  __shared__ JHChunkCounts s_cnt[JH_LANES];                // ... and what it holds
  __shared__ uint16_t s_list[JH_LANES];                    // the dirty chunks of a round
  __shared__ int s_wave[JH_LANES / 64];
  __shared__ uint32_t s_nentries;
  __shared__ uint8_t s_zz[64];
  __shared__ uint16_t s_dc12[3 * JH_LOOK12], s_ac12[3 * JH_LOOK12];   // 12-bit first-level tables, built below (48 KiB: the workgroup has the CU to itself anyway)
  const int f = blockIdx.x;
  if (B.payload[f] == nullptr) return;
  if (threadIdx.x < 64) s_zz[threadIdx.x] = (uint8_t)jh_zigzag((int)threadIdx.x);
  const uint32_t* __restrict__ prep = reinterpret_cast<const uint32_t*>(B.prepared[f]);
  if (threadIdx.x == 0) s_nentries = 0u;
  {
    // layout, geometry and the six tables into LDS (all multiples of 4 bytes, contiguous in the prepared payload)
    uint32_t* dl = reinterpret_cast<uint32_t*>(&s_layout);
    for (uint32_t i = threadIdx.x; i < sizeof(SfJpegLayout) / 4; i += JH_LANES) dl[i] = prep[i];
    const uint32_t* src = prep + sizeof(SfJpegLayout) / 4;
    uint32_t* dg = reinterpret_cast<uint32_t*>(&s_geom);
    for (uint32_t i = threadIdx.x; i < sizeof(SfJpegHuffGeom) / 4; i += JH_LANES) dg[i] = src[i];
    src += sizeof(SfJpegHuffGeom) / 4;
    uint32_t* dd = reinterpret_cast<uint32_t*>(&s_dc[0]);
    for (uint32_t i = threadIdx.x; i < 3 * sizeof(SfJpegHuffTable) / 4; i += JH_LANES) dd[i] = src[i];
    src += 3 * sizeof(SfJpegHuffTable) / 4;
    uint32_t* da = reinterpret_cast<uint32_t*>(&s_ac[0]);
    for (uint32_t i = threadIdx.x; i < 3 * sizeof(SfJpegHuffTable) / 4; i += JH_LANES) da[i] = src[i];
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 6u * JH_LOOK12; i += JH_LANES) {
    const uint32_t tb = i / JH_LOOK12, idx = i % JH_LOOK12;
    if (tb < 3u) s_dc12[i] = jh_look12_entry(s_dc[tb], idx);
    else s_ac12[i - 3u * JH_LOOK12] = jh_look12_entry(s_ac[tb - 3u], idx);
  }
  __syncthreads();
  const uint32_t* __restrict__ words = prep + (sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc)) / 4;
  const SfJpegHuffGeom& G = s_geom;
  const uint32_t nbits = G.ecs_bytes * 8u;
  uint32_t C, Bc;
  jh_geometry(nbits, C, Bc);
  const uint32_t c = threadIdx.x;
  const bool mine = c < C;
  const uint32_t limit = (c + 1 == C) ? nbits : (c + 1) * Bc;
  // ---- stage A: chunk states to their fixed point.  Round 0 decodes every chunk from a guess, round 1 nearly every chunk again (its left neighbour's end state
  // is news); from then on only the chunks whose neighbour did not fall in step inside its own chunk are dirty -- one in ten, then fewer.  A wave takes as long
  // over one dirty lane as over sixty-four, so the dirty chunks are DEALT OUT AGAIN every round: lane i takes the i-th dirty chunk (states and counts of all
  // chunks live in LDS), the late rounds run on one or two waves instead of on all sixteen (rounds 3-6 cost a full chunk decode each until round 6).
  if (mine) s_start[c] = JHState{c * Bc, 0, 0, 0u};
  JHNoEmit none;
  {
    JHCounts cnt;
    cnt.blocks = cnt.entries = 0; cnt.dc_sum[0] = cnt.dc_sum[1] = cnt.dc_sum[2] = 0; cnt.bad = 0;
    if (mine) s_end[c] = jh_run(G, s_dc, s_ac, s_dc12, s_ac12, words, s_start[c], limit, cnt, none);
    s_cnt[c] = JHChunkCounts{cnt.blocks, cnt.entries, cnt.dc_sum[0], cnt.dc_sum[1], cnt.dc_sum[2]};
  }
  for (uint32_t round = 0; round < C + 2u; round++) {
    __syncthreads();   // every end state of the round before is written
    bool changed = false;
    if (mine && c > 0) {
      const JHState ns = s_end[c - 1];
      changed = !jh_same(ns, s_start[c]);
      s_start[c] = ns;
    }
    int ndirty;
    const int at = block_exscan(changed ? 1 : 0, s_wave, ndirty);
    if (ndirty == 0) break;
    if (changed) s_list[at] = (uint16_t)c;
    __syncthreads();
    for (int i = (int)threadIdx.x; i < ndirty; i += JH_LANES) {
      const uint32_t d = s_list[i];
      JHCounts cnt;
      s_end[d] = jh_run(G, s_dc, s_ac, s_dc12, s_ac12, words, s_start[d], (d + 1 == C) ? nbits : (d + 1) * Bc, cnt, none);
      s_cnt[d] = JHChunkCounts{cnt.blocks, cnt.entries, cnt.dc_sum[0], cnt.dc_sum[1], cnt.dc_sum[2]};
    }
  }
  __syncthreads();
  const JHState start = mine ? s_start[c] : JHState{0u, 0, 0, 0u};
  const JHChunkCounts cnt = s_cnt[c];
  // ---- stage B: where every chunk's blocks, entries and DC predictors start
  int total_blocks_seen, total_entries, t0, t1, t2;
  const uint32_t ord0 = (uint32_t)block_exscan(mine ? (int)cnt.blocks : 0, s_wave, total_blocks_seen);
  const uint32_t ent0 = (uint32_t)block_exscan(mine ? (int)cnt.entries : 0, s_wave, total_entries);
  const int p0 = block_exscan(mine ? cnt.dc0 : 0, s_wave, t0);
  const int p1 = block_exscan(mine ? cnt.dc1 : 0, s_wave, t1);
  const int p2 = block_exscan(mine ? cnt.dc2 : 0, s_wave, t2);
  uint32_t* out = reinterpret_cast<uint32_t*>(B.payload[f]);
  uint32_t* table = out + sizeof(SfJpegLayout) / 4;
  uint32_t* entries = table + s_layout.nblocks;
  int32_t status = 0;
  if ((uint32_t)total_blocks_seen < G.total_blocks) status = -3;              // the segment ends before the picture does
  else if ((uint32_t)total_entries > B.max_entries[f]) status = -4;           // more coefficients than the payload holds
  if (status == 0 && mine && ord0 < G.total_blocks) {
    // ---- stage C: the same decode once more, written
    DeviceWriter w{s_layout, G, table, entries, &s_nentries, ord0, ent0, p0, p1, p2, s_zz};
    JHCounts again;
    (void)jh_run(G, s_dc, s_ac, s_dc12, s_ac12, words, start, limit, again, w);
    if (again.bad) status = -2;   // an invalid code on the true path
  }
  const int any_bad = __syncthreads_or(status != 0);
  if (any_bad)   // no block has entries: a grey picture instead of table words that point anywhere
    for (uint32_t i = threadIdx.x; i < s_layout.nblocks; i += JH_LANES) table[i] = 0u;
  // the layout header of the payload, with the number of entries
  for (uint32_t i = threadIdx.x; i < sizeof(SfJpegLayout) / 4; i += JH_LANES) out[i] = reinterpret_cast<const uint32_t*>(&s_layout)[i];
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<SfJpegLayout*>(out)->nentries = any_bad ? 0u : s_nentries;
  if (B.status && status != 0 && (status == -2 || threadIdx.x == 0)) {   // -3 / -4 are the same in every lane; -2 only in the lane that met the code
    B.status[2 * f] = status;
    B.status[2 * f + 1] = B.tag[f];
  }
}

}  // namespace

// The code object of this file is loaded by the runtime when one of its kernels is first used (milliseconds, inside a scan's first sf_fuse_run unless somebody asks
// earlier): the preparation thread of the frame pipeline asks (pipeline.hip, sf_run_resources_prepare_ex).
void jpeg_huff_gpu_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_jpeg_huff));
  (void)hipGetLastError();
}

// Entropy-decode up to 32 prepared pictures on `stream`: d_prepared[i] (jpeg_prepare_huff's payload, uploaded) -> d_payload[i] (what
// jpeg_gpu_reconstruct reads); max_entries[i] = room for entries in d_payload[i]; d_status (nullable): 2 * n ints the caller zeroed, written only
// for pictures that fail: {code < 0, tags[i]}.
int jpeg_gpu_huffman(hipStream_t stream, int n, const uint8_t* const* d_prepared, uint8_t* const* d_payload, const uint32_t* max_entries, const int32_t* tags,
                     int32_t* d_status) {
  if (n < 1 || n > JH_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "jpeg_gpu_huffman: %d frames", n);
  JpegHuffBatch b;
  for (int i = 0; i < JH_BATCH; i++) {
    b.prepared[i] = i < n ? d_prepared[i] : nullptr;
    b.payload[i] = i < n ? d_payload[i] : nullptr;
    b.max_entries[i] = i < n ? max_entries[i] : 0u;
    b.tag[i] = (i < n && tags) ? tags[i] : i;
  }
  b.status = d_status;
  hipLaunchKernelGGL(k_jpeg_huff, dim3(n), dim3(JH_LANES), 0, stream, b);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
}

// scanfuse_internal.h: one picture through the whole device path -- prepared on the host, entropy-decoded AND reconstructed on `device` -- for the
// parity tests (the bytes must be sf_jpeg_decode's).  SF_ERR_UNSUPPORTED for what the device decoder leaves to the host; SF_ERR_FORMAT when the
// device reports a corrupt stream.
SF_API int sf_jpeg_decode_gpu_huffman(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, int device, uint8_t* dst_rgb) {
  if (!data || !dst_rgb || width == 0 || height == 0) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sf::fail(SF_ERR_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  const uint64_t padded = (uint64_t)((width + 15) & ~15u) * ((height + 15) & ~15u);
  std::vector<uint32_t> host((sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) + bytes + 64) / 4 + 16);
  const int rc = jpeg_prepare_huff(data, bytes, width, height, reinterpret_cast<uint8_t*>(host.data()), host.size() * 4);
  if (rc != SF_OK) return rc;
  const SfJpegLayout* L = reinterpret_cast<const SfJpegLayout*>(host.data());
  const SfJpegHuffDesc* D = reinterpret_cast<const SfJpegHuffDesc*>(reinterpret_cast<const uint8_t*>(host.data()) + sizeof(SfJpegLayout));
  const size_t prep_b = sizeof(SfJpegLayout) + sizeof(SfJpegHuffDesc) + 4 * (size_t)D->ecs_words;
  const uint32_t max_entries = (uint32_t)(padded * 3 + 64);   // every coefficient of a 4:4:4 picture non-zero
  const size_t pay_b = sizeof(SfJpegLayout) + 4 * ((size_t)L->nblocks + max_entries), rgb_b = (size_t)width * height * 3;
  uint8_t *d_prep = nullptr, *d_pay = nullptr, *d_rgb = nullptr, *d_planes = nullptr;
  int32_t* d_status = nullptr;
  auto release = [&]() { (void)hipFree(d_prep); (void)hipFree(d_pay); (void)hipFree(d_rgb); (void)hipFree(d_planes); (void)hipFree(d_status); };
  hipError_t e = hipMalloc((void**)&d_prep, prep_b);
  if (e == hipSuccess) e = hipMalloc((void**)&d_pay, pay_b);
  if (e == hipSuccess) e = hipMalloc((void**)&d_rgb, rgb_b);
  if (e == hipSuccess) e = hipMalloc((void**)&d_planes, sf_jpeg_plane_bytes(*L));
  if (e == hipSuccess) e = hipMalloc((void**)&d_status, 8);
  if (e == hipSuccess) e = hipMemset(d_status, 0, 8);
  if (e == hipSuccess) e = hipMemcpy(d_prep, host.data(), prep_b, hipMemcpyHostToDevice);
  int out = SF_OK;
  int32_t status = 0;
  if (e == hipSuccess) {
    const uint8_t* pp = d_prep;
    out = jpeg_gpu_huffman(nullptr, 1, &pp, &d_pay, &max_entries, nullptr, d_status);
    if (out == SF_OK) {
      const uint8_t* pay = d_pay;
      out = jpeg_gpu_reconstruct(nullptr, 1, &pay, &d_rgb, &d_planes, L->nblocks, width, height);
    }
    if (out == SF_OK) e = hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost);
    if (out == SF_OK && e == hipSuccess) e = hipMemcpy(dst_rgb, d_rgb, rgb_b, hipMemcpyDeviceToHost);
  }
  release();
  if (e != hipSuccess) return sf::fail(SF_ERR_DEVICE, "sf_jpeg_decode_gpu_huffman: %s", hipGetErrorString(e));
  if (out == SF_OK && status != 0) return sf::fail(SF_ERR_FORMAT, "jpeg: the device's entropy decoder reports a corrupt or truncated stream (status %d)", status);
  return out;
}

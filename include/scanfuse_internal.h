/* scanfuse_internal.h -- NOT part of the drop-in boundary.
 *
 * Entry points libscanfuse.so exports for this repository's own bench.py, tools/ and tests/: measurement aids, scheduling
 * switches, a device self-test and the synthetic stream source.  Nothing a pipeline stage needs is declared here; the
 * product ABI is include/scanfuse.h.  (Round 1 kept these in the public header and read the switches from SF_* environment
 * variables inside sf_fuser_create; they now live behind this header and sf_fuser_tune.)
 */
#ifndef SCANFUSE_INTERNAL_H
#define SCANFUSE_INTERNAL_H

#include "scanfuse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scheduling switches of a fuser (results are bit-identical under all of them; tests/test_gpu_tsdf.py runs the matrix):
 *   "batch"       1..32  frames fused per pass over the voxel tiles (default 32: the frame mask of a block is one word; 1 = what sf_fuser_integrate gives a live stream)
 *   "overlap"     0/1    pre-pass / allocation / compaction of the next batch on a second stream (default 1)
 *   "xcd_walk"    0/1    each XCD walks one contiguous eighth of the block list (default 1)
 *   "pipe"        0/1    colourless one-frame passes run the software-pipelined persistent kernel (default 1)
 *   "pipe_wgs"    1..3   persistent workgroups per CU of that kernel (default 3)
 *   "pipe_overlap" -1/0/1 the next frame's pre-pass / allocation / compaction runs on the second stream beside that kernel: -1 = when the previous
 *                        pass touched more than 512 MiB of tiles (default), 0 never, 1 always
 *   "nt"          -1/0/1 that kernel's tile loads and stores non-temporal: -1 = when the previous pass touched more than 512 MiB of tiles (default)
 *   "front_cus"   0..128 the second stream owns that many CUs (spread over the chip), the main stream the rest (hipExtStreamCreateWithCUMask); 0 = shared
 *   "alloc_group" 1..32  consecutive frames one allocation workgroup walks (default 16: half of a 32-frame pass)
 *   "alloc_group_head" 0..32 the same for the FIRST pass of a multi-pass batch call, whose front chain nothing hides (default 4; 0 = like every pass)
 *   "alloc_ray"   0/1    the allocation kernel's occupancy bitmap in ray space (k_alloc_ray; default: whenever the voxel size lets the window hold a
 *                        pixel tile's rays: >= 2.5 mm voxels with the shipped camera) or as a 32^3-block cube anchored at the first ray (k_alloc)
 *   "prepass_fuse" 0/1   one colourless frame per pass: the allocation kernel converts the depth itself (default 1), no separate pre-pass launch
 *   "ramp"        0..32  frames of the FIRST pass of a sf_fuser_integrate_batch_device call (default 8; 0 = a full pass): nothing overlaps that
 *                        pass's pre-pass / allocation, so a short one starts the pipeline sooner
 *   "ramp_geo"    0/1    the passes behind the first one double (ramp, 2 ramp, 4 ramp, ... batch) instead of jumping to the batch size, and a call of more than
 *                        `ramp` but no more than `batch` frames is fused as two halves (default 1)
 *   "tail_wide"   0/1    the LAST pass of a sf_fuser_integrate_batch_device call (no front chain runs beside it) takes the variant of k_integrate that fuses
 *                        the tile in halves at 8 waves per SIMD (default 1: +0.8 % on a 20-frame call; slower for a pass as a whole when allocation runs beside it)
 *   "xrow"        0/1    passes of several frames run k_integrate in the x-row lane layout: a lane holds one x-row of the block (y = lane & 7, z = lane >> 3)
 *                        instead of two x-neighbours in four z-layers -- the same voxels, 18 fma fewer per lane and frame, and the gathers of one instruction
 *                        fall on two image rows instead of four or five (default 1)
 *   "front_prio"  -1/0/1 which front stream a pass's pre-pass / allocation / compaction goes down: -1 (default) a second one at the device's LOWEST priority beside the
 *                        persistent kernel of one frame per launch out of cache reach (at the highest priority the allocation's 72 KiB workgroups take the LDS the integrate
 *                        kernel's third workgroup per CU needs: 0.54 -> 0.62 of peak HBM shipped at 1 mm), the high-priority one for passes of several frames; 1 always high
 *                        (rounds 2-5), 0 always the second one
 *   "front_lo_lowest" 1/0 that second stream at the device's lowest priority (default) or at the main stream's (0.60-0.61 instead of 0.62).  For a process that ALSO runs
 *                        sf_fuse_run later: the first stream of a priority class opens that class's hardware queues for the life of the process, and sf_fuse_run's seven to
 *                        nine busy streams then run 12 % slower; set 0 there.  Before the stream's first use.
 *   "brick_cache" 0/1    the cube-window allocation kernels (voxels under 2.5 mm, or "alloc_ray" 0) ask the presence cache (one {tag, 64-bit mask} entry per 4x4x4-block brick: "this block is in the table and
 *                        older than this batch") before they probe the hash table (default 1; 0: every look-up probes the table, rounds 1-5).  The allocated
 *                        set and every birth frame are the same either way.
 * Synchronises the fuser.  SF_ERR_INVALID_ARG for an unknown key or a value out of range. */
int sf_fuser_tune(sf_fuser* f, const char* key, int value);

/* How many look-ups of the cube-window allocation kernels went to the hash table since the fuser was created or reset (32-bit, wraps): with the presence cache
 * on, what is left are new blocks, blocks of the previous batch and bricks whose cache entry another brick holds. */
int sf_fuser_alloc_probe_count(sf_fuser* f, uint64_t* out);

/* How many blocks the allocation kernels took to the global hash table one by one because a workgroup's LDS queue / hash set was full (the slow
 * path k_alloc_ray exists to avoid; the volume is the same either way).  tests/test_gpu_tsdf.py asserts 0 on the bench walk's corners. */
int sf_fuser_alloc_direct_count(sf_fuser* f, uint64_t* out);

/* Phases of the most recent sf_fuser_extract_mesh, milliseconds: [0] whole call, [1] live-block list, [2] count pass (k_mc), [3] scan + emit pass,
 * [4] vertex sort (radix sort of the 3T edge keys), [5] heads + scan + weld, [6] triangle sort + gather, [7] downloads (device side), [8] host: output
 * arrays allocated + downloads awaited; then [9] live blocks, [10] triangles, [11] welded vertices.  n <= 12 values are written. */
int sf_fuser_mc_timing(const sf_fuser* f, double* out, int n);

/* One zlib stream through the DEVICE inflate of the frame pipeline (csrc/inflate_gpu.hip): the bytes are sf_zlib_inflate's.  SF_ERR_UNSUPPORTED
 * for streams the device leaves to the host inflater (anything but ONE final fixed-Huffman block -- what the reference's writer and this
 * library's emit; expect_bytes not a multiple of 4), SF_ERR_FORMAT for corrupt streams and streams that inflate to another size. */
int sf_zlib_inflate_gpu(const void* src, uint64_t src_bytes, uint64_t expect_bytes, int device, void* dst);
/* The two kernels of that path timed apart (HIP events) on `count` <= 32 resident streams: microseconds per launch (tools/gpu/inflate_bench.py); skip (a -DSF_MEASURE_ABLATE build only, SF_ERR_UNSUPPORTED otherwise): 1 = the token kernel without its writing pass, 2 = without its scans either. */
int sf_zlib_inflate_gpu_bench(const void* const* srcs, const uint64_t* src_bytes, int count, uint64_t expect_bytes, int device, int repeats, int skip, double* us_tokens, double* us_copy);

/* Where the frames of the calling thread's last sf_fuse_run were decoded: out[0] zlib depth frames inflated on the device, out[1] by the host threads
 * (streams the device does not take, or SF_INFLATE_HOST), out[2] JPEG colour frames entropy-decoded on the device, out[3] by the host threads. */
int sf_fuse_run_device_counts(uint64_t out[4]);

/* One baseline-JPEG picture through the whole DEVICE path of the frame pipeline: headers parsed and the byte stuffing removed on the host, entropy
 * decoding (csrc/jpeg_huff_gpu.hip) and reconstruction (csrc/jpeg_gpu.hip) on GPU `device`; the bytes are sf_jpeg_decode's.  SF_ERR_UNSUPPORTED
 * for what the device's entropy decoder leaves to the host (restart intervals, sampling factors above 2), SF_ERR_FORMAT for a corrupt stream. */
int sf_jpeg_decode_gpu_huffman(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, int device, uint8_t* dst_rgb);

/* Kernel timing with HIP events on the fuser's stream: when enabled, every integrate launch is bracketed
 * by an event pair; sf_fuser_profile_read sums and clears them (synchronises). */
int sf_fuser_profile_enable(sf_fuser* f, int on);
int sf_fuser_profile_read(sf_fuser* f, double* integrate_ms, uint64_t* launches, uint64_t* blocks);

/* Synthetic stream source (benchmark input): renders frames
 * [first_frame, first_frame+n) of the `total_frames`-frame box-room walk as u16 millimetre depth directly
 * into device memory and returns the n camToWorld poses (n*16 floats, host).
 */
int sf_synth_room_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                         int width, int height, int noise, float* poses_out);
/* The same walk through a box room of room_m = {x, y, z} metres whose corner sits at origin_m (NULL: the world origin): the other scans
 * of SURVEY 8d config 4 (room size +-20 %) and the rooms of config 5's corridor world (one origin per room). */
int sf_synth_scan_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                         int width, int height, int noise, const double room_m[3], const double origin_m[3], float* poses_out);

/* The same with the scene and the noise model chosen ("real-entropy inputs").  noise 0: none; 1: the round-1 LCG ramp (kept
 * for the committed digests); 2: three LSBs hashed per pixel and frame.  scene 0: the empty box room; 1: the room furnished with the 48 boxes of
 * sf_synth_clutter_boxes(room, seed) -- wall furniture, a table island, shelves, lamps -- and, under noise 2, sensor holes (grazing
 * incidence, 0.4 % speckle).  scannet_amd/synth.py renders the same scenes on the host. */
int sf_synth_scene_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                          int width, int height, int noise, int scene, uint32_t seed, const double room_m[3], const double origin_m[3],
                          float* poses_out);
int sf_synth_clutter_boxes(const double room_m[3], uint32_t seed, float* lo_out /* 48 x 3 */, float* hi_out /* 48 x 3 */, int* n_out);

/* Decimation known-answer probe (tests/test_simplify_known_answers.py): of mesh `in` as the quadric edge collapse sees it before the first
 * collapse -- the summed quadric of the edge's end points {a00 a01 a02 a11 a12 a22, b0 b1 b2, c} (error(x) = x'Ax + b'x + c), the position
 * the collapsed vertex would take, the priority of the collapse, and ScaleFactor = 1e8 / diag^6.  Any output may be NULL. */
struct sf_mesh;
int sf_simplify_probe_edge(const struct sf_mesh* in, const sf_simplify_params* p, uint32_t v0, uint32_t v1, double quadric10[10], float position[3],
                           float* priority, double* scale_factor);

/* Device self-test: the hand-expanded correctly rounded divisions of the integrate and allocation kernels against the hardware's IEEE
 * division -- all 2^23 mantissas x 9 exponents for 1/x, 511 integer divisors x 2^21 numerators for n/m, 2^28 general operand pairs
 * (incl. near-exact and near-half-way quotients) for a/b.  All three counts must be 0 (tests/test_gpu_tsdf.py). */
int sf_selftest_division(int device, uint64_t* recip_mismatches, uint64_t* quot_mismatches, uint64_t* div_mismatches);

/* Measurement aid (bench.py roofline_single_frame.pattern_ceiling): the memory traffic of the most recent integrate pass
 * without its arithmetic -- every tile of that pass's list is read and (read_only == 0) written back unchanged, with the
 * integrate kernel's launch geometry; iters timed launches, average duration in microseconds.  The volume is unchanged. */
int sf_fuser_calib_tile_rmw(sf_fuser* f, int read_only, int iters, double* avg_us, uint32_t* tiles);
/* The same traffic taken apart (bench.py roofline.hbm_out_of_cache.rmw_decomposition).  mode bit 0: read only; bit 1: tiles 0 .. n - 1 of the pool -- one
 * contiguous span of the same size -- instead of the pass's scattered list; bits 2-3: log2 of the tiles a wave reads before it writes them back (1 / 2 / 4).
 * Every tile is written back as read: the volume is unchanged. */
int sf_fuser_calib_tile_rmw_ex(sf_fuser* f, int mode, int iters, double* avg_us, uint32_t* tiles);

/* PMC calibration stream (tools/pmc_calibrate.py): known-byte-count 16 B/lane RMW + read-only launches. */
int sf_calib_stream(int device, uint64_t bytes, int iters);

#ifdef __cplusplus
}
#endif
#endif

/*
 * scanfuse.h -- C ABI of the MI355X-native RGB-D integration hot path (libscanfuse.so).
 *
 * This is the drop-in boundary for the `improve` and `segment` stages of the reference pipeline
 * (Server/scan_processor.py:137-138,155-156).  Plain pointers and sizes only; no exception crosses the
 * ABI.  Every function returns SF_OK (0) or a negative sf_status; the message of the last failure on the
 * calling thread is available from sf_last_error().
 *
 * Each group below cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef SCANFUSE_H
#define SCANFUSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sf_status {
  SF_OK = 0,
  SF_ERR_INVALID_ARG = -1,
  SF_ERR_IO = -2,
  SF_ERR_FORMAT = -3,      /* bad .sens / PLY / zlib / parameter file */
  SF_ERR_UNSUPPORTED = -4, /* e.g. arithmetic-coded JPEG colour, PreserveBoundary in simplify.mlx */
  SF_ERR_DEVICE = -5,      /* HIP error, no GPU */
  SF_ERR_CAPACITY = -6,    /* SDF block heap or hash table exhausted */
  SF_ERR_BOUNDS = -7,
  SF_ERR_SKIPPED = -8      /* frame skipped: camToWorld is all -inf (tracking lost, sensorData.h:382) */
} sf_status;

const char* sf_last_error(void);
const char* sf_version(void);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction parameters.  Field names follow Server/tools/recons/zParametersScanNet.txt (the
 * mLib ParameterFile the reference passes to FriedLiver.exe / DepthSensing.exe as argv[1]).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_params {
  int32_t depth_width, depth_height;   /* size of the depth frames handed to integrate()          */
  float fx, fy, mx, my;                /* depth intrinsics (sensorData.h:305-312 layout)          */
  float depth_shift;                   /* m_depthShift, sensorData.h:895 (1000)                   */
  float depth_min, depth_max;          /* s_sensorDepthMin / Max            (:34-35)              */
  float voxel_size;                    /* s_SDFVoxelSize                    (:47)                 */
  float trunc_base, trunc_scale;       /* s_SDFTruncation / Scale           (:49-50)              */
  float max_integration_dist;          /* s_SDFMaxIntegrationDistance       (:51)                 */
  int32_t weight_sample;               /* s_SDFIntegrationWeightSample      (:52)                 */
  int32_t weight_max;                  /* s_SDFIntegrationWeightMax (:53), saturated at 255       */
  float mc_thresh_factor;              /* s_SDFMarchingCubeThreshFactor     (:48)                 */
  uint32_t hash_num_buckets;           /* s_hashNumBuckets                  (:56)                 */
  uint32_t hash_bucket_size;           /* HASH_BUCKET_SIZE (pound-defined upstream, :55) = 10     */
  uint32_t num_sdf_blocks;             /* s_hashNumSDFBlocks                (:57)                 */
  uint32_t mc_max_triangles;           /* s_marchingCubesMaxNumTriangles (:106); 0 = unlimited    */
  int32_t gc_enabled;                  /* s_garbageCollectionEnabled        (:83)                 */
  /* Colour frames at their own resolution (real ScanNet scans: 1296x968 colour, 640x480 depth, sensorData.h:1269-1272).
   * color_width == 0: rgb buffers are depth_width x depth_height.  Otherwise rgb buffers are color_width x color_height
   * and every depth pixel (x, y) takes the colour pixel under the same ray: u = (x - mx) / fx * cfx + cmx (nearest),
   * black outside the colour image -- depth and colour share the extrinsics in ScanNet's .sens files. */
  int32_t color_width, color_height;
  float cfx, cfy, cmx, cmy;            /* colour intrinsics (m_calibrationColor, sensorData.h:1264) */
  /* s_integrationWidth / s_integrationHeight (zParametersScanNet.txt:20-21, shipped as 320 x 240: "input depth gets re-sampled to this
   * width").  0 x 0: integrate at depth_width x depth_height.  Otherwise every frame is resampled to this size before anything else:
   * integration pixel (x, y) takes input pixel ((uint)(x * (depth_width - 1) / (integration_width - 1) + 0.5f), same in y) -- nearest, the
   * sampling convention of the reference's own resample kernels (AnnotationTools/Filter2dAnnotations/filter.cu:647-665) --, the intrinsics
   * follow it (fx * iw / dw, mx * (iw - 1) / (dw - 1): Calibrate/src/calibration.h:125-128), colour is looked up under the same ray. */
  int32_t integration_width, integration_height;
  /* Upstream-conformance switches (DESIGN.md section 6b "upstream vs this specification").  The TSDF algorithm lives in external code
   * (VoxelHashing / BundleFusion) that is not in the reference tree; SURVEY.md App. C is the specification built here and 0 selects it
   * everywhere.  Where the public upstream sources are remembered to differ, the other value reproduces THAT behaviour -- in the kernels,
   * in the CPU checker (oracle/tsdf_oracle.c) and in the float64 literal evaluator (oracle/spec_literal.py) alike -- so that a maintainer
   * who holds the Windows binaries can pick the semantics that match them.
   *   frustum_mode  0: a block is allocated / fused in a frame when its bounding sphere touches the view frustum (conservative: every voxel
   *                    that projects into the image is visited).
   *                 1: VoxelHashing's isSDFBlockInCameraFrustumApprox: the block CENTRE is projected, its normalised device coordinates
   *                    are scaled by 0.95 and must lie in [-1, 1]^2 x [0, 1] (z against s_sensorDepthMin / Max) -- border blocks whose
   *                    centre falls outside are neither allocated nor fused although some of their voxels project into the image.
   *   colour_round  0: running colour average (old + new) / 2 in integer arithmetic (App. C as written: truncation).
   *                 1: (uchar)(0.5f * old + 0.5f * new + 0.5f) per channel (combineVoxel upstream: round half up).
   *   colour_first  0: a voxel with weight 0 takes the observed colour as it is.
   *                 1: a voxel whose accumulated colour is black (r + g + b == 0) does (combineVoxel upstream).
   *   weight_mode   0: every observation weighs s_SDFIntegrationWeightSample (BundleFusion forces this; App. C).
   *                 1: VoxelHashing: (uchar)max(weightSample * 1.5f * (1 - (d - depthMin) / (depthMax - depthMin)), 1.0f).  With the shipped
   *                    weightSample = 1 (zParametersScanNet.txt:52) both give 1 for every depth. */
  int32_t frustum_mode, colour_round, colour_first, weight_mode;
  /*   weight_wrap   0: the 8-bit weight saturates at min(s_SDFIntegrationWeightMax, 255) (SURVEY App. C decision).
   *                 1: upstream stores min(weightMax, w0 + w1) into its `uchar weight` -- with the shipped weightMax = 99999999
   *                    (zParametersScanNet.txt:53) the 256th observation of a voxel WRAPS its weight to 0 and the running mean starts again.
   *                    weight_max is then taken as given (not clamped to 255) and the stored weight is the sum modulo 256. */
  int32_t weight_wrap;
} sf_params;

/* SURVEY 8d camera + zParametersScanNet.txt values with BASELINE.json's 4 mm / 2^19-bucket overrides */
void sf_params_default(sf_params* p);
/* Parse an mLib ParameterFile ("name = v1 v2 ...; // comment").  Unknown keys are ignored; keys that are
 * present override *p (call sf_params_default first).  Replaces GlobalAppState::readMembers upstream;
 * in-tree format examples: Alignment/src/globalAppState.h:8-41. */
int sf_params_load_file(const char* path, sf_params* p);
/* The five upstream-conformance switches at once: which = 1 "VoxelHashing" (DepthSensing.exe, the binary the `improve` stage runs,
 * Server/scan_processor.py:34-35,138): frustum_mode = colour_round = colour_first = weight_mode = weight_wrap = 1; which = 2 "BundleFusion"
 * (FriedLiver.exe, the `recons` stage, :27-29,126): the same with weight_mode = 0 (that code base computes the depth-dependent weight and then
 * forces it to 1); which = 0: SURVEY App. C (all 0, the default).  The upstream behaviour is AS REMEMBERED -- neither code base is in the
 * reference tree -- which is why it is a preset a maintainer can confirm with one mesh (INTEGRATION.md "Conformance packet"),
 * not the default.  Also a parameter-file key: s_scanfuseUpstream = 0 | 1 | 2 (applied before the five individual keys, which override it). */
int sf_params_upstream_preset(sf_params* p, int which);

/* ------------------------------------------------------------------------------------------------
 * Voxel-hash TSDF fuser.  Replaces the scene-representation calls of the external DepthSensing.exe /
 * FriedLiver.exe (call sites Server/scan_processor.py:126,138).  One fuser per HIP
 * device + stream; a handle is not thread-safe, distinct handles are independent.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_fuser sf_fuser;

typedef struct sf_stats {
  uint64_t frames_integrated, frames_skipped;
  uint32_t blocks_allocated;     /* live SDF blocks                                              */
  uint32_t heap_free;            /* free SDF blocks ("heapFreeCount" of processed.txt)           */
  uint32_t last_frame_blocks;    /* N_blk of the last integrate: allocated AND in the frustum    */
  uint32_t alloc_failures;       /* blocks that could not be allocated (heap / table exhausted)  */
  uint64_t total_frame_blocks;   /* sum of N_blk over all frames since create / reset_counters   */
  uint32_t hash_slots_used;
  uint32_t high_water;           /* 1 + highest heap block index ever handed out                 */
  uint64_t total_pass_tiles;     /* sum over the passes (of up to sf_fuser_batch_frames() frames) of the tiles the pass read and wrote:
                                    what temporal blocking moves through HBM, against total_frame_blocks for frame-by-frame fusion */
} sf_stats;

int sf_device_count(int* count);
int sf_fuser_create(const sf_params* p, int device, sf_fuser** out);
void sf_fuser_destroy(sf_fuser* f);

/* Host-buffer entry points: depth = W*H u16 (row-major, as decompressDepthAlloc returns it,
 * sensorData.h:943-946), rgb = W*H*3 u8 at depth resolution or NULL, pose = row-major camToWorld
 * (RGBDFrame::getCameraToWorld, sensorData.h:432).  The buffers have been read when the call returns (they may be reused or freed at once);
 * the fusion itself runs asynchronously (sf_fuser_sync / any later call orders behind it).  SF_ERR_SKIPPED for -inf poses. */
int sf_fuser_integrate(sf_fuser* f, const uint16_t* depth, const uint8_t* rgb, const float pose[16]);
int sf_fuser_deintegrate(sf_fuser* f, const uint16_t* depth, const uint8_t* rgb, const float pose[16]);

/* Device-buffer entry points (inputs already resident in HBM; used by bench.py and by sf_fuse_run). */
int sf_fuser_integrate_device(sf_fuser* f, const void* d_depth, const void* d_rgb, const float pose[16]);
int sf_fuser_deintegrate_device(sf_fuser* f, const void* d_depth, const void* d_rgb, const float pose[16]);
/* n frames laid out `frame_stride_bytes` apart starting at d_depth; poses = n*16 floats on the host; -inf poses are
 * skipped.  Frames are fused sf_fuser_batch_frames() at a time: one pass over the voxel tiles applies every frame of
 * the batch in frame order (temporal blocking) -- the result is bit-identical to frame-by-frame integration. */
int sf_fuser_integrate_batch_device(sf_fuser* f, const void* d_depth, uint64_t frame_stride_bytes,
                                    const float* poses, uint64_t n);
/* The same with a colour frame per depth frame (rgb: depth_width x depth_height x 3 bytes, or color_width x color_height x 3 when sf_params
 * gives a colour resolution), `rgb_stride_bytes` apart. */
int sf_fuser_integrate_batch_device_rgb(sf_fuser* f, const void* d_depth, uint64_t frame_stride_bytes, const void* d_rgb, uint64_t rgb_stride_bytes,
                                        const float* poses, uint64_t n);
int sf_fuser_batch_frames(const sf_fuser* f);   /* 32 */

/* Empty volume again (table, heap, tiles, counters, frame numbering); parameters, streams and tuning stay.  The reference's tools are
 * one process per scan (Server/scan_processor.py:137-141); a caller that fuses scan after scan keeps its allocation this way. */
int sf_fuser_reset(sf_fuser* f);
int sf_fuser_garbage_collect(sf_fuser* f, uint32_t* freed);
int sf_fuser_sync(sf_fuser* f);
int sf_fuser_stats(sf_fuser* f, sf_stats* out); /* synchronises */
void* sf_fuser_stream(sf_fuser* f);            /* the hipStream_t all work of this handle is queued on */


/* Copy out the live blocks (for parity checks): coords = n*3 int32 block coordinates, voxels = n*4096
 * bytes ({float sdf; uchar r,g,b,weight} x 512, index z*64+y*8+x).  Pass NULLs to query n only. */
int sf_fuser_export_blocks(sf_fuser* f, int32_t* coords, void* voxels, uint64_t capacity, uint64_t* n);

/* One large scan over several GPUs (BASELINE configs[4]): the block space is cut into slabs along `axis`; a fuser with a
 * slab set still sees every frame but only allocates (hence fuses) the blocks whose coordinate on `axis` lies in
 * [lo_block, hi_block).  axis < 0 removes the partition.  Before meshing, each fuser imports -- as GHOST blocks, which are
 * read as neighbours but never fused, meshed or garbage-collected -- the lowest block layer of the slab above it (marching
 * cubes needs the +1 voxel neighbours): sf_fuser_export_blocks_where(axis, lo, lo + 1) on the owner, an all-gather (RCCL
 * over xGMI when the buffers are device tensors: dst_on_device / src_on_device = 1), sf_fuser_import_blocks(ghost = 1) on
 * the neighbour.  Slabs along x concatenate into the canonical mesh (scannet_amd/partition.py). */
int sf_fuser_set_slab(sf_fuser* f, int axis, int32_t lo_block, int32_t hi_block);
int sf_fuser_export_blocks_where(sf_fuser* f, int axis, int32_t lo, int32_t hi, int include_ghosts, int32_t* coords, void* voxels,
                                 uint64_t capacity, uint64_t* n, int dst_on_device);   /* coords = voxels = NULL: count only */
int sf_fuser_import_blocks(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int ghost, int src_on_device);
/* The same partition as STRIPES: layers of `thickness_blocks` blocks along `axis`, dealt round-robin to the `world` fusers starting at
 * `origin_block` (owner = floor((c - origin) / thickness) mod world).  A slab per GPU leaves all GPUs but one idle while the camera
 * is inside one slab; stripes a fraction of the view frustum thick spread every frame over all of them, at the price of one
 * boundary layer per stripe in the exchange (1 / thickness of the blocks).
 * The exchange without host staging: sf_fuser_export_boundary writes the lowest layer of each of this fuser's slabs / stripes
 * (coords n x 3 int32, voxels n x 4096 bytes; NULLs: count only) -- into device memory with dst_on_device = 1 --, the payloads are
 * all-gathered (RCCL), and sf_fuser_import_ghosts keeps, of a gathered payload, exactly the blocks this fuser needs as ghosts (the
 * ones directly above a layer it owns); *imported = how many. */
int sf_fuser_set_stripes(sf_fuser* f, int axis, int32_t origin_block, int32_t thickness_blocks, int world, int rank);
int sf_fuser_export_boundary(sf_fuser* f, int32_t* coords, void* voxels, uint64_t capacity, uint64_t* n, int dst_on_device);
int sf_fuser_import_ghosts(sf_fuser* f, const int32_t* coords, const void* voxels, uint64_t n, int src_on_device, uint64_t* imported);

/* The exchange step itself for host programs WITHOUT a process group (bin/depthsensing --ranks N: N copies of one executable, one per GPU; callers that
 * have torch.distributed use scannet_amd/partition.py): rank r's boundary layers go to rank r - 1, rank r + 1's come in and the wanted ones are kept
 * as ghosts -- from device memory to device memory, nothing staged on the host.  Transport: RCCL (ncclSend / ncclRecv over xGMI; librccl.so is loaded
 * with dlopen when the first exchange is created) for ranks on distinct GPUs, a hipIpc mapping of the owner's buffer for ranks that share a device
 * (RCCL refuses those) or where librccl is absent; SF_EXCHANGE_AUTO picks -- identically on every rank, from notes the ranks leave in
 * `rendezvous_dir` (a directory only this run's ranks see; a file named "abort" in it makes every wait give up).  The north star's all-gather of the
 * boundary blocks is this ring shift with the blocks nobody but one neighbour needs left out (SURVEY 8e "cheaper equivalent").
 * SF_ERR_UNSUPPORTED: no device-to-device transport on every rank -- the caller falls back to its own (files). */
typedef struct sf_exchange sf_exchange;
enum { SF_EXCHANGE_AUTO = 0, SF_EXCHANGE_RCCL = 1, SF_EXCHANGE_IPC = 2 };
int sf_exchange_create(const char* rendezvous_dir, int rank, int ranks, int device, int transport, sf_exchange** out);
int sf_exchange_boundary(sf_exchange* x, sf_fuser* f, uint64_t* sent, uint64_t* received, uint64_t* kept);
const char* sf_exchange_transport(const sf_exchange* x);   /* what sf_exchange_create chose, for the log */
void sf_exchange_destroy(sf_exchange* x);

/* Fuse frames [first, last) of an opened .sens file (last = 0: to the end): a pool of `decode_threads` (0 = one
 * per core, at most 32 / 64) fills pinned buffers in frame order -- zlib depth frames as the reference's writer
 * stores them (one fixed-Huffman block, stb_image_write.h:733-736) are copied COMPRESSED and inflated on the GPU
 * (4 threads keep up); any other depth stream and JPEG / PNG colour are decoded by the pool (JPEG: entropy decoding
 * only, the rest on the GPU) --, copies and kernels are queued as frames become ready.  Replaces the frame loop around
 * RGBDFrameCacheRead (sensorData.h:1717-1831) and, inside it, decompressDepthAlloc / decompressColorAlloc
 * (sensorData.h:600-616, 693-709).  The fuser must have been created for the file's depth resolution.  Colour is fused
 * when it is stored at depth resolution or at the colour resolution given in sf_params.  Streams and the page-locked
 * pool are kept for the next call of the process.  The library never touches the process
 * environment: the run wants a hardware queue per stream (HIP's GPU_MAX_HW_QUEUES, default 4, the application's to
 * export before its first HIP call); on fewer it still returns SF_OK and leaves a "note: ..." text in sf_last_error().
 * A depth frame whose stream is corrupt fails the run with SF_ERR_FORMAT; when the device inflated it, the frame was
 * fused as "no measurement" (zero depth) before the failure reached the host -- never with another frame's pixels. */
typedef struct sf_run_stats {
  uint64_t frames_total, frames_integrated, frames_skipped;
  uint32_t decode_threads, color_fused;
  double seconds_total;       /* wall time of the call: first byte decoded -> last kernel complete */
  double seconds_decode_cpu;  /* decode time summed over the pool */
} sf_run_stats;
struct sf_sens;
int sf_fuse_run(sf_fuser* f, const struct sf_sens* s, uint64_t first, uint64_t last, int decode_threads, sf_run_stats* stats);
/* A hint about the calling thread's most recent successful sf_fuse_run that is NOT an error ("" when there is none): today, that the run drove more streams than
 * the process has hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at its first call: INTEGRATION.md section 4).  sf_last_error() stays empty on success. */
const char* sf_fuse_run_note(void);
/* Optional, for a process that fuses ONE scan (the pipeline's contract: one `DepthSensing.exe` per scan, Server/scan_processor.py:138): with the file open
 * and BEFORE sf_fuser_create, start making what sf_fuse_run will want for THIS file -- its side streams (a hardware queue each: ~5 ms), the page-locked ring,
 * the device ring, one pass of copies over both -- on a thread of its own, beside sf_fuser_create's own gigabytes of allocation.  A JPEG-colour scan wants
 * several times what a depth-only one does; without this call the first sf_fuse_run pays the difference inside its loop (RGB-D: 9.8 k frames/s in the first
 * run of a process against 16 k in the second).  Changes nothing a run computes. */
int sf_fuse_run_prepare(const struct sf_sens* s, const sf_params* p, int device);

/* Iso-surface extraction (marching cubes over all live blocks, exact weld): the `<id>_vh.ply` product of the
 * improve stage (Server/scan_processor.py:141, scan_stages.json:33-42).  Vertices are ordered by grid-edge key,
 * triangles by cube; the result is deterministic.  Free with sf_mesh_free; write with sf_mesh_write_ply. */
struct sf_mesh;
int sf_fuser_extract_mesh(sf_fuser* f, struct sf_mesh** out);

/* Host <-> device helpers so that callers without a HIP binding can stage inputs in HBM. */
int sf_device_malloc(int device, uint64_t bytes, void** out);
int sf_device_free(void* p);
int sf_device_upload(void* dst, const void* src, uint64_t bytes);
int sf_device_download(void* dst, const void* src, uint64_t bytes);

/* ------------------------------------------------------------------------------------------------
 * zlib codec used for .sens depth blobs.  Replaces stb::stbi_zlib_decode_malloc / stbi_zlib_compress as
 * called from SensReader/c++/src/sensorData.h:703-709 and :659-670.  Like the reference reader the
 * decoder validates the 2-byte zlib header and does NOT verify the trailing Adler-32.
 * ---------------------------------------------------------------------------------------------- */
int sf_zlib_inflate(const void* src, uint64_t src_bytes, void* dst, uint64_t dst_capacity, uint64_t* out_bytes);
int sf_zlib_deflate(const void* src, uint64_t src_bytes, void* dst, uint64_t dst_capacity, uint64_t* out_bytes);
uint64_t sf_zlib_deflate_bound(uint64_t src_bytes);

/* ------------------------------------------------------------------------------------------------
 * .sens v4 container.  Replaces ml::SensorData (SensReader/c++/src/sensorData.h:285-1936):
 *   sf_sens_open          SensorData(filename) / loadFromFile          :855-859, :1250-1290
 *   sf_sens_get_info      header members                               :1257-1273
 *   sf_sens_decode_depth  decompressDepthAlloc(frameIdx)               :943-946 (strict bounds check; the
 *                         reference's `>` off-by-one at :944 is not reproduced); caller owns dst (W*H u16)
 *   sf_sens_decode_color  decompressColorAlloc(frameIdx)               :933-936 (RAW and baseline JPEG)
 *   sf_sens_pose          m_frames[i].getCameraToWorld()               :432; *valid = 0 for the all -inf
 *                         "tracking lost" pose (:382, SensReader/c++/README.txt:55-57)
 *   sf_sens_create/add_frame/save   initDefault / addFrame / saveToFile  :891-921, :1101-1109
 *   sf_sens_set_pose      RGBDFrame::setCameraToWorld                  :436 (trajectory rewrite by `recons`)
 * The reference throws MLibException on open / version errors (:883-886, :1253-1255); here every failure is
 * a negative sf_status.  The file is memory-mapped; compressed blobs are never copied.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_sens sf_sens;

typedef struct sf_sens_info {
  uint32_t version;                    /* 4 */
  uint32_t color_width, color_height, depth_width, depth_height;
  int32_t color_compression;           /* -1 unknown, 0 raw, 1 png, 2 jpeg   (sensorData.h:346-351) */
  int32_t depth_compression;           /* -1 unknown, 0 raw u16, 1 zlib u16, 2 occi u16 (:352-357)  */
  float depth_shift;
  uint64_t num_frames, num_imu;
  float color_intrinsic[16], color_extrinsic[16], depth_intrinsic[16], depth_extrinsic[16]; /* row-major */
  char sensor_name[256];
} sf_sens_info;

typedef struct sf_sens_frame_meta_t {
  uint64_t timestamp_color, timestamp_depth;  /* microseconds (sensorData.h:440-457) */
  uint64_t color_bytes, depth_bytes;          /* compressed sizes */
} sf_sens_frame_meta_t;

int sf_sens_open(const char* path, sf_sens** out);
void sf_sens_close(sf_sens* s);
int sf_sens_get_info(const sf_sens* s, sf_sens_info* out);
int sf_sens_decode_depth(const sf_sens* s, uint64_t frame, uint16_t* dst);
int sf_sens_decode_color(const sf_sens* s, uint64_t frame, uint8_t* dst_rgb);
int sf_sens_pose(const sf_sens* s, uint64_t frame, float out16[16], int* valid);
/* SensorData::computeDepthImage(frameIdx) (:968-982): W*H floats in metres, (float)depth / depthShift, 0 -> 0.0f (its invalid value) */
int sf_sens_depth_image(const sf_sens* s, uint64_t frame, float* dst_metres);
int sf_sens_frame_meta(const sf_sens* s, uint64_t frame, sf_sens_frame_meta_t* out);
/* RGBDFrame::getColorCompressed / getDepthCompressed (sensorData.h:418-429): the frame's compressed blobs where they lie (any out pointer may be
 * NULL); valid until sf_sens_close or, for a file under construction, until the next frame is added. */
int sf_sens_frame_blobs(const sf_sens* s, uint64_t frame, const uint8_t** color, uint64_t* color_bytes, const uint8_t** depth, uint64_t* depth_bytes);
/* Writer.  `header` supplies everything but num_frames / num_imu.  color = W*H*3 RGB bytes for TYPE_RAW, an
 * already encoded blob for TYPE_JPEG / TYPE_PNG, or NULL / 0 for no colour; depth = W*H u16, compressed as
 * header->depth_compression says (0 raw, 1 zlib). */
int sf_sens_create(const sf_sens_info* header, sf_sens** out);
int sf_sens_add_frame(sf_sens* s, const uint8_t* color, uint64_t color_bytes, const uint16_t* depth, const float pose[16],
                      uint64_t timestamp_color, uint64_t timestamp_depth);
/* n depth-only frames (`frame_stride_bytes` apart; poses = n x 16 floats; depth time stamps timestamp0 + i * step) compressed on `threads`
 * threads (0 = every CPU this process may use) and appended in order -- the result is the file n sf_sens_add_frame calls would write. */
int sf_sens_add_depth_frames(sf_sens* s, const uint16_t* depth, uint64_t frame_stride_bytes, uint64_t n, const float* poses,
                             uint64_t timestamp0_us, uint64_t timestamp_step_us, int threads);
/* A frame whose colour and depth blobs are already in the container's compression types -- stored as given (what RGBDFrame::loadFromFile
 * keeps per frame, sensorData.h:743-754): transcoding, merging files, depth streams another writer compressed. */
int sf_sens_add_frame_blobs(sf_sens* s, const uint8_t* color, uint64_t color_bytes, const uint8_t* depth, uint64_t depth_bytes,
                            const float pose[16], uint64_t timestamp_color, uint64_t timestamp_depth);
int sf_sens_set_pose(sf_sens* s, uint64_t frame, const float pose[16]);
int sf_sens_save(const sf_sens* s, const char* path);
/* SensorData::loadFromImages(sourceFolder, basename = "frame-", colorEnding = "png") (sensorData.h:1468-1559, FreeImage builds only): a folder as
 * saveToImages / bin/sens writes it -- info.txt (or _info.txt), <basename>%06d.color.jpg|png (kept as the colour blob), .depth.png (16-bit grey; or
 * the .depth.pgm saveToImages really writes), .pose.txt -- into a .sens in memory (zlib depth, time stamps 0), frames until one is incomplete.
 * basename NULL: "frame-"; color_ending NULL: what frame 0 has.  scannet_amd/csrc/sens_images.cpp. */
int sf_sens_load_from_images(const char* folder, const char* basename, const char* color_ending, sf_sens** out);
/* SensorData::saveToImages(outputFolder, basename = "frame-") (sensorData.h:1380-1466): _info.txt + per frame <basename>%06d.color.jpg|png (the stored
 * blob; a TYPE_RAW frame as a PNG), .depth.pgm (16-bit big-endian, depth shift in the comment), .pose.txt -- the folder the reference writes, byte for
 * byte.  Frames are written by a pool of threads; progress (nullable) is called on the caller's thread with (frame, num_frames, user) in frame order.
 * bin/sens is this call plus the reference tool's stdout. */
int sf_sens_save_to_images(const sf_sens* s, const char* folder, const char* basename, void (*progress)(uint64_t, uint64_t, void*), void* user);
/* SensorData::saveToPointCloud(filename, frameFrom, frameTo) (:1564-1602, compiled only with mLib; the reference's statement of the unprojection, SURVEY 8a row
 * a6): every valid depth pixel of frames [frame_from, frame_to) (frame_to = 0: one frame) unprojected with K_depth^-1, moved by the frame's camera-to-world
 * (identity for a lost pose) and coloured by the pixel the colour camera sees there (alpha 255; (0,0,0,0) outside the colour image) -> a binary PLY point cloud. */
int sf_sens_save_point_cloud(const sf_sens* s, const char* ply_path, uint64_t frame_from, uint64_t frame_to, uint64_t* n_points);
/* Editing a file in memory, opened or under construction (then sf_sens_save):
 *   sf_sens_replace_depth   SensorData::replaceDepth(frameIdx, depth)  :948-955,499-502 (W*H u16, compressed with the file's type; depth time stamp -> 0
 *                           as freeDepth leaves it, :516-521) -- what the Calibrate stage does to every frame (Calibrate/src/calibration.h:303)
 *   sf_sens_replace_color   SensorData::replaceColor(frameIdx, color)  :957-964,505-508 (`color` as sf_sens_add_frame takes it; colour time stamp -> 0)
 *   sf_sens_append          SensorData::append(second)                 :1605-1624 (frames only, no IMU; ANY difference in frame sizes or compression
 *                           types is refused -- the reference's test joins its six comparisons with && and so lets almost everything through)
 *   sf_sens_equal           SensorData::operator==                     :1626-1650 (floats and doubles compared as numbers: -inf equals -inf, NaN nothing) */
int sf_sens_replace_depth(sf_sens* s, uint64_t frame, const uint16_t* depth);
int sf_sens_replace_color(sf_sens* s, uint64_t frame, const uint8_t* color, uint64_t color_bytes);
int sf_sens_append(sf_sens* s, const sf_sens* other);
int sf_sens_equal(const sf_sens* a, const sf_sens* b, int* equal);
/* SensorData::applyTransform(t) (:1047-1054): camera-to-world of every tracked frame <- t * m (row-major); all -inf poses are left alone */
int sf_sens_apply_transform(sf_sens* s, const float t[16]);

/* Frames streamed into a file as they arrive: SensorData::LiveSensorDataWriter (sensorData.h:1112-1246, _HAS_MLIB builds only there).  The header goes
 * out at open (a path that exists: overwritten, or -- overwrite = 0 -- the name's numeric suffix is counted up until it is free, :1118-1131;
 * sf_sens_writer_path says which); add copies the caller's buffers into a queue of at most cache_frames (0: 500, the reference's default) and blocks
 * while it is full; one background thread compresses and writes in order; close drains, writes "0 IMU frames" and patches the frame count (:1146-1157).
 * The file is byte for byte what the in-memory writer saves.  An error of any frame is returned by the next add and by close.
 * scannet_amd/csrc/sens_writer.cpp. */
typedef struct sf_sens_writer sf_sens_writer;
int sf_sens_writer_open(const sf_sens_info* header, const char* path, int overwrite, uint32_t cache_frames, sf_sens_writer** out);
const char* sf_sens_writer_path(const sf_sens_writer* w);
int sf_sens_writer_add_frame(sf_sens_writer* w, const uint8_t* color, uint64_t color_bytes, const uint16_t* depth, const float pose[16],
                             uint64_t timestamp_color, uint64_t timestamp_depth);
int sf_sens_writer_add_frame_blobs(sf_sens_writer* w, const uint8_t* color, uint64_t color_bytes, const uint8_t* depth, uint64_t depth_bytes,
                                   const float pose[16], uint64_t timestamp_color, uint64_t timestamp_depth);
int sf_sens_writer_close(sf_sens_writer* w, uint64_t* frames_written /*nullable*/);

/* IMU frames of a .sens under construction: 128 bytes each = rotationRate, acceleration, magneticField, attitude, gravity
 * (5 x 3 doubles) + u64 time stamp in microseconds (sensorData.h:796-803); addIMUFrame :923-926. */
int sf_sens_add_imu(sf_sens* s, const void* frame128);
/* Reading them back: sf_sens_imu = m_IMUFrames[index] (:1691); sf_sens_find_closest_imu = findClosestIMUFrame(frameIdx, basedOnRGB) (:1000-1044,
 * README.txt:49): nearest in time to the frame's colour (based_on_rgb) or depth time stamp; frame128 / index may be NULL. */
int sf_sens_imu(const sf_sens* s, uint64_t index, void* frame128);
int sf_sens_find_closest_imu(const sf_sens* s, uint64_t frame, int based_on_rgb, void* frame128, uint64_t* index);

/* ------------------------------------------------------------------------------------------------
 * ScannerApp captures and the `convert` stage (scannet_amd/csrc/occipital.cpp).  Replaces
 *   uplinksimple::decode / encode    ScannerApp/depth2pgm/uplinksimple_image-codecs.h:180-249 / :253-396
 *   uplinksimple::shift2depth        ScannerApp/depth2pgm/uplinksimple_shift2depth.h:9-90
 *   Converter convertToSens          Converter/main.cpp:16-179, MetaData Converter/src/metaData.h:11-72
 * sf_occ_decode is bounds-checked (SF_ERR_FORMAT on a truncated stream; the reference only asserts).  sf_occ_encode rejects
 * values above 2047 (the code has 11 bits; the reference would corrupt its own stream).
 * ---------------------------------------------------------------------------------------------- */
int sf_occ_decode(const uint8_t* stream, uint64_t stream_bytes, uint64_t num_elements, uint16_t* out_shift);
uint64_t sf_occ_encode_bound(uint64_t num_elements);
int sf_occ_encode(const uint16_t* shift_in, uint64_t num_elements, uint8_t* out, uint64_t out_capacity, uint64_t* out_bytes);
uint16_t sf_occ_shift2depth(uint16_t shift);
int sf_occ_shift2depth_buffer(uint16_t* buf, uint64_t n, int zero_invalid /* values >= shift2depth(0xffff) -> 0, Converter/main.cpp:89-93 */);

typedef struct sf_capture_meta {     /* <base>.txt (Converter/src/metaData.h:17-60) */
  uint32_t num_color_frames, num_depth_frames, num_imu;
  uint32_t color_width, color_height, depth_width, depth_height;
  float fx_color, fy_color, mx_color, my_color, fx_depth, fy_depth, mx_depth, my_depth;
  float color_to_depth_extrinsics[16];   /* row-major; identity when absent */
  int32_t has_extrinsics;
} sf_capture_meta;
typedef struct sf_convert_stats {
  uint64_t frames, depth_frames_in_capture, imu_frames, imu_skipped, depth_stream_bytes;
  uint32_t threads;
} sf_convert_stats;
typedef struct sf_capture sf_capture;
/* colour source of sf_capture_convert: *blob / *bytes for frame `frame` (valid until the next call); SF_OK or an sf_status */
typedef int (*sf_capture_color_fn)(void* user, uint64_t frame, const uint8_t** blob, uint64_t* bytes);
int sf_capture_open(const char* any_capture_file /* <base>.txt|.depth|.imu|.h264 */, sf_capture** out);
void sf_capture_close(sf_capture* c);
int sf_capture_get_meta(const sf_capture* c, sf_capture_meta* out);
int sf_capture_decode_depth(const sf_capture* c, uint64_t frame, uint16_t* dst_mm, uint64_t* timestamp_us /*nullable*/);
/* capture -> .sens: zlib depth, depthShift 1000, "StructureSensor", identity poses, depth time stamp on both streams,
 * IMU records with a zero time stamp skipped.  color_fn NULL: frames without colour; else color_compression 0 (raw RGB) or
 * 2 (JPEG blobs passed through). */
int sf_capture_convert(const sf_capture* c, const char* out_sens, sf_capture_color_fn color_fn, void* color_user, int color_compression,
                       int threads, sf_convert_stats* stats /*nullable*/);

/* ------------------------------------------------------------------------------------------------
 * `calibrate` stage, per-frame image operations on the GPU (scannet_amd/csrc/calibrate.hip).  Replaces the frame body of
 * Calibration::calibrateScan, Calibrate/src/calibration.h:253-307: colour undistortion (:185-223), distance-dependent
 * depth correction through the 3-D look-up table (:226-250, Grid3D::GetValue src/grid3d.cpp:119-151), depth undistortion,
 * the depth-to-colour warp that the reference draws with Direct3D 11 (src/aligner.h:21-87, shaders/aligner.hlsl), "invalidate
 * depth where we have no color" and the conversion back to u16 (:286-301).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_calib_params {     /* class Calib, calibration.h:10-72; parameter-file keys :22-48 */
  uint32_t color_width, color_height, depth_width, depth_height;
  float color_intrinsic[16], depth_intrinsic[16];   /* row-major 4x4, fx [0], fy [5], mx [2], my [6] */
  float depth_extrinsic[16];                        /* depthToColorExtrinsics */
  float color_dist[5], depth_dist[5];               /* k1..k5: k1 k2 k5 radial, k3 k4 tangential (:204-209) */
} sf_calib_params;
typedef struct sf_lut {              /* Grid3D, src/grid3d.h: xres*yres*zres floats, index (z*yres + y)*xres + x */
  int32_t xres, yres, zres;
  float max_dist;
  float* data;                       /* owned: sf_lut_free */
} sf_lut;
int sf_calib_params_load(const char* path, sf_calib_params* out);   /* "name = value" parameter file */
int sf_lut_load(const char* path, sf_lut* out);                     /* Grid3D::ReadFile, grid3d.cpp:362-406 */
void sf_lut_free(sf_lut* t);
typedef struct sf_calibrator sf_calibrator;
/* lut may be NULL (no distance correction).  depth_shift = the .sens header's (1000).  No CPU fallback. */
int sf_calibrator_create(const sf_calib_params* p, const sf_lut* lut, float depth_shift, int device, sf_calibrator** out);
void sf_calibrator_destroy(sf_calibrator* c);
int sf_calibrator_max_batch(void);   /* frames per call: 16 */
/* n frames per call, host buffers: rgb = colour_width*colour_height*3 bytes (or rgb_in == NULL: depth only), depth = W*H u16.
 * rgb_out receives the undistorted colour image, depth_out the corrected, undistorted depth aligned to the colour camera. */
int sf_calibrator_run(sf_calibrator* c, int n, const uint8_t* const* rgb_in, uint8_t* const* rgb_out, const uint16_t* const* depth_in,
                      uint16_t* const* depth_out);
/* the same on device buffers; kernel_us != NULL: synchronous, returns the duration of the batch's kernels in microseconds */
int sf_calibrator_run_device(sf_calibrator* c, int n, const void* const* d_rgb_in, void* const* d_rgb_out, const void* const* d_depth_in,
                             void* const* d_depth_out, float* kernel_us);

/* The stage end to end: Calibration::calibrateScan(inSens, outSens, params, table), calibration.h:87-137 (scannet_amd/csrc/
 * calib_stage.cpp): decode -> GPU -> re-encode, header rewritten as :119-129 say, input deleted when the names differ. */
typedef struct sf_calibrate_stats {
  uint64_t frames, frames_with_colour;
  int32_t skipped_existing;   /* input missing, output present: nothing done (:89-93)              */
  int32_t already_aligned;    /* depth extrinsic was the identity: file moved, not rewritten (:112-116) */
  double seconds_total;
  uint32_t threads;
} sf_calibrate_stats;
int sf_calibrate_sens(const char* in_sens, const char* out_sens, const char* params_txt, const char* lut_path /*nullable*/, int device,
                      int threads, sf_calibrate_stats* stats /*nullable*/);
/* Baseline JPEG of an RGB image (T.81 Annex K tables, IJG quality scaling; subsample != 0: 4:2:0).  Replaces the uplink encoder
 * behind RGBDFrame::compressColor(TYPE_JPEG), sensorData.h:565-596.  *out_bytes is set even when dst is too small. */
int sf_jpeg_encode(const uint8_t* rgb, uint32_t width, uint32_t height, int quality, int subsample, uint8_t* dst, uint64_t dst_capacity,
                   uint64_t* out_bytes);
/* Baseline JPEG -> RGB as RGBDFrame::decompressColorAlloc_stb does it (sensorData.h:609-616): sf_jpeg_decode on the host (what
 * sf_sens_decode_color runs for a TYPE_JPEG frame); sf_jpeg_decode_gpu = entropy decoding on the host, IDCT + chroma upsampling +
 * YCbCr -> RGB on the GPU -- the split sf_fuse_run uses for colour frames -- returning the same bytes. */
int sf_jpeg_decode(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, uint8_t* dst_rgb);
int sf_jpeg_decode_gpu(const uint8_t* data, uint64_t bytes, uint32_t width, uint32_t height, int device, uint8_t* dst_rgb);

/* ------------------------------------------------------------------------------------------------
 * 2-D annotation filter (scannet_amd/csrc/filter2d.hip).  Replaces the CUDA kernels AnnotationTools/Filter2dAnnotations/filter.cu
 * calls from Filter2dAnnotations.cpp:326-397 -- bilateralFilterFloatMap (:210-259), resampleFloatMap (:514-573), resampleUCharMap
 * (:647-676), filterAnnotations (:1020-1078), convertInstanceToLabel (:1082-1103) -- and the per-frame sequence around them.
 * Tables: FilterData::init (Filter2dAnnotations.cpp:52-61) with 256 / 80 / 256 entries; bins >= 80 cast no vote.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_filter2d sf_filter2d;
int sf_filter2d_create(int depth_width, int depth_height, int color_width, int color_height, int device, sf_filter2d** out);
void sf_filter2d_destroy(sf_filter2d* f);
int sf_filter2d_set_tables(sf_filter2d* f, const uint8_t instance_to_idx[256], const uint8_t idx_to_instance[80], const uint16_t instance_to_label[256]);
/* one frame: depth W_d*H_d u16 (mm), rgb W_c*H_c*3, instance_in W_c*H_c u8 (rendered annotation) -> instance_out u8, label_out u16 */
int sf_filter2d_frame(sf_filter2d* f, const uint16_t* depth, const uint8_t* rgb, const uint8_t* instance_in, uint8_t* instance_out,
                      uint16_t* label_out, float* kernel_us /*nullable*/);

/* PNG images of the annotation tools (scannet_amd/csrc/png.cpp): replaces FreeImageWrapper::loadImage / saveImage as
 * Filter2dAnnotations.cpp:340-341,400-401 uses them.  Non-interlaced, bit depth 8 / 16, grey (+alpha) and RGB(A) on read; grey on
 * write.  *data is malloc'ed (sf_free); 16-bit samples are in host byte order. */
int sf_png_read(const char* path, uint32_t* width, uint32_t* height, int* channels, int* bits, void** data);
int sf_png_write_gray(const char* path, const void* data, uint32_t width, uint32_t height, int bits);
/* grey (channels = 1) or RGB (channels = 3), 8 or 16 bits: the 16-bit depth PNGs of SensReader/python/SensorData.py:78-91 (pypng there) and the
 * colour PNG that SensorData::saveToImages makes of a TYPE_RAW colour frame (sensorData.h:1432-1440) */
int sf_png_write(const char* path, const void* data, uint32_t width, uint32_t height, int channels, int bits);
void sf_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * Annotation projection (scannet_amd/csrc/project.hip, annotations.cpp).  Replaces the Direct3D 11 render + read-back + filters of
 * AnnotationTools/ProjectAnnotations/Visualizer.cpp:57-193 (render) with shaders/drawAnnotations.hlsl:9-33, and the vertex labelling
 * of Visualizer.cpp:259-377.  Parameters: ProjectAnnotations/zParametersScan.txt:6,12-15.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_project_params {
  uint32_t color_width, color_height;      /* render target = the .sens colour size (Visualizer.cpp:51)      */
  uint32_t depth_width, depth_height;
  float fx, fy;                            /* m_calibrationColor.m_intrinsic(0,0) / (1,1) (:91)              */
  float depth_min, depth_max;              /* s_depthMin 0.1, s_depthMax 15.0                                */
  float depth_dist_thresh;                 /* s_depthDistThresh 0.2                                          */
  int32_t filter_using_original_depth;     /* s_filterUsingOrigialDepthImage false                           */
} sf_project_params;
typedef struct sf_projector sf_projector;
int sf_projector_create(const sf_project_params* params, int device, sf_projector** out);
void sf_projector_destroy(sf_projector* p);
int sf_projector_max_batch(void);
/* the mesh to draw, with the (instance, label) pair its m_Colors .z / .w hold (Visualizer.cpp:275,289); host pointers, copied */
int sf_projector_set_mesh(sf_projector* p, const float* xyz, uint64_t num_vertices, const uint32_t* triangles, uint64_t num_triangles,
                          const uint8_t* vertex_instance, const uint16_t* vertex_label);
/* n <= sf_projector_max_batch() frames.  cam2world: n x 16 row-major, first element -inf = no valid transform -> empty images
 * (:63,187-192).  orig_depth: n x depth_w*depth_h u16 millimetres (NULL: no depth-consistency filter).  Outputs n x color_w*color_h.
 * zcam_out (nullable): rendered depth in metres at colour resolution (:123-141). */
int sf_projector_run(sf_projector* p, int n, const float* cam2world, const uint16_t* orig_depth, uint8_t* instance_out, uint16_t* label_out,
                     float* zcam_out /*nullable*/, float* kernel_us /*nullable*/);
/* page-locked host buffers: images passed to / from sf_projector_run in such memory are copied at PCIe speed (pageable works, slower) */
int sf_host_alloc(uint64_t bytes, void** out);
void sf_host_free(void* p);
/* computeObjectIdsAndColorsPerVertex (:259-295) for the mesh the segs.json indexes: vertex -> (instance, label), 0 = unannotated */
int sf_annotation_vertex_ids(const char* segs_json, const char* aggregation_json, const char* label_map_tsv, uint64_t num_vertices,
                             uint8_t* vertex_instance, uint16_t* vertex_label, uint32_t* num_labels /*nullable*/);
/* propagateAnnotations (:297-377): ids of the decimated mesh carried to the high-resolution one (exact 3-nearest-neighbour search) */
int sf_annotation_propagate(const float* src_xyz, uint64_t src_vertices, const uint32_t* src_tris, uint64_t src_triangles, const uint8_t* src_instance,
                            const uint16_t* src_label, const float* dst_xyz, uint64_t dst_vertices, const uint32_t* dst_tris, uint64_t dst_triangles,
                            float normal_thresh, uint8_t* dst_instance, uint16_t* dst_label);

/* ------------------------------------------------------------------------------------------------
 * Triangle meshes and the PLY surface (README.md:45-46: binary little-endian PLY, vertex float x,y,z +
 * uchar red,green,blue,alpha, face list uchar int vertex_indices).  sf_ply_read replaces tinyply as the
 * Segmentator uses it (Segmentator/segmentator.cpp:131-141, tinyply.cpp:54-108,306-360): ascii / binary
 * little / big endian, x y z as 4-byte floats, triangle lists named vertex_indices or vertex_index with a
 * 4-byte index type; anything else is SF_ERR_FORMAT (tinyply throws or misreads).  .obj files are read like
 * tiny_obj_loader's first shape (segmentator.cpp:142-174).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_mesh sf_mesh;
int sf_ply_read(const char* path, sf_mesh** out);                 /* .ply or .obj by extension */
int sf_mesh_create(const float* xyz, const uint8_t* rgba /*nullable*/, uint64_t num_vertices, const uint32_t* tris,
                   uint64_t num_faces, sf_mesh** out);
int sf_mesh_counts(const sf_mesh* m, uint64_t* num_vertices, uint64_t* num_faces);
/* any destination may be NULL: xyz 3 floats, rgba 4 bytes, tris 3 u32, keys 1 u64 (marching-cubes meshes only) */
int sf_mesh_copy(const sf_mesh* m, float* xyz, uint8_t* rgba, uint32_t* tris, uint64_t* keys);
/* marching-cubes meshes only: the key of the cube each face came from (faces are in ascending key order) -- the meshes of a
 * partitioned scan merge into the one-GPU face order by a stable sort on it (scannet_amd/partition.py) */
int sf_mesh_copy_face_keys(const sf_mesh* m, uint64_t* face_keys);
/* One scan over several GPUs (BASELINE configs[4]; no reference counterpart: DepthSensing fuses a scan on one GPU, Server/scan_processor.py:138):
 * sf_mesh_create_keyed rebuilds a marching-cubes mesh from the arrays sf_mesh_copy / sf_mesh_copy_face_keys gave (another process's part);
 * sf_mesh_merge_parts turns the parts of a partitioned scan (sf_fuser_set_stripes / sf_fuser_set_slab, ghosts exchanged before meshing) into the
 * mesh ONE fuser would have extracted, byte for byte: vertices unique by edge key in key order (of the copies of a shared edge the lowest part's),
 * faces re-indexed and -- when every part has face keys -- stably sorted by cube key; without face keys they stay part after part (slabs in
 * rank order).  The merged mesh carries its keys, so merges nest.  bin/depthsensing --ranks N is the caller. */
int sf_mesh_create_keyed(const float* xyz, const uint8_t* rgba /*nullable*/, const uint64_t* keys, uint64_t num_vertices, const uint32_t* tris,
                         const uint64_t* face_keys /*nullable*/, uint64_t num_faces, sf_mesh** out);
int sf_mesh_merge_parts(const sf_mesh* const* parts, int n_parts, sf_mesh** out);
int sf_mesh_write_ply(const sf_mesh* m, const char* path);        /* the PLY surface above */
void sf_mesh_free(sf_mesh* m);

/* ------------------------------------------------------------------------------------------------
 * Mesh cleaning: `<id>_vh.ply` -> `<id>_vh_clean.ply`.  Replaces `meshlabserver -i in.ply -o out.ply -m vc -s clean.mlx`
 * (Server/scan_processor.py:134,143) for the filter scripts the pipeline ships -- Server/tools/meshclean/clean.mlx:3-10
 * and cleanLoRes.mlx:3-10: "Merge Close Vertices" (absolute Threshold 0.0010689), "Remove Duplicate Faces", "Remove
 * Isolated pieces (wrt Face Num.)" (MinComponentSize 7500 / 1000), "Remove Unreferenced Vertex".  Semantics: the VCG
 * algorithms behind those filters, restated in scannet_amd/csrc/clean.cpp.  simplify.mlx (scan_processor.py:144-145) =
 * quadric edge collapse (below) followed by the same four filters with MinComponentSize 1000.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sf_clean_stats {
  uint64_t vertices_in, faces_in;
  uint64_t vertices_merged;        /* vertices collapsed into a lower-index vertex                  */
  uint64_t faces_degenerate;       /* faces with a repeated vertex after merging                    */
  uint64_t faces_duplicate;
  uint64_t components_in, components_removed, faces_small_component;
  uint64_t vertices_unreferenced;
  uint64_t vertices_out, faces_out;
} sf_clean_stats;

/* "Quadric Edge Collapse Decimation" (Server/tools/meshclean/simplify.mlx:3-16), the first filter of the two
 * `meshlabserver ... -s simplify.mlx` calls of the decimate stage (Server/scan_processor.py:144-145).  Semantics: VCG's
 * TriEdgeCollapseQuadric restated in scannet_amd/csrc/simplify.cpp (parity unpinned: MeshLab is not in the tree). */
typedef struct sf_simplify_params {
  uint64_t target_faces;           /* simplify.mlx:4  TargetFaceNum (used when target_perc == 0)            */
  float target_perc;               /* :5  TargetPerc 0.2: keep this fraction of the faces                     */
  float quality_thr;               /* :6  QualityThr 0.3                                                      */
  int32_t preserve_boundary;       /* :7  false (true: SF_ERR_UNSUPPORTED)                                    */
  float boundary_weight;           /* :8  BoundaryWeight 1                                                    */
  int32_t preserve_normal;         /* :9  false (true: unsupported)                                           */
  int32_t preserve_topology;       /* :10 false (true: unsupported)                                           */
  int32_t optimal_placement;       /* :11 true                                                                */
  int32_t planar_quadric;          /* :12 false                                                               */
  int32_t quality_weight;          /* :13 false (true: unsupported)                                           */
  int32_t auto_clean;              /* :14 true                                                                */
} sf_simplify_params;
typedef struct sf_simplify_stats {
  uint64_t vertices_in, faces_in, target_faces;
  uint64_t collapses, stale_popped;
  uint64_t faces_zero_area, vertices_duplicate;   /* AutoClean */
  uint64_t vertices_out, faces_out;
  float max_priority;              /* largest scaled quadric error / quality of an executed collapse */
  uint32_t rounds;                 /* sf_mesh_simplify_gpu: rounds of independent collapses (0 from the sequential filter) */
} sf_simplify_stats;
void sf_simplify_default_params(sf_simplify_params* p);   /* the values simplify.mlx ships */
int sf_mesh_simplify(const sf_mesh* in, const sf_simplify_params* p, sf_mesh** out, sf_simplify_stats* stats /*nullable*/);

/* The same filter on HIP device `device` as rounds of independent collapses (scannet_amd/csrc/simplify_gpu.hip): same quadrics, placement,
 * priority and stop rule, a different order -- different triangles with the same guarantees (face budget, flat stays flat, closed stays
 * closed, deterministic) in a fraction of the time: the sequential filter is 22 s of a scan's 24 s of host time.  Opt-in; SF_ERR_DEVICE
 * without a GPU. */
int sf_mesh_simplify_gpu(const sf_mesh* in, const sf_simplify_params* p, int device, sf_mesh** out, sf_simplify_stats* stats /*nullable*/);

typedef struct sf_clean_script {   /* what a .mlx FilterScript asks for */
  int32_t merge_close_vertices, remove_duplicate_faces, remove_small_components, remove_unreferenced;
  float merge_distance;            /* clean.mlx:4  Threshold value (absolute)   */
  uint32_t min_component_faces;    /* clean.mlx:8  MinComponentSize             */
  int32_t simplify;                /* simplify.mlx: "Quadric Edge Collapse Decimation" comes first */
  sf_simplify_params simplify_params;
  sf_simplify_stats simplify_stats;  /* filled by sf_mesh_clean_script when simplify != 0 */
  int32_t simplify_device;           /* -1 (what sf_mlx_load sets): the sequential host filter; >= 0: sf_mesh_simplify_gpu on that device */
  int32_t clean_device;              /* -1 (what sf_mlx_load sets): the host cleaning filters; >= 0: sf_mesh_clean_gpu on that device (same output) */
} sf_clean_script;

int sf_mesh_clean(const sf_mesh* in, float merge_distance, uint32_t min_component_faces, sf_mesh** out, sf_clean_stats* stats /*nullable*/);
/* The same four filters on HIP device `device` (scannet_amd/csrc/clean_gpu.hip): identical arrays and statistics -- the greedy clustering
 * resolved in rounds down the index order, duplicate faces and connected components by radix sorts and a lock-free union-find --, a
 * few tens of milliseconds for a scan-sized mesh instead of 1.7 s of one host thread.  Opt-in; SF_ERR_DEVICE without a GPU. */
int sf_mesh_clean_gpu(const sf_mesh* in, float merge_distance, uint32_t min_component_faces, int device, sf_mesh** out, sf_clean_stats* stats /*nullable*/);
int sf_mlx_load(const char* mlx_path, sf_clean_script* out);
int sf_mesh_clean_script(const sf_mesh* in, sf_clean_script* script, sf_mesh** out, sf_clean_stats* stats /*nullable*/);

/* ------------------------------------------------------------------------------------------------
 * Segmentator: Felzenszwalb-Huttenlocher graph segmentation on vertex normals.  Replaces
 * Segmentator/segmentator.cpp: segment() :123-251 (normals :185-208, edge weights :211-229,
 * segment_graph :71-92, small-segment merge :237-243), writeToJSON :253-266, main :268-289.
 * segIndices are bit-exact with the reference binary (same union-find roots): the edge sort is libstdc++
 * std::sort with the reference's weight-only comparator, all float arithmetic is un-contracted IEEE fp32.
 * ---------------------------------------------------------------------------------------------- */
int sf_segment_mesh(const float* xyz, uint64_t num_vertices, const uint32_t* tris, uint64_t num_faces, float kThresh,
                    int segMinVerts, int32_t* segIndices_out);
/* The same labels with the two data-parallel stages -- vertex normals (a lane per vertex walks its faces in face order: the running mean's order is the
 * reference's) and edge weights -- on GPU `device` (csrc/segment_gpu.hip); the sort and the sweeps stay on the host.  Fails without a device. */
int sf_segment_mesh_gpu(const float* xyz, uint64_t num_vertices, const uint32_t* tris, uint64_t num_faces, float kThresh,
                        int segMinVerts, int device, int32_t* segIndices_out);
/* Reads mesh_path (.ply/.obj), segments, writes the JSON.  out_json NULL => reference naming:
 * <mesh minus extension>.<std::to_string(kThresh)>.segs.json (segmentator.cpp:282-286).  Prints nothing. */
int sf_segment_file(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments);
/* The same with what the drop-in CLI prints (segmentator.cpp:176-180, :285): counts4 = {vertexCount, verts.size(), faceCount,
 * faces.size()}, the path written, and whether an .obj held more than one shape (only the first is used, :166-169). */
int sf_segment_file_ex(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                       uint64_t* counts4, char* out_path, uint64_t out_path_cap, int* obj_multi);
/* sf_segment_file_ex with sf_segment_mesh_gpu inside (`bin/segmentator ... --gpu [device]`, not a flag of the reference's): the same JSON. */
int sf_segment_file_gpu(const char* mesh_path, float kThresh, int segMinVerts, const char* out_json, uint64_t* num_segments,
                        uint64_t* counts4, char* out_path, uint64_t out_path_cap, int* obj_multi, int device);



#ifdef __cplusplus
}
#endif
#endif /* SCANFUSE_H */

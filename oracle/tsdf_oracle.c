/*
 * oracle/tsdf_oracle.c -- CPU restatement of the voxel-hash TSDF fusion path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker (or as the timed CPU baseline), never as the thing shipped.
 *
 * PARITY UNPINNED (for the TSDF part).  The reference repository does not contain the TSDF
 * code: Server/scan_processor.py:27-35,126,138 only names the external, un-pinned binaries
 * FriedLiver.exe (github.com/niessner/BundleFusion) and DepthSensing.exe
 * (github.com/niessner/VoxelHashing); external/mLib is an empty submodule (.gitmodules:1-3).
 * There is no golden vector or test for this boundary in the reference, so this file is the
 * executable form of the specification in SURVEY.md Appendix C / DESIGN.md section 3 and is
 * pinned only by analytic known-answer tests (tests/test_oracle_tsdf.py).
 *
 * What IS anchored on in-tree reference code:
 *   - depth u16 -> metres: d = (float)depth / depthShift, 0 => invalid
 *       SensReader/c++/src/sensorData.h:968-977, :1575-1576
 *   - unprojection K^-1 (x*d, y*d, d), world = camToWorld * cam, no y flip, pixel centre at
 *     integer coordinates: sensorData.h:1568-1579; AnnotationTools/Filter2dAnnotations/filter.cu:74-91
 *   - row-major mat4f with translation in _m03,_m13,_m23: sensorData.h:186-196
 *   - invalid pose = all -inf: sensorData.h:382; SensReader/c++/README.txt:55-57
 *   - parameter values (trunc 0.06 + 0.02 z, max dist 4 m, depth range 0.1..6 m, weight
 *     sample 1, MC thresh factor 10): Server/tools/recons/zParametersScanNet.txt:34-35,47-53
 *
 * Every float operation below is an individually rounded IEEE-754 binary32 operation
 * (compile with -ffp-contract=off, no -ffast-math); fmaf() marks the places where the
 * specification asks for a fused multiply-add.  The GPU path has to reproduce these
 * operations one for one; the parity tests compare voxels bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_BLOCK 8
#define OR_BLOCK_VOXELS 512
#define OR_MAX_DDA_ITERS 1024

typedef struct {
  int32_t width, height;        /* depth image size */
  float fx, fy, mx, my;         /* depth intrinsics */
  float depth_shift;            /* u16 -> metres divisor (1000) */
  float depth_min, depth_max;   /* s_sensorDepthMin / Max */
  float voxel_size;             /* s_SDFVoxelSize */
  float trunc_base, trunc_scale;/* s_SDFTruncation, s_SDFTruncationScale */
  float max_integration_dist;   /* s_SDFMaxIntegrationDistance */
  int32_t weight_sample;        /* s_SDFIntegrationWeightSample */
  int32_t weight_max;           /* min(s_SDFIntegrationWeightMax, 255) */
  float mc_thresh_factor;       /* s_SDFMarchingCubeThreshFactor */
  /* upstream-conformance switches (include/scanfuse.h sf_params); 0 everywhere = SURVEY App. C */
  int32_t frustum_mode;         /* 1: block centre, NDC x 0.95 (VoxelHashing isSDFBlockInCameraFrustumApprox) */
  int32_t colour_round;         /* 1: (uchar)(0.5f a + 0.5f b + 0.5f) */
  int32_t colour_first;         /* 1: a black accumulated colour, not a zero weight, marks the first observation */
  int32_t weight_mode;          /* 1: depth-dependent observation weight (VoxelHashing) */
  int32_t weight_wrap;          /* 1: weight_max as given, the sum stored modulo 256 (upstream's uchar with weightMax = 99999999) */
} or_params;

typedef struct {
  float sdf;
  uint8_t r, g, b, w;
} or_voxel;

typedef struct {
  or_params p;
  /* open-addressing map: packed block key -> block slot */
  uint64_t* keys;
  int32_t* vals;
  uint64_t map_cap; /* power of two */
  /* block storage, insertion order */
  int32_t* coords;   /* 3 per block; x == INT32_MIN marks a freed slot */
  or_voxel* voxels;  /* 512 per block */
  int64_t n_slots, cap_slots;
  int64_t n_live;
  float* depthf;     /* scratch W*H */
  int32_t* frame_list; int64_t frame_list_cap;
  int threads;
} or_volume;

#define OR_EMPTY 0xFFFFFFFFFFFFFFFFull
#define OR_TOMB  0xFFFFFFFFFFFFFFFEull

static uint64_t pack_key(int32_t x, int32_t y, int32_t z) {
  return (((uint64_t)x & 0x1FFFFFull) << 42) | (((uint64_t)y & 0x1FFFFFull) << 21) | ((uint64_t)z & 0x1FFFFFull);
}
static uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k;
}

static int32_t map_find(const or_volume* v, uint64_t key) {
  uint64_t m = v->map_cap - 1, i = mix64(key) & m;
  for (;;) {
    uint64_t k = v->keys[i];
    if (k == key) return v->vals[i];
    if (k == OR_EMPTY) return -1;
    i = (i + 1) & m;
  }
}
static void map_put_raw(uint64_t* keys, int32_t* vals, uint64_t cap, uint64_t key, int32_t val) {
  uint64_t m = cap - 1, i = mix64(key) & m;
  while (keys[i] != OR_EMPTY && keys[i] != OR_TOMB) i = (i + 1) & m;
  keys[i] = key; vals[i] = val;
}
static void map_grow(or_volume* v) {
  uint64_t ncap = v->map_cap * 2;
  uint64_t* nk = (uint64_t*)malloc(ncap * sizeof(uint64_t));
  int32_t* nv = (int32_t*)malloc(ncap * sizeof(int32_t));
  for (uint64_t i = 0; i < ncap; i++) nk[i] = OR_EMPTY;
  for (uint64_t i = 0; i < v->map_cap; i++)
    if (v->keys[i] != OR_EMPTY && v->keys[i] != OR_TOMB) map_put_raw(nk, nv, ncap, v->keys[i], v->vals[i]);
  free(v->keys); free(v->vals);
  v->keys = nk; v->vals = nv; v->map_cap = ncap;
}
static void map_erase(or_volume* v, uint64_t key) {
  uint64_t m = v->map_cap - 1, i = mix64(key) & m;
  for (;;) {
    uint64_t k = v->keys[i];
    if (k == key) { v->keys[i] = OR_TOMB; return; }
    if (k == OR_EMPTY) return;
    i = (i + 1) & m;
  }
}

static int32_t block_insert(or_volume* v, int32_t x, int32_t y, int32_t z) {
  if ((uint64_t)(v->n_slots + 1) * 2 > v->map_cap) map_grow(v);
  if (v->n_slots == v->cap_slots) {
    int64_t nc = v->cap_slots ? v->cap_slots * 2 : 1024;
    v->coords = (int32_t*)realloc(v->coords, (size_t)nc * 3 * sizeof(int32_t));
    v->voxels = (or_voxel*)realloc(v->voxels, (size_t)nc * OR_BLOCK_VOXELS * sizeof(or_voxel));
    v->cap_slots = nc;
  }
  int32_t s = (int32_t)v->n_slots++;
  v->coords[3 * s] = x; v->coords[3 * s + 1] = y; v->coords[3 * s + 2] = z;
  memset(v->voxels + (size_t)s * OR_BLOCK_VOXELS, 0, OR_BLOCK_VOXELS * sizeof(or_voxel));
  map_put_raw(v->keys, v->vals, v->map_cap, pack_key(x, y, z), s);
  v->n_live++;
  return s;
}

or_volume* or_create(const or_params* p, int threads) {
  or_volume* v = (or_volume*)calloc(1, sizeof(or_volume));
  v->p = *p;
  if (v->p.weight_max > 255 && !v->p.weight_wrap) v->p.weight_max = 255; /* uchar weight: saturate (SURVEY App. C decision) */
  v->map_cap = 1u << 16;
  v->keys = (uint64_t*)malloc(v->map_cap * sizeof(uint64_t));
  v->vals = (int32_t*)malloc(v->map_cap * sizeof(int32_t));
  for (uint64_t i = 0; i < v->map_cap; i++) v->keys[i] = OR_EMPTY;
  v->depthf = (float*)malloc((size_t)p->width * p->height * sizeof(float));
  v->threads = threads > 0 ? threads : 1;
  return v;
}
void or_destroy(or_volume* v) {
  if (!v) return;
  free(v->keys); free(v->vals); free(v->coords); free(v->voxels); free(v->depthf); free(v->frame_list); free(v);
}

/* ---- spec 3.1: depth pre-pass (sensorData.h:968-977 + zParametersScanNet.txt:34-35) ---- */
void or_depth_to_float(const or_params* p, const uint16_t* depth, float* out) {
  const int n = p->width * p->height;
  for (int i = 0; i < n; i++) {
    const uint16_t u = depth[i];
    float d = (float)u / p->depth_shift;
    if (u == 0 || d < p->depth_min || d > p->depth_max) d = -INFINITY;
    out[i] = d;
  }
}

/* ---- spec 3.2: per-frame camera constants, computed in double and rounded once ---- */
typedef struct {
  float T[12];   /* camToWorld rows 0..2 */
  float Ti[12];  /* worldToCam rows 0..2 */
  float xa[2], xc[2], xr[2]; /* x side planes: fmaf(xa, pc.x, xc*pc.z) >= -xr */
  float ya[2], yc[2], yr[2];
  float radius;  /* bounding-sphere radius of one block */
  float zfar;    /* max_dist + trunc(max_dist) */
} or_frame;

static int frame_setup(const or_params* p, const float* pose, or_frame* f) {
  if (pose[0] == -INFINITY) return 0; /* tracking lost: skip (sensorData.h:382) */
  for (int i = 0; i < 12; i++) f->T[i] = pose[i];
  const double a00 = pose[0], a01 = pose[1], a02 = pose[2], t0 = pose[3];
  const double a10 = pose[4], a11 = pose[5], a12 = pose[6], t1 = pose[7];
  const double a20 = pose[8], a21 = pose[9], a22 = pose[10], t2 = pose[11];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  double inv[9];
  inv[0] = c00 / det;                       inv[1] = (a02 * a21 - a01 * a22) / det; inv[2] = (a01 * a12 - a02 * a11) / det;
  inv[3] = c01 / det;                       inv[4] = (a00 * a22 - a02 * a20) / det; inv[5] = (a02 * a10 - a00 * a12) / det;
  inv[6] = c02 / det;                       inv[7] = (a01 * a20 - a00 * a21) / det; inv[8] = (a00 * a11 - a01 * a10) / det;
  for (int r = 0; r < 3; r++) {
    const double ti = -(inv[3 * r] * t0 + inv[3 * r + 1] * t1 + inv[3 * r + 2] * t2);
    f->Ti[4 * r] = (float)inv[3 * r]; f->Ti[4 * r + 1] = (float)inv[3 * r + 1]; f->Ti[4 * r + 2] = (float)inv[3 * r + 2];
    f->Ti[4 * r + 3] = (float)ti;
  }
  const double rad = 4.0 * sqrt(3.0) * (double)p->voxel_size;
  f->radius = (float)rad;
  /* image bounds in continuous pixel coordinates.  The integrate rule casts (int)(u + 0.5f) and THEN tests the pixel, and
   * a C cast truncates towards zero: u + 0.5 in (-1, 0) lands on pixel 0.  So a voxel is "inside" for u in (-1.5, W - 0.5), and the
   * block test must not reject what the voxel rule would update: lower planes at -1.5, upper planes at W - 0.5 / H - 0.5. */
  const double fx = p->fx, fy = p->fy, mx = p->mx, my = p->my;
  const double xl = mx + 1.5, xh = ((double)p->width - 0.5) - mx;
  const double yl = my + 1.5, yh = ((double)p->height - 0.5) - my;
  f->xa[0] = (float)fx;  f->xc[0] = (float)xl; f->xr[0] = (float)(rad * sqrt(fx * fx + xl * xl));
  f->xa[1] = (float)-fx; f->xc[1] = (float)xh; f->xr[1] = (float)(rad * sqrt(fx * fx + xh * xh));
  f->ya[0] = (float)fy;  f->yc[0] = (float)yl; f->yr[0] = (float)(rad * sqrt(fy * fy + yl * yl));
  f->ya[1] = (float)-fy; f->yc[1] = (float)yh; f->yr[1] = (float)(rad * sqrt(fy * fy + yh * yh));
  f->zfar = p->max_integration_dist + fmaf(p->trunc_scale, p->max_integration_dist, p->trunc_base);
  return 1;
}

/* ---- spec 3.3: conservative block-in-frustum test (bounding sphere vs 4 side planes + z range) ---- */
static int block_in_frustum(const or_params* p, const or_frame* f, int32_t bx, int32_t by, int32_t bz) {
  const float cx = ((float)(8 * bx) + 3.5f) * p->voxel_size;
  const float cy = ((float)(8 * by) + 3.5f) * p->voxel_size;
  const float cz = ((float)(8 * bz) + 3.5f) * p->voxel_size;
  const float* M = f->Ti;
  const float px = fmaf(M[0], cx, fmaf(M[1], cy, fmaf(M[2], cz, M[3])));
  const float py = fmaf(M[4], cx, fmaf(M[5], cy, fmaf(M[6], cz, M[7])));
  const float pz = fmaf(M[8], cx, fmaf(M[9], cy, fmaf(M[10], cz, M[11])));
  if (p->frustum_mode == 1) {
    /* VoxelHashing DepthCameraData::isInCameraFrustumApprox as remembered from the public sources (not in /root/reference): the block
     * centre through cameraToKinectProj -- x, y to [-1, 1] over (W - 1), (H - 1), z to [0, 1] over the SENSOR depth range --, scaled by
     * 0.95, inside the unit box.  Every operation individually rounded, true divisions. */
    const float zn = ((pz - p->depth_min) / (p->depth_max - p->depth_min)) * 0.95f;
    if (!(zn >= 0.0f && zn <= 1.0f) || !(pz > 0.0f)) return 0;
    const float u = (px * p->fx) / pz + p->mx;
    const float v = (py * p->fy) / pz + p->my;
    const float wm1 = (float)(p->width - 1), hm1 = (float)(p->height - 1);
    const float nx = ((2.0f * u - wm1) / wm1) * 0.95f;
    const float ny = ((hm1 - 2.0f * v) / hm1) * 0.95f;
    return nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f;
  }
  if (!(pz > -f->radius)) return 0;
  if (!(pz < f->zfar + f->radius)) return 0;
  if (!(fmaf(f->xa[0], px, f->xc[0] * pz) >= -f->xr[0])) return 0;
  if (!(fmaf(f->xa[1], px, f->xc[1] * pz) >= -f->xr[1])) return 0;
  if (!(fmaf(f->ya[0], py, f->yc[0] * pz) >= -f->yr[0])) return 0;
  if (!(fmaf(f->ya[1], py, f->yc[1] * pz) >= -f->yr[1])) return 0;
  return 1;
}

static int32_t world_to_block(float w, float voxel) {
  const float q = w / voxel;
  const int32_t vi = (int32_t)(q >= 0.0f ? q + 0.5f : q - 0.5f); /* round half away from zero */
  return vi >> 3;                                                /* floor division by 8 */
}

/* ---- spec 3.4: allocation -- per valid pixel, 3-D DDA over blocks of [d - t, d + t] ---- */
typedef struct { int32_t* v; int64_t n, cap; } i32vec;
static void vec_push3(i32vec* a, int32_t x, int32_t y, int32_t z) {
  if (a->n + 3 > a->cap) { a->cap = a->cap ? a->cap * 2 : 4096; a->v = (int32_t*)realloc(a->v, (size_t)a->cap * sizeof(int32_t)); }
  a->v[a->n++] = x; a->v[a->n++] = y; a->v[a->n++] = z;
}

static void alloc_pixel(const or_volume* v, const or_frame* f, int x, int y, i32vec* out) {
  const or_params* p = &v->p;
  const float d = v->depthf[y * p->width + x];
  if (d == -INFINITY) return;
  if (!(d < p->max_integration_dist)) return;
  const float t = fmaf(p->trunc_scale, d, p->trunc_base);
  const float lo = fminf(p->max_integration_dist, d - t);
  const float hi = fminf(p->max_integration_dist, d + t);
  if (!(lo < hi)) return;
  const float kx = ((float)x - p->mx) / p->fx;
  const float ky = ((float)y - p->my) / p->fy;
  const float* T = f->T;
  float p0[3], p1[3];
  {
    const float cx = kx * lo, cy = ky * lo, cz = lo;
    for (int r = 0; r < 3; r++) p0[r] = fmaf(T[4 * r], cx, fmaf(T[4 * r + 1], cy, fmaf(T[4 * r + 2], cz, T[4 * r + 3])));
  }
  {
    const float cx = kx * hi, cy = ky * hi, cz = hi;
    for (int r = 0; r < 3; r++) p1[r] = fmaf(T[4 * r], cx, fmaf(T[4 * r + 1], cy, fmaf(T[4 * r + 2], cz, T[4 * r + 3])));
  }
  int32_t cur[3], bound[3], step[3];
  float tmax[3], tdelta[3];
  const float bsize = 8.0f * p->voxel_size;
  for (int c = 0; c < 3; c++) {
    const float dir = p1[c] - p0[c];
    cur[c] = world_to_block(p0[c], p->voxel_size);
    const int32_t e = world_to_block(p1[c], p->voxel_size);
    step[c] = dir > 0.0f ? 1 : (dir < 0.0f ? -1 : 0);
    bound[c] = e + step[c];
    if (step[c] == 0) { tmax[c] = INFINITY; tdelta[c] = INFINITY; }
    else {
      const int32_t nb = cur[c] + (step[c] > 0 ? 1 : 0);
      const float plane = ((float)(8 * nb) - 0.5f) * p->voxel_size;
      tmax[c] = (plane - p0[c]) / dir;
      tdelta[c] = ((float)step[c] * bsize) / dir;
    }
  }
  int32_t last[3] = {INT32_MIN, 0, 0};
  for (int it = 0; it < OR_MAX_DDA_ITERS; it++) {
    if (block_in_frustum(p, f, cur[0], cur[1], cur[2])) {
      if (!(cur[0] == last[0] && cur[1] == last[1] && cur[2] == last[2])) {
        if (map_find(v, pack_key(cur[0], cur[1], cur[2])) < 0) vec_push3(out, cur[0], cur[1], cur[2]);
        last[0] = cur[0]; last[1] = cur[1]; last[2] = cur[2];
      }
    }
    int c;
    if (tmax[0] < tmax[1] && tmax[0] < tmax[2]) c = 0;
    else if (tmax[2] < tmax[1]) c = 2;
    else c = 1;
    cur[c] += step[c];
    if (cur[c] == bound[c]) break;
    tmax[c] += tdelta[c];
  }
}

static void alloc_frame(or_volume* v, const or_frame* f) {
  const or_params* p = &v->p;
  const int nt = v->threads;
  i32vec* cand = (i32vec*)calloc((size_t)nt, sizeof(i32vec));
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int y = 0; y < p->height; y++) {
#ifdef _OPENMP
    i32vec* mine = &cand[omp_get_thread_num()];
#else
    i32vec* mine = &cand[0];
#endif
    for (int x = 0; x < p->width; x++) alloc_pixel(v, f, x, y, mine);
  }
  for (int t = 0; t < nt; t++) {
    for (int64_t i = 0; i < cand[t].n; i += 3) {
      const int32_t x = cand[t].v[i], y = cand[t].v[i + 1], z = cand[t].v[i + 2];
      if (map_find(v, pack_key(x, y, z)) < 0) block_insert(v, x, y, z);
    }
    free(cand[t].v);
  }
  free(cand);
}

/* weight_mode 1: (uchar)max(weightSample * 1.5f * (1 - depthZeroOne), 1.0f), depthZeroOne over the sensor depth range; at most 255 */
static int depth_weight(const or_params* p, float d) {
  const float z01 = (d - p->depth_min) / (p->depth_max - p->depth_min);
  const float wf = fmaxf(((float)p->weight_sample * 1.5f) * (1.0f - z01), 1.0f);
  const int w = wf >= 255.0f ? 255 : (int)wf;
  return w;
}
static uint8_t blend(const or_params* p, uint8_t a, uint8_t b) {
  if (p->colour_round) return (uint8_t)(0.5f * (float)a + 0.5f * (float)b + 0.5f);   /* combineVoxel upstream: round half up */
  return (uint8_t)((a + b) / 2);                                                     /* App. C: (v.color + c) / 2, integer division */
}

/* ---- spec 3.5: integrate / deintegrate one block ---- */
static void fuse_block(or_volume* v, const or_frame* f, int32_t slot, const uint8_t* rgb, int sign) {
  const or_params* p = &v->p;
  const int32_t bx = v->coords[3 * slot], by = v->coords[3 * slot + 1], bz = v->coords[3 * slot + 2];
  or_voxel* vox = v->voxels + (size_t)slot * OR_BLOCK_VOXELS;
  const float* M = f->Ti;
  for (int lz = 0; lz < 8; lz++) for (int ly = 0; ly < 8; ly++) {
    const float wy = (float)(8 * by + ly) * p->voxel_size;
    const float wz = (float)(8 * bz + lz) * p->voxel_size;
    const float ax = fmaf(M[1], wy, fmaf(M[2], wz, M[3]));
    const float ay = fmaf(M[5], wy, fmaf(M[6], wz, M[7]));
    const float az = fmaf(M[9], wy, fmaf(M[10], wz, M[11]));
    for (int lx = 0; lx < 8; lx++) {
      const float wx = (float)(8 * bx + lx) * p->voxel_size;
      const float pcx = fmaf(M[0], wx, ax);
      const float pcy = fmaf(M[4], wx, ay);
      const float pcz = fmaf(M[8], wx, az);
      if (!(pcz > 0.0f)) continue;
      const float rz = 1.0f / pcz;
      const float uf = fmaf(pcx * p->fx, rz, p->mx) + 0.5f;
      const float vf = fmaf(pcy * p->fy, rz, p->my) + 0.5f;
      /* App. C: pixel = (int)(... + 0.5f), then "skip if outside the image" -- the cast truncates towards zero, so (-1, 0) is pixel 0
       * (the range guard keeps the cast defined for far-away projections) */
      if (!(uf > -1.0f && uf < (float)p->width && vf > -1.0f && vf < (float)p->height)) continue;
      const int ix = (int)uf, iy = (int)vf;
      const float d = v->depthf[iy * p->width + ix];
      if (d == -INFINITY) continue;
      if (!(d < p->max_integration_dist)) continue;
      float sdf = d - pcz;
      const float t = fmaf(p->trunc_scale, d, p->trunc_base);
      if (sdf <= -t) continue;
      if (sdf > t) sdf = t;
      or_voxel* q = &vox[lz * 64 + ly * 8 + lx];
      const float wo = (float)q->w;
      const int wni = p->weight_mode == 1 ? depth_weight(p, d) : p->weight_sample;
      const float wn = (float)wni;
      if (sign > 0) {
        q->sdf = fmaf(q->sdf, wo, sdf * wn) / (wo + wn);
        if (rgb) {
          const uint8_t* c = rgb + 3 * (size_t)(iy * p->width + ix);
          const int first = p->colour_first ? (q->r == 0 && q->g == 0 && q->b == 0) : (q->w == 0);
          if (first) { q->r = c[0]; q->g = c[1]; q->b = c[2]; }
          else { q->r = blend(p, q->r, c[0]); q->g = blend(p, q->g, c[1]); q->b = blend(p, q->b, c[2]); }
        }
        int w = (int)q->w + wni;
        if (w > p->weight_max) w = p->weight_max;
        q->w = (uint8_t)w;
      } else {
        const int w = (int)q->w - wni;
        if (w <= 0) { q->sdf = 0.0f; q->r = q->g = q->b = 0; q->w = 0; }
        else { q->sdf = fmaf(q->sdf, wo, -(sdf * wn)) / (wo - wn); q->w = (uint8_t)w; }
      }
    }
  }
}

/* one frame: pre-pass, (alloc), compactify, fuse.  Returns #blocks in the frustum list, -1 if the pose is invalid. */
static int64_t run_frame(or_volume* v, const uint16_t* depth, const uint8_t* rgb, const float* pose, int sign) {
  or_frame f;
  if (!frame_setup(&v->p, pose, &f)) return -1;
  or_depth_to_float(&v->p, depth, v->depthf);
  if (sign > 0) alloc_frame(v, &f);
  if (v->frame_list_cap < v->n_slots) {
    v->frame_list_cap = v->n_slots * 2; v->frame_list = (int32_t*)realloc(v->frame_list, (size_t)v->frame_list_cap * sizeof(int32_t));
  }
  int64_t n = 0;
  for (int64_t s = 0; s < v->n_slots; s++) {
    if (v->coords[3 * s] == INT32_MIN) continue;
    if (block_in_frustum(&v->p, &f, v->coords[3 * s], v->coords[3 * s + 1], v->coords[3 * s + 2])) v->frame_list[n++] = (int32_t)s;
  }
#pragma omp parallel for num_threads(v->threads) schedule(dynamic, 64)
  for (int64_t i = 0; i < n; i++) fuse_block(v, &f, v->frame_list[i], rgb, sign);
  return n;
}

int64_t or_integrate(or_volume* v, const uint16_t* depth, const uint8_t* rgb, const float* pose) { return run_frame(v, depth, rgb, pose, +1); }
int64_t or_deintegrate(or_volume* v, const uint16_t* depth, const uint8_t* rgb, const float* pose) { return run_frame(v, depth, rgb, pose, -1); }

/* ---- spec 3.6: garbage collection ---- */
int64_t or_garbage_collect(or_volume* v) {
  const or_params* p = &v->p;
  const float thr = fmaf(p->trunc_scale, p->depth_max, p->trunc_base);
  int64_t freed = 0;
  for (int64_t s = 0; s < v->n_slots; s++) {
    if (v->coords[3 * s] == INT32_MIN) continue;
    const or_voxel* vox = v->voxels + (size_t)s * OR_BLOCK_VOXELS;
    float min_abs = INFINITY; int max_w = 0;
    for (int i = 0; i < OR_BLOCK_VOXELS; i++) {
      if (vox[i].w > 0) { const float a = fabsf(vox[i].sdf); if (a < min_abs) min_abs = a; }
      if (vox[i].w > max_w) max_w = vox[i].w;
    }
    if (max_w == 0 || min_abs >= thr) {
      map_erase(v, pack_key(v->coords[3 * s], v->coords[3 * s + 1], v->coords[3 * s + 2]));
      v->coords[3 * s] = INT32_MIN; v->n_live--; freed++;
    }
  }
  return freed;
}

int64_t or_num_blocks(const or_volume* v) { return v->n_live; }

/* export live blocks in slot order: coords [n][3] int32, voxels [n][512] 8-byte records */
int64_t or_export(const or_volume* v, int32_t* coords, void* voxels) {
  int64_t n = 0;
  for (int64_t s = 0; s < v->n_slots; s++) {
    if (v->coords[3 * s] == INT32_MIN) continue;
    if (coords) { coords[3 * n] = v->coords[3 * s]; coords[3 * n + 1] = v->coords[3 * s + 1]; coords[3 * n + 2] = v->coords[3 * s + 2]; }
    if (voxels) memcpy((uint8_t*)voxels + (size_t)n * 4096, v->voxels + (size_t)s * OR_BLOCK_VOXELS, 4096);
    n++;
  }
  return n;
}

/* lookups used by the marching-cubes oracle (oracle/mc_oracle.c) */
const or_params* or_get_params(const or_volume* v) { return &v->p; }
const void* or_find_block(const or_volume* v, int32_t x, int32_t y, int32_t z) {
  const int32_t s = map_find(v, pack_key(x, y, z));
  return s < 0 ? NULL : (const void*)(v->voxels + (size_t)s * OR_BLOCK_VOXELS);
}
int64_t or_num_slots(const or_volume* v) { return v->n_slots; }
int or_slot_coords(const or_volume* v, int64_t s, int32_t* xyz) {
  if (v->coords[3 * s] == INT32_MIN) return 0;
  xyz[0] = v->coords[3 * s]; xyz[1] = v->coords[3 * s + 1]; xyz[2] = v->coords[3 * s + 2];
  return 1;
}

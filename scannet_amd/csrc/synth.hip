// synth.hip -- synthetic RGB-D stream generator, device side.
//
// Benchmark input only: renders the analytic depth of the config-2 box room straight into HBM so that
// bench.py can hold a 5 578-frame 640x480 stream resident without touching the host.  Same conventions as
// scannet_amd/synth.py and the reference codec: u16 millimetres (depthShift 1000, sensorData.h:895), camera
// +z forward / x right / y down, pixel centres at integers (sensorData.h:1568-1579).  Not bit-identical
// to the numpy renderer (different summation order) and not meant to be: parity tests feed both sides the
// same u16 frames.
#include <hip/hip_runtime.h>

#include <cmath>

#include "common.h"

namespace {

struct SynthFrame {
  double R[9];
  double o[3];
};

// scene 1: axis-aligned boxes inside the room (furniture along the walls, a table island inside the walk, shelves, lamps)
constexpr int SYNTH_BOXES = 48;
struct SynthBoxes {
  int n;
  float lo[SYNTH_BOXES][3], hi[SYNTH_BOXES][3];   // room coordinates, metres (float is plenty for a generator)
};

// lowbias32: a full-avalanche 32-bit mix.  noise mode 2 takes the 3 noise LSBs (and the speckle holes of scene 1) of pixel i of frame f
// from hash(f * W * H + i): independent per pixel and frame -- the entropy of a real sensor's low bits.  Mode 1 (round 1 / 2) took ONE
// LCG step from consecutive seeds and used bits 24..26, which change every ~10 pixels: a ramp that zlib compresses 5x better than
// real depth ("Synthetic noise is a ramp").
__host__ __device__ inline uint32_t synth_hash(uint32_t s) {
  s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16;
  return s;
}

__global__ __launch_bounds__(256) void k_synth_room(uint16_t* __restrict__ out, int W, int H, double fx, double fy, double mx, double my,
                                                    double rx, double ry, double rz, SynthFrame fr, int noise, unsigned long long frame,
                                                    SynthBoxes boxes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int x = i % W, y = i / W;
  const double cx = ((double)x - mx) / fx, cy = ((double)y - my) / fy;
  const double room[3] = {rx, ry, rz};
  double t = INFINITY;
  double dir[3], inv[3];
  int axis = 0;   // axis of the surface hit (its normal): for the grazing-angle holes of scene 1
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double d = fr.R[3 * a] * cx + fr.R[3 * a + 1] * cy + fr.R[3 * a + 2];
    dir[a] = d;
    inv[a] = 1.0 / d;   // one division per axis: the 48 boxes below multiply (144 fp64 divisions per pixel made the generator slower than the fusion)
    double ta = INFINITY;
    if (d > 0) ta = (room[a] - fr.o[a]) / d;
    else if (d < 0) ta = (0.0 - fr.o[a]) / d;
    if (ta < t) { t = ta; axis = a; }
  }
  // boxes: slab test from outside; the entry face gives the normal.  Which box is hit is decided with the reciprocals; the depth of the hit is
  // then taken with the division the host renderer uses, so the two agree to the bit but for rays that graze an edge.
  int hit_box = -1;
  for (int b = 0; b < boxes.n; b++) {
    double tn = 0.0, tf = INFINITY;
    int an = 0;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double lo = boxes.lo[b][a], hi = boxes.hi[b][a];
      if (dir[a] == 0.0) { miss = miss || fr.o[a] < lo || fr.o[a] > hi; continue; }
      double t0 = (lo - fr.o[a]) * inv[a], t1 = (hi - fr.o[a]) * inv[a];
      if (t0 > t1) { const double q = t0; t0 = t1; t1 = q; }
      if (t0 > tn) { tn = t0; an = a; }
      tf = fmin(tf, t1);
    }
    if (!miss && tn < tf && tn > 0.0 && tn < t) { t = tn; axis = an; hit_box = b; }
  }
  if (hit_box >= 0) {
    const double face = dir[axis] > 0.0 ? (double)boxes.lo[hit_box][axis] : (double)boxes.hi[hit_box][axis];
    t = (face - fr.o[axis]) / dir[axis];
  }
  double mm = rint(t * 1000.0);
  long long v = (mm < 65535.0) ? (long long)mm : 0;  // also catches inf / nan
  if (noise == 1 && v > 0) {
    unsigned long long s = (frame * (unsigned long long)(W * H) + (unsigned long long)i) & 0xFFFFFFFFull;
    s = (s * 1664525ull + 1013904223ull) & 0xFFFFFFFFull;
    v += (long long)((s >> 24) & 7ull);
  } else if (noise == 2 && v > 0) {
    const uint32_t h = synth_hash((uint32_t)(frame * (unsigned long long)(W * H) + (unsigned long long)i));
    v += (long long)(h >> 29);
    if (boxes.n > 0) {
      // what a structured-light sensor does not return: surfaces seen at a grazing angle, and speckle drop-outs (0.4 % of the pixels)
      const double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
      if (fabs(dir[axis]) < 0.12 * len || (h & 0xFFu) == 0u) v = 0;
    }
  }
  out[i] = (uint16_t)v;
}

}  // namespace

// camToWorld of frame i of the closed rounded-rectangle walk (mirrors scannet_amd/synth.py:trajectory_pose)
static void trajectory_pose(uint64_t i, uint64_t n_frames, const double room[3], double pose[16]) {
  const double inset = 1.0, height = 1.5, corner = 0.5, pi = 3.14159265358979323846;
  const double lx = room[0] - 2 * inset, ly = room[1] - 2 * inset;
  const double per = 2 * (lx + ly);
  const double s = (double)(i % n_frames) / (double)n_frames * per;
  const double legs[4][2] = {{lx, 0.0}, {ly, pi / 2}, {lx, pi}, {ly, 3 * pi / 2}};
  double x = inset, y = inset, acc = 0.0, yaw = 0.0;
  for (int k = 0; k < 4; k++) {
    const double ln = legs[k][0], hd = legs[k][1];
    if (s < acc + ln || k == 3) {
      const double t = s - acc;
      x += std::cos(hd) * t;
      y += std::sin(hd) * t;
      const double nxt = hd + pi / 2, prv = hd - pi / 2;
      if (t > ln - corner) yaw = hd + (nxt - hd) * 0.5 * (t - (ln - corner)) / corner;
      else if (t < corner) yaw = hd + (prv - hd) * 0.5 * (corner - t) / corner;
      else yaw = hd;
      break;
    }
    x += std::cos(hd) * ln;
    y += std::sin(hd) * ln;
    acc += ln;
  }
  const double c = std::cos(yaw), sn = std::sin(yaw);
  const double m[16] = {sn, 0, c, x, -c, 0, sn, y, 0, -1, 0, height, 0, 0, 0, 1};
  for (int k = 0; k < 16; k++) pose[k] = m[k];
}

// The furniture of scene 1, from an integer LCG seeded with `seed` (scannet_amd/synth.py clutter_boxes draws the same numbers): everything
// keeps 0.35 m clear of the walk (1 m inset from the walls, camera at 1.5 m).
//   28 floor-standing boxes flush to the walls (7 per wall): depth 0.25..0.6 m, width 0.4..1.6 m, height 0.4..2.0 m
//    4 table-like boxes on the island inside the walk: height 0.4..1.2 m
//    8 shelves on the walls (2 per wall) at 1.2..2.2 m: depth 0.05..0.3 m
//    8 lamps under the ceiling: 0.3..0.8 m wide, 0.15..0.45 m tall
static void clutter_boxes(const double room[3], uint32_t seed, SynthBoxes& out) {
  uint32_t s = seed * 2654435761u + 12345u;
  auto rnd = [&]() {   // [0, 1) in steps of 2^-24
    s = s * 1664525u + 1013904223u;
    s ^= s >> 15;
    return (double)(s >> 8) / 16777216.0;
  };
  int k = 0;
  auto put = [&](double x0, double y0, double z0, double x1, double y1, double z1) {
    out.lo[k][0] = (float)x0; out.lo[k][1] = (float)y0; out.lo[k][2] = (float)z0;
    out.hi[k][0] = (float)x1; out.hi[k][1] = (float)y1; out.hi[k][2] = (float)z1;
    k++;
  };
  const double rx = room[0], ry = room[1], rz = room[2];
  for (int wall = 0; wall < 4; wall++) {          // 0: y = 0, 1: x = rx, 2: y = ry, 3: x = 0
    const double len = (wall & 1) ? ry : rx;
    for (int j = 0; j < 7; j++) {
      const double depth = 0.25 + 0.35 * rnd(), width = 0.4 + 1.2 * rnd(), height = 0.4 + 1.6 * rnd() * rnd();
      const double a0 = (len - width) * rnd();
      if (wall == 0) put(a0, 0, 0, a0 + width, depth, height);
      else if (wall == 1) put(rx - depth, a0, 0, rx, a0 + width, height);
      else if (wall == 2) put(a0, ry - depth, 0, a0 + width, ry, height);
      else put(0, a0, 0, depth, a0 + width, height);
    }
  }
  const double ix0 = 1.4, ix1 = rx - 1.4, iy0 = 1.4, iy1 = ry - 1.4;   // the island inside the walk
  for (int j = 0; j < 4; j++) {
    const double w = (0.2 + 0.6 * rnd()) * (ix1 - ix0), d = (0.3 + 0.7 * rnd()) * (iy1 - iy0), h = 0.4 + 0.8 * rnd();
    const double x0 = ix0 + (ix1 - ix0 - w) * rnd(), y0 = iy0 + (iy1 - iy0 - d) * rnd();
    put(x0, y0, 0, x0 + w, y0 + d, h);
  }
  for (int jj = 0; jj < 8; jj++) {                // shelves, two per wall
    const int j = jj & 3;
    const double len = (j & 1) ? ry : rx;
    const double depth = 0.05 + 0.25 * rnd(), width = 0.5 + 1.0 * rnd(), z0 = 1.2 + 0.7 * rnd(), th = 0.05 + 0.25 * rnd();
    const double a0 = (len - width) * rnd();
    if (j == 0) put(a0, 0, z0, a0 + width, depth, z0 + th);
    else if (j == 1) put(rx - depth, a0, z0, rx, a0 + width, z0 + th);
    else if (j == 2) put(a0, ry - depth, z0, a0 + width, ry, z0 + th);
    else put(0, a0, z0, depth, a0 + width, z0 + th);
  }
  for (int j = 0; j < 8; j++) {                   // lamps
    const double w = 0.3 + 0.5 * rnd(), h = 0.15 + 0.3 * rnd();
    const double x0 = (rx - w) * rnd(), y0 = (ry - w) * rnd();
    put(x0, y0, rz - h, x0 + w, y0 + w, rz);
  }
  out.n = k;
}

SF_API int sf_synth_clutter_boxes(const double room_m[3], uint32_t seed, float* lo_out, float* hi_out, int* n_out) {
  if (!room_m || !lo_out || !hi_out || !n_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  SynthBoxes b;
  clutter_boxes(room_m, seed, b);
  for (int i = 0; i < b.n; i++)
    for (int a = 0; a < 3; a++) { lo_out[3 * i + a] = b.lo[i][a]; hi_out[3 * i + a] = b.hi[i][a]; }
  *n_out = b.n;
  return SF_OK;
}

static int synth_scene(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames, int width, int height, int noise,
                       int scene, uint32_t seed, const double room_m[3], const double origin_m[3], float* poses_out);

// frames [first_frame, first_frame + n) of the `total_frames`-frame walk through a box room of `room` metres whose corner sits at `origin`
SF_API int sf_synth_scan_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                                int width, int height, int noise, const double room_m[3], const double origin_m[3], float* poses_out) {
  return synth_scene(d_depth, frame_stride_bytes, first_frame, n, total_frames, width, height, noise, 0, 0u, room_m, origin_m, poses_out);
}
// the same with the scene chosen: 0 = the empty box room, 1 = the room furnished by clutter_boxes(seed) with sensor holes (noise 2)
SF_API int sf_synth_scene_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                                 int width, int height, int noise, int scene, uint32_t seed, const double room_m[3], const double origin_m[3],
                                 float* poses_out) {
  if (scene < 0 || scene > 1 || noise < 0 || noise > 2) return sf::fail(SF_ERR_INVALID_ARG, "scene %d / noise %d", scene, noise);
  return synth_scene(d_depth, frame_stride_bytes, first_frame, n, total_frames, width, height, noise, scene, seed, room_m, origin_m, poses_out);
}

static int synth_scene(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames, int width, int height, int noise,
                       int scene, uint32_t seed, const double room_m[3], const double origin_m[3], float* poses_out) {
  if (!d_depth || !poses_out || !room_m || width <= 0 || height <= 0 || total_frames == 0) return sf::fail(SF_ERR_INVALID_ARG, "bad argument");
  if (!(room_m[0] > 2.5 && room_m[1] > 2.5 && room_m[2] > 1.6)) return sf::fail(SF_ERR_INVALID_ARG, "room too small for the walk (1 m inset, camera at 1.5 m)");
  const double room[3] = {room_m[0], room_m[1], room_m[2]};
  const double org[3] = {origin_m ? origin_m[0] : 0.0, origin_m ? origin_m[1] : 0.0, origin_m ? origin_m[2] : 0.0};
  const double fx = 577.87 * width / 640.0, mx = (width - 1) / 2.0, my = (height - 1) / 2.0;
  SynthBoxes boxes;
  boxes.n = 0;
  if (scene == 1) clutter_boxes(room, seed, boxes);
  for (uint64_t k = 0; k < n; k++) {
    double pose[16];
    trajectory_pose(first_frame + k, total_frames, room, pose);
    float* pf = poses_out + 16 * k;
    for (int q = 0; q < 16; q++) pf[q] = (float)pose[q];
    SynthFrame fr;
    // render from the float-rounded pose (room coordinates); the pose handed to the fuser carries the room's place in the world
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) fr.R[3 * r + c] = (double)pf[4 * r + c]; fr.o[r] = (double)pf[4 * r + 3]; }
    for (int r = 0; r < 3; r++) pf[4 * r + 3] = (float)((double)pf[4 * r + 3] + org[r]);
    uint16_t* out = (uint16_t*)((uint8_t*)d_depth + k * frame_stride_bytes);
    hipLaunchKernelGGL(k_synth_room, dim3((width * height + 255) / 256), dim3(256), 0, 0, out, width, height, fx, fx, mx, my, room[0],
                       room[1], room[2], fr, noise, (unsigned long long)(first_frame + k), boxes);
  }
  SF_HIP_CHECK(hipGetLastError());
  SF_HIP_CHECK(hipDeviceSynchronize());
  return SF_OK;
}

SF_API int sf_synth_room_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                                int width, int height, int noise, float* poses_out) {
  const double room[3] = {6.0, 4.0, 3.0};
  return sf_synth_scan_device(d_depth, frame_stride_bytes, first_frame, n, total_frames, width, height, noise, room, nullptr, poses_out);
}

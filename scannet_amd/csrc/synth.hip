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

__global__ __launch_bounds__(256) void k_synth_room(uint16_t* __restrict__ out, int W, int H, double fx, double fy, double mx, double my,
                                                    double rx, double ry, double rz, SynthFrame fr, int noise, unsigned long long frame) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int x = i % W, y = i / W;
  const double cx = ((double)x - mx) / fx, cy = ((double)y - my) / fy;
  const double room[3] = {rx, ry, rz};
  double t = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double d = fr.R[3 * a] * cx + fr.R[3 * a + 1] * cy + fr.R[3 * a + 2];
    if (d > 0) t = fmin(t, (room[a] - fr.o[a]) / d);
    else if (d < 0) t = fmin(t, (0.0 - fr.o[a]) / d);
  }
  double mm = rint(t * 1000.0);
  long long v = (mm < 65535.0) ? (long long)mm : 0;  // also catches inf / nan
  if (noise && v > 0) {
    unsigned long long s = (frame * (unsigned long long)(W * H) + (unsigned long long)i) & 0xFFFFFFFFull;
    s = (s * 1664525ull + 1013904223ull) & 0xFFFFFFFFull;
    v += (long long)((s >> 24) & 7ull);
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

// frames [first_frame, first_frame + n) of the `total_frames`-frame walk through a box room of `room` metres whose corner sits at `origin`
SF_API int sf_synth_scan_device(void* d_depth, uint64_t frame_stride_bytes, uint64_t first_frame, uint64_t n, uint64_t total_frames,
                                int width, int height, int noise, const double room_m[3], const double origin_m[3], float* poses_out) {
  if (!d_depth || !poses_out || !room_m || width <= 0 || height <= 0 || total_frames == 0) return sf::fail(SF_ERR_INVALID_ARG, "bad argument");
  if (!(room_m[0] > 2.5 && room_m[1] > 2.5 && room_m[2] > 1.6)) return sf::fail(SF_ERR_INVALID_ARG, "room too small for the walk (1 m inset, camera at 1.5 m)");
  const double room[3] = {room_m[0], room_m[1], room_m[2]};
  const double org[3] = {origin_m ? origin_m[0] : 0.0, origin_m ? origin_m[1] : 0.0, origin_m ? origin_m[2] : 0.0};
  const double fx = 577.87 * width / 640.0, mx = (width - 1) / 2.0, my = (height - 1) / 2.0;
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
                       room[1], room[2], fr, noise, (unsigned long long)(first_frame + k));
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

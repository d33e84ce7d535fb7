// calibrate.hip -- the `calibrate` stage's per-frame image operations on gfx950.
//
// Replaces the frame body of Calibration::calibrateScan (Calibrate/src/calibration.h:253-307), which runs on the host under
// OpenMP with the depth-to-colour warp as a Direct3D 11 draw call inside an `omp critical` (calibration.h:140-148,
// src/aligner.h:21-87, shaders/aligner.hlsl):
//   k_undistort_rgb     Calibration::undistort on the colour image            calibration.h:185-223 (:264-268)
//   k_depth_prepare     undistortDistance (3-D look-up table, trilinear)      calibration.h:226-250, grid3d.cpp:119-151
//                       + u16 -> metres + Calibration::undistort on depth     calibration.h:275-281, fused: the table correction
//                       is pointwise on the SOURCE pixel, so the undistorted image gathers corrected source pixels directly
//                       + the mesh vertex of the pixel (aligner.hlsl:66-103), once per pixel instead of once per quad corner
//   k_raster_quads      Aligner::depthToColor: the depth image as a quad mesh projected into the colour camera and drawn
//                       with a depth test                                     aligner.hlsl:52-166, aligner.h:24-81
//   k_finalize          depth buffer -> metres, "invalidate depth where we have no color", -> u16   calibration.h:286-301
// B frames per launch (blockIdx.z): a 640x480 frame is ~10 MB of traffic, a few microseconds of HBM time, so a stage that
// launched per frame would be launch-bound.  Everything is HBM / L2-bound integer and fp32 work: no LDS staging (no reuse),
// coalesced 4-pixel (12-byte) colour stores, the depth buffer lives in HBM and is hit with 32-bit atomicMin (non-negative
// floats order like their bit patterns).
//
// The arithmetic is the statement-by-statement restatement the CPU checker holds (see its header for what is and is not
// pinned: the reference's warp is whatever its GPU's rasteriser produced; here it is a software rasteriser with the
// Direct3D 11 rules -- pixel centres at +0.5, 1/256-pixel vertex snapping, top-left fill rule, LESS test, buffer cleared to 1).
// Built with -ffp-contract=off; fp32 division is IEEE.  mLib's math::round is taken as floor(x + 0.5).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"
#include "jpeg_idct.h"

namespace {

constexpr int CALIB_MAX_BATCH = 16;
constexpr float DEPTH_WORLD_MIN = 0.1f;   // aligner.hlsl:3-4
constexpr float DEPTH_WORLD_MAX = 10.0f;

struct CalibK {
  int w, h, cw, ch;
  float Kd[16], Kc[16], Kinv[16], E[16];
  float cdist[5], ddist[5];
  float shift;
  int lx, ly, lz;       // look-up table resolution (0: no table)
  float lmax;
};

struct CalibBatch {
  const uint8_t* rgb_in[CALIB_MAX_BATCH];
  uint8_t* rgb_out[CALIB_MAX_BATCH];
  const uint16_t* depth_in[CALIB_MAX_BATCH];
  uint16_t* depth_out[CALIB_MAX_BATCH];
};

__device__ inline int round_i(float x) { return (int)floorf(x + 0.5f); }

// calibration.h:192-212
__device__ inline void sample_loc(const float* K, const float* c, unsigned x, unsigned y, int& sx, int& sy) {
  const float nx = ((float)x - K[2]) / K[0];
  const float ny = ((float)y - K[6]) / K[5];
  const float r2 = nx * nx + ny * ny;
  const float radial = 1.0f + r2 * c[0] + r2 * r2 * c[1] + r2 * r2 * r2 * c[4];
  float lx = nx * radial, ly = ny * radial;
  lx += 2.0f * c[2] * nx * ny + c[3] * (r2 + 2.0f * nx * nx);
  ly += c[2] * (r2 + 2.0f * ny * ny) + 2.0f * c[3] * nx * ny;
  lx = lx * K[0] + K[2];
  ly = ly * K[5] + K[6];
  sx = round_i(lx);
  sy = round_i(ly);
}

// four output pixels (12 bytes) per lane: three dword stores, fully coalesced across the wave
__global__ __launch_bounds__(256) void k_undistort_rgb(CalibBatch B, CalibK P) {
  const uint8_t* __restrict__ src = B.rgb_in[blockIdx.z];
  uint8_t* __restrict__ dst = B.rgb_out[blockIdx.z];
  const size_t n = (size_t)P.cw * P.ch;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t i0 = 4 * g;
  if (i0 >= n) return;
  uint8_t px[12];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const size_t i = i0 + k;
    uint8_t r = 0, gch = 0, b = 0;
    if (i < n) {
      const unsigned x = (unsigned)(i % (size_t)P.cw), y = (unsigned)(i / (size_t)P.cw);
      int sx, sy;
      sample_loc(P.Kc, P.cdist, x, y, sx, sy);
      if (sx >= 0 && sx < P.cw && sy >= 0 && sy < P.ch) {
        const uint8_t* s = src + 3 * ((size_t)sy * P.cw + sx);
        r = s[0]; gch = s[1]; b = s[2];
      }
    }
    px[3 * k] = r; px[3 * k + 1] = gch; px[3 * k + 2] = b;
  }
  if (i0 + 4 <= n && ((uintptr_t)dst & 3) == 0) {
    uint32_t* o = reinterpret_cast<uint32_t*>(dst + 3 * i0);
    o[0] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)px[3] << 24);
    o[1] = (uint32_t)px[4] | ((uint32_t)px[5] << 8) | ((uint32_t)px[6] << 16) | ((uint32_t)px[7] << 24);
    o[2] = (uint32_t)px[8] | ((uint32_t)px[9] << 8) | ((uint32_t)px[10] << 16) | ((uint32_t)px[11] << 24);
  } else {
    for (int k = 0; k < 12 && 3 * i0 + k < 3 * n; k++) dst[3 * i0 + k] = px[k];
  }
}

// grid3d.cpp:119-151
__device__ inline float lut_value(const float* __restrict__ t, int xr, int yr, int zr, float x, float y, float z) {
  const int x1 = (int)x, y1 = (int)y, z1 = (int)z;
  int x2 = x1 + 1, y2 = y1 + 1, z2 = z1 + 1;
  if (x2 >= xr) x2 = x1;
  if (y2 >= yr) y2 = y1;
  if (z2 >= zr) z2 = z1;
  const float dx = x - (float)x1, dy = y - (float)y1, dz = z - (float)z1;
  auto G = [&](int a, int b, int c) { return t[((size_t)c * yr + b) * xr + a]; };
  float v = 0.0f;
  v += G(x1, y1, z1) * (1.0f - dx) * (1.0f - dy) * (1.0f - dz);
  v += G(x1, y1, z2) * (1.0f - dx) * (1.0f - dy) * dz;
  v += G(x1, y2, z1) * (1.0f - dx) * dy * (1.0f - dz);
  v += G(x1, y2, z2) * (1.0f - dx) * dy * dz;
  v += G(x2, y1, z1) * dx * (1.0f - dy) * (1.0f - dz);
  v += G(x2, y1, z2) * dx * (1.0f - dy) * dz;
  v += G(x2, y2, z1) * dx * dy * (1.0f - dz);
  v += G(x2, y2, z2) * dx * dy * dz;
  return v;
}

struct Vert { float px, py, z; bool ok; };
__device__ inline Vert quad_vertex(float d, const CalibK& P, int x, int y);

// per OUTPUT depth pixel: where it samples the distorted image, that source pixel's table-corrected depth in metres;
// also clears the depth buffer of the draw
__global__ __launch_bounds__(256) void k_depth_prepare(CalibBatch B, CalibK P, const float* __restrict__ lut, float* __restrict__ und_all,
                                                       uint32_t* __restrict__ zbuf_all, float4* __restrict__ vert_all) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = P.w * P.h;
  if (i >= n) return;
  const uint16_t* __restrict__ raw = B.depth_in[blockIdx.z];
  float* und = und_all + (size_t)blockIdx.z * n;
  uint32_t* zbuf = zbuf_all + (size_t)blockIdx.z * n;
  const unsigned x = (unsigned)(i % P.w), y = (unsigned)(i / P.w);
  int sx, sy;
  sample_loc(P.Kd, P.ddist, x, y, sx, sy);
  float v = 0.0f;  // invalid value of the depth image (calibration.h:276)
  if (sx >= 0 && sx < P.w && sy >= 0 && sy < P.h) {
    uint16_t u = raw[(size_t)sy * P.w + sx];
    if (P.lx > 0) {  // calibration.h:226-250 on source pixel (sx, sy)
      const float xbin = (float)(P.w / P.lx), ybin = (float)(P.h / P.ly);
      const float zbin = (float)P.lz / P.lmax;
      const float depth = (float)u / P.shift;
      const float zidx = fminf(depth * zbin, (float)P.lz - 1.0f);
      const float multiplier = 1.0f / lut_value(lut, P.lx, P.ly, P.lz, (float)sx / xbin, (float)sy / ybin, zidx);
      const float nd = depth * multiplier * P.shift;
      u = !(nd >= 0.0f) ? (uint16_t)0 : (nd >= 65535.0f ? (uint16_t)65535 : (uint16_t)nd);
    }
    v = (float)u / P.shift;
  }
  und[i] = v;
  zbuf[i] = __float_as_uint(1.0f);
  // the mesh vertex of this pixel (aligner.hlsl:66-103), computed once here instead of by each of the four quads that share it
  const Vert q = quad_vertex(v, P, (int)x, (int)y);
  vert_all[(size_t)blockIdx.z * n + i] = make_float4(q.px, q.py, q.z, q.ok ? 1.0f : 0.0f);
}

// aligner.hlsl:52-103 + viewport transform
__device__ inline Vert quad_vertex(float d, const CalibK& P, int x, int y) {
  const float ax = (float)x * d, ay = (float)y * d;
  const float* Ki = P.Kinv;
  const float* E = P.E;
  const float* Kc = P.Kc;
  const float cx = Ki[0] * ax + Ki[1] * ay + Ki[2] * d + Ki[3] * d;
  const float cy = Ki[4] * ax + Ki[5] * ay + Ki[6] * d + Ki[7] * d;
  const float cz = Ki[8] * ax + Ki[9] * ay + Ki[10] * d + Ki[11] * d;
  float wx = E[0] * cx + E[1] * cy + E[2] * cz + E[3];
  float wy = E[4] * cx + E[5] * cy + E[6] * cz + E[7];
  float wz = E[8] * cx + E[9] * cy + E[10] * cz + E[11];
  const float ww = E[12] * cx + E[13] * cy + E[14] * cz + E[15];
  wx /= ww; wy /= ww; wz /= ww;
  const float qx = Kc[0] * wx + Kc[1] * wy + Kc[2] * wz + Kc[3];
  const float qy = Kc[4] * wx + Kc[5] * wy + Kc[6] * wz + Kc[7];
  const float qz = Kc[8] * wx + Kc[9] * wy + Kc[10] * wz + Kc[11];
  const float ux = qx / qz, uy = qy / qz;
  const float fx = (ux / (float)(P.cw - 1)) * 2.0f - 1.0f;
  const float fy = 1.0f - (uy / ((float)P.ch - 1.0f)) * 2.0f;
  const float fz = (qz - DEPTH_WORLD_MIN) / (DEPTH_WORLD_MAX - DEPTH_WORLD_MIN);
  Vert v;
  v.ok = !(fx < -1.0f || fx > 1.0f) && !(fy < -1.0f || fy > 1.0f) && !(fz < 0.0f || fz > 1.0f);
  v.px = (fx + 1.0f) * 0.5f * (float)P.w;
  v.py = (1.0f - fy) * 0.5f * (float)P.h;
  v.z = fz;
  return v;
}

__device__ inline long long snap(float p) {
  const float s = p * 256.0f;
  if (!(s > -1.0e9f && s < 1.0e9f)) return LLONG_MIN;
  return (long long)floorf(s + 0.5f);
}

__device__ inline void raster_tri(uint32_t* __restrict__ zbuf, int w, int h, const Vert& a, const Vert& b, const Vert& c) {
  long long x0 = snap(a.px), y0 = snap(a.py), x1 = snap(b.px), y1 = snap(b.py), x2 = snap(c.px), y2 = snap(c.py);
  if (x0 == LLONG_MIN || y0 == LLONG_MIN || x1 == LLONG_MIN || y1 == LLONG_MIN || x2 == LLONG_MIN || y2 == LLONG_MIN) return;
  float z0 = a.z, z1 = b.z, z2 = c.z;
  long long area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
  if (area == 0) return;
  if (area < 0) {
    long long t;
    float tz;
    t = x1; x1 = x2; x2 = t;
    t = y1; y1 = y2; y2 = t;
    tz = z1; z1 = z2; z2 = tz;
    area = -area;
  }
  long long minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2)), miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  long long i0 = (minx - 128 + 255) >> 8, i1 = (maxx - 128) >> 8, j0 = (miny - 128 + 255) >> 8, j1 = (maxy - 128) >> 8;
  if (i0 < 0) i0 = 0;
  if (j0 < 0) j0 = 0;
  if (i1 > w - 1) i1 = w - 1;
  if (j1 > h - 1) j1 = h - 1;
  const long long ex0 = x1 - x0, ex1 = x2 - x1, ex2 = x0 - x2, ey0 = y1 - y0, ey1 = y2 - y1, ey2 = y0 - y2;
  const bool tl0 = (ey0 == 0 && ex0 > 0) || (ey0 < 0), tl1 = (ey1 == 0 && ex1 > 0) || (ey1 < 0), tl2 = (ey2 == 0 && ex2 > 0) || (ey2 < 0);
  const float fa = (float)area;
  for (long long j = j0; j <= j1; j++)
    for (long long i = i0; i <= i1; i++) {
      const long long px = 256 * i + 128, py = 256 * j + 128;
      const long long e0 = ex0 * (py - y0) - ey0 * (px - x0);
      const long long e1 = ex1 * (py - y1) - ey1 * (px - x1);
      const long long e2 = ex2 * (py - y2) - ey2 * (px - x2);
      if (e0 < 0 || e1 < 0 || e2 < 0) continue;
      if ((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2)) continue;
      const float z = ((float)e1 * z0 + (float)e2 * z1 + (float)e0 * z2) / fa;
      atomicMin(&zbuf[(size_t)j * w + i], __float_as_uint(z));  // z >= +0: float order = unsigned order of the bits
    }
}

// one lane per quad (x, y), x < w-1, y < h-1 (aligner.hlsl:131-173)
__global__ __launch_bounds__(256) void k_raster_quads(CalibK P, const float* __restrict__ und_all, const float4* __restrict__ vert_all,
                                                      uint32_t* __restrict__ zbuf_all) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = P.w * P.h;
  if (i >= n) return;
  const int x = i % P.w, y = i / P.w;
  if (x >= P.w - 1 || y >= P.h - 1) return;
  const float* __restrict__ depth = und_all + (size_t)blockIdx.z * n;
  uint32_t* zbuf = zbuf_all + (size_t)blockIdx.z * n;
  const float d0 = depth[(size_t)y * P.w + x], d1 = depth[(size_t)(y + 1) * P.w + x], d2 = depth[(size_t)y * P.w + x + 1], d3 = depth[(size_t)(y + 1) * P.w + x + 1];
  if (d0 <= DEPTH_WORLD_MIN || d1 <= DEPTH_WORLD_MIN || d2 <= DEPTH_WORLD_MIN || d3 <= DEPTH_WORLD_MIN) return;
  if (d0 == -INFINITY || d1 == -INFINITY || d2 == -INFINITY || d3 == -INFINITY) return;
  const float dmax = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)), dmin = fminf(fminf(d0, d1), fminf(d2, d3));
  const float dm = 0.5f * (dmax + dmin);
  if (dmax - dmin > 0.01f + 0.05f * dm) return;  // aligner.h:31-32, aligner.hlsl:154
  const float4* __restrict__ vert = vert_all + (size_t)blockIdx.z * n;
  auto load = [&](int xx, int yy) {
    const float4 q = vert[(size_t)yy * P.w + xx];
    return Vert{q.x, q.y, q.z, q.w != 0.0f};
  };
  const Vert v0 = load(x, y + 1), v1 = load(x, y), v2 = load(x + 1, y + 1), v3 = load(x + 1, y);
  if (!(v0.ok && v1.ok && v2.ok && v3.ok)) return;
  raster_tri(zbuf, P.w, P.h, v0, v1, v2);
  raster_tri(zbuf, P.w, P.h, v1, v2, v3);
}

// aligner.h:78-81 + calibration.h:286-301
__global__ __launch_bounds__(256) void k_finalize(CalibBatch B, CalibK P, const uint32_t* __restrict__ zbuf_all) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = P.w * P.h;
  if (i >= n) return;
  const float zb = __uint_as_float(zbuf_all[(size_t)blockIdx.z * n + i]);
  float d = zb == 1.0f ? 0.0f : DEPTH_WORLD_MIN + zb * (DEPTH_WORLD_MAX - DEPTH_WORLD_MIN);
  const uint8_t* rgb = B.rgb_out[blockIdx.z];
  if (rgb != nullptr) {
    const int x = i % P.w, y = i / P.w;
    const float sw = (float)(P.w - 1) / (float)(P.cw - 1), sh = (float)(P.h - 1) / (float)(P.ch - 1);
    int cx = round_i((float)x / sw), cy = round_i((float)y / sh);
    if (cx > P.cw - 1) cx = P.cw - 1;
    if (cy > P.ch - 1) cy = P.ch - 1;
    const uint8_t* p = rgb + 3 * ((size_t)cy * P.cw + cx);
    if (p[0] == 0 && p[1] == 0 && p[2] == 0) d = 0.0f;
  }
  const int r = round_i(d * P.shift);
  B.depth_out[blockIdx.z][i] = (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
}

}  // namespace

struct sf_calibrator {
  sf_calib_params p;
  CalibK k;
  int device = 0;
  hipStream_t stream = nullptr;
  float* lut = nullptr;
  float* und = nullptr;      // CALIB_MAX_BATCH x W*H undistorted depth in metres
  uint32_t* zbuf = nullptr;  // CALIB_MAX_BATCH x W*H depth buffer
  float4* vert = nullptr;    // CALIB_MAX_BATCH x W*H mesh vertices {target x, target y, projected z, valid}
  // staging for the host-pointer entry point
  uint8_t *d_rgb_in = nullptr, *d_rgb_out = nullptr;
  uint8_t *d_pay = nullptr, *d_planes = nullptr;   // JPEG coefficient payloads and plane scratch (allocated on first use)
  size_t pay_cap = 0, planes_cap = 0;
  uint16_t *d_depth_in = nullptr, *d_depth_out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

// ---- parameter / table files ---------------------------------------------------------------------------------------
SF_API int sf_calib_params_load(const char* path, sf_calib_params* out) {
  if (!path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::ifstream f(path);
  if (!f) return sf::fail(SF_ERR_IO, "no calibration param file: %s", path);
  std::memset(out, 0, sizeof(*out));
  for (int i = 0; i < 4; i++) out->color_intrinsic[5 * i] = out->depth_intrinsic[5 * i] = out->depth_extrinsic[5 * i] = 1.0f;  // Calib::reset, calibration.h:62-70
  std::string line;
  while (std::getline(f, line)) {
    const size_t c = line.find("//");
    if (c != std::string::npos) line.resize(c);
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == ';' || line.back() == '\t')) line.pop_back();
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string name = line.substr(0, eq), value = line.substr(eq + 1);
    auto trim = [](std::string& s) {
      const size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
      s = a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    trim(name); trim(value);
    const float v = (float)std::atof(value.c_str());
    // calibration.h:22-48
    if (name == "colorWidth") out->color_width = (uint32_t)v;
    else if (name == "colorHeight") out->color_height = (uint32_t)v;
    else if (name == "fx_color") out->color_intrinsic[0] = v;
    else if (name == "fy_color") out->color_intrinsic[5] = v;
    else if (name == "mx_color") out->color_intrinsic[2] = v;
    else if (name == "my_color") out->color_intrinsic[6] = v;
    else if (name == "depthWidth") out->depth_width = (uint32_t)v;
    else if (name == "depthHeight") out->depth_height = (uint32_t)v;
    else if (name == "fx_depth") out->depth_intrinsic[0] = v;
    else if (name == "fy_depth") out->depth_intrinsic[5] = v;
    else if (name == "mx_depth") out->depth_intrinsic[2] = v;
    else if (name == "my_depth") out->depth_intrinsic[6] = v;
    else if (name.size() == 8 && name[0] == 'k' && name[1] >= '1' && name[1] <= '5' && name.compare(2, 6, "_color") == 0) out->color_dist[name[1] - '1'] = v;
    else if (name.size() == 8 && name[0] == 'k' && name[1] >= '1' && name[1] <= '5' && name.compare(2, 6, "_depth") == 0) out->depth_dist[name[1] - '1'] = v;
    else if (name == "depthToColorExtrinsics") {
      std::istringstream is(value);
      float e[16];
      int k = 0;
      while (k < 16 && (is >> e[k])) k++;
      if (k == 16) std::memcpy(out->depth_extrinsic, e, sizeof(e));
    }
  }
  if (!out->depth_width || !out->depth_height || !out->color_width || !out->color_height)
    return sf::fail(SF_ERR_FORMAT, "%s: image dimensions missing", path);
  return SF_OK;
}

// Grid3D::ReadFile (grid3d.cpp:362-406): int xres, yres, zres; float maxDist; xres*yres*zres floats
SF_API int sf_lut_load(const char* path, sf_lut* out) {
  if (!path || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  std::ifstream f(path, std::ios::binary);
  if (!f) return sf::fail(SF_ERR_IO, "Could not open file %s for reading!", path);
  int32_t res[3];
  float maxd;
  if (!f.read((char*)res, 12) || !f.read((char*)&maxd, 4)) return sf::fail(SF_ERR_FORMAT, "Unable to read resolution from file %s", path);
  if (res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || (int64_t)res[0] * res[1] * res[2] > (1ll << 28) || !(maxd > 0.0f))
    return sf::fail(SF_ERR_FORMAT, "%s: implausible table header %d x %d x %d, max distance %g", path, res[0], res[1], res[2], (double)maxd);
  const size_t n = (size_t)res[0] * res[1] * res[2];
  float* data = (float*)std::malloc(n * 4);
  if (!data) return sf::fail(SF_ERR_IO, "out of memory");
  if (!f.read((char*)data, (std::streamsize)(n * 4))) { std::free(data); return sf::fail(SF_ERR_FORMAT, "Unable to read grid values from file %s", path); }
  out->xres = res[0]; out->yres = res[1]; out->zres = res[2];
  out->max_dist = maxd;
  out->data = data;
  return SF_OK;
}
SF_API void sf_lut_free(sf_lut* t) {
  if (t && t->data) { std::free(t->data); t->data = nullptr; }
}

// ---- calibrator ------------------------------------------------------------------------------------------------------
SF_API void sf_calibrator_destroy(sf_calibrator* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (void* p : {(void*)c->lut, (void*)c->und, (void*)c->zbuf, (void*)c->vert, (void*)c->d_rgb_in, (void*)c->d_rgb_out, (void*)c->d_depth_in, (void*)c->d_depth_out, (void*)c->d_pay, (void*)c->d_planes})
    if (p) (void)hipFree(p);
  if (c->e0) (void)hipEventDestroy(c->e0);
  if (c->e1) (void)hipEventDestroy(c->e1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

SF_API int sf_calibrator_create(const sf_calib_params* p, const sf_lut* lut, float depth_shift, int device, sf_calibrator** out) {
  if (!p || !out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (!p->depth_width || !p->depth_height || p->color_width < 2 || p->color_height < 2 || !(depth_shift > 0.0f) || !(p->depth_intrinsic[0] != 0.0f) ||
      !(p->depth_intrinsic[5] != 0.0f) || !(p->color_intrinsic[0] != 0.0f) || !(p->color_intrinsic[5] != 0.0f))
    return sf::fail(SF_ERR_INVALID_ARG, "invalid calibration parameters");
  if (lut && lut->data && (lut->xres > (int)p->depth_width || lut->yres > (int)p->depth_height))
    return sf::fail(SF_ERR_INVALID_ARG, "look-up table is finer than the depth image");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return sf::fail(SF_ERR_DEVICE, "no HIP device: libscanfuse has no CPU fallback, the calibrator needs an MI355X");
  if (device < 0 || device >= ndev) return sf::fail(SF_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  SF_HIP_CHECK(hipSetDevice(device));
  sf_calibrator* c = new sf_calibrator();
  c->p = *p;
  c->device = device;
  CalibK& k = c->k;
  std::memset(&k, 0, sizeof(k));
  k.w = (int)p->depth_width; k.h = (int)p->depth_height; k.cw = (int)p->color_width; k.ch = (int)p->color_height;
  std::memcpy(k.Kd, p->depth_intrinsic, 64);
  std::memcpy(k.Kc, p->color_intrinsic, 64);
  std::memcpy(k.E, p->depth_extrinsic, 64);
  std::memcpy(k.cdist, p->color_dist, 20);
  std::memcpy(k.ddist, p->depth_dist, 20);
  {  // mat4f::getInverse of the (upper triangular) depth intrinsic, in double, rounded once
    const double fx = k.Kd[0], sk = k.Kd[1], mx = k.Kd[2], fy = k.Kd[5], my = k.Kd[6];
    k.Kinv[0] = (float)(1.0 / fx);
    k.Kinv[1] = (float)(-sk / (fx * fy));
    k.Kinv[2] = (float)((sk * my - mx * fy) / (fx * fy));
    k.Kinv[5] = (float)(1.0 / fy);
    k.Kinv[6] = (float)(-my / fy);
    k.Kinv[10] = 1.0f;
    k.Kinv[15] = 1.0f;
  }
  k.shift = depth_shift;
  const size_t n = (size_t)k.w * k.h, cn = (size_t)k.cw * k.ch;
#define CAL_ALLOC(ptr, bytes)                                                                                       \
  do {                                                                                                              \
    const hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                                       \
    if (e_ != hipSuccess) { sf_calibrator_destroy(c); return sf::fail(SF_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e_)); } \
  } while (0)
  if (lut && lut->data) {
    k.lx = lut->xres; k.ly = lut->yres; k.lz = lut->zres; k.lmax = lut->max_dist;
    const size_t ln = (size_t)k.lx * k.ly * k.lz;
    CAL_ALLOC(c->lut, ln * 4);
    SF_HIP_CHECK(hipMemcpy(c->lut, lut->data, ln * 4, hipMemcpyHostToDevice));
  }
  CAL_ALLOC(c->und, n * 4 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->zbuf, n * 4 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->vert, n * 16 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->d_rgb_in, cn * 3 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->d_rgb_out, cn * 3 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->d_depth_in, n * 2 * CALIB_MAX_BATCH);
  CAL_ALLOC(c->d_depth_out, n * 2 * CALIB_MAX_BATCH);
#undef CAL_ALLOC
  SF_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreate(&c->e0));
  SF_HIP_CHECK(hipEventCreate(&c->e1));
  *out = c;
  return SF_OK;
}

SF_API int sf_calibrator_max_batch(void) { return CALIB_MAX_BATCH; }

// n <= 16 frames, device pointers.  rgb_in / rgb_out NULL (all frames): depth only, nothing invalidated by black pixels.
// Asynchronous on the calibrator's stream; kernel_us (nullable) makes it synchronous and returns the four launches' duration.
SF_API int sf_calibrator_run_device(sf_calibrator* c, int n, const void* const* d_rgb_in, void* const* d_rgb_out, const void* const* d_depth_in,
                                    void* const* d_depth_out, float* kernel_us) {
  if (!c || !d_depth_in || !d_depth_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > CALIB_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (limit %d)", n, CALIB_MAX_BATCH);
  SF_HIP_CHECK(hipSetDevice(c->device));
  const bool rgb = d_rgb_in != nullptr && d_rgb_out != nullptr && d_rgb_in[0] != nullptr;
  CalibBatch b;
  std::memset(&b, 0, sizeof(b));
  for (int j = 0; j < n; j++) {
    b.rgb_in[j] = rgb ? (const uint8_t*)d_rgb_in[j] : nullptr;
    b.rgb_out[j] = rgb ? (uint8_t*)d_rgb_out[j] : nullptr;
    b.depth_in[j] = (const uint16_t*)d_depth_in[j];
    b.depth_out[j] = (uint16_t*)d_depth_out[j];
    if (!b.depth_in[j] || !b.depth_out[j] || (rgb && (!b.rgb_in[j] || !b.rgb_out[j]))) return sf::fail(SF_ERR_INVALID_ARG, "NULL frame pointer in batch");
  }
  const CalibK& k = c->k;
  const int np = k.w * k.h;
  const size_t cn = (size_t)k.cw * k.ch;
  if (kernel_us) SF_HIP_CHECK(hipEventRecord(c->e0, c->stream));
  if (rgb) hipLaunchKernelGGL(k_undistort_rgb, dim3((unsigned)((cn / 4 + 1 + 255) / 256), 1, n), dim3(256), 0, c->stream, b, k);
  hipLaunchKernelGGL(k_depth_prepare, dim3((np + 255) / 256, 1, n), dim3(256), 0, c->stream, b, k, c->lut, c->und, c->zbuf, c->vert);
  hipLaunchKernelGGL(k_raster_quads, dim3((np + 255) / 256, 1, n), dim3(256), 0, c->stream, k, c->und, c->vert, c->zbuf);
  hipLaunchKernelGGL(k_finalize, dim3((np + 255) / 256, 1, n), dim3(256), 0, c->stream, b, k, c->zbuf);
  SF_HIP_CHECK(hipGetLastError());
  if (kernel_us) {
    SF_HIP_CHECK(hipEventRecord(c->e1, c->stream));
    SF_HIP_CHECK(hipEventSynchronize(c->e1));
    float ms = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, c->e0, c->e1));
    *kernel_us = ms * 1e3f;
  }
  return SF_OK;
}

// n <= 16 frames, host pointers (rgb: colour_width*colour_height*3 bytes per frame or NULL; depth: W*H u16).  Synchronous.
SF_API int sf_calibrator_run(sf_calibrator* c, int n, const uint8_t* const* rgb_in, uint8_t* const* rgb_out, const uint16_t* const* depth_in,
                             uint16_t* const* depth_out) {
  if (!c || !depth_in || !depth_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > CALIB_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (limit %d)", n, CALIB_MAX_BATCH);
  SF_HIP_CHECK(hipSetDevice(c->device));
  const CalibK& k = c->k;
  const size_t nb = (size_t)k.w * k.h * 2, cb = (size_t)k.cw * k.ch * 3;
  const bool rgb = rgb_in != nullptr && rgb_out != nullptr && rgb_in[0] != nullptr;
  const void* di[CALIB_MAX_BATCH];
  void* dout[CALIB_MAX_BATCH];
  const void* ri[CALIB_MAX_BATCH];
  void* ro[CALIB_MAX_BATCH];
  for (int j = 0; j < n; j++) {
    if (!depth_in[j] || !depth_out[j] || (rgb && (!rgb_in[j] || !rgb_out[j]))) return sf::fail(SF_ERR_INVALID_ARG, "NULL frame pointer in batch");
    di[j] = (uint8_t*)c->d_depth_in + j * nb;
    dout[j] = (uint8_t*)c->d_depth_out + j * nb;
    ri[j] = c->d_rgb_in + j * cb;
    ro[j] = c->d_rgb_out + j * cb;
    SF_HIP_CHECK(hipMemcpyAsync((void*)di[j], depth_in[j], nb, hipMemcpyHostToDevice, c->stream));
    if (rgb) SF_HIP_CHECK(hipMemcpyAsync((void*)ri[j], rgb_in[j], cb, hipMemcpyHostToDevice, c->stream));
  }
  const int rc = sf_calibrator_run_device(c, n, rgb ? ri : nullptr, rgb ? ro : nullptr, di, dout, nullptr);
  if (rc != SF_OK) return rc;
  for (int j = 0; j < n; j++) {
    SF_HIP_CHECK(hipMemcpyAsync(depth_out[j], dout[j], nb, hipMemcpyDeviceToHost, c->stream));
    if (rgb) SF_HIP_CHECK(hipMemcpyAsync(rgb_out[j], ro[j], cb, hipMemcpyDeviceToHost, c->stream));
  }
  SF_HIP_CHECK(hipStreamSynchronize(c->stream));
  return SF_OK;
}

int jpeg_gpu_reconstruct(hipStream_t stream, int n, const uint8_t* const* d_payload, uint8_t* const* d_rgb, uint8_t* const* d_planes, uint32_t max_blocks,
                         uint32_t max_width, uint32_t max_height);  // jpeg_gpu.hip

// bytes a frame's coefficient payload may take in sf_calibrator_run_payload (header + block table + as many entries as the pixels have bytes)
size_t calibrator_payload_capacity(const sf_calibrator* c) {
  const size_t padded = (size_t)((c->k.cw + 15) & ~15) * (size_t)((c->k.ch + 15) & ~15);
  return (sizeof(SfJpegLayout) + padded * 3 / 64 * 4 + (size_t)c->k.cw * c->k.ch * 3 + 255) & ~(size_t)255;
}

// sf_calibrator_run where a colour frame may arrive as an entropy-decoded JPEG payload (jpeg.cpp: jpeg_decode_coef) instead of pixels:
// payload[j] != NULL -> its coefficients are uploaded and reconstructed on the GPU straight into the stage's input buffer, the same
// bytes the host decoder would have produced; payload[j] == NULL -> rgb_in[j] as in sf_calibrator_run.  Internal to the calibrate stage.
int calibrator_run_payload(sf_calibrator* c, int n, const uint8_t* const* rgb_in, const uint8_t* const* payload, const uint32_t* payload_bytes,
                           uint8_t* const* rgb_out, const uint16_t* const* depth_in, uint16_t* const* depth_out) {
  if (!c || !rgb_in || !payload || !payload_bytes || !rgb_out || !depth_in || !depth_out) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > CALIB_MAX_BATCH) return sf::fail(SF_ERR_INVALID_ARG, "batch of %d frames (limit %d)", n, CALIB_MAX_BATCH);
  SF_HIP_CHECK(hipSetDevice(c->device));
  const CalibK& k = c->k;
  const size_t nb = (size_t)k.w * k.h * 2, cb = (size_t)k.cw * k.ch * 3;
  if (!c->d_pay) {
    c->pay_cap = calibrator_payload_capacity(c);
    c->planes_cap = ((size_t)((k.cw + 15) & ~15) * (size_t)((k.ch + 15) & ~15) * 3 + 255) & ~(size_t)255;
    SF_HIP_CHECK(hipMalloc((void**)&c->d_pay, c->pay_cap * CALIB_MAX_BATCH));
    SF_HIP_CHECK(hipMalloc((void**)&c->d_planes, c->planes_cap * CALIB_MAX_BATCH));
  }
  const void* di[CALIB_MAX_BATCH];
  void* dout[CALIB_MAX_BATCH];
  const void* ri[CALIB_MAX_BATCH];
  void* ro[CALIB_MAX_BATCH];
  const uint8_t* pp[CALIB_MAX_BATCH];
  uint8_t* pr[CALIB_MAX_BATCH];
  uint8_t* pl[CALIB_MAX_BATCH];
  int np = 0;
  uint32_t max_blocks = 0;
  for (int j = 0; j < n; j++) {
    if (!depth_in[j] || !depth_out[j] || !rgb_out[j] || (!payload[j] && !rgb_in[j])) return sf::fail(SF_ERR_INVALID_ARG, "NULL frame pointer in batch");
    di[j] = (uint8_t*)c->d_depth_in + j * nb;
    dout[j] = (uint8_t*)c->d_depth_out + j * nb;
    ri[j] = c->d_rgb_in + j * cb;
    ro[j] = c->d_rgb_out + j * cb;
    SF_HIP_CHECK(hipMemcpyAsync((void*)di[j], depth_in[j], nb, hipMemcpyHostToDevice, c->stream));
    if (payload[j]) {
      const SfJpegLayout* L = reinterpret_cast<const SfJpegLayout*>(payload[j]);
      if (payload_bytes[j] > c->pay_cap || L->width != k.cw || L->height != k.ch) return sf::fail(SF_ERR_INVALID_ARG, "frame %d: coefficient payload does not fit this calibrator", j);
      uint8_t* d = c->d_pay + (size_t)j * c->pay_cap;
      SF_HIP_CHECK(hipMemcpyAsync(d, payload[j], payload_bytes[j], hipMemcpyHostToDevice, c->stream));
      pp[np] = d; pr[np] = (uint8_t*)ri[j]; pl[np] = c->d_planes + (size_t)j * c->planes_cap;
      np++;
      max_blocks = std::max(max_blocks, L->nblocks);
    } else {
      SF_HIP_CHECK(hipMemcpyAsync((void*)ri[j], rgb_in[j], cb, hipMemcpyHostToDevice, c->stream));
    }
  }
  if (np > 0) {
    const int rcj = jpeg_gpu_reconstruct(c->stream, np, pp, pr, pl, max_blocks, (uint32_t)k.cw, (uint32_t)k.ch);
    if (rcj != SF_OK) return rcj;
  }
  const int rc = sf_calibrator_run_device(c, n, ri, ro, di, dout, nullptr);
  if (rc != SF_OK) return rc;
  for (int j = 0; j < n; j++) {
    SF_HIP_CHECK(hipMemcpyAsync(depth_out[j], dout[j], nb, hipMemcpyDeviceToHost, c->stream));
    SF_HIP_CHECK(hipMemcpyAsync(rgb_out[j], ro[j], cb, hipMemcpyDeviceToHost, c->stream));
  }
  SF_HIP_CHECK(hipStreamSynchronize(c->stream));
  return SF_OK;
}


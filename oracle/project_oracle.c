// project_oracle.c -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), never the product path).
//
// CPU restatement of the per-frame body of AnnotationTools/ProjectAnnotations/Visualizer.cpp (render(), :57-186): draw the
// labelled mesh from the frame's camera with the flat-colour shader (shaders/drawAnnotations.hlsl:9-33, `nointerpolation`
// colour = the triangle's first vertex), read back colour + depth, drop labels that disagree with the sensor depth (:147-164) and
// labels that hold less than 20 % of their 5x5 neighbourhood (:166-186).
//
// PARITY UNPINNED for the rendering conventions: the reference draws through mLib's D3D11 wrapper (github.com/niessner/mLib,
// a submodule with no recorded commit, absent from the tree) on whatever GPU the host had, and has no test or golden image for
// this path.  What this file fixes, and the HIP path must reproduce bit for bit:
//   * view matrix = rigid inverse of camera-to-world from its normalised columns (Cameraf(m, ...), Visualizer.cpp:40-41,64-65);
//   * projection = fx, fy with the principal point at the centre of the render target (visionToGraphicsProj gets only
//     intrinsic(0,0) and (1,1), :91), image y down, z_ndc = f/(f-n) - f n/((f-n) z) (Direct3D, left-handed);
//   * Direct3D 11 rasterisation rules: pixel centres at +0.5, vertices snapped to 1/256 pixel, top-left fill rule, depth test
//     LESS on a float buffer cleared to 1, depth = the plane through the snapped vertices evaluated at the pixel centre (anchored at the
//     first pixel of the triangle's bounding box, see raster_tri), no back-face culling, near-plane clipping of triangles that cross z_ndc = 0 (clip
//     points computed from the inside vertex towards the outside one), fragments with z_ndc > 1 discarded; on equal depth
//     the triangle drawn first (lower index) wins;
//   * camera-space depth z = B / (A - z_ndc), A = f/(f-n), B = n f/(f-n) (the reference inverts m_camera.getProj(), :123-134);
//   * the rendered depth is resampled to the depth resolution by nearest neighbour and converted to millimetres by
//     truncation (BaseImage::getResized + DepthImage16(., 1000.0f), :142 -- mLib internals);
//   * the 5x5 vote reads the labels as they were before the vote (the reference filters in place under `omp parallel for`,
//     :168-185, so its result depends on thread timing).
// Built with -ffp-contract=off; every expression below is evaluated left to right in fp32.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint32_t color_width, color_height, depth_width, depth_height;
  float fx, fy;                 // colour intrinsics (0,0) and (1,1)
  float depth_min, depth_max;   // s_depthMin / s_depthMax (zParametersScan.txt:12-13)
  float depth_dist_thresh;      // s_depthDistThresh (:14)
  int32_t filter_using_original_depth;   // s_filterUsingOrigialDepthImage (:6)
} or_project_params;

// worldViewProj of Visualizer.cpp:91-93 for a camera-to-world matrix (row-major), see the header for the conventions
void or_project_matrix(const float* c, float fx, float fy, uint32_t W, uint32_t H, float n, float f, float* M) {
  float ax[3][3];
  for (int k = 0; k < 3; k++) {
    const float x = c[k], y = c[4 + k], z = c[8 + k];
    const float len = sqrtf(x * x + y * y + z * z);
    ax[k][0] = x / len; ax[k][1] = y / len; ax[k][2] = z / len;
  }
  const float ex = c[3], ey = c[7], ez = c[11];
  float view[3][4];
  for (int k = 0; k < 3; k++) {
    view[k][0] = ax[k][0]; view[k][1] = ax[k][1]; view[k][2] = ax[k][2];
    view[k][3] = -(ax[k][0] * ex + ax[k][1] * ey + ax[k][2] * ez);
  }
  const float p00 = 2.0f * fx / (float)W, p11 = -(2.0f * fy / (float)H);
  const float A = f / (f - n), B = (n * f) / (f - n);
  for (int j = 0; j < 4; j++) {
    M[j] = p00 * view[0][j];
    M[4 + j] = p11 * view[1][j];
    M[8 + j] = A * view[2][j];
    M[12 + j] = view[2][j];
  }
  M[11] = M[11] - B;
}

typedef struct { float x, y, z, w; } clipv;
typedef struct { float px, py, z; } scrv;

static long long snap(float p) {
  const float s = p * 256.0f;
  if (!(s > -1.0e9f && s < 1.0e9f)) return INT64_MIN;
  return (long long)floorf(s + 0.5f);
}

static void raster_tri(uint64_t* zbuf, int w, int h, scrv a, scrv b, scrv c, uint32_t id) {
  long long x0 = snap(a.px), y0 = snap(a.py), x1 = snap(b.px), y1 = snap(b.py), x2 = snap(c.px), y2 = snap(c.py);
  if (x0 == INT64_MIN || y0 == INT64_MIN || x1 == INT64_MIN || y1 == INT64_MIN || x2 == INT64_MIN || y2 == INT64_MIN) return;
  float z0 = a.z, z1 = b.z, z2 = c.z;
  long long area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
  if (area == 0) return;
  if (area < 0) {
    long long t; float tz;
    t = x1; x1 = x2; x2 = t;
    t = y1; y1 = y2; y2 = t;
    tz = z1; z1 = z2; z2 = tz;
    area = -area;
  }
  long long minx = x0 < x1 ? x0 : x1; if (x2 < minx) minx = x2;
  long long maxx = x0 > x1 ? x0 : x1; if (x2 > maxx) maxx = x2;
  long long miny = y0 < y1 ? y0 : y1; if (y2 < miny) miny = y2;
  long long maxy = y0 > y1 ? y0 : y1; if (y2 > maxy) maxy = y2;
  long long i0 = (minx - 128 + 255) >> 8, i1 = (maxx - 128) >> 8, j0 = (miny - 128 + 255) >> 8, j1 = (maxy - 128) >> 8;
  if (i0 < 0) i0 = 0;
  if (j0 < 0) j0 = 0;
  if (i1 > w - 1) i1 = w - 1;
  if (j1 > h - 1) j1 = h - 1;
  const long long ex0 = x1 - x0, ex1 = x2 - x1, ex2 = x0 - x2, ey0 = y1 - y0, ey1 = y2 - y1, ey2 = y0 - y2;
  const int tl0 = (ey0 == 0 && ex0 > 0) || (ey0 < 0), tl1 = (ey1 == 0 && ex1 > 0) || (ey1 < 0), tl2 = (ey2 == 0 && ex2 > 0) || (ey2 < 0);
  if (i1 < i0 || j1 < j0) return;
  // the depth plane through the three snapped vertices, anchored at the first pixel of the (viewport-clamped) bounding box: with the
  // edge functions e0 + e1 + e2 = area, z = z0 + (e2 (z1 - z0) + e0 (z2 - z0)) / area; per pixel z = (zc + di dzdx) + dj dzdy
  const long long bx = 256 * i0 + 128, by = 256 * j0 + 128;
  const long long q0 = ex0 * (by - y0) - ey0 * (bx - x0), q2 = ex2 * (by - y2) - ey2 * (bx - x2);
  const float fa = (float)area, dz1 = z1 - z0, dz2 = z2 - z0;
  const float zc = z0 + ((float)q2 * dz1 + (float)q0 * dz2) / fa;
  const float dzdx = ((float)(-(ey2 * 256)) * dz1 + (float)(-(ey0 * 256)) * dz2) / fa;
  const float dzdy = ((float)(ex2 * 256) * dz1 + (float)(ex0 * 256) * dz2) / fa;
  for (long long j = j0; j <= j1; j++)
    for (long long i = i0; i <= i1; i++) {
      const long long px = 256 * i + 128, py = 256 * j + 128;
      const long long e0 = ex0 * (py - y0) - ey0 * (px - x0);
      const long long e1 = ex1 * (py - y1) - ey1 * (px - x1);
      const long long e2 = ex2 * (py - y2) - ey2 * (px - x2);
      if (e0 < 0 || e1 < 0 || e2 < 0) continue;
      if ((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2)) continue;
      const float z = (zc + (float)(i - i0) * dzdx) + (float)(j - j0) * dzdy;
      if (!(z >= 0.0f && z <= 1.0f)) continue;
      uint32_t zb;
      memcpy(&zb, &z, 4);
      const uint64_t key = ((uint64_t)zb << 32) | id;
      uint64_t* p = &zbuf[(size_t)j * w + i];
      if (key < *p) *p = key;
    }
}

static scrv to_screen(clipv c, float W, float H) {
  scrv s;
  s.px = (c.x / c.w + 1.0f) * 0.5f * W;
  s.py = (1.0f - c.y / c.w) * 0.5f * H;
  s.z = c.z / c.w;
  return s;
}

static clipv clip_point(clipv in, clipv out) {
  const float t = in.z / (in.z - out.z);
  clipv q;
  q.x = in.x + t * (out.x - in.x);
  q.y = in.y + t * (out.y - in.y);
  q.z = 0.0f;
  q.w = in.w + t * (out.w - in.w);
  return q;
}

static float camera_z(float d, float n, float f) {
  if (d == 0.0f || d == 1.0f) return 0.0f;
  const float A = f / (f - n), B = (n * f) / (f - n);
  const float z = B / (A - d);
  return (z >= n && z <= f) ? z : 0.0f;
}

// One frame.  zcam_out (nullable): the camera-space depth image at colour resolution (Visualizer.cpp:127-141).
int or_project_frame(const or_project_params* P, const float* xyz, uint64_t V, const uint32_t* tris, uint64_t F, const uint8_t* vinst,
                     const uint16_t* vlabel, const float* cam2world, const uint16_t* orig_depth, uint8_t* inst_out, uint16_t* label_out,
                     float* zcam_out) {
  const int w = (int)P->color_width, h = (int)P->color_height, dw = (int)P->depth_width, dh = (int)P->depth_height;
  const size_t n = (size_t)w * h;
  memset(inst_out, 0, n);
  memset(label_out, 0, 2 * n);
  if (zcam_out) memset(zcam_out, 0, 4 * n);
  if (cam2world[0] == -INFINITY) return 0;  // Visualizer.cpp:63,187-192
  float M[16];
  or_project_matrix(cam2world, P->fx, P->fy, P->color_width, P->color_height, P->depth_min, P->depth_max, M);
  uint64_t* zbuf = (uint64_t*)malloc(8 * n);
  clipv* cv = (clipv*)malloc(sizeof(clipv) * (V ? V : 1));
  if (!zbuf || !cv) { free(zbuf); free(cv); return -1; }
  memset(zbuf, 0xff, 8 * n);
  for (uint64_t i = 0; i < V; i++) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    cv[i].x = M[0] * x + M[1] * y + M[2] * z + M[3];
    cv[i].y = M[4] * x + M[5] * y + M[6] * z + M[7];
    cv[i].z = M[8] * x + M[9] * y + M[10] * z + M[11];
    cv[i].w = M[12] * x + M[13] * y + M[14] * z + M[15];
  }
  const float Wf = (float)w, Hf = (float)h;
  for (uint64_t t = 0; t < F; t++) {
    const uint32_t ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
    if (ia >= V || ib >= V || ic >= V) continue;
    const clipv c[3] = {cv[ia], cv[ib], cv[ic]};
    const int in[3] = {c[0].z >= 0.0f, c[1].z >= 0.0f, c[2].z >= 0.0f};
    const int nin = in[0] + in[1] + in[2];
    if (nin == 0) continue;
    if (!(c[0].z == c[0].z && c[1].z == c[1].z && c[2].z == c[2].z)) continue;
    clipv poly[4];
    int np = 0;
    if (nin == 3) { poly[0] = c[0]; poly[1] = c[1]; poly[2] = c[2]; np = 3; }
    else {
      for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3;
        if (in[k]) poly[np++] = c[k];
        if (in[k] != in[k1]) poly[np++] = in[k] ? clip_point(c[k], c[k1]) : clip_point(c[k1], c[k]);
      }
    }
    scrv s[4];
    for (int k = 0; k < np; k++) s[k] = to_screen(poly[k], Wf, Hf);
    raster_tri(zbuf, w, h, s[0], s[1], s[2], (uint32_t)t);
    if (np == 4) raster_tri(zbuf, w, h, s[0], s[2], s[3], (uint32_t)t);
  }
  // colour buffer -> ids (:104-114), depth buffer -> camera z (:123-141)
  uint8_t* inst1 = (uint8_t*)calloc(n, 1);
  uint16_t* lab1 = (uint16_t*)calloc(n, 2);
  float* zc = (float*)calloc(n, 4);
  for (size_t i = 0; i < n; i++) {
    const uint64_t key = zbuf[i];
    if (key == UINT64_MAX) continue;
    const uint32_t t = (uint32_t)key, zb = (uint32_t)(key >> 32);
    const uint32_t v0 = tris[3 * (size_t)t];
    inst1[i] = vinst[v0];
    lab1[i] = vlabel[v0];
    float d;
    memcpy(&d, &zb, 4);
    zc[i] = camera_z(d, P->depth_min, P->depth_max);
  }
  if (zcam_out) memcpy(zcam_out, zc, 4 * n);
  // depth consistency (:144-164)
  if (orig_depth) {
    const float sw = (float)(dw - 1) / (float)(w - 1), sh = (float)(dh - 1) / (float)(h - 1);
    const float rw = (float)(w - 1) / (float)(dw - 1), rh = (float)(h - 1) / (float)(dh - 1);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const size_t i = (size_t)y * w + x;
        if (lab1[i] == 0) continue;
        const unsigned dx = (unsigned)roundf(sw * (float)x), dy = (unsigned)roundf(sh * (float)y);
        int sx = (int)roundf((float)dx * rw), sy = (int)roundf((float)dy * rh);   // nearest-neighbour resample of the rendered depth
        if (sx > w - 1) sx = w - 1;
        if (sy > h - 1) sy = h - 1;
        const uint16_t drndr = (uint16_t)(zc[(size_t)sy * w + sx] * 1000.0f);
        const uint16_t dorig = orig_depth[(size_t)dy * dw + dx];
        if ((P->filter_using_original_depth && dorig == 0) ||
            (drndr != 0 && dorig != 0 && fabsf((float)((int)drndr - (int)dorig) * 0.001f) > P->depth_dist_thresh + 0.01f * (float)dorig)) {
          lab1[i] = 0;
          inst1[i] = 0;
        }
      }
  }
  // 5x5 vote (:166-186), reading the pre-vote labels
  const int radius = 2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const size_t i = (size_t)y * w + x;
      const uint16_t v = lab1[i];
      if (v == 0) continue;
      unsigned count = 0, total = 0;
      for (int yy = y - radius; yy <= y + radius; yy++)
        for (int xx = x - radius; xx <= x + radius; xx++)
          if (xx >= 0 && xx < w && yy >= 0 && yy < h) {
            total++;
            if (lab1[(size_t)yy * w + xx] == v) count++;
          }
      if ((float)count / (float)total < 0.2f) continue;
      label_out[i] = v;
      inst_out[i] = inst1[i];
    }
  free(zbuf); free(cv); free(inst1); free(lab1); free(zc);
  return 0;
}

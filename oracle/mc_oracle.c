/*
 * oracle/mc_oracle.c -- CPU restatement of iso-surface extraction (marching cubes + exact weld).
 *
 * TEST INFRASTRUCTURE ONLY (see tsdf_oracle.c header).  PARITY UNPINNED: the reference tree holds no
 * marching-cubes code (it lives in the external DepthSensing.exe, Server/scan_processor.py:34-35,138);
 * the only in-tree anchors are s_SDFMarchingCubeThreshFactor and s_marchingCubesMaxNumTriangles
 * (Server/tools/recons/zParametersScanNet.txt:48,106) and the PLY surface the result is written to
 * (README.md:45-46).  Spec: DESIGN.md section 3.7.
 *
 * Canonical mesh: one vertex per sign-changing grid edge, identified by the integer key
 * (gx, gy, gz, axis) of the edge's lower voxel; vertices sorted by key; triangles in (cube key, table
 * order).  Two implementations of this spec produce byte-identical arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mc_tables.h"

typedef struct { float sdf; uint8_t r, g, b, w; } mc_voxel;
typedef struct {
  int32_t width, height; float fx, fy, mx, my; float depth_shift; float depth_min, depth_max; float voxel_size;
  float trunc_base, trunc_scale; float max_integration_dist; int32_t weight_sample, weight_max; float mc_thresh_factor;
} mc_params;

/* provided by tsdf_oracle.c */
typedef struct or_volume or_volume;
const mc_params* or_get_params(const or_volume* v);
const void* or_find_block(const or_volume* v, int32_t x, int32_t y, int32_t z);
int64_t or_num_slots(const or_volume* v);
int or_slot_coords(const or_volume* v, int64_t s, int32_t* xyz);

#define KEY_BIAS (1 << 19)
static uint64_t edge_key(int32_t gx, int32_t gy, int32_t gz, int axis) {
  return ((uint64_t)(uint32_t)(gx + KEY_BIAS) << 42) | ((uint64_t)(uint32_t)(gy + KEY_BIAS) << 22) |
         ((uint64_t)(uint32_t)(gz + KEY_BIAS) << 2) | (uint64_t)axis;
}

typedef struct { uint64_t cube; uint64_t k[3]; } soup_tri;
typedef struct { uint64_t key; float p[3]; uint8_t c[3]; } soup_vert;

typedef struct {
  soup_tri* tris; int64_t nt, ct;
  soup_vert* verts; int64_t nv, cv;
  /* canonical result */
  float* pos; uint8_t* col; uint64_t* vkeys; int64_t n_verts;
  int32_t* idx; int64_t n_tris;
} mc_mesh;

static int cmp_vert(const void* a, const void* b) {
  const uint64_t x = ((const soup_vert*)a)->key, y = ((const soup_vert*)b)->key;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static int cmp_tri(const void* a, const void* b) {
  const soup_tri* x = (const soup_tri*)a; const soup_tri* y = (const soup_tri*)b;
  if (x->cube != y->cube) return x->cube < y->cube ? -1 : 1;
  return 0; /* qsort is not stable: the emitter below appends a per-cube ordinal into the low bits of `cube` */
}

static const mc_voxel* fetch(const or_volume* v, int32_t gx, int32_t gy, int32_t gz) {
  const mc_voxel* blk = (const mc_voxel*)or_find_block(v, gx >> 3, gy >> 3, gz >> 3);
  if (!blk) return NULL;
  return &blk[(gz & 7) * 64 + (gy & 7) * 8 + (gx & 7)];
}

mc_mesh* or_mc_extract(const or_volume* v) {
  const mc_params* p = or_get_params(v);
  const float voxel = p->voxel_size;
  const float thresh = p->mc_thresh_factor * voxel;
  mc_mesh* m = (mc_mesh*)calloc(1, sizeof(mc_mesh));
  const int64_t ns = or_num_slots(v);
  for (int64_t s = 0; s < ns; s++) {
    int32_t b[3];
    if (!or_slot_coords(v, s, b)) continue;
    for (int lz = 0; lz < 8; lz++) for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
      const int32_t g[3] = {8 * b[0] + lx, 8 * b[1] + ly, 8 * b[2] + lz};
      const mc_voxel* cv[8];
      int ok = 1, cs = 0;
      for (int i = 0; i < 8 && ok; i++) {
        cv[i] = fetch(v, g[0] + MC_CORNER_OFF[i][0], g[1] + MC_CORNER_OFF[i][1], g[2] + MC_CORNER_OFF[i][2]);
        if (!cv[i] || cv[i]->w == 0 || !(fabsf(cv[i]->sdf) <= thresh)) ok = 0;
        else if (cv[i]->sdf < 0.0f) cs |= 1 << i;
      }
      if (!ok) continue;
      const int nt = MC_NUM_TRIS[cs];
      if (!nt) continue;
      const uint64_t cube = edge_key(g[0], g[1], g[2], 0);
      for (int t = 0; t < nt; t++) {
        if (m->nt == m->ct) { m->ct = m->ct ? m->ct * 2 : 65536; m->tris = (soup_tri*)realloc(m->tris, (size_t)m->ct * sizeof(soup_tri)); }
        soup_tri* st = &m->tris[m->nt++];
        st->cube = ((cube >> 2) << 3) | (uint64_t)t; /* (gx,gy,gz) lexicographic, then table order (qsort is unstable) */
        for (int k = 0; k < 3; k++) {
          const int e = MC_TRIS[cs][3 * t + k];
          int a = MC_EDGE_CORNERS[e][0], c = MC_EDGE_CORNERS[e][1];
          int axis = 0;
          for (int q = 0; q < 3; q++) if (MC_CORNER_OFF[a][q] != MC_CORNER_OFF[c][q]) axis = q;
          if (MC_CORNER_OFF[a][axis] > MC_CORNER_OFF[c][axis]) { const int tmp = a; a = c; c = tmp; }
          const int32_t gl[3] = {g[0] + MC_CORNER_OFF[a][0], g[1] + MC_CORNER_OFF[a][1], g[2] + MC_CORNER_OFF[a][2]};
          const float dl = cv[a]->sdf, dh = cv[c]->sdf;
          const float mu = dl / (dl - dh);
          if (m->nv == m->cv) { m->cv = m->cv ? m->cv * 2 : 65536; m->verts = (soup_vert*)realloc(m->verts, (size_t)m->cv * sizeof(soup_vert)); }
          soup_vert* sv = &m->verts[m->nv++];
          sv->key = edge_key(gl[0], gl[1], gl[2], axis);
          for (int q = 0; q < 3; q++) sv->p[q] = (q == axis) ? ((float)gl[q] + mu) * voxel : (float)gl[q] * voxel;
          const uint8_t cl[3] = {cv[a]->r, cv[a]->g, cv[a]->b}, ch[3] = {cv[c]->r, cv[c]->g, cv[c]->b};
          for (int q = 0; q < 3; q++) sv->c[q] = (uint8_t)(fmaf(mu, (float)ch[q] - (float)cl[q], (float)cl[q]) + 0.5f);
          st->k[k] = sv->key;
        }
      }
    }
  }
  /* weld: sort + unique by key */
  qsort(m->verts, (size_t)m->nv, sizeof(soup_vert), cmp_vert);
  int64_t nu = 0;
  for (int64_t i = 0; i < m->nv; i++) if (i == 0 || m->verts[i].key != m->verts[i - 1].key) m->verts[nu++] = m->verts[i];
  m->n_verts = nu;
  m->pos = (float*)malloc((size_t)(nu ? nu : 1) * 12); m->col = (uint8_t*)malloc((size_t)(nu ? nu : 1) * 3);
  m->vkeys = (uint64_t*)malloc((size_t)(nu ? nu : 1) * 8);
  for (int64_t i = 0; i < nu; i++) {
    memcpy(m->pos + 3 * i, m->verts[i].p, 12); memcpy(m->col + 3 * i, m->verts[i].c, 3); m->vkeys[i] = m->verts[i].key;
  }
  qsort(m->tris, (size_t)m->nt, sizeof(soup_tri), cmp_tri);
  m->n_tris = m->nt;
  m->idx = (int32_t*)malloc((size_t)(m->nt ? m->nt : 1) * 12);
  for (int64_t i = 0; i < m->nt; i++) for (int k = 0; k < 3; k++) {
    const uint64_t key = m->tris[i].k[k];
    int64_t lo = 0, hi = nu - 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (m->vkeys[mid] < key) lo = mid + 1; else hi = mid; }
    m->idx[3 * i + k] = (int32_t)lo;
  }
  free(m->tris); m->tris = NULL; free(m->verts); m->verts = NULL;
  return m;
}

int64_t or_mc_num_verts(const mc_mesh* m) { return m->n_verts; }
int64_t or_mc_num_tris(const mc_mesh* m) { return m->n_tris; }
void or_mc_copy(const mc_mesh* m, float* pos, uint8_t* col, int32_t* idx, uint64_t* keys) {
  if (pos) memcpy(pos, m->pos, (size_t)m->n_verts * 12);
  if (col) memcpy(col, m->col, (size_t)m->n_verts * 3);
  if (idx) memcpy(idx, m->idx, (size_t)m->n_tris * 12);
  if (keys) memcpy(keys, m->vkeys, (size_t)m->n_verts * 8);
}
void or_mc_free(mc_mesh* m) { if (!m) return; free(m->pos); free(m->col); free(m->vkeys); free(m->idx); free(m); }

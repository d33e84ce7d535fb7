// oracle/simplify_rounds_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The checker of scannet_amd/csrc/simplify_gpu.hip: the `decimate` stage's "Quadric Edge Collapse Decimation" (Server/tools/meshclean/simplify.mlx:3-16,
// run twice by Server/scan_processor.py:144-146) as the GPU runs it -- ROUNDS OF INDEPENDENT COLLAPSES -- restated as a plain sequential program.  The
// rule is a property of the mesh, not of thread timing, so one thread walking edges in index order must arrive at the same surviving vertices, the
// same positions and the same triangles, bit for bit:
//
//   per round  1. the unique undirected edges (v0 < v1) of the live faces, in key order (v0 << 32 | v1), each with its face count (1 = border)
//              2. per edge: Q = Q(v0) + Q(v1); x = the minimiser of Q nearest the edge midpoint (adjugate inverse when det > 1e-6 trace^3, else the
//                 pseudo-inverse through a cyclic Jacobi eigen-decomposition, eigenvalues below 1e-9 of the largest dropped); priority =
//                 scale * Q(x), floored at 1e-15, divided by min(QualityThr, worst VCG quality of the faces around the pair after the move), as a
//                 float; the LINK CONDITION (common neighbours of v0 and v1 == faces on the edge; rings longer than 96 reject) makes it +inf
//              3. tau = the priority of rank min(E - 1, n + n / 2 + 64) among this round's priorities, n = (faces still to remove + 1) / 2
//                 (after a round without winners: no threshold); never above 3e38
//              4. three passes: every candidate (priority <= tau, neither end point inside the closed rings of an earlier pass's winner) writes its
//                 key -- priority bits << 32 | scramble(edge index + salt), salt = (round * 3 + pass) * 0x9E3779B9 -- over the closed 1-rings of both
//                 end points, keeping the minimum per vertex; a candidate WINS iff both its end points still hold its key; winners mark their rings
//              5. all winners collapse when the face budget allows it; in the last round the winners in key order until the budget is reached
//              6. collapse (v0 -> v1): faces holding both die, v0's other faces are re-pointed to v1, v1 moves to x, Q(v1) += Q(v0), v0 is deleted
//   then AutoClean: zero-area faces, vertices with bit-identical positions merged to the lowest index, faces that lost a corner, unreferenced vertices.
//
// Initial quadrics: per vertex, over its faces in face order, the plane quadric of the UN-normalised normal (area-weighted) plus, for a border edge
// at the vertex (or any edge with PlanarQuadric), the quadric of the plane through the edge perpendicular to the face, weight 0.5 * BoundaryWeight
// (/ 100 off the border).  All of it in binary64 with every operation rounded separately (-ffp-contract=off), face quality in binary32 as CMeshO does.
//
// parity unpinned against MeshLab itself (not in /root/reference, no meshlabserver here): this pins the GPU implementation to its own specification.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double a[6], b[3], c; } Quad;

static void q_zero(Quad* q) { memset(q, 0, sizeof(*q)); }
static void q_plane(Quad* q, const double n[3], double off) {
  q->a[0] = n[0] * n[0]; q->a[1] = n[0] * n[1]; q->a[2] = n[0] * n[2];
  q->a[3] = n[1] * n[1]; q->a[4] = n[1] * n[2]; q->a[5] = n[2] * n[2];
  q->b[0] = -2.0 * off * n[0]; q->b[1] = -2.0 * off * n[1]; q->b[2] = -2.0 * off * n[2];
  q->c = off * off;
}
static void q_add(Quad* q, const Quad* r) {
  for (int i = 0; i < 6; i++) q->a[i] += r->a[i];
  for (int i = 0; i < 3; i++) q->b[i] += r->b[i];
  q->c += r->c;
}
static double q_apply(const Quad* q, const double p[3]) {
  return p[0] * p[0] * q->a[0] + 2 * p[0] * p[1] * q->a[1] + 2 * p[0] * p[2] * q->a[2] + p[0] * q->b[0] + p[1] * p[1] * q->a[3] + 2 * p[1] * p[2] * q->a[4] +
         p[1] * q->b[1] + p[2] * p[2] * q->a[5] + p[2] * q->b[2] + q->c;
}

// cyclic Jacobi on the symmetric 3x3 matrix {a0 a1 a2; a1 a3 a4; a2 a4 a5}: eigenvalues w, eigenvectors the columns of V
static void jacobi3(const double a[6], double w[3], double V[3][3]) {
  double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-18 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) { const double x = A[k][p], y = A[k][q]; A[k][p] = cs * x - sn * y; A[k][q] = sn * x + cs * y; }
        for (int k = 0; k < 3; k++) { const double x = A[p][k], y = A[q][k]; A[p][k] = cs * x - sn * y; A[q][k] = sn * x + cs * y; }
        for (int k = 0; k < 3; k++) { const double x = V[k][p], y = V[k][q]; V[k][p] = cs * x - sn * y; V[k][q] = sn * x + cs * y; }
      }
  }
  for (int i = 0; i < 3; i++) w[i] = A[i][i];
}

static void minimiser(const Quad* q, const double mid[3], double x[3]) {
  const double* a = q->a;
  const double c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, tr = a[0] + a[3] + a[5];
  if (det > 1e-6 * tr * tr * tr && tr > 0.0) {
    const double c11 = a[0] * a[5] - a[2] * a[2], c12 = a[1] * a[2] - a[0] * a[4], c22 = a[0] * a[3] - a[1] * a[1];
    const double r0 = -0.5 * q->b[0], r1 = -0.5 * q->b[1], r2 = -0.5 * q->b[2], inv = 1.0 / det;
    x[0] = (c00 * r0 + c01 * r1 + c02 * r2) * inv;
    x[1] = (c01 * r0 + c11 * r1 + c12 * r2) * inv;
    x[2] = (c02 * r0 + c12 * r1 + c22 * r2) * inv;
    return;
  }
  double w[3], V[3][3];
  jacobi3(a, w, V);
  const double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
  const double Am[3] = {a[0] * mid[0] + a[1] * mid[1] + a[2] * mid[2], a[1] * mid[0] + a[3] * mid[1] + a[4] * mid[2], a[2] * mid[0] + a[4] * mid[1] + a[5] * mid[2]};
  const double r[3] = {-0.5 * q->b[0] - Am[0], -0.5 * q->b[1] - Am[1], -0.5 * q->b[2] - Am[2]};
  x[0] = mid[0]; x[1] = mid[1]; x[2] = mid[2];
  if (!(wmax > 0.0)) return;
  for (int k = 0; k < 3; k++) {
    if (!(fabs(w[k]) > 1e-9 * wmax)) continue;
    const double proj = (V[0][k] * r[0] + V[1][k] * r[1] + V[2][k] * r[2]) / w[k];
    for (int i = 0; i < 3; i++) x[i] += V[i][k] * proj;
  }
}

// vcg::Quality: 2 area / longest edge squared, in float
static float tri_quality(const float* p0, const float* p1, const float* p2) {
  const float d10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const float d20[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const float d12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const float x[3] = {d10[1] * d20[2] - d10[2] * d20[1], d10[2] * d20[0] - d10[0] * d20[2], d10[0] * d20[1] - d10[1] * d20[0]};
  const float a = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (a == 0) return 0;
  float b = d10[0] * d10[0] + d10[1] * d10[1] + d10[2] * d10[2];
  if (b == 0) return 0;
  float t = d20[0] * d20[0] + d20[1] * d20[1] + d20[2] * d20[2];
  if (b < t) b = t;
  t = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
  if (b < t) b = t;
  return a / b;
}

typedef struct { uint32_t k[3], v; } Rec;   // a live vertex: the bits of its canonical position, its index
static int cmp_rec(const void* pa, const void* pb) {
  const Rec* a = (const Rec*)pa;
  const Rec* b = (const Rec*)pb;
  for (int c = 0; c < 3; c++)
    if (a->k[c] != b->k[c]) return a->k[c] < b->k[c] ? -1 : 1;
  return a->v < b->v ? -1 : a->v > b->v;
}
static int cmp_u64(const void* a, const void* b) { const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static int cmp_u32(const void* a, const void* b) { const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

typedef struct {
  float* pos; uint32_t* tri; uint8_t* alive; uint8_t* vdel; Quad* Q;
  uint32_t* vbeg; uint32_t* vcnt; uint32_t* corner;   // per vertex: its corners 3 f + j in ascending order
  uint64_t* ukey; uint32_t* ucnt; uint32_t E;
  uint32_t V, F;
} M;

static uint32_t edge_count(const M* m, uint32_t a, uint32_t b) {
  const uint64_t key = ((uint64_t)(a < b ? a : b) << 32) | (uint64_t)(a < b ? b : a);
  uint32_t lo = 0, hi = m->E;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (m->ukey[mid] < key) lo = mid + 1; else hi = mid; }
  return (lo < m->E && m->ukey[lo] == key) ? m->ucnt[lo] : 0u;
}
static uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint64_t lock_key(float p, uint32_t e, uint32_t salt) {
  uint32_t h = e + salt;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return ((uint64_t)bits_of(p) << 32) | h;
}
#define MAX_RING 96

// out arrays sized for V vertices / F faces.  Returns 0, or -1 on out of memory / bad input.
int or_simplify_rounds(const float* pos_in, uint64_t V64, const uint32_t* tri_in, uint64_t F64, const uint8_t* rgba_in /* nullable */, float target_perc,
                       uint64_t target_faces, float quality_thr, float boundary_weight, int optimal, int planar, int auto_clean, float* pos_out,
                       uint32_t* tri_out, uint8_t* rgba_out, uint64_t* V_out, uint64_t* F_out, uint32_t* rounds_out, uint64_t* collapses_out) {
  const uint32_t V = (uint32_t)V64, F = (uint32_t)F64;
  M m;
  memset(&m, 0, sizeof(m));
  m.V = V; m.F = F;
  m.pos = (float*)malloc((size_t)V * 12 + 12); m.tri = (uint32_t*)malloc((size_t)F * 12 + 12); m.alive = (uint8_t*)malloc((size_t)F + 1); m.vdel = (uint8_t*)calloc((size_t)V + 1, 1);
  m.Q = (Quad*)malloc(((size_t)V + 1) * sizeof(Quad)); m.vbeg = (uint32_t*)malloc(((size_t)V + 2) * 4); m.vcnt = (uint32_t*)malloc(((size_t)V + 1) * 4);
  m.corner = (uint32_t*)malloc((size_t)F * 12 + 12); m.ukey = (uint64_t*)malloc((size_t)F * 24 + 24); m.ucnt = (uint32_t*)malloc((size_t)F * 12 + 12);
  uint64_t* ekey = (uint64_t*)malloc((size_t)F * 24 + 24);
  float* pri = (float*)malloc((size_t)F * 12 + 12);
  float* xs = (float*)malloc((size_t)F * 36 + 36);
  uint32_t* psort = (uint32_t*)malloc((size_t)F * 12 + 12);
  uint64_t* lock = (uint64_t*)malloc(((size_t)V + 1) * 8);
  uint8_t* taken = (uint8_t*)malloc((size_t)V + 1);
  uint64_t* win = (uint64_t*)malloc((size_t)F * 24 + 24);
  if (!m.pos || !m.tri || !m.alive || !m.vdel || !m.Q || !m.vbeg || !m.vcnt || !m.corner || !m.ukey || !m.ucnt || !ekey || !pri || !xs || !psort || !lock || !taken || !win) return -1;
  memcpy(m.pos, pos_in, (size_t)V * 12);
  memcpy(m.tri, tri_in, (size_t)F * 12);
  uint64_t nalive = 0;
  for (uint32_t f = 0; f < F; f++) {
    const uint32_t* t = &m.tri[3 * (size_t)f];
    if (t[0] >= V || t[1] >= V || t[2] >= V) return -1;
    m.alive[f] = !(t[0] == t[1] || t[1] == t[2] || t[0] == t[2]);   // faces with a repeated vertex take no part
    nalive += m.alive[f];
  }
  uint64_t target = target_faces;
  if (target_perc != 0.0f) target = (uint64_t)((double)F * (double)target_perc);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t v = 0; v < V; v++)
    for (int k = 0; k < 3; k++) { const double c = m.pos[3 * (size_t)v + k]; if (c < lo[k]) lo[k] = c; if (c > hi[k]) hi[k] = c; }
  double scale = 1.0;
  if (V) {
    const double diag = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    scale = diag > 0.0 ? 1e8 * pow(1.0 / diag, 6.0) : 1.0;
  }
  uint32_t rounds = 0;
  uint64_t collapses = 0;
  int first = 1, stalled = 0;
  while (F > 0 && V > 0 && nalive > target) {
    // ---- 1. corners per vertex (ascending corner id) and the unique edges of the live faces
    memset(m.vcnt, 0, (size_t)V * 4);
    for (uint32_t f = 0; f < F; f++)
      if (m.alive[f]) for (int j = 0; j < 3; j++) m.vcnt[m.tri[3 * (size_t)f + j]]++;
    uint32_t run = 0;
    for (uint32_t v = 0; v < V; v++) { m.vbeg[v] = run; run += m.vcnt[v]; m.vcnt[v] = 0; }
    size_t ne = 0;
    for (uint32_t f = 0; f < F; f++) {
      if (!m.alive[f]) continue;
      const uint32_t* t = &m.tri[3 * (size_t)f];
      for (int j = 0; j < 3; j++) {
        m.corner[m.vbeg[t[j]] + m.vcnt[t[j]]++] = 3 * f + (uint32_t)j;
        const uint32_t a = t[j], b = t[(j + 1) % 3];
        ekey[ne++] = ((uint64_t)(a < b ? a : b) << 32) | (uint64_t)(a < b ? b : a);
      }
    }
    qsort(ekey, ne, 8, cmp_u64);
    uint32_t E = 0;
    for (size_t i = 0; i < ne; i++) {
      if (E && m.ukey[E - 1] == ekey[i]) m.ucnt[E - 1]++;
      else { m.ukey[E] = ekey[i]; m.ucnt[E] = 1; E++; }
    }
    m.E = E;
    if (E == 0) break;
    if (first) {
      // ---- initial quadrics
      for (uint32_t v = 0; v < V; v++) {
        Quad acc;
        q_zero(&acc);
        for (uint32_t i = 0; i < m.vcnt[v]; i++) {
          const uint32_t f = m.corner[m.vbeg[v] + i] / 3;
          const uint32_t* t = &m.tri[3 * (size_t)f];
          double p0[3], p1[3], p2[3];
          for (int k = 0; k < 3; k++) { p0[k] = m.pos[3 * (size_t)t[0] + k]; p1[k] = m.pos[3 * (size_t)t[1] + k]; p2[k] = m.pos[3 * (size_t)t[2] + k]; }
          const double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
          const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
          Quad q;
          q_plane(&q, n, n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
          q_add(&acc, &q);
          for (int j = 0; j < 3; j++) {
            const uint32_t a = t[j], b = t[(j + 1) % 3];
            if (a != v && b != v) continue;
            const int border = edge_count(&m, a, b) == 1;
            if (!border && !planar) continue;
            double pa[3], pb[3], d[3];
            for (int k = 0; k < 3; k++) { pa[k] = m.pos[3 * (size_t)a + k]; pb[k] = m.pos[3 * (size_t)b + k]; d[k] = pb[k] - pa[k]; }
            const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (!(dl > 0.0)) continue;
            for (int k = 0; k < 3; k++) d[k] /= dl;
            const double wgt = border ? 0.5 * (double)boundary_weight : 0.5 * (double)boundary_weight / 100.0;
            const double bn[3] = {(n[1] * d[2] - n[2] * d[1]) * wgt, (n[2] * d[0] - n[0] * d[2]) * wgt, (n[0] * d[1] - n[1] * d[0]) * wgt};
            Quad bq;
            q_plane(&bq, bn, bn[0] * pa[0] + bn[1] * pa[1] + bn[2] * pa[2]);
            q_add(&acc, &bq);
          }
        }
        m.Q[v] = acc;
      }
      first = 0;
    }
    // ---- 2. priority, placement and link condition per edge
    for (uint32_t e = 0; e < E; e++) {
      const uint32_t v0 = (uint32_t)(m.ukey[e] >> 32), v1 = (uint32_t)m.ukey[e];
      Quad q = m.Q[v0];
      q_add(&q, &m.Q[v1]);
      float x[3];
      if (!optimal) { for (int k = 0; k < 3; k++) x[k] = m.pos[3 * (size_t)v1 + k]; }
      else {
        double p0[3], p1[3], mid[3], xd[3];
        for (int k = 0; k < 3; k++) { p0[k] = m.pos[3 * (size_t)v0 + k]; p1[k] = m.pos[3 * (size_t)v1 + k]; mid[k] = 0.5 * (p0[k] + p1[k]); }
        minimiser(&q, mid, xd);
        for (int k = 0; k < 3; k++) x[k] = (float)xd[k];
        if (!(x[0] == x[0] && x[1] == x[1] && x[2] == x[2])) {   // NaN: the best of midpoint and end points
          const double qm = q_apply(&q, mid), q0 = q_apply(&q, p0), q1 = q_apply(&q, p1);
          const double* best = mid;
          if (q0 < qm) best = p0;
          if (q1 < qm && q1 < q0) best = p1;
          for (int k = 0; k < 3; k++) x[k] = (float)best[k];
        }
      }
      double min_qual = 1e300;
      uint32_t ring[MAX_RING];
      int nring = 0, reject = 0, shared_faces = 0;
      for (int side = 0; side < 2; side++) {
        const uint32_t a = side ? v1 : v0, other = side ? v0 : v1;
        for (uint32_t i = 0; i < m.vcnt[a]; i++) {
          const uint32_t f = m.corner[m.vbeg[a] + i] / 3;
          const uint32_t* t = &m.tri[3 * (size_t)f];
          const int has_other = t[0] == other || t[1] == other || t[2] == other;
          if (has_other) { if (side == 0) shared_faces++; }
          else {
            const float* p[3];
            for (int k = 0; k < 3; k++) p[k] = t[k] == a ? x : &m.pos[3 * (size_t)t[k]];
            const double qt = tri_quality(p[0], p[1], p[2]);
            if (qt < min_qual) min_qual = qt;
          }
          if (side == 0)
            for (int k = 0; k < 3; k++)
              if (t[k] != a) { if (nring < MAX_RING) ring[nring++] = t[k]; else reject = 1; }
        }
      }
      if (!reject) {
        int common = 0;
        for (uint32_t i = 0; i < m.vcnt[v1]; i++) {
          const uint32_t c = m.corner[m.vbeg[v1] + i], f = c / 3, j = c % 3;
          const uint32_t* t = &m.tri[3 * (size_t)f];
          const uint32_t nxt = t[(j + 1) % 3], prv = t[(j + 2) % 3];
          uint32_t cand[2];
          int nc = 0;
          cand[nc++] = nxt;
          if (edge_count(&m, prv, v1) == 1) cand[nc++] = prv;
          for (int q2 = 0; q2 < nc; q2++) {
            if (cand[q2] == v0) continue;
            int in0 = 0;
            for (int r = 0; r < nring; r++) in0 = in0 || ring[r] == cand[q2];
            common += in0;
          }
        }
        if (common != shared_faces) reject = 1;
      }
      const double xd[3] = {x[0], x[1], x[2]};
      double err = scale * q_apply(&q, xd);
      if (min_qual > quality_thr) min_qual = quality_thr;
      if (err < 1e-15) err = 1e-15;
      if (quality_thr > 0.0f) err = min_qual > 0.0 ? err / min_qual : 1e300;
      float pf = err > 3.0e38 ? 3.0e38f : (float)err;
      if (reject) pf = INFINITY;
      pri[e] = pf;
      xs[3 * (size_t)e] = x[0]; xs[3 * (size_t)e + 1] = x[1]; xs[3 * (size_t)e + 2] = x[2];
    }
    // ---- 3. the threshold
    const uint64_t needed = (nalive - target + 1) / 2;
    float tau = INFINITY;
    if (stalled == 0) {
      for (uint32_t e = 0; e < E; e++) psort[e] = bits_of(pri[e]);   // positive floats and +inf: the bit patterns order them
      qsort(psort, E, 4, cmp_u32);
      uint64_t kth = needed + needed / 2 + 64;
      if (kth > (uint64_t)E - 1) kth = (uint64_t)E - 1;
      memcpy(&tau, &psort[kth], 4);
    }
    if (!(tau < 3.0e38f)) tau = 3.0e38f;
    // ---- 4. winners of three passes
    uint32_t nwin = 0;
    uint64_t removed_all = 0;
    memset(taken, 0, V);
    for (int pass = 0; pass < 3; pass++) {
      const uint32_t salt = (rounds * 3u + (uint32_t)pass) * 0x9E3779B9u;
      for (uint32_t v = 0; v < V; v++) lock[v] = ~0ull;
      for (uint32_t e = 0; e < E; e++) {
        if (!(pri[e] <= tau)) continue;
        const uint32_t v0 = (uint32_t)(m.ukey[e] >> 32), v1 = (uint32_t)m.ukey[e];
        if (taken[v0] | taken[v1]) continue;
        const uint64_t key = lock_key(pri[e], e, salt);
        if (key < lock[v0]) lock[v0] = key;
        if (key < lock[v1]) lock[v1] = key;
        for (int side = 0; side < 2; side++) {
          const uint32_t a = side ? v1 : v0;
          for (uint32_t i = 0; i < m.vcnt[a]; i++) {
            const uint32_t* t = &m.tri[3 * (size_t)(m.corner[m.vbeg[a] + i] / 3)];
            for (int k = 0; k < 3; k++)
              if (t[k] != a && key < lock[t[k]]) lock[t[k]] = key;
          }
        }
      }
      const uint32_t w0 = nwin;
      for (uint32_t e = 0; e < E; e++) {   // decided on the locks alone: marks of THIS pass's winners count from the next pass on
        if (!(pri[e] <= tau)) continue;
        const uint32_t v0 = (uint32_t)(m.ukey[e] >> 32), v1 = (uint32_t)m.ukey[e];
        const uint64_t key = lock_key(pri[e], e, salt);
        if (lock[v0] != key || lock[v1] != key) continue;
        win[nwin++] = ((uint64_t)bits_of(pri[e]) << 32) | e;
        removed_all += m.ucnt[e];
      }
      for (uint32_t i = w0; i < nwin; i++) {
        const uint32_t e = (uint32_t)win[i], v0 = (uint32_t)(m.ukey[e] >> 32), v1 = (uint32_t)m.ukey[e];
        taken[v0] = taken[v1] = 1;
        for (int side = 0; side < 2; side++) {
          const uint32_t a = side ? v1 : v0;
          for (uint32_t k2 = 0; k2 < m.vcnt[a]; k2++) {
            const uint32_t* t = &m.tri[3 * (size_t)(m.corner[m.vbeg[a] + k2] / 3)];
            for (int k = 0; k < 3; k++) taken[t[k]] = 1;
          }
        }
      }
    }
    rounds++;
    if (nwin == 0) {
      if (stalled++ >= 1) break;
      continue;
    }
    stalled = 0;
    // ---- 5. the budget
    uint32_t take = nwin;
    if (nalive - removed_all >= target) {
      nalive -= removed_all;
    } else {
      qsort(win, nwin, 8, cmp_u64);
      uint64_t removed = 0;
      take = 0;
      while (take < nwin && nalive - removed > target) { removed += m.ucnt[(uint32_t)win[take]]; take++; }
      nalive -= removed;
    }
    collapses += take;
    // ---- 6. collapse
    for (uint32_t i = 0; i < take; i++) {
      const uint32_t e = (uint32_t)win[i], v0 = (uint32_t)(m.ukey[e] >> 32), v1 = (uint32_t)m.ukey[e];
      for (uint32_t k = 0; k < m.vcnt[v0]; k++) {
        const uint32_t c = m.corner[m.vbeg[v0] + k], f = c / 3, j = c % 3;
        uint32_t* t = &m.tri[3 * (size_t)f];
        if (t[0] == v1 || t[1] == v1 || t[2] == v1) m.alive[f] = 0;
        else t[j] = v1;
      }
      q_add(&m.Q[v1], &m.Q[v0]);   // Q(v0) + Q(v1): addition of doubles commutes, the sum is the GPU's
      for (int k = 0; k < 3; k++) m.pos[3 * (size_t)v1 + k] = xs[3 * (size_t)e + k];
      m.vdel[v0] = 1;
    }
  }
  // ---- AutoClean and compaction
  uint32_t* target_of = (uint32_t*)malloc(((size_t)V + 1) * 4);
  uint32_t* order = (uint32_t*)malloc(((size_t)V + 1) * 4);
  uint8_t* used = (uint8_t*)calloc((size_t)V + 1, 1);
  uint32_t* remap = (uint32_t*)malloc(((size_t)V + 1) * 4);
  if (!target_of || !order || !used || !remap) return -1;
  for (uint32_t v = 0; v < V; v++) target_of[v] = v;
  if (auto_clean) {
    for (uint32_t f = 0; f < F; f++) {
      if (!m.alive[f]) continue;
      const uint32_t* t = &m.tri[3 * (size_t)f];
      const float *a = &m.pos[3 * (size_t)t[0]], *b = &m.pos[3 * (size_t)t[1]], *c = &m.pos[3 * (size_t)t[2]];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      if (!(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 0.0f)) m.alive[f] = 0;
    }
    // vertices with identical positions (-0 == +0) go to the lowest index: sort the live vertices by (x, y, z bits of the canonical value, index)
    uint32_t nl = 0;
    for (uint32_t v = 0; v < V; v++) if (!m.vdel[v]) order[nl++] = v;
    Rec* rec = (Rec*)malloc(((size_t)nl + 1) * sizeof(Rec));
    if (!rec) return -1;
    for (uint32_t i = 0; i < nl; i++) {
      for (int c = 0; c < 3; c++) { const float f = m.pos[3 * (size_t)order[i] + c] + 0.0f; memcpy(&rec[i].k[c], &f, 4); }
      rec[i].v = order[i];
    }
    qsort(rec, nl, sizeof(Rec), cmp_rec);
    for (uint32_t i = 1; i < nl; i++)
      if (rec[i].k[0] == rec[i - 1].k[0] && rec[i].k[1] == rec[i - 1].k[1] && rec[i].k[2] == rec[i - 1].k[2]) target_of[rec[i].v] = target_of[rec[i - 1].v];
    free(rec);
  }
  uint64_t nf = 0;
  for (uint32_t f = 0; f < F; f++) {
    if (!m.alive[f]) continue;
    const uint32_t a = target_of[m.tri[3 * (size_t)f]], b = target_of[m.tri[3 * (size_t)f + 1]], c = target_of[m.tri[3 * (size_t)f + 2]];
    if (auto_clean && (a == b || b == c || a == c)) continue;
    tri_out[3 * nf] = a; tri_out[3 * nf + 1] = b; tri_out[3 * nf + 2] = c;
    used[a] = used[b] = used[c] = 1;
    nf++;
  }
  uint32_t w = 0;
  for (uint32_t v = 0; v < V; v++) {
    const int keep = auto_clean ? used[v] != 0 : !m.vdel[v];
    remap[v] = keep ? w++ : 0xFFFFFFFFu;
  }
  for (uint32_t v = 0; v < V; v++) {
    if (remap[v] == 0xFFFFFFFFu) continue;
    memcpy(&pos_out[3 * (size_t)remap[v]], &m.pos[3 * (size_t)v], 12);
    if (rgba_in && rgba_out) memcpy(&rgba_out[4 * (size_t)remap[v]], &rgba_in[4 * (size_t)v], 4);
  }
  for (uint64_t i = 0; i < 3 * nf; i++) tri_out[i] = remap[tri_out[i]];
  *V_out = w; *F_out = nf; *rounds_out = rounds; *collapses_out = collapses;
  free(m.pos); free(m.tri); free(m.alive); free(m.vdel); free(m.Q); free(m.vbeg); free(m.vcnt); free(m.corner); free(m.ukey); free(m.ucnt);
  free(ekey); free(pri); free(xs); free(psort); free(lock); free(taken); free(win); free(target_of); free(order); free(used); free(remap);
  return 0;
}

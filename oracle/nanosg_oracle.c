/* oracle/nanosg_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement (float) of the reference's two-level traversal: examples/nanosg/nanosg.h
 * (Matrix :56-241, XformBoundingBox :246-302, Node::Update :397-437, NodeBBoxIntersector :599-665,
 * Scene::Traverse :773-870) over nanort.h's ListNodeIntersections (:2608-2692) and the per-node
 * BVHAccel::Traverse already restated in nanort_oracle_body.inc.  Pinned against oracle/_ref/libnanosg_ref.so
 * (tests/test_scene_oracle.py) and the golden fixture tests/golden/scene_ref.npz.
 *
 * The listing of node-AABB hits does not depend on the shape of the top-level BVH (every ancestor box
 * contains the leaf box and the slab arithmetic is monotone), so it is restated as a scan over the nodes.
 * Order among EQUAL t_min entries comes out of a std::priority_queue in the reference and is not restated
 * (it only matters when two instances give exactly the same world t).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* from nanort_oracle.c */
void orc_traverse_f32(const void *nodes, const uint32_t *indices, const void *verts, size_t stride,
                      const uint32_t *faces, const void *rays, uint64_t n, const uint32_t *trace_opt, void *hits,
                      uint8_t *mask, uint64_t *counters);

typedef struct {
  float org[3], dir[3], min_t, max_t;
  uint32_t type;
} sg_ray;
typedef struct {
  float u, v, t;
  uint32_t prim_id;
} sg_local_hit;
typedef struct {
  float t, u, v;
  uint32_t prim_id, node_id;
} sg_hit;

typedef struct {
  /* inputs */
  const void *nodes;
  const uint32_t *indices;
  const float *verts;
  const uint32_t *faces;
  float local_xform[4][4];
  float lbmin[3], lbmax[3];
  /* derived by sgo_node_update */
  float xform[4][4], inv_xform[4][4], inv_xform33[4][4];
  float xbmin[3], xbmax[3];
} sg_node;

/* Matrix::Mult — nanosg.h:221-230 */
static void mat_mult(float dst[4][4], const float m0[4][4], const float m1[4][4]) {
  int i, j, k;
  for (i = 0; i < 4; ++i)
    for (j = 0; j < 4; ++j) {
      dst[i][j] = 0;
      for (k = 0; k < 4; ++k) dst[i][j] += m0[k][j] * m1[i][k];
    }
}

/* Matrix::MultV — nanosg.h:232-240 */
static void mat_multv(float dst[3], const float m[4][4], const float v[3]) {
  float tmp[3];
  tmp[0] = m[0][0] * v[0] + m[1][0] * v[1] + m[2][0] * v[2] + m[3][0];
  tmp[1] = m[0][1] * v[0] + m[1][1] * v[1] + m[2][1] * v[2] + m[3][1];
  tmp[2] = m[0][2] * v[0] + m[1][2] * v[1] + m[2][2] * v[2] + m[3][2];
  dst[0] = tmp[0];
  dst[1] = tmp[1];
  dst[2] = tmp[2];
}

/* Matrix::Inverse (Cramer's rule) — nanosg.h:92-203 */
static void mat_inverse(float m[4][4]) {
  int i, j;
  float tmp[12], tsrc[16], det;
  for (i = 0; i < 4; i++) {
    tsrc[i] = m[i][0];
    tsrc[i + 4] = m[i][1];
    tsrc[i + 8] = m[i][2];
    tsrc[i + 12] = m[i][3];
  }
  tmp[0] = tsrc[10] * tsrc[15];
  tmp[1] = tsrc[11] * tsrc[14];
  tmp[2] = tsrc[9] * tsrc[15];
  tmp[3] = tsrc[11] * tsrc[13];
  tmp[4] = tsrc[9] * tsrc[14];
  tmp[5] = tsrc[10] * tsrc[13];
  tmp[6] = tsrc[8] * tsrc[15];
  tmp[7] = tsrc[11] * tsrc[12];
  tmp[8] = tsrc[8] * tsrc[14];
  tmp[9] = tsrc[10] * tsrc[12];
  tmp[10] = tsrc[8] * tsrc[13];
  tmp[11] = tsrc[9] * tsrc[12];
  m[0][0] = tmp[0] * tsrc[5] + tmp[3] * tsrc[6] + tmp[4] * tsrc[7];
  m[0][0] -= tmp[1] * tsrc[5] + tmp[2] * tsrc[6] + tmp[5] * tsrc[7];
  m[0][1] = tmp[1] * tsrc[4] + tmp[6] * tsrc[6] + tmp[9] * tsrc[7];
  m[0][1] -= tmp[0] * tsrc[4] + tmp[7] * tsrc[6] + tmp[8] * tsrc[7];
  m[0][2] = tmp[2] * tsrc[4] + tmp[7] * tsrc[5] + tmp[10] * tsrc[7];
  m[0][2] -= tmp[3] * tsrc[4] + tmp[6] * tsrc[5] + tmp[11] * tsrc[7];
  m[0][3] = tmp[5] * tsrc[4] + tmp[8] * tsrc[5] + tmp[11] * tsrc[6];
  m[0][3] -= tmp[4] * tsrc[4] + tmp[9] * tsrc[5] + tmp[10] * tsrc[6];
  m[1][0] = tmp[1] * tsrc[1] + tmp[2] * tsrc[2] + tmp[5] * tsrc[3];
  m[1][0] -= tmp[0] * tsrc[1] + tmp[3] * tsrc[2] + tmp[4] * tsrc[3];
  m[1][1] = tmp[0] * tsrc[0] + tmp[7] * tsrc[2] + tmp[8] * tsrc[3];
  m[1][1] -= tmp[1] * tsrc[0] + tmp[6] * tsrc[2] + tmp[9] * tsrc[3];
  m[1][2] = tmp[3] * tsrc[0] + tmp[6] * tsrc[1] + tmp[11] * tsrc[3];
  m[1][2] -= tmp[2] * tsrc[0] + tmp[7] * tsrc[1] + tmp[10] * tsrc[3];
  m[1][3] = tmp[4] * tsrc[0] + tmp[9] * tsrc[1] + tmp[10] * tsrc[2];
  m[1][3] -= tmp[5] * tsrc[0] + tmp[8] * tsrc[1] + tmp[11] * tsrc[2];
  tmp[0] = tsrc[2] * tsrc[7];
  tmp[1] = tsrc[3] * tsrc[6];
  tmp[2] = tsrc[1] * tsrc[7];
  tmp[3] = tsrc[3] * tsrc[5];
  tmp[4] = tsrc[1] * tsrc[6];
  tmp[5] = tsrc[2] * tsrc[5];
  tmp[6] = tsrc[0] * tsrc[7];
  tmp[7] = tsrc[3] * tsrc[4];
  tmp[8] = tsrc[0] * tsrc[6];
  tmp[9] = tsrc[2] * tsrc[4];
  tmp[10] = tsrc[0] * tsrc[5];
  tmp[11] = tsrc[1] * tsrc[4];
  m[2][0] = tmp[0] * tsrc[13] + tmp[3] * tsrc[14] + tmp[4] * tsrc[15];
  m[2][0] -= tmp[1] * tsrc[13] + tmp[2] * tsrc[14] + tmp[5] * tsrc[15];
  m[2][1] = tmp[1] * tsrc[12] + tmp[6] * tsrc[14] + tmp[9] * tsrc[15];
  m[2][1] -= tmp[0] * tsrc[12] + tmp[7] * tsrc[14] + tmp[8] * tsrc[15];
  m[2][2] = tmp[2] * tsrc[12] + tmp[7] * tsrc[13] + tmp[10] * tsrc[15];
  m[2][2] -= tmp[3] * tsrc[12] + tmp[6] * tsrc[13] + tmp[11] * tsrc[15];
  m[2][3] = tmp[5] * tsrc[12] + tmp[8] * tsrc[13] + tmp[11] * tsrc[14];
  m[2][3] -= tmp[4] * tsrc[12] + tmp[9] * tsrc[13] + tmp[10] * tsrc[14];
  m[3][0] = tmp[2] * tsrc[10] + tmp[5] * tsrc[11] + tmp[1] * tsrc[9];
  m[3][0] -= tmp[4] * tsrc[11] + tmp[0] * tsrc[9] + tmp[3] * tsrc[10];
  m[3][1] = tmp[8] * tsrc[11] + tmp[0] * tsrc[8] + tmp[7] * tsrc[10];
  m[3][1] -= tmp[6] * tsrc[10] + tmp[9] * tsrc[11] + tmp[1] * tsrc[8];
  m[3][2] = tmp[6] * tsrc[9] + tmp[11] * tsrc[11] + tmp[3] * tsrc[8];
  m[3][2] -= tmp[10] * tsrc[11] + tmp[2] * tsrc[8] + tmp[7] * tsrc[9];
  m[3][3] = tmp[10] * tsrc[10] + tmp[4] * tsrc[8] + tmp[9] * tsrc[9];
  m[3][3] -= tmp[8] * tsrc[9] + tmp[11] * tsrc[0] + tmp[5] * tsrc[8];
  det = tsrc[0] * m[0][0] + tsrc[1] * m[0][1] + tsrc[2] * m[0][2] + tsrc[3] * m[0][3];
  det = 1.0f / det;
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) m[j][i] *= det;
}

/* XformBoundingBox — nanosg.h:246-302 */
static void xform_bbox(float xbmin[3], float xbmax[3], const float bmin[3], const float bmax[3], const float m[4][4]) {
  float b[8][3], xb[8][3];
  int i, k;
  for (i = 0; i < 8; i++) {
    b[i][0] = (i & 1) ? bmax[0] : bmin[0];
    b[i][1] = (i & 2) ? bmax[1] : bmin[1];
    b[i][2] = (i & 4) ? bmax[2] : bmin[2];
    mat_multv(xb[i], m, b[i]);
  }
  for (k = 0; k < 3; k++) xbmin[k] = xbmax[k] = xb[0][k];
  for (i = 1; i < 8; i++)
    for (k = 0; k < 3; k++) {
      xbmin[k] = (xbmin[k] < xb[i][k]) ? xbmin[k] : xb[i][k]; /* std::min(xb, xbmin) */
      xbmax[k] = (xb[i][k] < xbmax[k]) ? xbmax[k] : xb[i][k]; /* std::max(xb, xbmax) */
    }
}

/* Node::Update with an identity parent — nanosg.h:397-437, Scene::Commit :708-715 */
void sgo_node_update(sg_node *n) {
  float ident[4][4];
  int i, j;
  for (i = 0; i < 4; i++)
    for (j = 0; j < 4; j++) ident[i][j] = (i == j) ? 1.0f : 0.0f;
  mat_mult(n->xform, ident, n->local_xform);
  xform_bbox(n->xbmin, n->xbmax, n->lbmin, n->lbmax, n->xform);
  memcpy(n->inv_xform, n->xform, sizeof(n->xform));
  mat_inverse(n->inv_xform);
  memcpy(n->inv_xform33, n->xform, sizeof(n->xform));
  n->inv_xform33[3][0] = 0.0f;
  n->inv_xform33[3][1] = 0.0f;
  n->inv_xform33[3][2] = 0.0f;
  mat_inverse(n->inv_xform33);
}

typedef struct {
  float t_min, t_max;
  uint32_t node;
} sg_nodehit;

static int nodehit_cmp(const void *a, const void *b) {
  const sg_nodehit *x = (const sg_nodehit *)a, *y = (const sg_nodehit *)b;
  if (x->t_min < y->t_min) return -1;
  if (x->t_min > y->t_min) return 1;
  return (x->node > y->node) - (x->node < y->node);
}

/* Does node `i` appear in ListNodeIntersections' result set? nanort.h:2649-2671 + 2285-2325 for the leaf
 * box, then NodeBBoxIntersector::Intersect nanosg.h:603-639 for the interval. */
static int node_interval(const sg_ray *ray, const sg_node *nd, float *tmin_out, float *tmax_out) {
  int k, sign[3];
  float inv_safe[3], tmin, tmax, tn[3], tf[3];
  for (k = 0; k < 3; k++) {
    const float d = ray->dir[k];
    sign[k] = d < 0.0f ? 1 : 0;
    if (fabsf(d) < FLT_EPSILON)
      inv_safe[k] = INFINITY * ((d < 0.0f) ? -1.0f : 1.0f);
    else
      inv_safe[k] = 1.0f / d;
  }
  tmin = ray->min_t;
  tmax = ray->max_t; /* hit_t never shrinks in ListNodeIntersections */
  for (k = 0; k < 3; k++) {
    const float lo = sign[k] ? nd->xbmax[k] : nd->xbmin[k], hi = sign[k] ? nd->xbmin[k] : nd->xbmax[k];
    const float t0 = (lo - ray->org[k]) * inv_safe[k];
    const float t1 = (hi - ray->org[k]) * inv_safe[k] * 1.00000024f;
    tmin = (t0 > tmin) ? t0 : tmin;
    tmax = (t1 < tmax) ? t1 : tmax;
  }
  if (!(tmin <= tmax)) return 0;
  /* the interval the scene graph sorts by: plain 1/dir, no MaxMult, no [min_t, max_t] clipping */
  for (k = 0; k < 3; k++) {
    const float inv = 1.0f / ray->dir[k];
    const float lo = sign[k] ? nd->xbmax[k] : nd->xbmin[k], hi = sign[k] ? nd->xbmin[k] : nd->xbmax[k];
    tn[k] = (lo - ray->org[k]) * inv;
    tf[k] = (hi - ray->org[k]) * inv;
  }
  tmin = (tn[1] > tn[0]) ? tn[1] : tn[0];
  tmin = (tn[2] > tmin) ? tn[2] : tmin;
  tmax = (tf[1] < tf[0]) ? tf[1] : tf[0];
  tmax = (tf[2] < tmax) ? tf[2] : tmax;
  if (!(tmin <= tmax)) return 0;
  *tmin_out = tmin;
  *tmax_out = tmax;
  return 1;
}

/* Scene::Traverse — nanosg.h:773-870 (kMaxIntersections = 64). */
void sgo_traverse(const sg_node *nodes, uint32_t num_nodes, const sg_ray *rays, uint64_t n, sg_hit *hits,
                  uint8_t *mask) {
  sg_nodehit *list = (sg_nodehit *)malloc(sizeof(sg_nodehit) * (num_nodes ? num_nodes : 1));
  uint64_t r;
  for (r = 0; r < n; r++) {
    const sg_ray *ray = &rays[r];
    uint32_t cnt = 0, i;
    float t_nearest = FLT_MAX;
    int has_hit = 0;
    sg_hit best;
    best.t = ray->max_t;
    best.u = best.v = 0.0f;
    best.prim_id = best.node_id = 0xFFFFFFFFu;
    for (i = 0; i < num_nodes; i++) {
      float a, b;
      if (node_interval(ray, &nodes[i], &a, &b)) {
        list[cnt].t_min = a;
        list[cnt].t_max = b;
        list[cnt].node = i;
        cnt++;
      }
    }
    qsort(list, cnt, sizeof(sg_nodehit), nodehit_cmp);
    if (cnt > 64) cnt = 64; /* the 64 nearest by t_min survive the priority queue (nanort.h:2594-2601) */
    for (i = 0; i < cnt; i++) {
      const sg_node *nd = &nodes[list[i].node];
      sg_ray lr;
      sg_local_hit lh;
      uint8_t m = 0;
      if (t_nearest < list[i].t_min) continue; /* early cull :795 */
      mat_multv(lr.org, nd->inv_xform, ray->org);
      mat_multv(lr.dir, nd->inv_xform33, ray->dir);
      lr.min_t = 0.0f; /* Ray() defaults: the world ray's interval is NOT propagated (:806 TODO) */
      lr.max_t = FLT_MAX;
      lr.type = 0;
      /* default BVHTraceOptions: the cull_back_face argument never reaches Traverse (:790-791 vs :817) */
      orc_traverse_f32(nd->nodes, nd->indices, nd->verts, 12, nd->faces, &lr, 1, NULL, &lh, &m, NULL);
      if (m) {
        float lp[3], wp[3], po[3], t_world;
        int k;
        for (k = 0; k < 3; k++) lp[k] = lr.org[k] + lh.t * lr.dir[k];
        mat_multv(wp, nd->xform, lp);
        for (k = 0; k < 3; k++) po[k] = wp[k] - ray->org[k];
        t_world = sqrtf(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]);
        if (t_world < t_nearest) {
          t_nearest = t_world;
          has_hit = 1;
          best.t = t_world;
          best.u = lh.u;
          best.v = lh.v;
          best.prim_id = lh.prim_id;
          best.node_id = list[i].node;
        }
      }
    }
    hits[r] = best;
    if (mask) mask[r] = (uint8_t)has_hit;
  }
  free(list);
}

/* MODEL (not a restatement) of the single-pass scene walk of nanort_amd/csrc/traverse.hip (k_scene_walk): the instances whose
 * boxes a ray enters are visited in ANY order — here a seeded shuffle, or the sorted order with local disorder — and no list is
 * kept.  Rules:
 *   rank(i)   = (t_min_i, i), the order Scene::Traverse visits its list in (nanosg.h:793-795 over nanort.h:2676-2687);
 *   winner    = the visited hit with the smallest (t_world, rank)  [the reference: strict '<' in rank order, nanosg.h:838];
 *   may skip i whenever the current winner w has t_world_w < t_min_i AND rank(w) < rank(i)  [then the reference's early
 *             cull (:795) has already fired at or before i]; the model skips such an instance or not by a coin;
 *   certificate: at most 64 instances traced, and every OTHER hit lies at or beyond the winner's own box entry
 *             (t2 >= t_min_w, t2 = second smallest t_world) — then the winner sits in the prefix of the list the reference
 *             processes and nothing outside that prefix can beat it (scenes).  A ray without the certificate
 *             is re-done by the listing path.
 * tests/test_scene_walk_model.py: every certified ray equals sgo_traverse bit for bit, whatever the order and the coins. */
void sgo_traverse_unordered_model(const sg_node *nodes, uint32_t num_nodes, const sg_ray *rays, uint64_t n, uint32_t seed,
                                  int roughly_front_to_back, sg_hit *hits, uint8_t *mask, uint8_t *certified) {
  sg_nodehit *list = (sg_nodehit *)malloc(sizeof(sg_nodehit) * (num_nodes ? num_nodes : 1));
  uint64_t r;
  for (r = 0; r < n; r++) {
    const sg_ray *ray = &rays[r];
    uint32_t cnt = 0, i, traced = 0;
    uint64_t rng = ((uint64_t)seed << 32) ^ (r * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull;
    float best_t = FLT_MAX, best_tmin = 0.0f, t2 = INFINITY;
    uint32_t best_id = 0;
    int has_hit = 0;
    sg_hit best;
    best.t = ray->max_t;
    best.u = best.v = 0.0f;
    best.prim_id = best.node_id = 0xFFFFFFFFu;
    for (i = 0; i < num_nodes; i++) {
      float a, b;
      if (node_interval(ray, &nodes[i], &a, &b)) {
        list[cnt].t_min = a;
        list[cnt].t_max = b;
        list[cnt].node = i;
        cnt++;
      }
    }
    if (roughly_front_to_back) { /* what a near-child-first walk of a tree produces: sorted up to local disorder */
      qsort(list, cnt, sizeof(sg_nodehit), nodehit_cmp);
      for (i = 0; i + 1 < cnt; i++) {
        uint32_t k;
        sg_nodehit tmp;
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        k = i + (uint32_t)((rng >> 33) % 4u);
        if (k >= cnt) k = cnt - 1;
        tmp = list[i];
        list[i] = list[k];
        list[k] = tmp;
      }
    } else {
      for (i = cnt; i > 1; i--) { /* Fisher-Yates */
        uint32_t k;
        sg_nodehit tmp;
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        k = (uint32_t)((rng >> 33) % i);
        tmp = list[i - 1];
        list[i - 1] = list[k];
        list[k] = tmp;
      }
    }
    for (i = 0; i < cnt; i++) {
      const sg_node *nd = &nodes[list[i].node];
      const float tmin_i = list[i].t_min;
      const uint32_t id_i = list[i].node;
      sg_ray lr;
      sg_local_hit lh;
      uint8_t m = 0;
      if (has_hit && best_t < tmin_i && (best_tmin < tmin_i || (best_tmin == tmin_i && best_id < id_i))) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        if (roughly_front_to_back || ((rng >> 40) & 1u)) continue; /* allowed to skip: the coin decides (the kernel's order: always) */
      }
      traced++;
      mat_multv(lr.org, nd->inv_xform, ray->org);
      mat_multv(lr.dir, nd->inv_xform33, ray->dir);
      lr.min_t = 0.0f;
      lr.max_t = FLT_MAX;
      lr.type = 0;
      orc_traverse_f32(nd->nodes, nd->indices, nd->verts, 12, nd->faces, &lr, 1, NULL, &lh, &m, NULL);
      if (m) {
        float lp[3], wp[3], po[3], t_world;
        int k, wins;
        for (k = 0; k < 3; k++) lp[k] = lr.org[k] + lh.t * lr.dir[k];
        mat_multv(wp, nd->xform, lp);
        for (k = 0; k < 3; k++) po[k] = wp[k] - ray->org[k];
        t_world = sqrtf(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]);
        wins = t_world < best_t ||
               (has_hit && t_world == best_t && (tmin_i < best_tmin || (tmin_i == best_tmin && id_i < best_id)));
        if (wins) {
          if (has_hit && best_t < t2) t2 = best_t; /* the old winner becomes the runner-up */
          best_t = t_world;
          best_tmin = tmin_i;
          best_id = id_i;
          has_hit = 1;
          best.t = t_world;
          best.u = lh.u;
          best.v = lh.v;
          best.prim_id = lh.prim_id;
          best.node_id = id_i;
        } else if (t_world < t2) {
          t2 = t_world; /* (a NaN distance is neither a winner nor a runner-up: the reference ignores it too) */
        }
      }
    }
    hits[r] = best;
    if (mask) mask[r] = (uint8_t)has_hit;
    /* 1: certified; else why not — 2: more than 64 traced, 4: another hit in front of the winner's box entry */
    certified[r] = (uint8_t)((traced <= 64u && (!has_hit || t2 >= best_tmin)) ? 1u : ((traced > 64u ? 2u : 0u) | ((has_hit && !(t2 >= best_tmin)) ? 4u : 0u)));
  }
  free(list);
}

int sgo_sizeof_node(void) { return (int)sizeof(sg_node); }

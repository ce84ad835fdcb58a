/* oracle/cylinder_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's cylinder custom primitive traced through BVHAccel<float>::Traverse:
 * examples/cylinder_primitive/main.cc (solve2e :61-90, CylinderIntersector::Intersect :237-343, PostTraversal
 * :367-418) over nanort.h (Traverse :2487-2556, TestLeafNode :2374-2407, IntersectRayAABB :2285-2325,
 * vsafe_inverse :442-461, vnormalize :383-398, vdot :410-412).  Pinned bit-for-bit against the unmodified example
 * (oracle/ref_cylinder_shim.cc) by tests/test_cylinder_oracle.py; only tests/ may use it.
 * Record: {u, v, normal[3], t, prim_id}; t is the intersector's t (the example's PostTraversal never writes it). */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  float bmin[3], bmax[3];
  int32_t flag, axis;
  uint32_t data[2];
} cyo_node;
typedef struct {
  float org[3], dir[3], min_t, max_t;
  uint32_t type;
} cyo_ray;
typedef struct {
  float u, v, normal[3], t;
  uint32_t prim_id;
} cyo_hit;

static float cyo_safe_inv(float v) {
  if (fabsf(v) < FLT_EPSILON) return INFINITY * ((v < 0.0f) ? -1.0f : 1.0f);
  return 1.0f / v;
}

static int cyo_slab(float min_t, float max_t, const float bmin[3], const float bmax[3], const float org[3],
                    const float inv[3], const int sign[3]) {
  float tmn[3], tmx[3], tmin, tmax;
  int k;
  for (k = 0; k < 3; k++) {
    const float mn = sign[k] ? bmax[k] : bmin[k], mx = sign[k] ? bmin[k] : bmax[k];
    tmn[k] = (mn - org[k]) * inv[k];
    tmx[k] = (mx - org[k]) * inv[k] * 1.00000024f;
  }
  tmin = (tmn[0] > min_t) ? tmn[0] : min_t;
  tmin = (tmn[1] > tmin) ? tmn[1] : tmin;
  tmin = (tmn[2] > tmin) ? tmn[2] : tmin;
  tmax = (tmx[0] < max_t) ? tmx[0] : max_t;
  tmax = (tmx[1] < tmax) ? tmx[1] : tmax;
  tmax = (tmx[2] < tmax) ? tmx[2] : tmax;
  return tmin <= tmax;
}

static float cyo_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cyo_sub(float o[3], const float a[3], const float b[3]) {
  o[0] = a[0] - b[0];
  o[1] = a[1] - b[1];
  o[2] = a[2] - b[2];
}
/* vnormalize — nanort.h:383-398 */
static void cyo_normalize(float o[3], const float a[3]) {
  const float len = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  o[0] = a[0];
  o[1] = a[1];
  o[2] = a[2];
  if (fabsf(len) > FLT_EPSILON) {
    const float inv_len = 1.0f / len;
    o[0] *= inv_len;
    o[1] *= inv_len;
    o[2] *= inv_len;
  }
}

/* solve2e — main.cc:61-90 */
static int cyo_solve2e(float root[2], float A, float B, float C) {
  if (fabsf(A) <= 1.0e-6f) {
    root[0] = -C / B;
    return 1;
  } else {
    const float D = B * B - A * C;
    if (D < 0) {
      return 0;
    } else if (D == 0) {
      root[0] = -B / A;
      return 1;
    } else {
      float x1 = (fabsf(B) + sqrtf(D)) / A, x2;
      if (B >= 0.0) x1 = -x1;
      x2 = C / (A * x1);
      if (x1 > x2) {
        const float tmp = x1;
        x1 = x2;
        x2 = tmp;
      }
      root[0] = x1;
      root[1] = x2;
      return 2;
    }
  }
}

typedef struct { /* the intersector's mutable members */
  int hit_cap;
  float u_param, v_param;
} cyo_state;

/* CylinderIntersector::Intersect — main.cc:237-343 */
static int cyo_intersect(const float *verts, const float *radii, int test_cap, const float org[3], const float dir[3],
                         uint32_t range0, uint32_t range1, cyo_state *s, float *t_inout, uint32_t prim) {
  const float kEPS = 1.0e-6f;
  const float *p0 = &verts[3 * (2 * prim + 0)], *p1 = &verts[3 * (2 * prim + 1)];
  float r0, r1, tmax, rr, d[3], m[3], md, nd, dd, capT = FLT_MAX, nn, mn, A, k, C, B, root[2] = {0.0f, 0.0f};
  int hitCap = 0, nRet;
  if (prim < range0 || prim >= range1) return 0;
  r0 = radii[2 * prim + 0];
  r1 = radii[2 * prim + 1];
  tmax = *t_inout;
  rr = (r0 < r1) ? r1 : r0; /* std::max(r0, r1) */
  cyo_sub(d, p1, p0);
  cyo_sub(m, org, p0);
  md = cyo_dot(m, d);
  nd = cyo_dot(dir, d);
  dd = cyo_dot(d, d);
  if (test_cap) {
    float t01[3], dN0[3], dN1[3], rd[3];
    cyo_sub(t01, p0, p1);
    cyo_normalize(dN0, t01);
    dN1[0] = -dN0[0];
    dN1[1] = -dN0[1];
    dN1[2] = -dN0[2];
    cyo_normalize(rd, dir);
    if (fabs((double)cyo_dot(dir, dN0)) > kEPS) {
      const float p0D = -cyo_dot(p0, dN0), p1D = -cyo_dot(p1, dN1);
      const float p0T = -(cyo_dot(org, dN0) + p0D) / cyo_dot(rd, dN0);
      const float p1T = -(cyo_dot(org, dN1) + p1D) / cyo_dot(rd, dN1);
      float q0[3], q1[3], e0[3], e1[3], qp0Sqr, qp1Sqr;
      int kk;
      for (kk = 0; kk < 3; kk++) {
        q0[kk] = org[kk] + rd[kk] * p0T; /* ray_org_ + p0T * rd: operator*(T, real3) computes v * f */
        q1[kk] = org[kk] + rd[kk] * p1T;
      }
      cyo_sub(e0, q0, p0);
      cyo_sub(e1, q1, p1);
      qp0Sqr = cyo_dot(e0, e0);
      qp1Sqr = cyo_dot(e1, e1);
      if (p0T > 0.0 && p0T < tmax && (qp0Sqr < rr * rr)) {
        s->hit_cap = hitCap = 1;
        capT = p0T;
        *t_inout = capT;
        s->u_param = sqrtf(qp0Sqr);
        s->v_param = 0;
      }
      if (p1T > 0.0 && p1T < tmax && p1T < capT && (qp1Sqr < rr * rr)) {
        s->hit_cap = hitCap = 1;
        capT = p1T;
        *t_inout = capT;
        s->u_param = sqrtf(qp1Sqr);
        s->v_param = 1.0;
      }
    }
  }
  if (md <= 0.0 && nd <= 0.0) return hitCap;
  if (md >= dd && nd >= 0.0) return hitCap;
  nn = cyo_dot(dir, dir);
  mn = cyo_dot(m, dir);
  A = dd * nn - nd * nd;
  k = cyo_dot(m, m) - rr * rr;
  C = dd * k - md * md;
  B = dd * mn - nd * md;
  nRet = cyo_solve2e(root, A, B, C);
  if (nRet) {
    const float t = root[0];
    if (0 <= t && t <= tmax && t <= capT) {
      float sv = md + t * nd;
      sv /= dd;
      if (0 <= sv && sv <= 1) {
        s->hit_cap = hitCap = 0;
        *t_inout = t;
        s->u_param = 0;
        s->v_param = sv;
        return 1;
      }
    }
  }
  return hitCap;
}

static int cyo_traverse_one(const cyo_node *nodes, const uint32_t *indices, const float *verts, const float *radii,
                            int test_cap, const cyo_ray *ray, uint32_t range0, uint32_t range1, cyo_hit *out) {
  uint32_t stack[512], best = 0xFFFFFFFFu;
  int sp = 0, sign[3], k, hit;
  float inv[3], hit_t = ray->max_t, t_best = ray->max_t;
  cyo_state st = {0, 0.0f, 0.0f};
  stack[0] = 0;
  for (k = 0; k < 3; k++) {
    sign[k] = ray->dir[k] < 0.0f ? 1 : 0;
    inv[k] = cyo_safe_inv(ray->dir[k]);
  }
  while (sp >= 0) {
    const cyo_node *node = &nodes[stack[sp]];
    sp--;
    if (cyo_slab(ray->min_t, hit_t, node->bmin, node->bmax, ray->org, inv, sign)) {
      if (node->flag == 0) {
        const int near = sign[node->axis];
        stack[++sp] = node->data[1 - near];
        stack[++sp] = node->data[near];
      } else {
        uint32_t i;
        float t = t_best;
        int any = 0;
        for (i = 0; i < node->data[0]; i++) {
          const uint32_t prim = indices[node->data[1] + i];
          float local_t = t;
          if (cyo_intersect(verts, radii, test_cap, ray->org, ray->dir, range0, range1, &st, &local_t, prim)) {
            t = local_t;
            t_best = t;
            best = prim;
            any = 1;
          }
        }
        if (any) hit_t = t_best;
      }
    }
  }
  hit = t_best < ray->max_t;
  if (hit) { /* PostTraversal — main.cc:367-418 */
    const float *p0 = &verts[3 * (2 * best + 0)], *p1 = &verts[3 * (2 * best + 1)];
    const float v = st.v_param;
    float center[3], position[3], n[3], d01[3];
    cyo_sub(d01, p1, p0);
    for (k = 0; k < 3; k++) {
      center[k] = p0[k] + v * d01[k];
      position[k] = ray->org[k] + ray->dir[k] * t_best; /* ray_org_ + t_ * ray_dir_ */
    }
    if (st.hit_cap) {
      float c[3], pc[3];
      for (k = 0; k < 3; k++) c[k] = d01[k] * 0.5f + p0[k]; /* 0.5f * (p1 - p0) + p0 */
      cyo_normalize(n, d01);
      cyo_sub(pc, position, c);
      if (cyo_dot(pc, n) > 0.0) {
      } else {
        n[0] = -n[0];
        n[1] = -n[1];
        n[2] = -n[2];
      }
    } else {
      float pc[3];
      cyo_sub(pc, position, center);
      cyo_normalize(n, pc);
    }
    out->u = st.u_param;
    out->v = st.v_param;
    out->normal[0] = n[0];
    out->normal[1] = n[1];
    out->normal[2] = n[2];
    out->t = t_best;
    out->prim_id = best;
  } else {
    memset(out, 0, sizeof(*out));
    out->t = ray->max_t;
    out->prim_id = 0xFFFFFFFFu;
  }
  return hit;
}

void cyo_traverse(const void *nodes, const uint32_t *indices, const float *verts, const float *radii, int test_cap,
                  const void *rays, uint64_t n, uint32_t range0, uint32_t range1, void *hits, uint8_t *mask) {
  uint64_t i;
  for (i = 0; i < n; i++) {
    const int h = cyo_traverse_one((const cyo_node *)nodes, indices, verts, radii, test_cap, (const cyo_ray *)rays + i,
                                   range0, range1, (cyo_hit *)hits + i);
    if (mask) mask[i] = (uint8_t)h;
  }
}

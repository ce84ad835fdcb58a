/* oracle/sphere_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's sphere ("particle") custom primitive traced through
 * BVHAccel<float>::Traverse: examples/particle_primitive/main.cc (SphereIntersector::Intersect :174-236,
 * PostTraversal :262-277, GenerateRandomSpheres :295-325) over nanort.h (Traverse :2487-2556, TestLeafNode
 * :2374-2407, IntersectRayAABB :2285-2325, vsafe_inverse :442-461, vnormalize :383-398, vdot :410-412).
 * Pinned bit-for-bit against the unmodified example (oracle/ref_sphere_shim.cc) by
 * tests/test_sphere_oracle.py; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it. */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE /* M_PI under -std=c99 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  float bmin[3], bmax[3];
  int32_t flag, axis;
  uint32_t data[2];
} spo_node; /* nanort::BVHNode<float>, nanort.h:498-550 */
typedef struct {
  float org[3], dir[3], min_t, max_t;
  uint32_t type;
} spo_ray; /* nanort.h:474-496 */
typedef struct {
  float u, v, t;
  uint32_t prim_id;
} spo_hit; /* SphereIntersection, main.cc:149-159 (same layout as TriangleIntersection) */

/* vsafe_inverse, the non-C++11 arm — nanort.h:442-461 */
static float spo_safe_inv(float v) {
  if (fabsf(v) < FLT_EPSILON) return INFINITY * ((v < 0.0f) ? -1.0f : 1.0f);
  return 1.0f / v;
}

/* IntersectRayAABB — nanort.h:2285-2325 */
static int spo_slab(float min_t, float max_t, const float bmin[3], const float bmax[3], const float org[3],
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

static float spo_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* SphereIntersector::Intersect — main.cc:174-236 */
static int spo_intersect(const float *centers, const float *radii, const float org[3], const float dir[3],
                         uint32_t range0, uint32_t range1, float *t_inout, uint32_t prim) {
  float oc[3], a, b, c, disc, t0, t1, t;
  const float radius = radii[prim];
  if (prim < range0 || prim >= range1) return 0;
  oc[0] = org[0] - centers[3 * prim + 0];
  oc[1] = org[1] - centers[3 * prim + 1];
  oc[2] = org[2] - centers[3 * prim + 2];
  a = spo_dot(dir, dir);
  b = 2.0f * spo_dot(dir, oc);
  c = spo_dot(oc, oc) - radius * radius;
  disc = b * b - 4.0f * a * c;
  if (disc < 0.0f) {
    return 0;
  } else if (fabsf(disc) < FLT_EPSILON) {
    t0 = t1 = -0.5f * (b / a);
  } else {
    const float ds = sqrtf(disc);
    float q;
    if (b < 0)
      q = (-b - ds) / 2.0f;
    else
      q = (-b + ds) / 2.0f;
    t0 = q / a;
    t1 = c / q;
  }
  if (t0 > t1) {
    const float tmp = t0;
    t0 = t1;
    t1 = tmp;
  }
  if (t1 < 0) return 0;
  t = (t0 < 0) ? t1 : t0;
  if (t > *t_inout) return 0;
  *t_inout = t;
  return 1;
}

/* Traverse + TestLeafNode + PostTraversal for one ray */
static int spo_traverse_one(const spo_node *nodes, const uint32_t *indices, const float *centers,
                            const float *radii, const spo_ray *ray, uint32_t range0, uint32_t range1,
                            spo_hit *out) {
  uint32_t stack[512], best = 0xFFFFFFFFu;
  int sp = 0, sign[3], k, hit;
  float inv[3], hit_t = ray->max_t, t_best = ray->max_t; /* Update(hit_t, -1): nanort.h:2501 */
  stack[0] = 0;
  for (k = 0; k < 3; k++) {
    sign[k] = ray->dir[k] < 0.0f ? 1 : 0;
    inv[k] = spo_safe_inv(ray->dir[k]);
  }
  while (sp >= 0) {
    const spo_node *node = &nodes[stack[sp]];
    sp--;
    if (spo_slab(ray->min_t, hit_t, node->bmin, node->bmax, ray->org, inv, sign)) {
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
          if (spo_intersect(centers, radii, ray->org, ray->dir, range0, range1, &local_t, prim)) {
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
  hit = t_best < ray->max_t; /* strict: nanort.h:2552 */
  if (hit) {                 /* PostTraversal — main.cc:262-277 */
    float n[3], len;
    for (k = 0; k < 3; k++) n[k] = (ray->org[k] + t_best * ray->dir[k]) - centers[3 * best + k];
    len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (fabsf(len) > FLT_EPSILON) {
      const float inv_len = 1.0f / len;
      n[0] *= inv_len;
      n[1] *= inv_len;
      n[2] *= inv_len;
    }
    out->t = t_best;
    out->prim_id = best;
    out->u = (float)(atan2((double)n[0], (double)n[2]) + M_PI) * 0.5f * (float)(1.0 / M_PI);
    out->v = (float)(acos((double)n[1]) / M_PI);
  } else { /* the reference leaves *isect untouched; the oracle writes a fixed miss record */
    out->u = out->v = 0.0f;
    out->t = ray->max_t;
    out->prim_id = 0xFFFFFFFFu;
  }
  return hit;
}

void spo_traverse(const void *nodes, const uint32_t *indices, const float *centers, const float *radii,
                  const void *rays, uint64_t n, uint32_t range0, uint32_t range1, void *hits, uint8_t *mask) {
  uint64_t i;
  for (i = 0; i < n; i++) {
    const int h = spo_traverse_one((const spo_node *)nodes, indices, centers, radii, (const spo_ray *)rays + i,
                                   range0, range1, (spo_hit *)hits + i);
    if (mask) mask[i] = (uint8_t)h;
  }
}

// nanort_amd/csrc/scene.hip — two-level (instanced) traversal on the GPU: SURVEY.md §8(f) row 3.
//
// Replaces nanosg::Scene<float, M>::Commit / Traverse (reference examples/nanosg/nanosg.h:700-870) and the
// BVHAccel::ListNodeIntersections it rests on (reference nanort.h:2608-2692).  Structure:
//
//   k_scene_list     per ray: which node boxes does the ray enter, sorted by entry distance, at most 64
//                    (the listing does not depend on the shape of the reference's top-level BVH: every
//                    ancestor box contains the leaf box and the slab arithmetic is monotone, so it is a scan
//                    over the node table);
//   for list position j (front to back) — one host synchronisation per round, independent of the number of nodes:
//     k_scene_count    per node: how many rays have it as their j-th entry and survive the early cull
//                      (t_nearest < t_min, nanosg.h:795);
//     k_scene_gather   those rays, compacted into one segment per node and transformed into the node's space;
//     for every node with a non-empty segment:
//       k_traverse_wide  the single-level kernel, unchanged, over the node's own tree (nrtTraverseBatchDevice);
//       k_scene_apply    world-space distance of each local hit, strict-nearer update of the ray's result.
//
// The per-node arithmetic (Matrix::Mult / Inverse / MultV, XformBoundingBox, the two slab tests) follows the
// reference operation for operation; this file is compiled with the same no-contraction / IEEE flags.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct NodeDev { // per-instance table in HBM
  float xbmin[3], xbmax[3]; // world AABB
  float inv_xform[4][4];    // world -> local (points)
  float inv_xform33[4][4];  // world -> local (directions)
  float xform[4][4];        // local -> world
};

constexpr int kMaxList = 64; // kMaxIntersections, nanosg.h:782

// Matrix::MultV — nanosg.h:232-240
__host__ __device__ inline void mult_v(float dst[3], const float m[4][4], const float v[3]) {
  const float t0 = m[0][0] * v[0] + m[1][0] * v[1] + m[2][0] * v[2] + m[3][0];
  const float t1 = m[0][1] * v[0] + m[1][1] * v[1] + m[2][1] * v[2] + m[3][1];
  const float t2 = m[0][2] * v[0] + m[1][2] * v[1] + m[2][2] * v[2] + m[3][2];
  dst[0] = t0;
  dst[1] = t1;
  dst[2] = t2;
}

// Does the ray enter node box `nd`, and over which interval?  First the BVH leaf's robust test
// (IntersectRayAABB, nanort.h:2285-2325, hit_t == ray.max_t throughout ListNodeIntersections), then
// NodeBBoxIntersector::Intersect (nanosg.h:603-639: plain reciprocal, no MaxMult, no clipping).
__device__ inline bool node_interval(const nrt_ray_f32 &r, const NodeDev &nd, float &t_min_out) {
  float tmin = r.min_t, tmax = r.max_t;
  float tn[3], tf[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float d = r.dir[k];
    const bool neg = d < 0.0f;
    float inv_safe;
    if (__builtin_fabsf(d) < 1.1920928955078125e-07f)
      inv_safe = __builtin_huge_valf() * (neg ? -1.0f : 1.0f);
    else
      inv_safe = 1.0f / d;
    const float lo = neg ? nd.xbmax[k] : nd.xbmin[k], hi = neg ? nd.xbmin[k] : nd.xbmax[k];
    const float t0 = (lo - r.org[k]) * inv_safe;
    const float t1 = (hi - r.org[k]) * inv_safe * 1.00000024f;
    tmin = (t0 > tmin) ? t0 : tmin;
    tmax = (t1 < tmax) ? t1 : tmax;
    const float inv = 1.0f / d;
    tn[k] = (lo - r.org[k]) * inv;
    tf[k] = (hi - r.org[k]) * inv;
  }
  if (!(tmin <= tmax)) return false;
  float a = (tn[1] > tn[0]) ? tn[1] : tn[0];
  a = (tn[2] > a) ? tn[2] : a;
  float b = (tf[1] < tf[0]) ? tf[1] : tf[0];
  b = (tf[2] < b) ? tf[2] : b;
  if (!(a <= b)) return false;
  t_min_out = a;
  return true;
}

// list_t / list_node: [kMaxList or num_nodes][n] (entry-major, so lane-consecutive accesses coalesce)
__global__ __launch_bounds__(256) void k_scene_list(const nrt_ray_f32 *__restrict__ rays, uint32_t n,
                                                    const NodeDev *__restrict__ nodes, uint32_t num_nodes, uint32_t cap,
                                                    float *__restrict__ list_t, uint32_t *__restrict__ list_node,
                                                    uint32_t *__restrict__ count, float *__restrict__ best_t,
                                                    nrt_scene_hit_f32 *__restrict__ best, uint32_t *max_count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const nrt_ray_f32 r = rays[i];
  uint32_t cnt = 0;
  for (uint32_t k = 0; k < num_nodes; k++) {
    float t;
    if (!node_interval(r, nodes[k], t)) continue;
    // insertion by (t_min, node id) into the sorted prefix kept in global memory; beyond `cap` the farthest drops
    uint32_t pos = cnt < cap ? cnt : cap;
    while (pos > 0) {
      const float pt = list_t[(size_t)(pos - 1) * n + i];
      if (pt <= t) break; // equal t_min: lower node id first (ids ascend with k)
      if (pos < cap) {
        list_t[(size_t)pos * n + i] = pt;
        list_node[(size_t)pos * n + i] = list_node[(size_t)(pos - 1) * n + i];
      }
      pos--;
    }
    if (pos < cap) {
      list_t[(size_t)pos * n + i] = t;
      list_node[(size_t)pos * n + i] = k;
    }
    if (cnt < cap) cnt++;
  }
  count[i] = cnt;
  best_t[i] = 3.402823466e+38f; // t_nearest = numeric_limits<T>::max(), nanosg.h:787
  nrt_scene_hit_f32 h;
  h.t = r.max_t;
  h.u = 0.0f;
  h.v = 0.0f;
  h.prim_id = 0xFFFFFFFFu;
  h.node_id = 0xFFFFFFFFu;
  best[i] = h;
  if (cnt) atomicMax(max_count, cnt);
}

// Does ray i visit its j-th listed node in round j?  (early cull, nanosg.h:795).  A ray has ONE node at list position j,
// so within a round no ray is handled twice and best_t[i] only changes through the ray's own node: the decisions of a
// whole round can be taken up front, for all nodes at once.
__device__ inline bool round_take(uint32_t i, uint32_t n, uint32_t j, const float *__restrict__ list_t,
                                  const uint32_t *__restrict__ list_node, const uint32_t *__restrict__ count,
                                  const float *__restrict__ best_t, uint32_t &node) {
  node = 0xFFFFFFFFu;
  if (i >= n || j >= count[i]) return false;
  if (best_t[i] < list_t[(size_t)j * n + i]) return false;
  node = list_node[(size_t)j * n + i];
  return true;
}

// One atomic per (wave, distinct node): lanes of a wave mostly share a node.  Returns the lane's rank among the lanes
// of its wave that hold the same node, and through `base` what the group's leader got back from the atomic.
__device__ inline uint32_t wave_group_add(bool take, uint32_t node, uint32_t *__restrict__ counters, uint32_t &base) {
  const unsigned lane = threadIdx.x & 63u;
  uint32_t rank = 0;
  base = 0;
  unsigned long long todo = __ballot(take);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const uint32_t lnode = __shfl(node, leader);
    const unsigned long long grp = __ballot(take && node == lnode);
    uint32_t b = 0;
    if ((int)lane == leader) b = atomicAdd(&counters[lnode], (uint32_t)__builtin_popcountll(grp));
    b = __shfl(b, leader);
    if (take && node == lnode) {
      base = b;
      rank = (uint32_t)__builtin_popcountll(grp & ((1ull << lane) - 1ull));
    }
    todo &= ~grp;
  }
  return rank;
}

// round j, step 1: how many rays go to each node
__global__ __launch_bounds__(256) void k_scene_count(uint32_t n, uint32_t j, const float *__restrict__ list_t,
                                                     const uint32_t *__restrict__ list_node,
                                                     const uint32_t *__restrict__ count, const float *__restrict__ best_t,
                                                     uint32_t *__restrict__ node_count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  uint32_t node, base;
  const bool take = round_take(i, n, j, list_t, list_node, count, best_t, node);
  wave_group_add(take, node, node_count, base);
}

// round j, step 2: compact the rays of every node into its segment [node_offset[k], node_offset[k] + node_count[k]) and
// transform them into the node's space.  node_cursor starts at zero.
__global__ __launch_bounds__(256) void k_scene_gather(const nrt_ray_f32 *__restrict__ rays, uint32_t n, uint32_t j,
                                                      const NodeDev *__restrict__ nodes, const float *__restrict__ list_t,
                                                      const uint32_t *__restrict__ list_node,
                                                      const uint32_t *__restrict__ count,
                                                      const float *__restrict__ best_t,
                                                      const uint32_t *__restrict__ node_offset,
                                                      uint32_t *__restrict__ node_cursor, uint32_t *__restrict__ sel_index,
                                                      nrt_ray_f32 *__restrict__ local_rays) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  uint32_t node, base;
  const bool take = round_take(i, n, j, list_t, list_node, count, best_t, node);
  const uint32_t rank = wave_group_add(take, node, node_cursor, base);
  if (!take) return;
  const uint32_t slot = node_offset[node] + base + rank;
  const nrt_ray_f32 r = rays[i];
  const NodeDev &nd = nodes[node];
  nrt_ray_f32 lr;
  mult_v(lr.org, nd.inv_xform, r.org);   // nanosg.h:807
  mult_v(lr.dir, nd.inv_xform33, r.dir); // nanosg.h:808
  lr.min_t = 0.0f;                       // Ray() defaults (nanort.h:477-487): the world interval is not propagated
  lr.max_t = 3.402823466e+38f;
  lr.type = 0;
  local_rays[slot] = lr;
  sel_index[slot] = i;
}

__global__ __launch_bounds__(256) void k_scene_apply(const nrt_ray_f32 *__restrict__ rays, uint32_t node,
                                                     const NodeDev *__restrict__ nodes, uint32_t sel_count,
                                                     const uint32_t *__restrict__ sel_index,
                                                     const nrt_ray_f32 *__restrict__ local_rays,
                                                     const nrt_hit_f32 *__restrict__ local_hits,
                                                     const uint8_t *__restrict__ local_mask, float *__restrict__ best_t,
                                                     nrt_scene_hit_f32 *__restrict__ best) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= sel_count || !local_mask[s]) return;
  const uint32_t i = sel_index[s];
  const nrt_ray_f32 lr = local_rays[s];
  const nrt_hit_f32 lh = local_hits[s];
  float lp[3], wp[3];
#pragma unroll
  for (int k = 0; k < 3; k++) lp[k] = lr.org[k] + lh.t * lr.dir[k]; // nanosg.h:823-825
  mult_v(wp, nodes[node].xform, lp);
  const float px = wp[0] - rays[i].org[0], py = wp[1] - rays[i].org[1], pz = wp[2] - rays[i].org[2];
  const float t_world = __builtin_sqrtf(px * px + py * py + pz * pz); // vlength, nanort.h:383-385
  if (t_world < best_t[i]) {                                           // strict, nanosg.h:838
    best_t[i] = t_world;
    nrt_scene_hit_f32 h;
    h.t = t_world;
    h.u = lh.u;
    h.v = lh.v;
    h.prim_id = lh.prim_id;
    h.node_id = node;
    best[i] = h;
  }
}

__global__ __launch_bounds__(256) void k_scene_mask(const float *__restrict__ best_t, uint32_t n, uint8_t *__restrict__ mask) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) mask[i] = best_t[i] < 3.402823466e+38f ? 1 : 0;
}

// ---- host-side restatement of the per-node update (nanosg.h:92-241, 246-302, 397-437) ----------------
void mat_mult(float dst[4][4], const float m0[4][4], const float m1[4][4]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      dst[i][j] = 0;
      for (int k = 0; k < 4; ++k) dst[i][j] += m0[k][j] * m1[i][k];
    }
}

void mat_inverse(float m[4][4]) { // Cramer's rule, same operation order as the reference
  float tmp[12], s[16], det;
  for (int i = 0; i < 4; i++) {
    s[i] = m[i][0];
    s[i + 4] = m[i][1];
    s[i + 8] = m[i][2];
    s[i + 12] = m[i][3];
  }
  tmp[0] = s[10] * s[15]; tmp[1] = s[11] * s[14]; tmp[2] = s[9] * s[15]; tmp[3] = s[11] * s[13];
  tmp[4] = s[9] * s[14]; tmp[5] = s[10] * s[13]; tmp[6] = s[8] * s[15]; tmp[7] = s[11] * s[12];
  tmp[8] = s[8] * s[14]; tmp[9] = s[10] * s[12]; tmp[10] = s[8] * s[13]; tmp[11] = s[9] * s[12];
  m[0][0] = tmp[0] * s[5] + tmp[3] * s[6] + tmp[4] * s[7];
  m[0][0] -= tmp[1] * s[5] + tmp[2] * s[6] + tmp[5] * s[7];
  m[0][1] = tmp[1] * s[4] + tmp[6] * s[6] + tmp[9] * s[7];
  m[0][1] -= tmp[0] * s[4] + tmp[7] * s[6] + tmp[8] * s[7];
  m[0][2] = tmp[2] * s[4] + tmp[7] * s[5] + tmp[10] * s[7];
  m[0][2] -= tmp[3] * s[4] + tmp[6] * s[5] + tmp[11] * s[7];
  m[0][3] = tmp[5] * s[4] + tmp[8] * s[5] + tmp[11] * s[6];
  m[0][3] -= tmp[4] * s[4] + tmp[9] * s[5] + tmp[10] * s[6];
  m[1][0] = tmp[1] * s[1] + tmp[2] * s[2] + tmp[5] * s[3];
  m[1][0] -= tmp[0] * s[1] + tmp[3] * s[2] + tmp[4] * s[3];
  m[1][1] = tmp[0] * s[0] + tmp[7] * s[2] + tmp[8] * s[3];
  m[1][1] -= tmp[1] * s[0] + tmp[6] * s[2] + tmp[9] * s[3];
  m[1][2] = tmp[3] * s[0] + tmp[6] * s[1] + tmp[11] * s[3];
  m[1][2] -= tmp[2] * s[0] + tmp[7] * s[1] + tmp[10] * s[3];
  m[1][3] = tmp[4] * s[0] + tmp[9] * s[1] + tmp[10] * s[2];
  m[1][3] -= tmp[5] * s[0] + tmp[8] * s[1] + tmp[11] * s[2];
  tmp[0] = s[2] * s[7]; tmp[1] = s[3] * s[6]; tmp[2] = s[1] * s[7]; tmp[3] = s[3] * s[5];
  tmp[4] = s[1] * s[6]; tmp[5] = s[2] * s[5]; tmp[6] = s[0] * s[7]; tmp[7] = s[3] * s[4];
  tmp[8] = s[0] * s[6]; tmp[9] = s[2] * s[4]; tmp[10] = s[0] * s[5]; tmp[11] = s[1] * s[4];
  m[2][0] = tmp[0] * s[13] + tmp[3] * s[14] + tmp[4] * s[15];
  m[2][0] -= tmp[1] * s[13] + tmp[2] * s[14] + tmp[5] * s[15];
  m[2][1] = tmp[1] * s[12] + tmp[6] * s[14] + tmp[9] * s[15];
  m[2][1] -= tmp[0] * s[12] + tmp[7] * s[14] + tmp[8] * s[15];
  m[2][2] = tmp[2] * s[12] + tmp[7] * s[13] + tmp[10] * s[15];
  m[2][2] -= tmp[3] * s[12] + tmp[6] * s[13] + tmp[11] * s[15];
  m[2][3] = tmp[5] * s[12] + tmp[8] * s[13] + tmp[11] * s[14];
  m[2][3] -= tmp[4] * s[12] + tmp[9] * s[13] + tmp[10] * s[14];
  m[3][0] = tmp[2] * s[10] + tmp[5] * s[11] + tmp[1] * s[9];
  m[3][0] -= tmp[4] * s[11] + tmp[0] * s[9] + tmp[3] * s[10];
  m[3][1] = tmp[8] * s[11] + tmp[0] * s[8] + tmp[7] * s[10];
  m[3][1] -= tmp[6] * s[10] + tmp[9] * s[11] + tmp[1] * s[8];
  m[3][2] = tmp[6] * s[9] + tmp[11] * s[11] + tmp[3] * s[8];
  m[3][2] -= tmp[10] * s[11] + tmp[2] * s[8] + tmp[7] * s[9];
  m[3][3] = tmp[10] * s[10] + tmp[4] * s[8] + tmp[9] * s[9];
  m[3][3] -= tmp[8] * s[9] + tmp[11] * s[0] + tmp[5] * s[8];
  det = s[0] * m[0][0] + s[1] * m[0][1] + s[2] * m[0][2] + s[3] * m[0][3];
  det = 1.0f / det;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) m[j][i] *= det;
}

void node_update(const float local[4][4], const float lbmin[3], const float lbmax[3], NodeDev *out) {
  float ident[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) ident[i][j] = (i == j) ? 1.0f : 0.0f;
  mat_mult(out->xform, ident, local);
  float xb[8][3];
  for (int i = 0; i < 8; i++) {
    const float b[3] = {(i & 1) ? lbmax[0] : lbmin[0], (i & 2) ? lbmax[1] : lbmin[1], (i & 4) ? lbmax[2] : lbmin[2]};
    mult_v(xb[i], out->xform, b);
  }
  for (int k = 0; k < 3; k++) out->xbmin[k] = out->xbmax[k] = xb[0][k];
  for (int i = 1; i < 8; i++)
    for (int k = 0; k < 3; k++) {
      out->xbmin[k] = std::min(xb[i][k], out->xbmin[k]);
      out->xbmax[k] = std::max(xb[i][k], out->xbmax[k]);
    }
  memcpy(out->inv_xform, out->xform, sizeof(out->xform));
  mat_inverse(out->inv_xform);
  memcpy(out->inv_xform33, out->xform, sizeof(out->xform));
  out->inv_xform33[3][0] = out->inv_xform33[3][1] = out->inv_xform33[3][2] = 0.0f;
  mat_inverse(out->inv_xform33);
}

thread_local std::string g_scene_create_error;

} // namespace

struct nrt_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  struct Inst {
    nrt_ctx *mesh;
    float local[4][4];
  };
  std::vector<Inst> insts;
  std::vector<NodeDev> host_nodes;
  bool committed = false;
  nrt::DevBuf d_nodes, d_rays, d_list_t, d_list_node, d_count, d_best_t, d_best, d_sel_index, d_local_rays, d_local_hits,
      d_local_mask, d_mask, d_scalars, d_node_counters; // d_scalars: [0] max_count
  std::vector<uint32_t> h_node_counters;
};

static nrt_status sfail(nrt_scene *s, nrt_status st, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s)
    s->err = buf;
  else
    g_scene_create_error = buf;
  return st;
}

#define SCHK(s, call)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      return sfail((s), NRT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" {

nrt_status nrtSceneCreate(int device, nrt_scene **out) {
  if (!out) return sfail(nullptr, NRT_ERR_INVALID, "nrtSceneCreate: out == NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return sfail(nullptr, NRT_ERR_DEVICE, "nrtSceneCreate: no HIP device visible");
  if (device < 0 || device >= ndev) return sfail(nullptr, NRT_ERR_INVALID, "nrtSceneCreate: device %d out of range", device);
  nrt_scene *s = new nrt_scene();
  s->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
    delete s;
    return sfail(nullptr, NRT_ERR_DEVICE, "nrtSceneCreate: stream creation failed");
  }
  *out = s;
  return NRT_OK;
}

void nrtSceneDestroy(nrt_scene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  nrt::DevBuf *bufs[] = {&s->d_nodes,     &s->d_rays,       &s->d_list_t,     &s->d_list_node,  &s->d_count,
                         &s->d_best_t,    &s->d_best,       &s->d_sel_index,  &s->d_local_rays, &s->d_local_hits,
                         &s->d_local_mask, &s->d_mask,       &s->d_scalars,    &s->d_node_counters};
  for (nrt::DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  (void)hipStreamDestroy(s->stream);
  delete s;
}

const char *nrtSceneLastError(const nrt_scene *s) { return s ? s->err.c_str() : g_scene_create_error.c_str(); }

nrt_status nrtSceneAddNode_f32(nrt_scene *s, nrt_ctx *mesh, const float local_xform[16], uint32_t *node_id_out) {
  if (!s) return NRT_ERR_INVALID;
  if (!mesh || !local_xform) return sfail(s, NRT_ERR_INVALID, "nrtSceneAddNode: NULL mesh or transform");
  uint64_t nn = 0, ni = 0;
  if (nrtTreeSize(mesh, &nn, &ni) != NRT_OK || nn == 0)
    return sfail(s, NRT_ERR_INVALID, "nrtSceneAddNode: the mesh context has no tree (call nrtBuild first)");
  nrt_scene::Inst in;
  in.mesh = mesh;
  memcpy(in.local, local_xform, sizeof(in.local));
  s->insts.push_back(in);
  s->committed = false;
  if (node_id_out) *node_id_out = (uint32_t)s->insts.size() - 1;
  return NRT_OK;
}

nrt_status nrtSceneCommit(nrt_scene *s) {
  if (!s) return NRT_ERR_INVALID;
  if (s->insts.empty()) return sfail(s, NRT_ERR_EMPTY, "nrtSceneCommit: empty scene (the reference's Commit() returns false)");
  SCHK(s, hipSetDevice(s->device));
  s->host_nodes.resize(s->insts.size());
  for (size_t i = 0; i < s->insts.size(); i++) {
    // local AABB = the root box of the node's tree (accel_.BoundingBox, nanosg.h:411)
    uint64_t nn = 0, ni = 0;
    nrtTreeSize(s->insts[i].mesh, &nn, &ni);
    std::vector<nrt_node_f32> nodes((size_t)nn);
    if (nrtGetTree_f32(s->insts[i].mesh, nodes.data(), nullptr) != NRT_OK)
      return sfail(s, NRT_ERR_PRECISION, "nrtSceneCommit: node %zu is not a built f32 mesh: %s", i, nrtLastError(s->insts[i].mesh));
    node_update(s->insts[i].local, nodes[0].bmin, nodes[0].bmax, &s->host_nodes[i]);
  }
  SCHK(s, nrt::devbuf_ensure(&s->d_nodes, s->host_nodes.size() * sizeof(NodeDev)));
  SCHK(s, hipMemcpy(s->d_nodes.p, s->host_nodes.data(), s->host_nodes.size() * sizeof(NodeDev), hipMemcpyHostToDevice));
  s->committed = true;
  return NRT_OK;
}

// Scene::GetBoundingBox (nanosg.h:761-769): the root box of the top-level tree, i.e. the plain union of the nodes'
// world boxes (nanort.h:1546-1567 pads nothing).
nrt_status nrtSceneBounds_f32(nrt_scene *s, float bmin[3], float bmax[3]) {
  if (!s || !bmin || !bmax) return NRT_ERR_INVALID;
  if (!s->committed) return sfail(s, NRT_ERR_INVALID, "nrtSceneBounds: commit the scene first");
  for (int k = 0; k < 3; k++) {
    bmin[k] = s->host_nodes[0].xbmin[k];
    bmax[k] = s->host_nodes[0].xbmax[k];
  }
  for (size_t i = 1; i < s->host_nodes.size(); i++)
    for (int k = 0; k < 3; k++) {
      bmin[k] = std::min(bmin[k], s->host_nodes[i].xbmin[k]);
      bmax[k] = std::max(bmax[k], s->host_nodes[i].xbmax[k]);
    }
  return NRT_OK;
}

// What Node::Update derives from the local transform (nanosg.h:397-437), for callers that finish the
// reference's Intersection record (P, Ns, Ng) on the host: xform, inv_xform, inv_xform33 and
// inv_transpose_xform33 = transpose(inv_xform33) (nanosg.h:430-432), 16 floats each, row-major T[4][4].
nrt_status nrtSceneNodeState_f32(nrt_scene *s, uint32_t node_id, float out[64]) {
  if (!s || !out) return NRT_ERR_INVALID;
  if (!s->committed) return sfail(s, NRT_ERR_INVALID, "nrtSceneNodeState: commit the scene first");
  if (node_id >= s->host_nodes.size()) return sfail(s, NRT_ERR_INVALID, "nrtSceneNodeState: node %u out of range", node_id);
  const NodeDev &nd = s->host_nodes[node_id];
  memcpy(out, nd.xform, 64);
  memcpy(out + 16, nd.inv_xform, 64);
  memcpy(out + 32, nd.inv_xform33, 64);
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) out[48 + 4 * j + i] = nd.inv_xform33[i][j];
  return NRT_OK;
}

} // extern "C"

// `device` = rays / hits_out / mask_out are device pointers (no PCIe traffic); the call itself stays synchronous: it
// reads one counter array back per list position to size the per-node launches.
static nrt_status scene_traverse(nrt_scene *s, const nrt_ray_f32 *rays, uint64_t n64, nrt_scene_hit_f32 *hits_out,
                                 uint8_t *mask_out, bool device) {
  if (!s) return NRT_ERR_INVALID;
  if (!s->committed) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: commit the scene first");
  if (n64 == 0) return NRT_OK;
  if (!rays || !hits_out) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: NULL rays/hits");
  if (n64 > 0x7FFFFFFFull) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: too many rays in one call");
  const uint32_t n = (uint32_t)n64, num_nodes = (uint32_t)s->insts.size();
  const uint32_t cap = std::min<uint32_t>(kMaxList, num_nodes);
  SCHK(s, hipSetDevice(s->device));
  if (!device) SCHK(s, nrt::devbuf_ensure(&s->d_rays, (size_t)n * sizeof(nrt_ray_f32)));
  SCHK(s, nrt::devbuf_ensure(&s->d_list_t, (size_t)cap * n * sizeof(float)));
  SCHK(s, nrt::devbuf_ensure(&s->d_list_node, (size_t)cap * n * sizeof(uint32_t)));
  SCHK(s, nrt::devbuf_ensure(&s->d_count, (size_t)n * sizeof(uint32_t)));
  SCHK(s, nrt::devbuf_ensure(&s->d_best_t, (size_t)n * sizeof(float)));
  if (!device) SCHK(s, nrt::devbuf_ensure(&s->d_best, (size_t)n * sizeof(nrt_scene_hit_f32)));
  SCHK(s, nrt::devbuf_ensure(&s->d_sel_index, (size_t)n * sizeof(uint32_t)));
  SCHK(s, nrt::devbuf_ensure(&s->d_local_rays, (size_t)n * sizeof(nrt_ray_f32)));
  SCHK(s, nrt::devbuf_ensure(&s->d_local_hits, (size_t)n * sizeof(nrt_hit_f32)));
  SCHK(s, nrt::devbuf_ensure(&s->d_local_mask, (size_t)n));
  if (!device || !mask_out) SCHK(s, nrt::devbuf_ensure(&s->d_mask, (size_t)n));
  SCHK(s, nrt::devbuf_ensure(&s->d_scalars, 64));
  uint32_t *d_max_count = (uint32_t *)s->d_scalars.p;
  const NodeDev *d_nodes = (const NodeDev *)s->d_nodes.p;
  const nrt_ray_f32 *d_rays = device ? rays : (const nrt_ray_f32 *)s->d_rays.p;
  nrt_scene_hit_f32 *d_best = device ? hits_out : (nrt_scene_hit_f32 *)s->d_best.p;
  uint8_t *d_mask = (device && mask_out) ? mask_out : (uint8_t *)s->d_mask.p;
  const unsigned grid = (n + 255u) / 256u;

  if (!device) SCHK(s, hipMemcpyAsync(s->d_rays.p, rays, (size_t)n * sizeof(nrt_ray_f32), hipMemcpyHostToDevice, s->stream));
  SCHK(s, hipMemsetAsync(s->d_scalars.p, 0, 64, s->stream));
  hipLaunchKernelGGL(k_scene_list, dim3(grid), dim3(256), 0, s->stream, d_rays, n, d_nodes, num_nodes, cap,
                     (float *)s->d_list_t.p, (uint32_t *)s->d_list_node.p, (uint32_t *)s->d_count.p, (float *)s->d_best_t.p,
                     d_best, d_max_count);
  SCHK(s, hipGetLastError());
  uint32_t max_count = 0;
  SCHK(s, hipMemcpyAsync(&max_count, d_max_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
  SCHK(s, hipStreamSynchronize(s->stream));

  // per-node counters of a round: [0, N) counts, [N, 2N) segment offsets, [2N, 3N) gather cursors
  SCHK(s, nrt::devbuf_ensure(&s->d_node_counters, (size_t)3 * num_nodes * sizeof(uint32_t)));
  uint32_t *d_node_count = (uint32_t *)s->d_node_counters.p, *d_node_offset = d_node_count + num_nodes,
           *d_node_cursor = d_node_offset + num_nodes;
  s->h_node_counters.resize((size_t)3 * num_nodes);
  uint32_t *h_count = s->h_node_counters.data(), *h_offset = h_count + num_nodes;
  for (uint32_t j = 0; j < max_count; j++) {
    SCHK(s, hipMemsetAsync(d_node_count, 0, (size_t)3 * num_nodes * sizeof(uint32_t), s->stream));
    hipLaunchKernelGGL(k_scene_count, dim3(grid), dim3(256), 0, s->stream, n, j, (const float *)s->d_list_t.p,
                       (const uint32_t *)s->d_list_node.p, (const uint32_t *)s->d_count.p, (const float *)s->d_best_t.p,
                       d_node_count);
    SCHK(s, hipGetLastError());
    SCHK(s, hipMemcpyAsync(h_count, d_node_count, (size_t)num_nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SCHK(s, hipStreamSynchronize(s->stream)); // the only host round trip of the round
    uint32_t total = 0;
    for (uint32_t k = 0; k < num_nodes; k++) {
      h_offset[k] = total;
      total += h_count[k];
    }
    if (total == 0) continue;
    SCHK(s, hipMemcpyAsync(d_node_offset, h_offset, (size_t)num_nodes * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_scene_gather, dim3(grid), dim3(256), 0, s->stream, d_rays, n, j, d_nodes, (const float *)s->d_list_t.p,
                       (const uint32_t *)s->d_list_node.p, (const uint32_t *)s->d_count.p, (const float *)s->d_best_t.p,
                       (const uint32_t *)d_node_offset, d_node_cursor, (uint32_t *)s->d_sel_index.p,
                       (nrt_ray_f32 *)s->d_local_rays.p);
    SCHK(s, hipGetLastError());
    for (uint32_t k = 0; k < num_nodes; k++) {
      const uint32_t m = h_count[k], off = h_offset[k];
      if (m == 0) continue;
      const nrt_ray_f32 *lr = (const nrt_ray_f32 *)s->d_local_rays.p + off;
      nrt_hit_f32 *lh = (nrt_hit_f32 *)s->d_local_hits.p + off;
      uint8_t *lm = (uint8_t *)s->d_local_mask.p + off;
      // the single-level kernel over node k's own tree, default trace options (nanosg.h:817)
      if (nrtTraverseBatchDevice_f32(s->insts[k].mesh, lr, m, nullptr, lh, lm, s->stream) != NRT_OK)
        return sfail(s, NRT_ERR_DEVICE, "nrtSceneTraverseBatch: node %u: %s", k, nrtLastError(s->insts[k].mesh));
      hipLaunchKernelGGL(k_scene_apply, dim3((m + 255u) / 256u), dim3(256), 0, s->stream, d_rays, k, d_nodes, m,
                         (const uint32_t *)s->d_sel_index.p + off, lr, (const nrt_hit_f32 *)lh, (const uint8_t *)lm,
                         (float *)s->d_best_t.p, d_best);
      SCHK(s, hipGetLastError());
    }
  }
  if (!device || mask_out) {
    hipLaunchKernelGGL(k_scene_mask, dim3(grid), dim3(256), 0, s->stream, (const float *)s->d_best_t.p, n, d_mask);
    SCHK(s, hipGetLastError());
  }
  if (!device) {
    SCHK(s, hipMemcpyAsync(hits_out, s->d_best.p, (size_t)n * sizeof(nrt_scene_hit_f32), hipMemcpyDeviceToHost, s->stream));
    if (mask_out) SCHK(s, hipMemcpyAsync(mask_out, s->d_mask.p, (size_t)n, hipMemcpyDeviceToHost, s->stream));
  }
  SCHK(s, hipStreamSynchronize(s->stream));
  return NRT_OK;
}

extern "C" {

nrt_status nrtSceneTraverseBatch_f32(nrt_scene *s, const nrt_ray_f32 *rays, uint64_t n, nrt_scene_hit_f32 *hits_out,
                                     uint8_t *mask_out) {
  return scene_traverse(s, rays, n, hits_out, mask_out, false);
}

nrt_status nrtSceneTraverseBatchDevice_f32(nrt_scene *s, const nrt_ray_f32 *d_rays, uint64_t n, nrt_scene_hit_f32 *d_hits_out,
                                           uint8_t *d_mask_out) {
  return scene_traverse(s, d_rays, n, d_hits_out, d_mask_out, true);
}

} // extern "C"

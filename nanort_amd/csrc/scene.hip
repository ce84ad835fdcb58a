// nanort_amd/csrc/scene.hip — two-level (instanced) traversal on the GPU: SURVEY.md §8(f) row 3.
//
// Replaces nanosg::Scene<float, M>::Commit / Traverse (reference examples/nanosg/nanosg.h:700-870) and the
// BVHAccel::ListNodeIntersections it rests on (reference nanort.h:2608-2692).  Two paths give the same records:
//
//   THE SINGLE-PASS WALK (scenes of kWalkMinNodes nodes or more): k_scene_walk (traverse.hip) walks the top-level tree and, on
//   reaching an instance, the instance's own tree in the same lane on the same stack — no per-ray list; a per-ray certificate
//   says when the record is provably the reference's, the other rays (few) are appended to a list and go through
//   THE LISTING PATH (small scenes; the rays handed over; tunable single_pass = 0):
//
//   k_scene_list*     per ray: which node boxes does the ray enter, sorted by entry distance, at most 64 — a walk
//                     over the TOP-LEVEL BVH of the instances' world boxes (built on the GPU by the ordinary builder
//                     with min_leaf_primitives = 1, as nanosg.h:730-735 does on the host).  The set of listed nodes
//                     does not depend on the shape of that tree (every ancestor box contains the leaf box and the slab
//                     arithmetic is monotone); scenes of a handful of nodes skip the tree (k_scene_list: a scan);
//   k_scene_trace     (traverse.hip) per ray: the reference's loop over that list — early cull, ray into the node's
//                     space, the node's own tree walked in the same lane, world distance, strict-nearer update —
//                     for the whole batch (or the subset handed over) in one launch: no per-node launches, no host round trips.
//
// The per-node arithmetic (Matrix::Mult / Inverse / MultV, XformBoundingBox, the two slab tests) follows the
// reference operation for operation; this file is compiled with the same no-contraction / IEEE flags.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#ifdef NRT_PROF
#include "../../include/nanort_hip_prof.h"
#endif

struct nrt_ctx;
nrt_status nrt_internal_tree_view(nrt_ctx *c, nrt::TreeViewF32 *out); // api.hip
uint64_t nrt_internal_generation(const nrt_ctx *c);                      // api.hip: counts the context's rebuilds
namespace nrt {
hipError_t launch_scene_trace(const SceneTraceArgs &args, unsigned grid, hipStream_t s); // traverse.hip
int scene_trace_blocks_per_cu();
hipError_t launch_scene_walk(const SceneWalkArgs &args, unsigned grid, hipStream_t s); // traverse.hip
int scene_walk_blocks_per_cu();
}

namespace {

struct NodeDev { // per-instance table in HBM
  float xbmin[3], xbmax[3]; // world AABB
  float inv_xform[4][4];    // world -> local (points)
  float inv_xform33[4][4];  // world -> local (directions)
  float xform[4][4];        // local -> world
};

constexpr int kMaxList = 64; // kMaxIntersections, nanosg.h:782
constexpr int kTopStack = 64; // per-ray stack of the top-level walk (a deeper top-level tree falls back to the scan)
constexpr uint32_t kScanMaxNodes = 8; // scenes of at most this many nodes are listed by the scan
constexpr unsigned kWalkBackoff = 15;   // listing-path calls after a batch of which the walk handed over more than a quarter
constexpr uint32_t kWalkMinNodes = 64; // scenes of at least this many nodes are traced by the single-pass walk (100 - 1 000 instances: 5-8 % faster than the
                                      // listing path, 10 000: 1.7x, 100 000: 9x; the 5-node fixture: 6-20 % slower — profiles/r04n_scene_walk.txt, r04t_*)

// Matrix::MultV — nanosg.h:232-240
__host__ __device__ inline void mult_v(float dst[3], const float m[4][4], const float v[3]) {
  const float t0 = m[0][0] * v[0] + m[1][0] * v[1] + m[2][0] * v[2] + m[3][0];
  const float t1 = m[0][1] * v[0] + m[1][1] * v[1] + m[2][1] * v[2] + m[3][1];
  const float t2 = m[0][2] * v[0] + m[1][2] * v[1] + m[2][2] * v[2] + m[3][2];
  dst[0] = t0;
  dst[1] = t1;
  dst[2] = t2;
}

// Does the ray enter node box `nd`, and over which interval?  nrt::scene_node_interval (common.h): the BVH leaf's robust test, then
// NodeBBoxIntersector::Intersect.
__device__ inline bool node_interval_box(const nrt_ray_f32 &r, const float xbmin[3], const float xbmax[3], float &t_min_out) {
  return nrt::scene_node_interval(r, xbmin, xbmax, t_min_out);
}
__device__ inline bool node_interval(const nrt_ray_f32 &r, const NodeDev &nd, float &t_min_out) {
  return nrt::scene_node_interval(r, nd.xbmin, nd.xbmax, t_min_out);
}

// A ray's candidate list while it is being collected: UNSORTED, at most `cap` entries — the cap nearest by (entry distance,
// node id), which is what BVHAccel::ListNodeIntersections keeps (nanort.h:2608-2692: a priority queue of kMaxIntersections).
// Adding a candidate is one store; only a ray that enters more than `cap` boxes pays for finding the farthest entry to
// replace.  k_scene_trace picks the candidates in (distance, id) order as it needs them (round 2 kept the list sorted by
// insertion in global memory: most of the listing kernel's time).
struct ListTail {
  uint32_t cnt = 0;
  bool far_known = false; // far_* describe the farthest entry of a full list
  float far_t = 0.0f;
  uint32_t far_id = 0, far_pos = 0;
};
__device__ inline void list_farthest(ListTail &st, uint32_t cap, const float *list_t, const uint32_t *list_node, uint32_t n, uint32_t i) {
  st.far_t = list_t[i];
  st.far_id = list_node[i];
  st.far_pos = 0;
  for (uint32_t q = 1; q < cap; q++) {
    const float qt = list_t[(size_t)q * n + i];
    const uint32_t qk = list_node[(size_t)q * n + i];
    if (qt > st.far_t || (qt == st.far_t && qk > st.far_id)) {
      st.far_t = qt;
      st.far_id = qk;
      st.far_pos = q;
    }
  }
  st.far_known = true;
}
__device__ inline void list_add(ListTail &st, float t, uint32_t k, uint32_t cap, float *__restrict__ list_t,
                                uint32_t *__restrict__ list_node, uint32_t n, uint32_t i) {
  if (st.cnt < cap) {
    list_t[(size_t)st.cnt * n + i] = t;
    list_node[(size_t)st.cnt * n + i] = k;
    st.cnt++;
    return;
  }
  if (!st.far_known) list_farthest(st, cap, list_t, list_node, n, i);
  if (t < st.far_t || (t == st.far_t && k < st.far_id)) { // nearer than the farthest kept: takes its place
    list_t[(size_t)st.far_pos * n + i] = t;
    list_node[(size_t)st.far_pos * n + i] = k;
    st.far_known = false;
  }
}

// list_t / list_node: [kMaxList or num_nodes][n] (entry-major, so lane-consecutive accesses coalesce)
__global__ __launch_bounds__(256) void k_scene_list(const nrt_ray_f32 *__restrict__ rays, uint32_t n,
                                                    const NodeDev *__restrict__ nodes, uint32_t num_nodes, uint32_t cap,
                                                    float *__restrict__ list_t, uint32_t *__restrict__ list_node,
                                                    uint32_t *__restrict__ count, const uint32_t *__restrict__ subset) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const nrt_ray_f32 r = rays[subset ? subset[i] : i]; // (a subset launch lists rays[subset[i]] into slot i of n)
  ListTail st;
  for (uint32_t k = 0; k < num_nodes; k++) {
    float t;
    if (!node_interval(r, nodes[k], t)) continue;
    list_add(st, t, k, cap, list_t, list_node, n, i);
  }
  count[i] = st.cnt;
}

// The same listing through the top-level BVH: reference-format nodes over the instances' world boxes, leaves name the
// instances through `top_indices`.  Inner boxes are tested as ListNodeIntersections tests them (IntersectRayAABB with
// hit_t == ray.max_t throughout, nanort.h:2651), leaves with node_interval; entries are kept sorted by (entry distance,
// node id) — the order the scan above produces — whatever order the walk finds them in.
__device__ inline bool top_box_hit(const nrt_ray_f32 &r, const float inv[3], const int sign[3], const float bmin[3], const float bmax[3]) {
  float tmin = r.min_t, tmax = r.max_t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float lo = sign[k] ? bmax[k] : bmin[k], hi = sign[k] ? bmin[k] : bmax[k];
    const float t0 = (lo - r.org[k]) * inv[k];
    const float t1 = (hi - r.org[k]) * inv[k] * 1.00000024f;
    tmin = (t0 > tmin) ? t0 : tmin;
    tmax = (t1 < tmax) ? t1 : tmax;
  }
  return tmin <= tmax;
}

__global__ __launch_bounds__(256) void k_scene_list_bvh(const nrt_ray_f32 *__restrict__ rays, uint32_t n,
                                                        const nrt_node_f32 *__restrict__ top_nodes,
                                                        const uint32_t *__restrict__ top_indices,
                                                        const NodeDev *__restrict__ nodes, uint32_t cap,
                                                        float *__restrict__ list_t, uint32_t *__restrict__ list_node,
                                                        uint32_t *__restrict__ count, const uint32_t *__restrict__ subset) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const nrt_ray_f32 r = rays[subset ? subset[i] : i];
  float inv[3];
  int sign[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float d = r.dir[k];
    sign[k] = d < 0.0f ? 1 : 0;
    inv[k] = (__builtin_fabsf(d) < 1.1920928955078125e-07f) ? __builtin_huge_valf() * (sign[k] ? -1.0f : 1.0f) : 1.0f / d;
  }
  uint32_t stack[kTopStack];
  int sp = 0;
  stack[0] = 0u;
  ListTail st;
  while (sp >= 0) {
    const nrt_node_f32 nd = top_nodes[stack[sp]];
    sp--;
    if (!top_box_hit(r, inv, sign, nd.bmin, nd.bmax)) continue;
    if (nd.flag == 0) {
      const int near = sign[nd.axis];
      stack[++sp] = nd.data[1 - near];
      stack[++sp] = nd.data[near];
      continue;
    }
    for (uint32_t q = 0; q < nd.data[0]; q++) {
      const uint32_t k = top_indices[nd.data[1] + q];
      float t;
      if (!node_interval(r, nodes[k], t)) continue;
      list_add(st, t, k, cap, list_t, list_node, n, i);
    }
  }
  count[i] = st.cnt;
}

// The same listing through the top-level tree's Wide4Node records (two tree levels per 128-byte fetch: a third of the
// dependent round trips of the walk above).  Which instances a ray lists does not depend on the walk: an instance is listed iff
// node_interval passes on its world box, and the tree only prunes — a subtree is skipped when the ray misses its box by the
// same IntersectRayAABB arithmetic, and a box that contains an entered box is entered (the slab arithmetic is monotone), so
// testing the four grandchild boxes instead of child-then-grandchild prunes the same subtrees.  A leaf of one instance needs
// no fetch at all: its slot's box IS the instance's world box (the builder's box of the zero-radius cylinder xbmin -> xbmax).
// PRUNE: keep entry distances on the stack, walk front to back and skip what lies beyond a full list (scenes of many
// instances, where rays enter more boxes than the list holds; on smaller scenes the bookkeeping costs 5-10 % and buys nothing).
template <bool PRUNE>
__global__ __launch_bounds__(256) void k_scene_list_w4(const nrt_ray_f32 *__restrict__ rays, uint32_t n,
                                                       const nrt::Wide4Node<float> *__restrict__ wide4,
                                                       const nrt_node_f32 *__restrict__ top_nodes, uint32_t packed_leaves,
                                                       const uint32_t *__restrict__ top_indices,
                                                       const NodeDev *__restrict__ nodes, uint32_t cap,
                                                       float *__restrict__ list_t, uint32_t *__restrict__ list_node,
                                                       uint32_t *__restrict__ count, const uint32_t *__restrict__ subset) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const nrt_ray_f32 r = rays[subset ? subset[i] : i];
  float inv[3], pinv[3];
  int sign[3];
  bool tame = PRUNE; // every direction component is an ordinary non-zero number and the origin is finite: entry distances are monotone in the box
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float d = r.dir[k];
    sign[k] = d < 0.0f ? 1 : 0;
    inv[k] = (__builtin_fabsf(d) < 1.1920928955078125e-07f) ? __builtin_huge_valf() * (sign[k] ? -1.0f : 1.0f) : 1.0f / d;
    pinv[k] = 1.0f / d; // NodeBBoxIntersector's plain reciprocal (nanosg.h:603-639)
    tame = tame && (__builtin_fabsf(d) >= 1.1920928955078125e-07f) && (__builtin_fabsf(pinv[k]) < __builtin_huge_valf()) &&
           (__builtin_fabsf(r.org[k]) < __builtin_huge_valf());
  }
  // entry distance of a box as node_interval computes it for an instance's box (unclipped, plain reciprocal): for a tame ray
  // it can only grow from a box to a box inside it, so a subtree whose box is entered beyond the farthest entry of a FULL
  // list holds nothing that could still make the list — the walk skips it (a ray through 100 000 instances lists its 64
  // nearest without visiting the rest)
  auto entry = [&](const float bmin[3], const float bmax[3]) -> float {
    float a = -__builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float lo = sign[k] ? bmax[k] : bmin[k];
      const float t = (lo - r.org[k]) * pinv[k];
      a = (t > a) ? t : a;
    }
    return a;
  };
  uint32_t stack[kTopStack];
  float stack_t[PRUNE ? kTopStack : 1]; // entry distance of the pushed record's box (tame rays; else unused)
  int sp = 0;
  stack[0] = 0u; // record 0 == the root branch
  stack_t[0] = -__builtin_huge_valf();
  sp = 1;
  ListTail st;
  while (sp > 0) {
    sp--;
    if constexpr (PRUNE) {
      if (tame && st.cnt == cap) { // the list is full: anything entered beyond its farthest entry cannot get in
        if (!st.far_known) list_farthest(st, cap, list_t, list_node, n, i);
        if (stack_t[sp] > st.far_t) continue;
      }
    }
    const nrt::Wide4Node<float> w = wide4[stack[sp]];
    uint32_t in_ref[4];
    float in_t[4];
    int nin = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t cj = w.c[j];
      if (cj == nrt::kWide4Empty) continue;
      const float bmin[3] = {w.bmin[0][j], w.bmin[1][j], w.bmin[2][j]}, bmax[3] = {w.bmax[0][j], w.bmax[1][j], w.bmax[2][j]};
      if (!(cj & nrt::kLeafBit)) {
        if (top_box_hit(r, inv, sign, bmin, bmax)) {
          in_ref[nin] = cj;
          in_t[nin] = PRUNE ? entry(bmin, bmax) : 0.0f;
          nin++;
        }
        continue;
      }
      uint32_t lcount, lfirst;
      if (packed_leaves) {
        lcount = ((cj & ~nrt::kLeafBit) >> nrt::kPackedFirstBits) + 1u;
        lfirst = cj & nrt::kPackedFirstMask;
      } else {
        const nrt_node_f32 &leaf = top_nodes[cj & ~nrt::kLeafBit];
        lcount = leaf.data[0];
        lfirst = leaf.data[1];
      }
      for (uint32_t q = 0; q < lcount; q++) {
        const uint32_t k = top_indices[lfirst + q];
        float t;
        if (lcount == 1u) {
          if (!node_interval_box(r, bmin, bmax, t)) continue;
        } else if (!node_interval(r, nodes[k], t)) {
          continue;
        }
        list_add(st, t, k, cap, list_t, list_node, n, i);
      }
    }
    // nearest box on top of the stack: the walk runs roughly front to back, the list fills with near entries first and the
    // far subtrees fall to the test above (insertion sort of at most four entries, farthest first)
    for (int x = 1; PRUNE && x < nin; x++)
      for (int y = x; y > 0 && in_t[y] > in_t[y - 1]; y--) {
        const float tt = in_t[y];
        in_t[y] = in_t[y - 1];
        in_t[y - 1] = tt;
        const uint32_t rr = in_ref[y];
        in_ref[y] = in_ref[y - 1];
        in_ref[y - 1] = rr;
      }
    for (int x = 0; x < nin; x++) {
      stack[sp] = in_ref[x];
      if constexpr (PRUNE) stack_t[sp] = in_t[x];
      sp++;
    }
  }
  count[i] = st.cnt;
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
  std::vector<std::pair<nrt_ctx *, uint64_t> > mesh_gens; // the distinct mesh contexts and their generations at Commit (the
                                                          // per-instance table caches their device addresses and flags)
  nrt_ctx *top = nullptr;        // top-level BVH over the nodes' world boxes (scenes of two nodes or more)
  nrt::TreeViewF32 top_view;
  bool use_top = false;          // the listing kernels walk the top-level tree (scenes of more than kScanMaxNodes nodes; else they scan the nodes)
  bool have_top = false;         // a top-level tree exists (two nodes or more): the single-pass walk uses it
  uint32_t max_inst_depth = 0;   // deepest instance tree: sizes the overflow stack of k_scene_trace
  nrt::DevBuf d_nodes, d_insts, d_rays, d_list_t, d_list_node, d_count, d_best, d_mask, d_spill, d_spill_tmin, d_cursor;
  nrt::DevBuf d_redo, d_redo_count, d_counters, d_open_top, d_meshes;
  bool walk_meshes_ok = false; // every mesh of the scene has the private layout k_scene_walk steps through (else: the listing path)
  unsigned count_loops = 0; // profiling build only (tunable "count_loops"): the next calls run the counting instantiation of k_scene_walk
  uint32_t *h_redo_count = nullptr; // page-locked: how many rays the single-pass walk left to the listing path
  uint64_t last_redone = 0;
  int last_path = 0; // 1: the last call went through the single-pass walk (0: the listing path alone)
  unsigned single_pass = 1;   // scenes of kWalkMinNodes nodes or more are traced by k_scene_walk (no per-ray list); 0: always listing +
                              // k_scene_trace; 2: k_scene_walk for every scene of two nodes or more
  unsigned walk_blocks_per_cu = 0;
  unsigned walk_leaf_items = 1; // the walk's leaf phase over items (traverse.hip leaf_items_one_trip; records bit-identical) when every mesh's leaves hold <= 4 records
  unsigned max_mesh_leaf = 0;   // most records a leaf of any instanced mesh holds (set by commit)
  unsigned walk_trav_min = 24, walk_refill_min = 24; // the walk's own phase thresholds (profiles/r04t_scene_walk_thresholds.txt: 8 / 56, the listing path's, cost it 10-13 %)
  unsigned walk_backoff = 0;  // calls left that skip the walk (see scene_traverse)
  unsigned walk_backoff_pct = 25; // share of a batch handed to the listing path above which the next kWalkBackoff calls skip the walk (tunable)
  unsigned walk_min = kWalkMinNodes; // scenes of at least this many nodes are traced by the walk (tunable "walk_min")
  unsigned prune_min = 32768; // scenes of at least this many instances are listed by the pruning walk (k_scene_list_w4<true>)
  unsigned scan_max = kScanMaxNodes; // scenes of at most this many nodes are listed by testing every world box (no top-level tree); tunable, <= kMaxList (next Commit)
  unsigned fuse_scan = 1; // scenes of at most kScanMaxNodes nodes: k_scene_trace lists a ray's instances itself (one launch); 0: k_scene_list in a launch of its own (the A/B)
  unsigned trav_min = 8;
  unsigned cand_min = 1, cand_busy_max = 64; // batching of the per-instance steps of k_scene_trace (env NRT_SCENE_CAND / NRT_SCENE_CAND_BUSY; 1 / 64: none)
  unsigned trace_blocks_per_cu = 0, num_cus = 0, refill_min = 56; // persistent grid of k_scene_trace (env NRT_SCENE_REFILL; 16-48 measured slower on small scenes, 64 slower on 10 000 instances)
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
  nrt::DevBuf *bufs[] = {&s->d_nodes, &s->d_insts, &s->d_rays, &s->d_list_t, &s->d_list_node, &s->d_count,
                         &s->d_best,  &s->d_mask,  &s->d_spill, &s->d_spill_tmin, &s->d_cursor, &s->d_redo, &s->d_redo_count, &s->d_counters, &s->d_open_top, &s->d_meshes};
  for (nrt::DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  if (s->h_redo_count) (void)hipHostFree(s->h_redo_count);
  if (s->top) nrtDestroy(s->top);
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
  // per distinct mesh context (instances share them): where its tree lives, and its root box
  struct MeshInfo {
    nrt::TreeViewF32 tv;
    nrt_node_f32 root;
  };
  std::vector<std::pair<nrt_ctx *, MeshInfo> > meshes;
  s->max_mesh_leaf = 0;
  std::unordered_map<nrt_ctx *, uint32_t> mesh_index;
  std::vector<uint32_t> mesh_of(s->insts.size());
  for (size_t i = 0; i < s->insts.size(); i++) {
    std::unordered_map<nrt_ctx *, uint32_t>::iterator it = mesh_index.find(s->insts[i].mesh);
    uint32_t m = it == mesh_index.end() ? (uint32_t)meshes.size() : it->second;
    if (it == mesh_index.end()) {
      mesh_index[s->insts[i].mesh] = m;
      MeshInfo mi;
      if (nrt_internal_tree_view(s->insts[i].mesh, &mi.tv) != NRT_OK || mi.tv.prim_kind != (uint32_t)nrt::kPrimTriangles)
        return sfail(s, NRT_ERR_PRECISION, "nrtSceneCommit: node %zu is not a built f32 triangle mesh: %s", i, nrtLastError(s->insts[i].mesh));
      SCHK(s, hipMemcpy(&mi.root, mi.tv.nodes, sizeof(mi.root), hipMemcpyDeviceToHost));
      s->max_mesh_leaf = std::max(s->max_mesh_leaf, (unsigned)mi.tv.max_leaf_count);
      meshes.push_back(std::make_pair(s->insts[i].mesh, mi));
    }
    mesh_of[i] = m;
    // local AABB = the root box of the node's tree (accel_.BoundingBox, nanosg.h:411)
    node_update(s->insts[i].local, meshes[m].second.root.bmin, meshes[m].second.root.bmax, &s->host_nodes[i]);
  }
  SCHK(s, nrt::devbuf_ensure(&s->d_nodes, s->host_nodes.size() * sizeof(NodeDev)));
  SCHK(s, hipMemcpy(s->d_nodes.p, s->host_nodes.data(), s->host_nodes.size() * sizeof(NodeDev), hipMemcpyHostToDevice));
  // per-instance table of k_scene_trace: where each node's tree lives + its three matrices
  std::vector<nrt::SceneInst> table(s->insts.size());
  s->max_inst_depth = 0;
  for (size_t i = 0; i < s->insts.size(); i++) {
    const nrt::TreeViewF32 &tv = meshes[mesh_of[i]].second.tv;
    nrt::SceneInst &e = table[i];
    e.wide = tv.wide;
    e.wide4 = (tv.tree_nested && tv.root_is_branch) ? tv.wide4 : nullptr; // two levels per step need nested boxes (traverse.hip)
    e.tris = tv.prims;
    e.nodes = tv.nodes;
    e.packed_leaves = tv.packed_leaves;
    e.root_is_branch = tv.root_is_branch;
    e.tree_nested = tv.tree_nested;
    e.pad = 0;
    memcpy(e.inv_xform, s->host_nodes[i].inv_xform, sizeof(e.inv_xform));
    memcpy(e.inv_xform33, s->host_nodes[i].inv_xform33, sizeof(e.inv_xform33));
    memcpy(e.xform, s->host_nodes[i].xform, sizeof(e.xform));
    memcpy(e.xbmin, s->host_nodes[i].xbmin, sizeof(e.xbmin));
    memcpy(e.xbmax, s->host_nodes[i].xbmax, sizeof(e.xbmax));
    e.id = (uint32_t)i;
    e.pad2 = 0;
    // deepest stack a walk of this tree can need: one pending sibling per level, three per two levels when stepping two
    s->max_inst_depth = std::max(s->max_inst_depth, e.wide4 ? 3u * (tv.tree_depth / 2u + 1u) : tv.tree_depth);
  }
  SCHK(s, nrt::devbuf_ensure(&s->d_insts, table.size() * sizeof(nrt::SceneInst)));
  SCHK(s, hipMemcpy(s->d_insts.p, table.data(), table.size() * sizeof(nrt::SceneInst), hipMemcpyHostToDevice));
  // top-level BVH over the world boxes (nanosg.h:726-735: BVHAccel over the nodes with min_leaf_primitives = 1), built
  // by the ordinary GPU builder: a box is handed over as a zero-radius "cylinder" from its low to its high corner, whose
  // bounding box is exactly the box.  Which nodes a ray lists does not depend on this tree's shape.
  s->use_top = false;
  s->have_top = false;
  s->walk_backoff = 0;
  // ... only where something reads it: the listing kernels of scenes of more than kScanMaxNodes nodes, the single-pass walk from
  // walk_min nodes (or forced, single_pass = 2).  A handful of nodes is scanned: no context, no build, no read-back for them
  // (a tunable that makes the walk eligible later commits the scene again: scene_traverse).
  const bool want_top = s->insts.size() > s->scan_max || (s->single_pass && (s->insts.size() >= s->walk_min || s->single_pass > 1));
  if (s->insts.size() >= 2 && want_top) {
    if (!s->top && nrtCreate(s->device, &s->top) != NRT_OK)
      return sfail(s, NRT_ERR_DEVICE, "nrtSceneCommit: top-level context: %s", nrtLastError(nullptr));
    std::vector<float> ends(6 * s->insts.size()), radii(2 * s->insts.size(), 0.0f);
    for (size_t i = 0; i < s->insts.size(); i++)
      for (int k = 0; k < 3; k++) {
        ends[6 * i + k] = s->host_nodes[i].xbmin[k];
        ends[6 * i + 3 + k] = s->host_nodes[i].xbmax[k];
      }
    nrt_build_options_f32 o;
    memset(&o, 0, sizeof(o));
    o.cost_t_aabb = 0.2f;
    o.min_leaf_primitives = 1;
    o.max_tree_depth = 256;
    o.bin_size = 64;
    o.shallow_depth = 4;
    o.min_primitives_for_parallel_build = 1024 * 8;
    if (nrtSetCylinders_f32(s->top, ends.data(), radii.data(), (uint32_t)s->insts.size(), 0) != NRT_OK ||
        nrtBuild_f32(s->top, &o, nullptr, nullptr) != NRT_OK || nrt_internal_tree_view(s->top, &s->top_view) != NRT_OK)
      return sfail(s, NRT_ERR_DEVICE, "nrtSceneCommit: top-level build: %s", nrtLastError(s->top));
    // the single-pass walk opens an instance with ONE fetch: a 128-byte line per instance, in the order of the tree's index array
    {
      uint64_t nn = 0, ni = 0;
      if (nrtTreeSize(s->top, &nn, &ni) != NRT_OK || ni != s->insts.size())
        return sfail(s, NRT_ERR_DEVICE, "nrtSceneCommit: top-level index array: %s", nrtLastError(s->top));
      std::vector<uint32_t> order(ni);
      if (nrtGetTree_f32(s->top, nullptr, order.data()) != NRT_OK)
        return sfail(s, NRT_ERR_DEVICE, "nrtSceneCommit: top-level index array: %s", nrtLastError(s->top));
      std::vector<nrt::SceneOpen> open(ni); // (the 128-byte line an opening reads; the full record, by id, only when the instance was hit)
      for (size_t q = 0; q < ni; q++) {
        const nrt::SceneInst &e = table[order[q]];
        nrt::SceneOpen &o = open[q];
        for (int r = 0; r < 4; r++)
          for (int c = 0; c < 3; c++) {
            o.inv[r][c] = e.inv_xform[r][c];
            o.inv33[r][c] = e.inv_xform33[r][c];
          }
        memcpy(o.xbmin, e.xbmin, sizeof(o.xbmin));
        memcpy(o.xbmax, e.xbmax, sizeof(o.xbmax));
        o.id = e.id;
        o.mesh = mesh_of[order[q]];
      }
      SCHK(s, nrt::devbuf_ensure(&s->d_open_top, open.size() * sizeof(nrt::SceneOpen)));
      SCHK(s, hipMemcpy(s->d_open_top.p, open.data(), open.size() * sizeof(nrt::SceneOpen), hipMemcpyHostToDevice));
      std::vector<nrt::SceneMesh> mt(meshes.size());
      s->walk_meshes_ok = true; // the walk steps two levels at a time through packed leaf references, from a branch root whose children lie inside it
      for (size_t m = 0; m < meshes.size(); m++) {
        const nrt::TreeViewF32 &tv = meshes[m].second.tv;
        const nrt_node_f32 &root = meshes[m].second.root;
        memset(&mt[m], 0, sizeof(mt[m]));
        mt[m].wide4 = tv.wide4;
        mt[m].tris = tv.prims;
        const bool one_leaf = !tv.root_is_branch && root.flag != 0 && root.data[0] >= 1 && root.data[0] <= nrt::kPackedMaxCount &&
                              root.data[1] <= nrt::kPackedFirstMask;
        if (one_leaf) {
          mt[m].root_leaf = 1;
          mt[m].leaf_ref = ((root.data[0] - 1u) << nrt::kPackedFirstBits) | root.data[1];
          memcpy(mt[m].bmin, root.bmin, sizeof(mt[m].bmin));
          memcpy(mt[m].bmax, root.bmax, sizeof(mt[m].bmax));
        }
        s->walk_meshes_ok = s->walk_meshes_ok && tv.prims && (one_leaf || (tv.wide4 && tv.tree_nested && tv.root_is_branch && tv.packed_leaves));
      }
      SCHK(s, nrt::devbuf_ensure(&s->d_meshes, mt.size() * sizeof(nrt::SceneMesh)));
      SCHK(s, hipMemcpy(s->d_meshes.p, mt.data(), mt.size() * sizeof(nrt::SceneMesh), hipMemcpyHostToDevice));
    }
    s->have_top = true;                                                                              // the single-pass walk's
    s->use_top = s->insts.size() > s->scan_max && s->top_view.tree_depth + 2 < (uint32_t)kTopStack; // the listing kernels' (else: the scan)
  }
  s->mesh_gens.clear();
  for (size_t m = 0; m < meshes.size(); m++) s->mesh_gens.push_back(std::make_pair(meshes[m].first, meshes[m].second.tv.generation));
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

// The listing path: one of the listing kernels + k_scene_trace, enqueued on the scene's stream, over the whole batch
// (`subset` == nullptr) or over the rays named by `subset` (the ones the single-pass walk left over).
static nrt_status scene_list_and_trace(nrt_scene *s, const nrt_ray_f32 *d_rays, uint32_t n, const uint32_t *subset,
                                       nrt_scene_hit_f32 *d_hits, uint8_t *d_mask) {
  const uint32_t num_nodes = (uint32_t)s->insts.size();
  const uint32_t cap = std::min<uint32_t>(kMaxList, num_nodes);
  SCHK(s, nrt::devbuf_ensure(&s->d_list_t, (size_t)cap * n * sizeof(float)));
  SCHK(s, nrt::devbuf_ensure(&s->d_list_node, (size_t)cap * n * sizeof(uint32_t)));
  SCHK(s, nrt::devbuf_ensure(&s->d_count, (size_t)n * sizeof(uint32_t)));
  const unsigned grid = (n + 255u) / 256u; // the listing kernels: one ray per thread
  const unsigned trace_grid = std::min(grid, s->num_cus * s->trace_blocks_per_cu); // the trace kernel: every block resident
  const uint32_t levels = s->max_inst_depth + 2 > (uint32_t)nrt::kSceneLdsStack ? s->max_inst_depth + 2 - nrt::kSceneLdsStack : 0;
  if (levels) {
    SCHK(s, nrt::devbuf_ensure(&s->d_spill, (size_t)levels * trace_grid * 256u * sizeof(uint32_t)));
    SCHK(s, nrt::devbuf_ensure(&s->d_spill_tmin, (size_t)levels * trace_grid * 256u * sizeof(float)));
  }
  SCHK(s, hipMemsetAsync(s->d_cursor.p, 0, (size_t)nrt::kMaxParts * nrt::kCursorStrideWords * sizeof(uint32_t), s->stream));
  const NodeDev *d_nodes = (const NodeDev *)s->d_nodes.p;
  const bool fused_scan = !s->use_top && s->fuse_scan && num_nodes <= (uint32_t)kMaxList; // (every entered box fits the list: no replacement logic needed in the trace kernel)
  if (s->use_top && s->top_view.wide4 && s->top_view.root_is_branch && s->top_view.tree_nested &&
      3u * (s->top_view.tree_depth / 2u + 1u) + 2u < (uint32_t)kTopStack)
  {
    if (num_nodes >= s->prune_min) // (rays can enter more boxes than the list holds: the pruning walk)
      hipLaunchKernelGGL(k_scene_list_w4<true>, dim3(grid), dim3(256), 0, s->stream, d_rays, n, (const nrt::Wide4Node<float> *)s->top_view.wide4,
                         s->top_view.nodes, s->top_view.packed_leaves, s->top_view.indices, d_nodes, cap, (float *)s->d_list_t.p,
                         (uint32_t *)s->d_list_node.p, (uint32_t *)s->d_count.p, subset);
    else
      hipLaunchKernelGGL(k_scene_list_w4<false>, dim3(grid), dim3(256), 0, s->stream, d_rays, n, (const nrt::Wide4Node<float> *)s->top_view.wide4,
                         s->top_view.nodes, s->top_view.packed_leaves, s->top_view.indices, d_nodes, cap, (float *)s->d_list_t.p,
                         (uint32_t *)s->d_list_node.p, (uint32_t *)s->d_count.p, subset);
  }
  else if (s->use_top)
    hipLaunchKernelGGL(k_scene_list_bvh, dim3(grid), dim3(256), 0, s->stream, d_rays, n, s->top_view.nodes, s->top_view.indices,
                       d_nodes, cap, (float *)s->d_list_t.p, (uint32_t *)s->d_list_node.p, (uint32_t *)s->d_count.p, subset);
  else if (!fused_scan)
    hipLaunchKernelGGL(k_scene_list, dim3(grid), dim3(256), 0, s->stream, d_rays, n, d_nodes, num_nodes, cap,
                       (float *)s->d_list_t.p, (uint32_t *)s->d_list_node.p, (uint32_t *)s->d_count.p, subset);
  // (else: a handful of nodes — k_scene_trace tests every world box itself as it fetches a ray: no listing launch)
  SCHK(s, hipGetLastError());
  nrt::SceneTraceArgs a;
  a.scan_nodes = fused_scan ? num_nodes : 0u;
  a.rays = d_rays;
  a.n = n;
  a.insts = (const nrt::SceneInst *)s->d_insts.p;
  a.list_t = (float *)s->d_list_t.p;
  a.list_node = (uint32_t *)s->d_list_node.p;
  a.count = (const uint32_t *)s->d_count.p;
  a.hits = d_hits;
  a.mask = d_mask;
  a.spill = (uint32_t *)s->d_spill.p;
  a.spill_tmin = (float *)s->d_spill_tmin.p;
  a.spill_stride = trace_grid * 256u;
  a.cursor = (uint32_t *)s->d_cursor.p;
  a.num_parts = std::max(1u, std::min(8u, trace_grid));
  a.refill_min = s->refill_min;
  a.trav_min = s->trav_min;
  a.cand_min = s->cand_min;
  a.cand_busy_max = s->cand_busy_max;
  a.subset = subset;
  SCHK(s, nrt::launch_scene_trace(a, trace_grid, s->stream));
  return NRT_OK;
}

// `device` = rays / hits_out / mask_out are device pointers (no PCIe traffic).  Scenes with a top-level tree: ONE launch of the
// single-pass walk (k_scene_walk), then — only if it left rays over — the listing path on those; other scenes (a handful of
// nodes) and single_pass = 0: the listing path on the whole batch.  The call returns with everything finished (the scene owns
// the per-ray scratch).
static nrt_status scene_traverse(nrt_scene *s, const nrt_ray_f32 *rays, uint64_t n64, nrt_scene_hit_f32 *hits_out,
                                 uint8_t *mask_out, bool device) {
  if (!s) return NRT_ERR_INVALID;
  if (!s->committed) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: commit the scene first");
  if (n64 == 0) return NRT_OK;
  if (!rays || !hits_out) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: NULL rays/hits");
  if (n64 > 0x7FFFFFFFull) return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: too many rays in one call");
  // the instance table holds device addresses, layout flags and stack depths of the mesh contexts' trees as they were at
  // Commit: a context rebuilt or re-set since then may have moved or re-shaped them — refuse instead of walking stale memory
  for (size_t m = 0; m < s->mesh_gens.size(); m++)
    if (nrt_internal_generation(s->mesh_gens[m].first) != s->mesh_gens[m].second)
      return sfail(s, NRT_ERR_INVALID, "nrtSceneTraverseBatch: a mesh context was rebuilt or re-set since nrtSceneCommit (commit the scene again)");
  const uint32_t n = (uint32_t)n64;
  SCHK(s, hipSetDevice(s->device));
  if (!device) SCHK(s, nrt::devbuf_ensure(&s->d_rays, (size_t)n * sizeof(nrt_ray_f32)));
  if (!device) SCHK(s, nrt::devbuf_ensure(&s->d_best, (size_t)n * sizeof(nrt_scene_hit_f32)));
  if (!device && mask_out) SCHK(s, nrt::devbuf_ensure(&s->d_mask, (size_t)n));
  if (s->trace_blocks_per_cu == 0) {
    s->trace_blocks_per_cu = (unsigned)nrt::scene_trace_blocks_per_cu();
    s->walk_blocks_per_cu = (unsigned)nrt::scene_walk_blocks_per_cu();
    hipDeviceProp_t prop;
    SCHK(s, hipGetDeviceProperties(&prop, s->device));
    s->num_cus = (unsigned)prop.multiProcessorCount;
    if (nrt::env_overrides_allowed()) {
    if (const char *e = getenv("NRT_SCENE_REFILL")) s->refill_min = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_TRAV")) s->trav_min = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_CAND")) s->cand_min = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_CAND_BUSY")) s->cand_busy_max = (unsigned)std::min(65, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_PRUNE_MIN")) s->prune_min = (unsigned)std::max(0, atoi(e)); // (debugging / tests: the pruning walk on small scenes)
    if (const char *e = getenv("NRT_SCENE_WALK_TRAV")) s->walk_trav_min = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_WALK_REFILL")) s->walk_refill_min = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NRT_SCENE_WALK")) s->single_pass = (unsigned)std::min(2, std::max(0, atoi(e)));
    }
  }
  // the walk became eligible after Commit (nrtSceneSetTunable / the variables above) on a scene committed without its top-level tree
  if (s->single_pass && !s->have_top && s->insts.size() >= 2 && s->insts.size() <= s->scan_max &&
      (s->insts.size() >= s->walk_min || s->single_pass > 1)) {
    const nrt_status st = nrtSceneCommit(s);
    if (st) return st;
  }
  SCHK(s, nrt::devbuf_ensure(&s->d_cursor, (size_t)nrt::kMaxParts * nrt::kCursorStrideWords * sizeof(uint32_t)));
  const nrt_ray_f32 *d_rays = device ? rays : (const nrt_ray_f32 *)s->d_rays.p;
  nrt_scene_hit_f32 *d_hits = device ? hits_out : (nrt_scene_hit_f32 *)s->d_best.p;
  uint8_t *d_mask = device ? mask_out : (mask_out ? (uint8_t *)s->d_mask.p : nullptr);
  if (!device) SCHK(s, hipMemcpyAsync(s->d_rays.p, rays, (size_t)n * sizeof(nrt_ray_f32), hipMemcpyHostToDevice, s->stream));

  s->last_redone = 0;
  s->last_path = 0;
  const bool eligible = s->single_pass && s->have_top && (s->insts.size() >= s->walk_min || s->single_pass > 1) && s->walk_meshes_ok && s->top_view.wide4 &&
                        s->top_view.root_is_branch && s->top_view.tree_nested && s->top_view.packed_leaves;
  // a batch the walk could not certify for the most part (direction vectors far shorter than 1: the reference's cull then compares
  // a distance with a parameter and fires early, nanosg.h:795) costs more than the listing path alone: after one, the next
  // kWalkBackoff calls go straight to the listing path, then the walk is tried again
  const bool walk = eligible && (s->single_pass > 1 || s->walk_backoff == 0);
  if (eligible && !walk) s->walk_backoff--;
  if (walk) {
    s->last_path = 1;
    if (!s->h_redo_count) SCHK(s, hipHostMalloc((void **)&s->h_redo_count, sizeof(uint32_t), hipHostMallocDefault));
    SCHK(s, nrt::devbuf_ensure(&s->d_redo, (size_t)n * sizeof(uint32_t)));
    SCHK(s, nrt::devbuf_ensure(&s->d_redo_count, sizeof(uint32_t)));
    const unsigned grid = std::min((n + 255u) / 256u, s->num_cus * s->walk_blocks_per_cu);
    // the lane's stack: top-level entries (three per two levels of the top-level tree, the instances of one leaf) below the open instance's
    const uint32_t need = 3u * (s->top_view.tree_depth / 2u + 1u) + 2u + nrt::kPackedMaxCount + s->max_inst_depth + 2u;
    const uint32_t levels = need > (uint32_t)nrt::kSceneWalkLdsStack ? need - nrt::kSceneWalkLdsStack : 0;
    if (levels) {
      SCHK(s, nrt::devbuf_ensure(&s->d_spill, (size_t)levels * grid * 256u * sizeof(uint32_t)));
      SCHK(s, nrt::devbuf_ensure(&s->d_spill_tmin, (size_t)levels * grid * 256u * sizeof(float)));
    }
    SCHK(s, hipMemsetAsync(s->d_cursor.p, 0, (size_t)nrt::kMaxParts * nrt::kCursorStrideWords * sizeof(uint32_t), s->stream));
    SCHK(s, hipMemsetAsync(s->d_redo_count.p, 0, sizeof(uint32_t), s->stream));
    nrt::SceneWalkArgs w;
    w.rays = d_rays;
    w.n = n;
    w.open_top = (const nrt::SceneOpen *)s->d_open_top.p;
    w.meshes = (const nrt::SceneMesh *)s->d_meshes.p;
    w.insts = (const nrt::SceneInst *)s->d_insts.p;
    w.top_wide4 = (const nrt::Wide4Node<float> *)s->top_view.wide4;
    w.hits = d_hits;
    w.mask = d_mask;
    w.spill = (uint32_t *)s->d_spill.p;
    w.spill_tmin = (float *)s->d_spill_tmin.p;
    w.spill_stride = grid * 256u;
    w.cursor = (uint32_t *)s->d_cursor.p;
    w.num_parts = std::max(1u, std::min(8u, grid));
    w.refill_min = s->walk_refill_min;
    w.trav_min = s->walk_trav_min;
    w.cand_min = s->cand_min;
    w.cand_busy_max = s->cand_busy_max;
    w.leaf_items = (s->walk_leaf_items && s->max_mesh_leaf <= 4u) ? 1u : 0u;
    w.redo = (uint32_t *)s->d_redo.p;
    w.redo_count = (uint32_t *)s->d_redo_count.p;
    w.counters = nullptr;
#ifdef NRT_PROF
    if (s->count_loops) {
      SCHK(s, nrt::devbuf_ensure(&s->d_counters, 16 * sizeof(unsigned long long)));
      SCHK(s, hipMemsetAsync(s->d_counters.p, 0, 16 * sizeof(unsigned long long), s->stream));
      w.counters = (unsigned long long *)s->d_counters.p;
    }
#endif
    SCHK(s, nrt::launch_scene_walk(w, grid, s->stream));
    SCHK(s, hipMemcpyAsync(s->h_redo_count, s->d_redo_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SCHK(s, hipStreamSynchronize(s->stream));
    s->last_redone = *s->h_redo_count;
    if ((uint64_t)*s->h_redo_count * 100u > (uint64_t)n * s->walk_backoff_pct) s->walk_backoff = kWalkBackoff;
    if (*s->h_redo_count) {
      const nrt_status st = scene_list_and_trace(s, d_rays, *s->h_redo_count, (const uint32_t *)s->d_redo.p, d_hits, d_mask);
      if (st != NRT_OK) return st;
    }
  } else {
    const nrt_status st = scene_list_and_trace(s, d_rays, n, nullptr, d_hits, d_mask);
    if (st != NRT_OK) return st;
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

nrt_status nrtSceneSetTunable(nrt_scene *s, const char *name, int value) {
  if (!s || !name) return NRT_ERR_INVALID;
  const std::string k(name);
  const unsigned lanes = (unsigned)std::min(64, std::max(1, value));
  if (k == "single_pass") s->single_pass = (unsigned)std::min(2, std::max(0, value));
  else if (k == "trav_min") s->trav_min = lanes;
  else if (k == "refill_min") s->refill_min = lanes;
  else if (k == "walk_trav_min") s->walk_trav_min = lanes;
  else if (k == "walk_leaf_items") s->walk_leaf_items = value ? 1u : 0u;
  else if (k == "walk_refill_min") s->walk_refill_min = lanes;
  else if (k == "cand_min") s->cand_min = lanes;
  else if (k == "cand_busy_max") s->cand_busy_max = (unsigned)std::min(65, std::max(1, value));
  else if (k == "fuse_scan") s->fuse_scan = value != 0;
  else if (k == "scan_max") s->scan_max = (unsigned)std::min(kMaxList, std::max(1, value));
  else if (k == "prune_min") s->prune_min = (unsigned)std::max(0, value);
  else if (k == "walk_min") s->walk_min = (unsigned)std::max(2, value);
  else if (k == "walk_backoff_pct") s->walk_backoff_pct = (unsigned)std::min(100, std::max(0, value));
#ifdef NRT_PROF
  else if (k == "count_loops") s->count_loops = value != 0;
#endif
  else return sfail(s, NRT_ERR_INVALID, "nrtSceneSetTunable: unknown tunable '%s'", name);
  return NRT_OK;
}

uint64_t nrtSceneLastRedone(const nrt_scene *s) { return s ? s->last_redone : 0; }
int nrtSceneLastPath(const nrt_scene *s) { return s ? s->last_path : 0; }

#ifdef NRT_PROF // libnanort_hip_prof.so only (include/nanort_hip_prof.h)
int nrtSceneDebugCounters(nrt_scene *s, unsigned long long *out, int cap) {
  if (!s || !out || cap < 16 || !s->d_counters.p) return 1;
  if (hipStreamSynchronize(s->stream) != hipSuccess) return 1;
  return hipMemcpy(out, s->d_counters.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

} // extern "C"

// nanort_amd/csrc/wide8.hip — construction of the 8-wide compressed layout (common.h: Wide8Node, W8Rec) from the
// reference-format BVHNode array of a context (nanort.h:498-550; the array BVHAccel::Build emits, :1892-2149).
//
// One persistent kernel walks the binary tree breadth-first through a device-side queue: queue position i IS wide node i,
// the entry is the binary branch the node is made from.  A thread claims a position, waits (polling, never blocking its
// wave) until the entry has been published, opens the branch's subtree into at most eight children, places them in the
// eight slots, quantises their boxes, reserves consecutive positions for the inner children and consecutive leaf blocks
// for the leaf children (two atomics), publishes the children and writes the record.  The order in which positions are
// handed out depends on the run; everything a record contains is a pure function of its binary branch, so two runs differ
// by a renumbering only (tests/test_gpu_wide8.py compares with the CPU model record by record through `root`).
//
// The rules (greedy opening, slot affinity, outward quantisation in double) are restated one for one by the CPU model the
// tests check against (the wide8 model under the test infrastructure).
#include "common.h"

#include <algorithm>

namespace nrt {

namespace {

constexpr uint32_t kW8Empty = 0xFFFFFFFFu;

__device__ __forceinline__ float w8_area(const nrt_node_f32 &n) {
  const float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
  return (dx * dy + dy * dz) + dz * dx;
}

__device__ __forceinline__ uint8_t w8_axis_exponent(float p, float bmax) {
  const float ext = bmax - p;
  const float s0 = ext / 255.0f;
  const uint32_t bits = __float_as_uint(s0);
  uint32_t eb = (bits >> 23) & 0xFFu;
  if (!(ext >= 0.0f) || eb == 0xFFu) return 254;
  if ((bits & 0x7FFFFFu) != 0u) eb++;
  if (eb < 1u) eb = 1u;
  while (eb < 254u && !((double)p + 255.0 * (double)__uint_as_float(eb << 23) >= (double)bmax)) eb++;
  return (uint8_t)eb;
}
__device__ __forceinline__ uint8_t w8_quant_lo(float p, uint8_t e, float cmin) {
  const double s = (double)__uint_as_float((uint32_t)e << 23);
  double q = floor(((double)cmin - (double)p) / s);
  if (!(q >= 0.0)) q = 0.0;
  if (q > 255.0) q = 255.0;
  int qi = (int)q;
  while (qi > 0 && !((double)p + (double)qi * s <= (double)cmin)) qi--;
  return (uint8_t)qi;
}
__device__ __forceinline__ uint8_t w8_quant_hi(float p, uint8_t e, float cmax) {
  const double s = (double)__uint_as_float((uint32_t)e << 23);
  double q = ceil(((double)cmax - (double)p) / s);
  if (!(q <= 255.0)) q = 255.0;
  if (q < 0.0) q = 0.0;
  int qi = (int)q;
  while (qi < 255 && !((double)p + (double)qi * s >= (double)cmax)) qi++;
  return (uint8_t)qi;
}

struct W8BuildArgs {
  const nrt_node_f32 *nodes;
  const LeafTri<float> *tris; // leaf-ordered triangle records of the context (slot s <-> indices[s])
  Wide8Node *out;
  W8Rec *recs;
  uint32_t *queue;   // [cap_nodes], kW8Empty until published
  W8BuildState *st;
  uint32_t cap_nodes, cap_recs;
};

// One wide node: everything the record holds, from binary branch `root`.
__device__ void w8_make_node(const W8BuildArgs &a, uint32_t my, uint32_t root) {
  const nrt_node_f32 *nodes = a.nodes;
  const nrt_node_f32 rn = nodes[root];
  uint32_t kids[8];
  float area[8];
  bool branch[8];
  int n = 2;
  kids[0] = rn.data[0];
  kids[1] = rn.data[1];
  for (int i = 0; i < 2; i++) {
    const nrt_node_f32 k = nodes[kids[i]];
    branch[i] = k.flag == 0;
    area[i] = w8_area(k);
  }
  // open the branch of largest surface area (ties: the earliest in the list) until 8 children or only leaves
  while (n < 8) {
    int best = -1;
    float best_area = 0.f;
    for (int i = 0; i < n; i++) {
      if (!branch[i]) continue;
      const float av = area[i];
      if (best < 0 || av > best_area) {
        best = i;
        best_area = av;
      }
    }
    if (best < 0) break;
    const nrt_node_f32 b = nodes[kids[best]];
    for (int i = n; i > best + 1; i--) {
      kids[i] = kids[i - 1];
      area[i] = area[i - 1];
      branch[i] = branch[i - 1];
    }
    kids[best] = b.data[0];
    kids[best + 1] = b.data[1];
    for (int i = best; i < best + 2; i++) {
      const nrt_node_f32 k = nodes[kids[i]];
      branch[i] = k.flag == 0;
      area[i] = w8_area(k);
    }
    n++;
  }
  // slots: the (child, slot) pair of largest affinity first
  float d[8][3];
  uint32_t flag_count[8], first[8]; // leaf: count (> 0 marks it below through is_leaf) / first slot
  bool is_leaf[8];
  float cmin[8][3], cmax[8][3];
  for (int c = 0; c < n; c++) {
    const nrt_node_f32 k = nodes[kids[c]];
    is_leaf[c] = k.flag != 0;
    flag_count[c] = k.data[0];
    first[c] = k.data[1];
    for (int x = 0; x < 3; x++) {
      cmin[c][x] = k.bmin[x];
      cmax[c][x] = k.bmax[x];
      const float cn = (rn.bmin[x] + rn.bmax[x]) * 0.5f;
      const float cc = (k.bmin[x] + k.bmax[x]) * 0.5f;
      d[c][x] = cc - cn;
    }
  }
  int kid_in_slot[8];
  for (int s = 0; s < 8; s++) kid_in_slot[s] = -1;
  {
    bool child_done[8];
    for (int c = 0; c < 8; c++) child_done[c] = false;
    for (int round = 0; round < n; round++) {
      int bc = -1, bs = -1;
      float best = 0.f;
      for (int c = 0; c < n; c++) {
        if (child_done[c]) continue;
        for (int s = 0; s < 8; s++) {
          if (kid_in_slot[s] >= 0) continue;
          const float av = (((s & 1) ? d[c][0] : -d[c][0]) + ((s & 2) ? d[c][1] : -d[c][1])) + ((s & 4) ? d[c][2] : -d[c][2]);
          if (bc < 0 || av > best) {
            bc = c;
            bs = s;
            best = av;
          }
        }
      }
      child_done[bc] = true;
      kid_in_slot[bs] = bc;
    }
  }
  Wide8Node o;
  o.root = root;
  o.pad[0] = o.pad[1] = 0;
  for (int x = 0; x < 3; x++) {
    o.p[x] = rn.bmin[x];
    o.e[x] = w8_axis_exponent(rn.bmin[x], rn.bmax[x]);
  }
  uint32_t imask = 0, lmask = 0, ninner = 0, nleaf = 0, max_count = 0;
  for (int s = 0; s < 8; s++) {
    const int c = kid_in_slot[s];
    for (int x = 0; x < 3; x++) {
      o.qlo[x][s] = 255;
      o.qhi[x][s] = 0;
    }
    if (c < 0) continue;
    for (int x = 0; x < 3; x++) {
      o.qlo[x][s] = w8_quant_lo(o.p[x], o.e[x], cmin[c][x]);
      o.qhi[x][s] = w8_quant_hi(o.p[x], o.e[x], cmax[c][x]);
    }
    if (is_leaf[c]) {
      lmask |= 1u << s;
      nleaf++;
      max_count = flag_count[c] > max_count ? flag_count[c] : max_count;
    } else {
      imask |= 1u << s;
      ninner++;
    }
  }
  const uint32_t stride = max_count + 1u;
  o.imask = (uint8_t)imask;
  o.lmask = (uint8_t)lmask;
  o.stride = (uint8_t)stride;
  uint32_t child_base = 0, leaf_base = 0;
  if (ninner) {
    (void)atomicAdd(&a.st->pending, ninner); // before any child becomes visible: `pending` never under-counts
    child_base = atomicAdd(&a.st->tail, ninner);
  }
  if (nleaf) leaf_base = atomicAdd(&a.st->rec_tail, nleaf * stride);
  const bool fits = child_base + ninner <= a.cap_nodes && leaf_base + nleaf * stride + 2u <= a.cap_recs && stride < 32u &&
                    child_base + ninner < (1u << kW8BaseBits) && leaf_base + nleaf * stride < (1u << kW8BaseBits);
  if (!fits) { // (cannot happen with the capacities api.hip computes; a walk never sees the result: `failed` is checked on the host)
    atomicExch(&a.st->failed, 1u);
    if (ninner) (void)atomicSub(&a.st->pending, ninner);
    o.imask = o.lmask = 0;
    o.child_base = o.leaf_base = 0;
    a.out[my] = o;
    return;
  }
  o.child_base = child_base;
  o.leaf_base = leaf_base;
  uint32_t ri = 0, rl = 0;
  for (int s = 0; s < 8; s++) {
    const int c = kid_in_slot[s];
    if (c < 0) continue;
    if (!is_leaf[c]) {
      (void)atomicExch(a.queue + child_base + ri, kids[c]);
      ri++;
    } else {
      W8Rec *r = a.recs + leaf_base + rl * stride;
      W8Rec box;
      for (int x = 0; x < 3; x++) {
        box.w[x] = __float_as_uint(cmin[c][x]);
        box.w[3 + x] = __float_as_uint(cmax[c][x]);
      }
      box.w[6] = flag_count[c];
      box.w[7] = box.w[8] = box.w[9] = 0u;
      r[0] = box;
      const W8Rec *src = reinterpret_cast<const W8Rec *>(a.tris) + first[c];
      for (uint32_t j = 0; j < flag_count[c]; j++) r[1 + j] = src[j];
      rl++;
    }
  }
  a.out[my] = o;
}

__global__ __launch_bounds__(256) void k_w8_build(const W8BuildArgs a) {
  // One loop with ONE back edge, taken by the whole wave until every lane is done: a lane that polls and a lane that builds
  // go through the same iteration.  (Written with a `continue` on the polling path the compiler nested the polling into an
  // inner loop of its own, and the lane holding the root — masked off until that loop ends — never got to publish what the
  // others were polling for.)
  uint32_t my = kW8Empty, polls = 0;
  bool alive = true;
  while (__ballot(alive) != 0ull) {
    uint32_t root = kW8Empty;
    if (alive) {
      if (my == kW8Empty) my = atomicAdd(&a.st->head, 1u);
      if (my >= a.cap_nodes) {
        alive = false; // beyond any node this tree can have
      } else {
        // (compare-and-swap as the polling read: performed at the memory side, so a value published from another CU or XCD
        // is seen; an idempotent atomicOr / atomicAdd of 0 is folded into a load by the compiler)
        root = atomicCAS(a.queue + my, kW8Empty, kW8Empty);
        if (root == kW8Empty) {
          // not published (yet).  `pending` counts the nodes published or about to be and not finished: once it is zero
          // every position that will ever be published has been processed, and this one never will be.
          // (looked at every 16th poll only: every waiting thread reads this ONE word, and same-address atomics serialise)
          if ((++polls & 15u) == 0u && atomicCAS(&a.st->pending, 0xFFFFFFFFu, 0xFFFFFFFFu) == 0u) {
            alive = false;
          } else if (polls > (1u << 24)) { // (a safety net, never reached: give up instead of hanging the device)
            atomicExch(&a.st->failed, 2u);
            alive = false;
          }
        }
      }
    }
    const bool work = alive && root != kW8Empty;
    if (work) {
      w8_make_node(a, my, root);
      (void)atomicSub(&a.st->pending, 1u);
      my = kW8Empty;
    }
    if (__ballot(work) == 0ull) __builtin_amdgcn_s_sleep(16); // nobody in this wave had anything to do: back off
  }
}

__global__ void k_w8_init(uint32_t *queue, W8BuildState *st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->head = 0u;
    st->tail = 1u;
    st->pending = 1u;
    st->rec_tail = 0u;
    st->failed = 0u;
    queue[0] = 0u;
  }
}

} // namespace

// Enqueue the construction on stream `s`.  `queue` (cap_nodes u32) and `st` are scratch; when the stream has drained,
// st->tail is the number of records, st->rec_tail (+ 2 records of slack the walk may read) the number of leaf records used.
hipError_t launch_w8_build(const nrt_node_f32 *nodes, const LeafTri<float> *tris, Wide8Node *out, W8Rec *recs, uint32_t *queue,
                           W8BuildState *st, uint32_t cap_nodes, uint32_t cap_recs, int num_cus, hipStream_t s) {
  hipError_t e = hipMemsetAsync(queue, 0xFF, (size_t)cap_nodes * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_w8_init, dim3(1), dim3(64), 0, s, queue, st);
  W8BuildArgs a;
  a.nodes = nodes;
  a.tris = tris;
  a.out = out;
  a.recs = recs;
  a.queue = queue;
  a.st = st;
  a.cap_nodes = cap_nodes;
  a.cap_recs = cap_recs;
  // a persistent grid that is certainly resident (waiting threads poll; nothing may wait for a block that cannot start)
  const unsigned blocks = (unsigned)std::max(1, std::min(num_cus * 2, (int)((cap_nodes + 255u) / 256u)));
  hipLaunchKernelGGL(k_w8_build, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

} // namespace nrt

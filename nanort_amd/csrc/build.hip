// nanort_amd/csrc/build.hip — binned-SAH BVH construction on gfx950.
//
// Replaces the reference's BVHAccel<T>::Build (nanort.h:1892-2149: recursive
// top-down builder, ContributeBinBuffer :1314-1367, FindCutFromBinBuffer
// :1381-1430, std::partition :1841) with a two-phase GPU builder that emits the
// reference's node format and invariants (nanort.h:498-550, 1797-1799,
// 1859-1885):
//
//   k_prim_records   per-primitive AABB + centroid (TriangleMesh::
//                    BoundingBoxAndCenter, nanort.h:958-971) and scene bounds.
//   TOP PHASE        level-synchronous, for nodes with more than kSmall
//                    primitives: k_bin (LDS bin reduction per 2048-primitive
//                    chunk, flushed with integer-ordered atomics), k_split
//                    (one wave per node, lane == bin, shuffle prefix/suffix
//                    sweeps of the SAH cost), k_partition (stable, ballot-rank
//                    scatter of the primitive records into the other buffer),
//                    k_level_setup (children of the level just partitioned +
//                    the next level's active list).  Four kernels per level:
//                    bins and accumulators are handed on clean by their
//                    consumers (clean_bins) instead of being re-initialised.
//   SUBTREE PHASE    one wave per node with <= kSmall primitives builds the
//                    whole subtree out of LDS (lane == split candidate).
//   RELAYOUT         subtree sizes bottom-up, DFS pre-order indices top-down,
//                    splice of the per-wave subtrees (the GPU analogue of the
//                    reference's shallow-tree splice, nanort.h:2040-2059), so
//                    that root == node 0 and left child == parent + 1.
//
// Differences from the reference builder, all deliberate:
//   * all three axes are binned (the reference's guard at nanort.h:1357 bins
//     X only, which degrades its trees on grid-like meshes);
//   * bins span the node's CENTROID bounds (not its AABB), so a split exists
//     whenever two centroids differ;
//   * one rule — bin(centroid) < split_bin — is used for both the cost sweep
//     and the partition (the reference mixes centroid*1/3 and sum<pos*3,
//     SURVEY.md Appendix A item 10);
//   * the partition is stable and the whole build is deterministic.
// Hit records do not depend on tree topology, which is what
// parity is judged on.
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

namespace nrt {

enum : unsigned { kBuildMorton = 1u, kBuildSubtreeDfs = 2u }; // gpu_build's build_flags

struct BuildResult {
  uint64_t num_nodes;
  uint32_t max_depth, num_leaves, num_branches, max_leaf_count;
};

constexpr int kSmall = 256;     // nodes at or below this many primitives are binned with kSmallBins bins (part of the tree's definition)
#ifndef NRT_BUILD_HANDOFF
#define NRT_BUILD_HANDOFF 256
#endif
// Nodes at or below this many primitives leave the level-synchronous top phase for the one-wave-per-node subtree phase.
// Both phases take the same decisions for a node (same bins — see node_bins —, same cost, same tie rules, same leaf rule),
// so this is a scheduling knob: any value <= kSmall gives the same tree (tools/tree_hash.py).
constexpr int kHandoff = NRT_BUILD_HANDOFF;
static_assert(kHandoff <= kSmall && kHandoff >= 64, "hand-off size");
#ifndef NRT_SUBTREE_REC_LDS
#define NRT_SUBTREE_REC_LDS 0 // 1: k_subtree copies its node's primitive records into LDS (10 KB per wave: 10 waves per CU instead of 22; measured slower, profiles/r02j_build_subtree_ab.txt)
#endif
#ifndef NRT_BIN_REPL
#define NRT_BIN_REPL 4 // copies of k_bin's LDS bins (fp64: at most 2), neighbouring lanes on different copies: 1 M 1.418 -> 1.314 ms, fp64 2.13 -> 2.08 (1 / 2 / 4 / 8 copies: 1.414 / 1.343 / 1.312 / 1.330; profiles/r05l_bin_repl_variants.txt)
#endif
#ifndef NRT_BIN_PRELOAD
#define NRT_BIN_PRELOAD 1 // k_bin requests a lane's records together instead of one by one (profiles/r03D_build_variants.txt: 10M-triangle build 14.5 -> 11.3 ms, 1M unchanged)
#endif
#ifndef NRT_BUILD_TILE
#define NRT_BUILD_TILE 2048
#endif
constexpr int kTile = NRT_BUILD_TILE; // primitives per top-phase chunk (256 threads x kTile / 256 rounds); any value gives the same tree
constexpr int kMaxBins = 64;    // top phase: lane == bin
constexpr int kSmallBins = 16;  // subtree phase: 3 x 15 candidates == 45 lanes
constexpr uint32_t kMedian = 0xFFFFFFFFu;
constexpr int kSceneReplicas = 16; // k_prim_records spreads its per-block atomics on the scene bounds over this many copies
[[maybe_unused]] constexpr int kSubStack = 48; // pending high-side children per subtree wave of k_subtree (LDS; profiling build)
constexpr int kSubStackSafe = 36; // above this many, splits are forced to the object median (depth <= log2 n more)

enum : uint32_t { KIND_SPLIT = 0, KIND_SMALL = 1, KIND_LEAF = 2 };

// ---- order-preserving integer images of floating-point values ---------------
template <typename T>
struct Ord;
template <>
struct Ord<float> {
  typedef uint32_t U;
  static __host__ __device__ __forceinline__ U enc(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  static __host__ __device__ __forceinline__ float dec(U e) {
    uint32_t u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
  }
  static __host__ __device__ __forceinline__ U lowest() { return 0u; }
  static __host__ __device__ __forceinline__ U highest() { return 0xFFFFFFFFu; }
};
template <>
struct Ord<double> {
  typedef unsigned long long U;
  static __host__ __device__ __forceinline__ U enc(double f) {
    unsigned long long u;
    __builtin_memcpy(&u, &f, 8);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
  }
  static __host__ __device__ __forceinline__ double dec(U e) {
    unsigned long long u = (e & 0x8000000000000000ull) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
    double f;
    __builtin_memcpy(&f, &u, 8);
    return f;
  }
  static __host__ __device__ __forceinline__ U lowest() { return 0ull; }
  static __host__ __device__ __forceinline__ U highest() { return 0xFFFFFFFFFFFFFFFFull; }
};

template <typename T>
struct Lim;
template <>
struct Lim<float> {
  static __device__ __forceinline__ float max() { return 3.402823466e+38f; }
  static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
};
template <>
struct Lim<double> {
  static __device__ __forceinline__ double max() { return 1.7976931348623157e+308; }
  static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
};

template <typename T>
__device__ __forceinline__ T tmin(T a, T b) {
  return (b < a) ? b : a;
}
template <typename T>
__device__ __forceinline__ T tmax(T a, T b) {
  return (a < b) ? b : a;
}

// Primitive record carried (and physically partitioned) through the build.
template <typename T>
struct alignas(8) PrimRec {
  T bmin[3];
  T bmax[3];
  T c[3];
  uint32_t prim;
};
static_assert(sizeof(PrimRec<float>) == 40, "PrimRec<float>");
static_assert(sizeof(PrimRec<double>) == 80, "PrimRec<double>");

template <typename T>
struct TopNode {
  T bmin[3], bmax[3]; // node AABB
  T cmin[3], cmax[3]; // centroid bounds
  uint32_t l, r;      // primitive range
  uint32_t depth;
  uint32_t kind;
  int32_t axis;
  uint32_t split_bin; // kMedian: object-median fallback (reference nanort.h:1849)
  uint32_t nleft;
  uint32_t child0;    // top index of the low-side child; high side is child0 + 1
  uint32_t size;      // nodes in this subtree
  uint32_t dfs;       // final node index
  uint32_t buf;       // record buffer holding [l, r) once the node stops splitting
  uint32_t chunk_base, nchunks;
  uint32_t parent;    // top index of the parent | kHighChild when this is its high-side child; kNoParent for node 0
};
constexpr uint32_t kHighChild = 0x80000000u, kNoParent = 0x7FFFFFFFu;

template <typename T>
struct BoundsAcc { // integer-ordered images: bmin[3] bmax[3] cmin[3] cmax[3]
  typename Ord<T>::U v[12];
};

constexpr int kMaxTopLevels = 120; // top-phase levels recorded for the relayout

// Device-resident state of the top phase: the host launches level after level with
// upper-bound grids and reads this back only to decide when to stop.
struct LevelInfo {
  uint32_t num_active;  // SPLIT nodes of the level being processed
  uint32_t num_chunks;
  uint32_t num_small;   // running count of subtree tasks (all levels)
  uint32_t max_depth;   // stats
  uint32_t num_leaves;
  uint32_t num_branches;
  uint32_t max_leaf_count;
  uint32_t error;       // 1: top array capacity exceeded
  uint32_t cand_begin, cand_end; // top nodes created by the previous level (candidates for this one)
  uint32_t top_count;   // top nodes allocated so far
  uint32_t child_base;  // first top index of the children created by the level being processed
  uint32_t top_cap;
  uint32_t num_levels;  // levels recorded in level_begin
  uint32_t num_nodes;   // nodes of the finished tree (k_layout)
  uint32_t level_begin[kMaxTopLevels + 2];
};

template <typename T>
__device__ __forceinline__ T bin_scale(T lo, T hi, int K) {
  const T ext = hi - lo;
  return (ext > T(0)) ? T(K) / ext : T(0);
}
// Bins of a node of n primitives: `kpack` carries the build's bin count for large nodes (low byte) and the one for nodes
// of at most kSmall primitives (second byte) — the subtree phase's lane == (axis, bin) layout holds 16.
__device__ __forceinline__ int node_bins(int kpack, uint32_t n) { return n <= (uint32_t)kSmall ? ((kpack >> 8) & 0xFF) : (kpack & 0xFF); }
template <typename T>
__device__ __forceinline__ int bin_of(T c, T lo, T scale, int K) {
  int i = (int)((c - lo) * scale);
  i = i < 0 ? 0 : i;
  return i > K - 1 ? K - 1 : i;
}
template <typename T>
__device__ __forceinline__ T half_area(const T mn[3], const T mx[3]) {
  const T a = mx[0] - mn[0], b = mx[1] - mn[1], c = mx[2] - mn[2];
  return a * b + b * c + c * a; // CalculateSurfaceArea / 2 (nanort.h:1278-1283)
}

__device__ __forceinline__ unsigned lane_id_b() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---- DPP (data-parallel primitive) moves inside 16-lane rows: VALU operand modifiers, no LDS
// crossbar (ds_bpermute) round trip.  row_shr:n = 0x110+n, row_shl:n = 0x100+n; a lane without a
// source keeps `old`, so `old` = the identity of the operation gives a clean scan step.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, dpp_u32<CTRL>(__builtin_bit_cast(uint32_t, old), __builtin_bit_cast(uint32_t, src)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double old, double src) {
  const unsigned long long o = __builtin_bit_cast(unsigned long long, old), v = __builtin_bit_cast(unsigned long long, src);
  const uint32_t lo = dpp_u32<CTRL>((uint32_t)o, (uint32_t)v), hi = dpp_u32<CTRL>((uint32_t)(o >> 32), (uint32_t)(v >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src) {
  return dpp_u32<CTRL>(old, src);
}

// One scan step over a {count, AABB} record inside each 16-lane row.
template <typename T, int CTRL>
__device__ __forceinline__ void row_scan_step(uint32_t &cnt, T mn[3], T mx[3]) {
  cnt += dpp_mov<CTRL>(0u, cnt);
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = tmin(mn[d], dpp_mov<CTRL>(Lim<T>::max(), mn[d]));
    mx[d] = tmax(mx[d], dpp_mov<CTRL>(-Lim<T>::max(), mx[d]));
  }
}
// inclusive prefix (lanes 0..i of the row) / suffix (lanes i..15 of the row)
template <typename T>
__device__ __forceinline__ void row_prefix(uint32_t &cnt, T mn[3], T mx[3]) {
  row_scan_step<T, 0x111>(cnt, mn, mx);
  row_scan_step<T, 0x112>(cnt, mn, mx);
  row_scan_step<T, 0x114>(cnt, mn, mx);
  row_scan_step<T, 0x118>(cnt, mn, mx);
}
template <typename T>
__device__ __forceinline__ void row_suffix(uint32_t &cnt, T mn[3], T mx[3]) {
  row_scan_step<T, 0x101>(cnt, mn, mx);
  row_scan_step<T, 0x102>(cnt, mn, mx);
  row_scan_step<T, 0x104>(cnt, mn, mx);
  row_scan_step<T, 0x108>(cnt, mn, mx);
}
// all-reduce min / max over the 64 lanes: 4 DPP steps inside rows (quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror), then two cross-row exchanges.
template <typename T>
__device__ __forceinline__ T wave_min(T x) {
  x = tmin(x, dpp_mov<0xB1>(x, x));
  x = tmin(x, dpp_mov<0x4E>(x, x));
  x = tmin(x, dpp_mov<0x141>(x, x));
  x = tmin(x, dpp_mov<0x140>(x, x));
  x = tmin(x, __shfl_xor(x, 16));
  x = tmin(x, __shfl_xor(x, 32));
  return x;
}
template <typename T>
__device__ __forceinline__ T wave_max(T x) {
  x = tmax(x, dpp_mov<0xB1>(x, x));
  x = tmax(x, dpp_mov<0x4E>(x, x));
  x = tmax(x, dpp_mov<0x141>(x, x));
  x = tmax(x, dpp_mov<0x140>(x, x));
  x = tmax(x, __shfl_xor(x, 16));
  x = tmax(x, __shfl_xor(x, 32));
  return x;
}

// ---------------------------------------------------------------------------
// primitive records + scene bounds
// ---------------------------------------------------------------------------
// Wave-uniform broadcast of lane `src` (an SGPR): v_readlane, no LDS crossbar round trip.
__device__ __forceinline__ uint32_t lane_bcast(uint32_t x, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, src); }
__device__ __forceinline__ float lane_bcast(float x, int src) {
  return __builtin_bit_cast(float, lane_bcast(__builtin_bit_cast(uint32_t, x), src));
}
__device__ __forceinline__ double lane_bcast(double x, int src) {
  const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
  const uint32_t lo = lane_bcast((uint32_t)v, src), hi = lane_bcast((uint32_t)(v >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long lane_bcast(unsigned long long v, int src) {
  const uint32_t lo = lane_bcast((uint32_t)v, src), hi = lane_bcast((uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov(unsigned long long old, unsigned long long src) {
  const uint32_t lo = dpp_u32<CTRL>((uint32_t)old, (uint32_t)src), hi = dpp_u32<CTRL>((uint32_t)(old >> 32), (uint32_t)(src >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// The subtree kernel is bound by VALU issue (a wave instruction costs the same for 5 active lanes as for 64), so its
// scans and reductions run on the ORDER-PRESERVING INTEGER IMAGES of the values (Ord<T>): for fp32 the compiler then
// folds each DPP move into the v_min_u32 / v_max_u32 that consumes it — one instruction per scan step and value where
// the float form (compare + select on a separately moved operand) takes three.
template <typename U>
__device__ __forceinline__ U umin_(U a, U b) {
  return a < b ? a : b;
}
template <typename U>
__device__ __forceinline__ U umax_(U a, U b) {
  return a > b ? a : b;
}
template <typename T, int CTRL>
__device__ __forceinline__ void row_scan_step_e(uint32_t &cnt, typename Ord<T>::U mn[3], typename Ord<T>::U mx[3]) {
  cnt += dpp_mov<CTRL>(0u, cnt);
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = umin_(mn[d], dpp_mov<CTRL>(Ord<T>::highest(), mn[d]));
    mx[d] = umax_(mx[d], dpp_mov<CTRL>(Ord<T>::lowest(), mx[d]));
  }
}
template <typename T>
__device__ __forceinline__ void row_prefix_e(uint32_t &cnt, typename Ord<T>::U mn[3], typename Ord<T>::U mx[3]) {
  row_scan_step_e<T, 0x111>(cnt, mn, mx);
  row_scan_step_e<T, 0x112>(cnt, mn, mx);
  row_scan_step_e<T, 0x114>(cnt, mn, mx);
  row_scan_step_e<T, 0x118>(cnt, mn, mx);
}
template <typename T>
__device__ __forceinline__ void row_suffix_e(uint32_t &cnt, typename Ord<T>::U mn[3], typename Ord<T>::U mx[3]) {
  row_scan_step_e<T, 0x101>(cnt, mn, mx);
  row_scan_step_e<T, 0x102>(cnt, mn, mx);
  row_scan_step_e<T, 0x104>(cnt, mn, mx);
  row_scan_step_e<T, 0x108>(cnt, mn, mx);
}
// All-reduce min / max with a wave-uniform result: 4 DPP steps leave every lane with its row's value (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror; `old` = the identity so that the move folds into the min / max),
// the four rows are combined through scalar registers.
template <typename U>
__device__ __forceinline__ U wave_umin(U x) {
  x = umin_(x, dpp_mov<0xB1>((U)~(U)0, x));
  x = umin_(x, dpp_mov<0x4E>((U)~(U)0, x));
  x = umin_(x, dpp_mov<0x141>((U)~(U)0, x));
  x = umin_(x, dpp_mov<0x140>((U)~(U)0, x));
  return umin_(umin_(lane_bcast(x, 0), lane_bcast(x, 16)), umin_(lane_bcast(x, 32), lane_bcast(x, 48)));
}
template <typename U>
__device__ __forceinline__ U wave_umax(U x) {
  x = umax_(x, dpp_mov<0xB1>((U)0, x));
  x = umax_(x, dpp_mov<0x4E>((U)0, x));
  x = umax_(x, dpp_mov<0x141>((U)0, x));
  x = umax_(x, dpp_mov<0x140>((U)0, x));
  return umax_(umax_(lane_bcast(x, 0), lane_bcast(x, 16)), umax_(lane_bcast(x, 32), lane_bcast(x, 48)));
}
template <typename T>
__device__ __forceinline__ T wave_min_u(T x) {
  return Ord<T>::dec(wave_umin<typename Ord<T>::U>(Ord<T>::enc(x)));
}
template <typename T>
__device__ __forceinline__ T wave_max_u(T x) {
  return Ord<T>::dec(wave_umax<typename Ord<T>::U>(Ord<T>::enc(x)));
}

// Inclusive scans over all 64 lanes (k_split: lane == bin, up to 64 bins): the row scans above, then each row takes
// the totals of the rows before (prefix) / after (suffix) it, which travel through scalar registers.
template <typename U, bool MIN>
__device__ __forceinline__ U row_carry(U x, U t_a, U t_b, U t_c, unsigned row, bool prefix) {
  // prefix: t_a, t_b, t_c = totals of rows 0, 1, 2;  suffix: totals of rows 1, 2, 3
  const U id = MIN ? (U) ~(U)0 : (U)0;
  auto op = [](U p, U q) { return MIN ? umin_(p, q) : umax_(p, q); };
  U c;
  if (prefix)
    c = row == 0 ? id : (row == 1 ? t_a : (row == 2 ? op(t_a, t_b) : op(op(t_a, t_b), t_c)));
  else
    c = row == 3 ? id : (row == 2 ? t_c : (row == 1 ? op(t_b, t_c) : op(op(t_a, t_b), t_c)));
  return op(x, c);
}
template <typename T>
__device__ __forceinline__ void wave_prefix_e(uint32_t &cnt, typename Ord<T>::U mn[3], typename Ord<T>::U mx[3], unsigned lane) {
  typedef typename Ord<T>::U U;
  row_prefix_e<T>(cnt, mn, mx);
  const unsigned row = lane >> 4;
  const uint32_t c0 = lane_bcast(cnt, 15), c1 = lane_bcast(cnt, 31), c2 = lane_bcast(cnt, 47);
  cnt += row == 0 ? 0u : (row == 1 ? c0 : (row == 2 ? c0 + c1 : c0 + c1 + c2));
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = row_carry<U, true>(mn[d], lane_bcast(mn[d], 15), lane_bcast(mn[d], 31), lane_bcast(mn[d], 47), row, true);
    mx[d] = row_carry<U, false>(mx[d], lane_bcast(mx[d], 15), lane_bcast(mx[d], 31), lane_bcast(mx[d], 47), row, true);
  }
}
template <typename T>
__device__ __forceinline__ void wave_suffix_e(uint32_t &cnt, typename Ord<T>::U mn[3], typename Ord<T>::U mx[3], unsigned lane) {
  typedef typename Ord<T>::U U;
  row_suffix_e<T>(cnt, mn, mx);
  const unsigned row = lane >> 4;
  const uint32_t c1 = lane_bcast(cnt, 16), c2 = lane_bcast(cnt, 32), c3 = lane_bcast(cnt, 48);
  cnt += row == 3 ? 0u : (row == 2 ? c3 : (row == 1 ? c2 + c3 : c1 + c2 + c3));
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = row_carry<U, true>(mn[d], lane_bcast(mn[d], 16), lane_bcast(mn[d], 32), lane_bcast(mn[d], 48), row, false);
    mx[d] = row_carry<U, false>(mx[d], lane_bcast(mx[d], 16), lane_bcast(mx[d], 32), lane_bcast(mx[d], 48), row, false);
  }
}
// value of lane - 1 (lane 0: `first`): DPP wave_shr:1
template <typename U>
__device__ __forceinline__ U wave_shr1(U first, U x) {
  return dpp_mov<0x138>(first, x);
}

template <typename E>
struct __attribute__((packed, aligned(4))) Vec3Of { // three consecutive elements of a tight xyz / ijk array (element-aligned only)
  E x, y, z;
};

template <typename T>
__global__ __launch_bounds__(256) void k_prim_records(const T *__restrict__ verts,
                                                      const uint32_t *__restrict__ faces,
                                                      const T *__restrict__ radii, bool cylinders, uint32_t n,
                                                      const uint32_t *__restrict__ prim_map,
                                                      PrimRec<T> *__restrict__ recs,
                                                      BoundsAcc<T> *__restrict__ scene) {
  typedef typename Ord<T>::U U;
  __shared__ U s_acc[12];
  if (threadIdx.x < 12) s_acc[threadIdx.x] = (threadIdx.x % 6 < 3) ? Ord<T>::highest() : Ord<T>::lowest();
  __syncthreads();
  T lo[3], hi[3], clo[3], chi[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    lo[k] = clo[k] = Lim<T>::max();
    hi[k] = chi[k] = -Lim<T>::max();
  }
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    PrimRec<T> r;
    const T third = T(1) / T(3);
    // three indices, then three vertices, each as ONE 12/24-byte load (dword-aligned vectors) instead of nine scalar gathers
    Vec3Of<uint32_t> f = {0, 0, 0};
    Vec3Of<T> v0 = {T(0), T(0), T(0)}, v1 = v0, v2 = v0;
    if (faces) {
      f = *reinterpret_cast<const Vec3Of<uint32_t> *>(faces + 3 * (size_t)i);
      v0 = *reinterpret_cast<const Vec3Of<T> *>(verts + 3 * (size_t)f.x);
      v1 = *reinterpret_cast<const Vec3Of<T> *>(verts + 3 * (size_t)f.y);
      v2 = *reinterpret_cast<const Vec3Of<T> *>(verts + 3 * (size_t)f.z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (faces) { // triangles
        const T p0 = k == 0 ? v0.x : (k == 1 ? v0.y : v0.z), p1 = k == 0 ? v1.x : (k == 1 ? v1.y : v1.z),
                p2 = k == 0 ? v2.x : (k == 1 ? v2.y : v2.z);
        r.bmin[k] = tmin(p0, tmin(p1, p2)); // nanort.h:967-968
        r.bmax[k] = tmax(p0, tmax(p1, p2));
        r.c[k] = ((p0 + p1) + p2) * third; // nanort.h:970
      } else if (!cylinders) { // spheres: SphereGeometry::BoundingBoxAndCenter (examples/particle_primitive/main.cc:124-136)
        const T c = verts[3 * (size_t)i + k], rad = radii[i];
        r.bmin[k] = c - rad;
        r.bmax[k] = c + rad;
        r.c[k] = c;
      } else { // cylinders: CylinderGeometry::BoundingBoxAndCenter (examples/cylinder_primitive/main.cc:166-205)
        const T a0 = verts[3 * (size_t)(2 * i) + k], a1 = verts[3 * (size_t)(2 * i + 1) + k];
        const T r0 = radii[2 * (size_t)i], r1 = radii[2 * (size_t)i + 1];
        r.bmin[k] = tmin(a1 - r1, a0 - r0); // std::min(second, first): identical unless NaN
        r.bmax[k] = tmax(a1 + r1, a0 + r0);
        r.c[k] = (a0 + a1) / T(2.0);
      }
      lo[k] = tmin(lo[k], r.bmin[k]);
      hi[k] = tmax(hi[k], r.bmax[k]);
      clo[k] = tmin(clo[k], r.c[k]);
      chi[k] = tmax(chi[k], r.c[k]);
    }
    r.prim = prim_map ? prim_map[i] : i; // (cylinder SEGMENTS carry their cylinder's id: k_cylinder_segments)
    recs[i] = r;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = tmin(lo[k], __shfl_xor(lo[k], off));
      hi[k] = tmax(hi[k], __shfl_xor(hi[k], off));
      clo[k] = tmin(clo[k], __shfl_xor(clo[k], off));
      chi[k] = tmax(chi[k], __shfl_xor(chi[k], off));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      atomicMin(&s_acc[k], Ord<T>::enc(lo[k]));
      atomicMax(&s_acc[3 + k], Ord<T>::enc(hi[k]));
      atomicMin(&s_acc[6 + k], Ord<T>::enc(clo[k]));
      atomicMax(&s_acc[9 + k], Ord<T>::enc(chi[k]));
    }
  }
  __syncthreads();
  // 2048 blocks hammering the same 12 words with device-scope atomics serialise at the memory side (~40 us of this
  // kernel's 60): each block reduces into one of kSceneReplicas copies, combined by combine_scene() before use
  BoundsAcc<T> *rep = scene + 1 + (blockIdx.x % kSceneReplicas);
  if (threadIdx.x < 12) {
    if (threadIdx.x % 6 < 3)
      atomicMin(&rep->v[threadIdx.x], s_acc[threadIdx.x]);
    else
      atomicMax(&rep->v[threadIdx.x], s_acc[threadIdx.x]);
  }
}

template <typename T>
struct GBins { // per active node, integer-ordered, accumulated with global atomics
  uint32_t count[3][kMaxBins];
  typename Ord<T>::U bmin[3][kMaxBins][3];
  typename Ord<T>::U bmax[3][kMaxBins][3];
};

// Bins and child accumulators are kept CLEAN between uses instead of being re-initialised by a kernel per level: whoever
// consumes slot `a` of a level with A active nodes resets slots a and a + A (k_split: the bins; k_level_setup: the child
// accumulators), so slots [0, 2A) — all the next level can use — are clean whatever the memory held before; slot 0 is
// reset by k_init_scene.
template <typename T>
__device__ __forceinline__ void clean_bins(GBins<T> *g, int k, unsigned bin) {
  g->count[k][bin] = 0;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    g->bmin[k][bin][d] = Ord<T>::highest();
    g->bmax[k][bin][d] = Ord<T>::lowest();
  }
}
template <typename T>
__device__ __forceinline__ void clean_acc(BoundsAcc<T> *acc) {
  typedef typename Ord<T>::U U;
#pragma unroll
  for (int j = 0; j < 12; j++) acc->v[j] = (j % 6 < 3) ? (U)Ord<T>::highest() : (U)Ord<T>::lowest();
}

// TOP LEVELS: a level of at most kRepNodes active nodes whose chunks all add into ONE node's bins and child accumulators sends
// hundreds of blocks' atomics through the same few cache lines (level 0 of a 1 M-triangle build: 489 blocks on 84 lines of bins
// and 2 of accumulators, ~75 requests per microsecond and line: +20 us on k_bin, +7 on k_partition).  Nodes of at least
// kRepMinChunks chunks (131 072 primitives) on such a level therefore accumulate into kRep COPIES — chunk c of the node into copy c % kRep — kept
// behind the regular slots (gbins[max_active ...], child_acc[2 max_active ...]); the consumer (k_split / make_children) folds
// the copies and hands them on clean like the regular slots.  Sums, minima and maxima of integers: the result is the same.
#ifndef NRT_TOP_REP
#define NRT_TOP_REP 4
#endif
constexpr uint32_t kRep = NRT_TOP_REP, kRepNodes = 8, kRepMinChunks = 64;
__device__ __forceinline__ bool rep_node(uint32_t num_active, uint32_t nchunks) {
  return kRep > 1u && num_active <= kRepNodes && nchunks >= kRepMinChunks;
}

template <typename T>
__global__ void k_init_scene(BoundsAcc<T> *scene, LevelInfo *info, uint32_t top_cap, GBins<T> *gbins, BoundsAcc<T> *child_acc,
                             uint32_t max_active) {
  if (blockIdx.x > 0) { // blocks 1 .. kRepNodes * kRep: one copy of the top levels' bins each (+ its two child accumulators)
    const uint32_t c = blockIdx.x - 1u;
    for (int k = 0; k < 3; k++) clean_bins<T>(&gbins[max_active + c], k, threadIdx.x);
    if (threadIdx.x < 2) clean_acc<T>(&child_acc[2u * max_active + 2u * c + threadIdx.x]);
    return;
  }
  if (threadIdx.x < 12)
    for (int r = 0; r <= kSceneReplicas; r++) scene[r].v[threadIdx.x] = (threadIdx.x % 6 < 3) ? Ord<T>::highest() : Ord<T>::lowest();
  for (int k = 0; k < 3; k++) clean_bins<T>(&gbins[0], k, threadIdx.x); // 64 threads == kMaxBins
  if (threadIdx.x < 2) clean_acc<T>(&child_acc[threadIdx.x]);
  if (threadIdx.x == 0) {
    info->num_active = 0;
    info->num_chunks = 0;
    info->num_small = 0;
    info->max_depth = 0;
    info->num_leaves = 0;
    info->num_branches = 0;
    info->max_leaf_count = 0;
    info->error = 0;
    info->cand_begin = 0;
    info->cand_end = 1;
    info->top_count = 1;
    info->child_base = 1;
    info->top_cap = top_cap;
    info->num_levels = 1;
    info->level_begin[0] = 0;
    info->level_begin[1] = 1;
  }
}

// The reference's leaf rule (nanort.h:1781-1783): a range of at most min_leaf_primitives, or one at the depth cap, is a leaf.
struct LeafRule {
  uint32_t max_depth, leaf_max; // leaf_max = max(min_leaf_primitives, 1)
};
template <typename T>
__device__ __forceinline__ uint32_t classify(uint32_t n, uint32_t depth, LeafRule rule) {
  if (n <= (uint32_t)kHandoff) return KIND_SMALL;
  if (depth >= rule.max_depth || n <= rule.leaf_max) return KIND_LEAF; // the rule applied to a node too large for one wave
  return KIND_SPLIT;
}

// scene[0] <- min / max over the replicas k_prim_records reduced into (idempotent); called by threads 0..11 of one block
template <typename T>
__device__ __forceinline__ void combine_scene(BoundsAcc<T> *scene, unsigned j) {
  typename Ord<T>::U x = scene[0].v[j];
  for (int r = 1; r <= kSceneReplicas; r++) {
    const typename Ord<T>::U y = scene[r].v[j];
    x = (j % 6 < 3) ? (y < x ? y : x) : (y > x ? y : x);
  }
  scene[0].v[j] = x;
}
template <typename T>
__global__ void k_combine_scene(BoundsAcc<T> *scene) {
  if (threadIdx.x < 12) combine_scene<T>(scene, threadIdx.x);
}

template <typename T>
__global__ void k_make_root(BoundsAcc<T> *scene, uint32_t n, LeafRule rule, uint32_t buf, TopNode<T> *top,
                            uint32_t *small_list, LevelInfo *info) {
  if (threadIdx.x < 12) combine_scene<T>(scene, threadIdx.x);
  __syncthreads();
  if (threadIdx.x != 0) return;
  TopNode<T> t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    t.bmin[k] = Ord<T>::dec(scene->v[k]);
    t.bmax[k] = Ord<T>::dec(scene->v[3 + k]);
    t.cmin[k] = Ord<T>::dec(scene->v[6 + k]);
    t.cmax[k] = Ord<T>::dec(scene->v[9 + k]);
  }
  t.l = 0;
  t.r = n;
  t.depth = 0;
  t.kind = classify<T>(n, 0, rule);
  t.axis = 0;
  t.split_bin = kMedian;
  t.nleft = 0;
  t.child0 = 0;
  t.size = 1;
  t.dfs = 0;
  t.buf = buf;
  t.chunk_base = 0;
  t.nchunks = 0;
  t.parent = kNoParent;
  top[0] = t;
  if (t.kind == KIND_SMALL) small_list[atomicAdd(&info->num_small, 1u)] = 0;
}

// ---------------------------------------------------------------------------
// Morton pre-pass: 30-bit code of the centroid inside the scene's centroid bounds, then a
// stable LSD radix sort of (code, record index) pairs, 8 bits per pass.  Scatter is done
// wave by wave: each wave owns a contiguous 1024-key slice of the block's tile and ranks
// one row of 64 keys at a time with __ballot ("which lanes hold my digit"), so equal
// digits keep their order without any atomics.
// ---------------------------------------------------------------------------
constexpr int kSortTile = 4096; // keys per block (256 threads x 16 rows of 64 per wave)

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_morton_keys(const PrimRec<T> *__restrict__ recs, uint32_t n,
                                                     const BoundsAcc<T> *__restrict__ scene,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t q[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T lo = Ord<T>::dec(scene->v[6 + k]), hi = Ord<T>::dec(scene->v[9 + k]);
    q[k] = (uint32_t)bin_of<T>(recs[i].c[k], lo, bin_scale<T>(lo, hi, 1024), 1024);
  }
  keys[i] = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
  vals[i] = i;
}

// block_hist is digit-major: [256 digits][num_blocks]
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t *__restrict__ keys, uint32_t n, int shift,
                                                    uint32_t *__restrict__ block_hist, uint32_t num_blocks) {
  __shared__ uint32_t s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t k = threadIdx.x; k < (uint32_t)kSortTile; k += 256u) {
    const uint32_t i = base + k;
    if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  block_hist[(size_t)threadIdx.x * num_blocks + blockIdx.x] = s_h[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_radix_scan(uint32_t *hist, uint32_t count) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < count; base += 1024u) {
    const uint32_t i = base + tid;
    const uint32_t v = i < count ? hist[i] : 0u;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off);
      if (lane >= (unsigned)off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t pre = s_carry;
    for (unsigned j = 0; j < w; j++) pre += s_wave[j];
    if (i < count) hist[i] = pre + inc - v;
    __syncthreads();
    if (tid == 1023) s_carry = pre + inc;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                       const uint32_t *__restrict__ vals_in,
                                                       uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                       uint32_t n, int shift, const uint32_t *__restrict__ block_hist,
                                                       uint32_t num_blocks) {
  __shared__ uint32_t s_off[4][256]; // running output offset per wave per digit
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t base = blockIdx.x * kSortTile + w * (kSortTile / 4);
  for (int d = lane; d < 256; d += 64) s_off[w][d] = 0;
  __syncthreads();
  // per-wave digit counts of its slice
  for (uint32_t r = 0; r < (uint32_t)kSortTile / 4; r += 64u) {
    const uint32_t i = base + r + lane;
    if (i < n) atomicAdd(&s_off[w][(keys_in[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  { // exclusive prefix over the 4 waves + the block's global offset, one digit per thread
    const uint32_t g = block_hist[(size_t)tid * num_blocks + blockIdx.x];
    uint32_t run = g;
    for (int ww = 0; ww < 4; ww++) {
      const uint32_t c = s_off[ww][tid];
      s_off[ww][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t r = 0; r < (uint32_t)kSortTile / 4; r += 64u) {
    const uint32_t i = base + r + lane;
    const bool valid = i < n;
    uint32_t key = 0, val = 0, d = 0;
    if (valid) {
      key = keys_in[i];
      val = vals_in[i];
      d = (key >> shift) & 255u;
    }
    // lanes holding the same digit as me (invalid lanes form their own group via bit 8)
    unsigned long long same = valid ? __ballot(valid) : ~__ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    if (valid) {
      const uint32_t rank = (uint32_t)__builtin_popcountll(same & lt);
      const uint32_t off = s_off[w][d];
      keys_out[off + rank] = key;
      vals_out[off + rank] = val;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (same & lt) == 0ull) s_off[w][d] += (uint32_t)__builtin_popcountll(same); // group leader
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_records(const PrimRec<T> *__restrict__ in,
                                                        const uint32_t *__restrict__ order, uint32_t n,
                                                        PrimRec<T> *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] = in[order[i]];
}

// ---------------------------------------------------------------------------
// top phase
// ---------------------------------------------------------------------------

// One block: compacts the SPLIT nodes among top[cand_begin, cand_end) into the
// active list (in order) and assigns each its chunks.
// The two children of active node `a` of the level just partitioned: ranges from the split, boxes from the accumulators
// k_partition reduced into (then reset, see clean_bins).
template <typename T>
__device__ __forceinline__ void make_children(TopNode<T> *top, const uint32_t *active, BoundsAcc<T> *child_acc, uint32_t a,
                                              uint32_t num_active, uint32_t max_active, LeafRule rule, uint32_t dst_buf,
                                              uint32_t *small_list, LevelInfo *info) {
  const TopNode<T> p = top[active[a]];
  bool small[2];
  uint32_t ci2[2];
  const bool folded = rep_node(num_active, p.nchunks); // (a top level: the node's chunks reduced into kRep copies)
  for (uint32_t c = 0; c < 2; c++) {
    TopNode<T> t;
    BoundsAcc<T> &acc = child_acc[2 * a + c];
    typename Ord<T>::U m[12];
#pragma unroll
    for (int j = 0; j < 12; j++) m[j] = acc.v[j];
    if (folded) { // (two copies are requested together before they are used and cleaned only then: kRep / 2 round trips, not kRep)
      static_assert(kRep % 2u == 0u || kRep == 1u, "copies are folded in pairs");
      static_assert((kRep & (kRep - 1u)) == 0u, "a copy is selected with & (kRep - 1): NRT_TOP_REP must be a power of two");
      for (uint32_t r0 = 0; r0 < kRep; r0 += 2u) {
        BoundsAcc<T> *ar = &child_acc[2u * max_active + 2u * (a * kRep + r0) + c]; // (copy r0 + 1 of this child: two records on)
        typename Ord<T>::U x[2][12];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int j = 0; j < 12; j++) x[q][j] = ar[2 * q].v[j];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int j = 0; j < 12; j++) m[j] = (j % 6 < 3) ? (x[q][j] < m[j] ? x[q][j] : m[j]) : (x[q][j] > m[j] ? x[q][j] : m[j]);
        clean_acc<T>(&ar[0]);
        clean_acc<T>(&ar[2]);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      t.bmin[k] = Ord<T>::dec(m[k]);
      t.bmax[k] = Ord<T>::dec(m[3 + k]);
      t.cmin[k] = Ord<T>::dec(m[6 + k]);
      t.cmax[k] = Ord<T>::dec(m[9 + k]);
    }
    clean_acc<T>(&acc);
    if (a + num_active < max_active) clean_acc<T>(&child_acc[2 * (a + num_active) + c]);
    t.l = c == 0 ? p.l : p.l + p.nleft;
    t.r = c == 0 ? p.l + p.nleft : p.r;
    t.depth = p.depth + 1;
    t.kind = classify<T>(t.r - t.l, t.depth, rule);
    t.axis = 0;
    t.split_bin = kMedian;
    t.nleft = 0;
    t.child0 = 0;
    t.size = 1;
    t.dfs = 0;
    t.buf = dst_buf;
    t.chunk_base = 0;
    t.nchunks = 0;
    t.parent = active[a] | (c ? kHighChild : 0u);
    const uint32_t ci = p.child0 + c;
    top[ci] = t;
    small[c] = t.kind == KIND_SMALL;
    ci2[c] = ci;
  }
  // the subtree tasks of the wave's lanes are appended with ONE atomic (late levels hand over thousands of tasks: one
  // device-scope atomic per task on one word was most of a narrow level's k_level_setup)
  const unsigned long long m0 = __ballot(small[0]), m1 = __ballot(small[1]), live = __ballot(true);
  const uint32_t n0 = (uint32_t)__builtin_popcountll(m0), total = n0 + (uint32_t)__builtin_popcountll(m1);
  if (total) {
    const int leader = __builtin_ctzll(live);
    const unsigned lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&info->num_small, total);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (small[0]) small_list[base + (uint32_t)__builtin_popcountll(m0 & below)] = ci2[0];
    if (small[1]) small_list[base + n0 + (uint32_t)__builtin_popcountll(m1 & below)] = ci2[1];
  }
}

// Wide levels (more than kNarrowLevel active nodes) create their children with a grid of their own; narrow levels
// do it at the head of k_level_setup and save the launch.
#ifndef NRT_NARROW_LEVEL
#define NRT_NARROW_LEVEL 256 // (1024 -> 256 with the wave-aggregated task append: 1 M 1.582 -> 1.557 ms, fp64 2.176 -> 2.142, 10 M 11.08 -> 10.94; profiles/r05j_build_children_variants.txt)
#endif
constexpr size_t kNarrowLevel = NRT_NARROW_LEVEL;
template <typename T>
__global__ __launch_bounds__(256) void k_children(TopNode<T> *top, const uint32_t *__restrict__ active, BoundsAcc<T> *child_acc,
                                                  uint32_t max_active, LeafRule rule, uint32_t dst_buf,
                                                  uint32_t *small_list, LevelInfo *info) {
  const uint32_t a = blockIdx.x * 256u + threadIdx.x;
  const uint32_t num_active = info->num_active;
  if (a >= num_active) return;
  make_children<T>(top, active, child_acc, a, num_active, max_active, rule, dst_buf, small_list, info);
}

template <typename T>
__global__ __launch_bounds__(1024) void k_level_setup(TopNode<T> *top, uint32_t *active, uint32_t *chunk_base,
                                                       LevelInfo *info, BoundsAcc<T> *child_acc, uint32_t max_active,
                                                       LeafRule rule, uint32_t dst_buf, uint32_t *small_list,
                                                       int with_children) {
  // phase A: finish the previous level — its active list is still in `active` — by creating its children
  if (with_children) {
    const uint32_t prev_active = info->num_active;
    for (uint32_t a = threadIdx.x; a < prev_active; a += 1024u)
      make_children<T>(top, active, child_acc, a, prev_active, max_active, rule, dst_buf, small_list, info);
    __syncthreads(); // (block-wide: the children are visible to the scan below, and `active` may be rewritten)
  }
  // phase B: the new level's active list and chunk ranges
  const uint32_t cand_begin = info->cand_begin, cand_end = info->cand_end;
  __shared__ uint32_t s_wave[2][16];
  __shared__ uint32_t s_carry[2];
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 2) s_carry[tid] = 0;
  __syncthreads();
  for (uint32_t base = cand_begin; base < cand_end; base += 1024u) {
    const uint32_t i = base + tid;
    uint32_t is_active = 0, nch = 0;
    if (i < cand_end && top[i].kind == KIND_SPLIT) {
      is_active = 1;
      nch = (top[i].r - top[i].l + kTile - 1) / kTile;
    }
    uint32_t a = is_active, c = nch; // inclusive wave scans
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t ta = __shfl_up(a, off), tc = __shfl_up(c, off);
      if (lane >= (unsigned)off) {
        a += ta;
        c += tc;
      }
    }
    if (lane == 63) {
      s_wave[0][w] = a;
      s_wave[1][w] = c;
    }
    __syncthreads();
    uint32_t pa = s_carry[0], pc = s_carry[1];
    for (unsigned j = 0; j < w; j++) {
      pa += s_wave[0][j];
      pc += s_wave[1][j];
    }
    if (is_active) {
      const uint32_t slot = pa + a - 1;
      active[slot] = i;
      chunk_base[slot] = pc + c - nch;
      top[i].chunk_base = pc + c - nch;
      top[i].nchunks = nch;
    }
    __syncthreads();
    if (tid == 1023) {
      s_carry[0] = pa + a;
      s_carry[1] = pc + c;
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t A = s_carry[0];
    const uint32_t tc = info->top_count;
    if (A && ((unsigned long long)tc + 2ull * A > info->top_cap || info->num_levels >= (uint32_t)kMaxTopLevels)) {
      info->error = 1; // the host retries with a larger top array
      A = 0;
    }
    info->num_active = A;
    info->num_chunks = A ? s_carry[1] : 0;
    info->child_base = tc;
    info->cand_begin = tc;
    info->cand_end = tc + 2 * A;
    info->top_count = tc + 2 * A;
    if (A) {
      info->num_levels += 1;
      info->level_begin[info->num_levels] = tc + 2 * A;
    }
  }
}

__device__ __forceinline__ uint32_t find_task(const uint32_t *chunk_base, uint32_t num_active, uint32_t chunk) {
  // last a with chunk_base[a] <= chunk
  uint32_t lo = 0, hi = num_active;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (chunk_base[mid] <= chunk)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// The same search by one full wave, 64 probes per round (two rounds up to 4096 active nodes, where the binary search is a
// chain of twelve dependent loads at the head of every k_bin / k_partition block).
__device__ __forceinline__ uint32_t find_task_wave(const uint32_t *chunk_base, uint32_t num_active, uint32_t chunk, unsigned lane) {
  uint32_t lo = 0, hi = num_active; // chunk_base[lo] <= chunk, and (hi == num_active or chunk_base[hi] > chunk)
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 63u) >> 6;
    const uint32_t idx = lo + lane * step;
    const bool ok = idx < hi && chunk_base[idx] <= chunk; // true for a prefix of the lanes (lane 0 included)
    const uint32_t j = (uint32_t)__builtin_popcountll(__ballot(ok)) - 1u;
    lo += j * step;
    hi = hi < lo + step ? hi : lo + step;
  }
  return lo;
}

// The cut search of one node by one full wave, lane == bin: prefix (left) and suffix (right) sweeps of count and AABB,
// cost = nL*SA(L) + nR*SA(R) as in FindCutFromBinBuffer (nanort.h:1393-1422), argmin over 3 x (K-1) candidates (ties:
// lowest axis, then lowest bin).  cnt3 / mn3 / mx3: this lane's bin of each axis (integer images; an empty bin's bounds
// are ignored).  Shared by k_split (bins from the node's global slot) and k_bin (a node that fits one chunk: bins
// straight from LDS).
template <typename T>
__device__ __forceinline__ void eval_split(int K, unsigned lane, const uint32_t cnt3[3], const typename Ord<T>::U mn3[3][3],
                                           const typename Ord<T>::U mx3[3][3], int &best_axis, uint32_t &best_bin,
                                           uint32_t &best_nl) {
  typedef typename Ord<T>::U U;
  T best_cost = Lim<T>::inf();
  best_axis = 0;
  best_bin = kMedian;
  best_nl = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // scans on the integer images (see row_scan_step_e): DPP + scalar registers, no LDS crossbar
    const uint32_t cnt = cnt3[k];
    U pmn[3], pmx[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      pmn[d] = cnt ? mn3[k][d] : Ord<T>::highest();
      pmx[d] = cnt ? mx3[k][d] : Ord<T>::lowest();
    }
    uint32_t pc = cnt, sc = cnt;
    U smn[3] = {pmn[0], pmn[1], pmn[2]}, smx[3] = {pmx[0], pmx[1], pmx[2]};
    wave_prefix_e<T>(pc, pmn, pmx, lane); // inclusive over lanes 0..lane
    wave_suffix_e<T>(sc, smn, smx, lane); // inclusive over lanes lane..63
    // candidate s == lane (1..K-1): left = bins [0, s), right = bins [s, K)
    const uint32_t nl = wave_shr1<uint32_t>(0u, pc);
    T cost = Lim<T>::inf();
    {
      T lmn[3], lmx[3], rmn[3], rmx[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        lmn[d] = Ord<T>::dec(wave_shr1<U>(Ord<T>::highest(), pmn[d]));
        lmx[d] = Ord<T>::dec(wave_shr1<U>(Ord<T>::lowest(), pmx[d]));
        rmn[d] = Ord<T>::dec(smn[d]);
        rmx[d] = Ord<T>::dec(smx[d]);
      }
      if (lane >= 1 && (int)lane < K && nl > 0 && sc > 0) cost = T(nl) * half_area<T>(lmn, lmx) + T(sc) * half_area<T>(rmn, rmx);
    }
    if (!(cost == cost)) cost = Lim<T>::inf(); // a NaN cost never wins
    // wave argmin, ties -> lowest lane
    const U ecost = Ord<T>::enc(cost);
    const U ebest = wave_umin<U>(ecost);
    const T c = Ord<T>::dec(ebest);
    const int who = (int)__builtin_ctzll(__ballot(ecost == ebest));
    if (c < best_cost) { // ties -> lowest axis
      best_cost = c;
      best_axis = k;
      best_bin = (uint32_t)who;
      best_nl = lane_bcast(nl, who);
    }
  }
}

// LDS bin reduction of one chunk; flush to the node's global bins; per-chunk
// per-bin counts kept for the stable partition's offsets.  A node that fits ONE chunk has all its bins right here:
// the block's first wave does the node's cut search on the spot (no global bins, no histogram, nothing left for k_split).
template <typename T>
__global__ __launch_bounds__(256) void k_bin(TopNode<T> *top, const uint32_t *__restrict__ active,
                                             const uint32_t *__restrict__ chunk_base, const LevelInfo *info,
                                             const PrimRec<T> *__restrict__ recs, int kpack, GBins<T> *gbins,
                                             uint32_t *__restrict__ chunk_hist, uint32_t *chunk_left_base, uint32_t max_active) {
  if (blockIdx.x >= info->num_chunks) return; // grids are upper bounds
  const uint32_t num_active = info->num_active;
  typedef typename Ord<T>::U U;
  // The bins are kept in R copies, neighbouring lanes on different ones (the copy index is the fastest-running one: the R
  // copies of a word lie in R different banks): lanes of a wave that end in the same bin — the rule on coherent input —
  // queue R ways less on one LDS word; the copies are folded into copy 0 before anything reads the bins.
  constexpr int R = sizeof(T) == 4 ? NRT_BIN_REPL : (NRT_BIN_REPL > 2 ? 2 : NRT_BIN_REPL);
  static_assert(R >= 1 && (R & (R - 1)) == 0, "a lane's copy is threadIdx.x & (R - 1): NRT_BIN_REPL must be a power of two");
  __shared__ uint32_t s_cnt[3][kMaxBins][R];
  __shared__ U s_min[3][kMaxBins][3][R];
  __shared__ U s_max[3][kMaxBins][3][R];
  __shared__ uint32_t s_task;
  const unsigned rep = threadIdx.x & (unsigned)(R - 1);
  const uint32_t chunk = blockIdx.x;
  if (threadIdx.x < 64u) {
    const uint32_t t = find_task_wave(chunk_base, num_active, chunk, threadIdx.x);
    if (threadIdx.x == 0) s_task = t;
  }
  for (int i = threadIdx.x; i < 3 * kMaxBins * R; i += 256) {
    const int k = i / (kMaxBins * R), b = (i / R) % kMaxBins, r = i % R;
    s_cnt[k][b][r] = 0;
    for (int d = 0; d < 3; d++) {
      s_min[k][b][d][r] = Ord<T>::highest();
      s_max[k][b][d][r] = Ord<T>::lowest();
    }
  }
  __syncthreads();
  const uint32_t a = s_task;
  TopNode<T> &nd = top[active[a]];
  const uint32_t begin = nd.l + (chunk - chunk_base[a]) * kTile;
  const uint32_t end = (nd.r - begin < (uint32_t)kTile) ? nd.r : begin + kTile;
  const int K = node_bins(kpack, nd.r - nd.l);
  const bool whole_node = nd.nchunks == 1u;
  T lo[3], sc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    lo[k] = nd.cmin[k];
    sc[k] = bin_scale<T>(nd.cmin[k], nd.cmax[k], K);
  }
  // Each lane takes 8 CONSECUTIVE primitives and merges runs that fall into the same bin in registers before touching LDS.
  // Meshes arrive spatially coherent (neighbouring triangles are neighbours in the array), so at the top levels — few
  // nodes, wide bins — whole tiles land in one or two bins of an axis and every LDS atomic of a wave would hit the same
  // address (serialised 64 ways).  Run merging cuts those atomics 8x there and costs a compare per axis where the input
  // is incoherent (deep levels).
  static_assert(kTile % 256 == 0, "k_bin: a whole number of primitives per lane");
  constexpr uint32_t kPerLane = kTile / 256;
  {
    int pb[3] = {-1, -1, -1};
    uint32_t pc[3] = {0, 0, 0};
    U pmin[3][3], pmax[3][3];
    // (a partly filled chunk — the usual case a few levels down — is still spread over all 256 lanes: fewer records per lane)
    const uint32_t per = (end - begin + 255u) >> 8;
    const uint32_t p0 = begin + threadIdx.x * per;
    auto flush = [&](int k) {
      atomicAdd(&s_cnt[k][pb[k]][rep], pc[k]);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        atomicMin(&s_min[k][pb[k]][d][rep], pmin[k][d]);
        atomicMax(&s_max[k][pb[k]][d][rep], pmax[k][d]);
      }
    };
    auto bin_rec = [&](const PrimRec<T> &r) {
      U emin[3], emax[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        emin[d] = Ord<T>::enc(r.bmin[d]);
        emax[d] = Ord<T>::enc(r.bmax[d]);
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int b = bin_of<T>(r.c[k], lo[k], sc[k], K);
        if (b == pb[k]) {
          pc[k]++;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            pmin[k][d] = emin[d] < pmin[k][d] ? emin[d] : pmin[k][d];
            pmax[k][d] = emax[d] > pmax[k][d] ? emax[d] : pmax[k][d];
          }
        } else {
          if (pb[k] >= 0) flush(k);
          pb[k] = b;
          pc[k] = 1;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            pmin[k][d] = emin[d];
            pmax[k][d] = emax[d];
          }
        }
      }
    };
#if NRT_BIN_PRELOAD
    // The lane's records are requested together, kHold at a time, before the first of them is binned: a chunk is about all
    // the work a CU gets at the top levels (two blocks per CU), so a load issued right before its use is an exposed round
    // trip — eight of them per lane where this pays one (fp64: two).
    constexpr uint32_t kHold = sizeof(T) == 4 ? (kPerLane < 8u ? kPerLane : 8u) : (kPerLane < 4u ? kPerLane : 4u);
    static_assert(kPerLane % kHold == 0, "k_bin: whole batches");
    for (uint32_t h0 = 0; h0 < per; h0 += kHold) {
      PrimRec<T> rr[kHold];
#pragma unroll
      for (uint32_t q = 0; q < kHold; q++)
        if (h0 + q < per && p0 + h0 + q < end) rr[q] = recs[p0 + h0 + q];
#pragma unroll
      for (uint32_t q = 0; q < kHold; q++)
        if (h0 + q < per && p0 + h0 + q < end) bin_rec(rr[q]);
    }
#else
    for (uint32_t it = 0; it < per; it++) {
      const uint32_t p = p0 + it;
      if (p >= end) break;
      bin_rec(recs[p]);
    }
#endif
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (pb[k] >= 0) flush(k);
  }
  __syncthreads();
  if (R > 1) { // fold the copies into copy 0
    for (int i = threadIdx.x; i < 3 * kMaxBins; i += 256) {
      const int k = i / kMaxBins, b = i % kMaxBins;
      uint32_t c = s_cnt[k][b][0];
#pragma unroll
      for (int r = 1; r < R; r++) c += s_cnt[k][b][r];
      s_cnt[k][b][0] = c;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        U mn = s_min[k][b][d][0], mx = s_max[k][b][d][0];
#pragma unroll
        for (int r = 1; r < R; r++) {
          const U a_ = s_min[k][b][d][r], b_ = s_max[k][b][d][r];
          mn = a_ < mn ? a_ : mn;
          mx = b_ > mx ? b_ : mx;
        }
        s_min[k][b][d][0] = mn;
        s_max[k][b][d][0] = mx;
      }
    }
    __syncthreads();
  }
  if (whole_node) { // the node's complete bins are in LDS: split it here (k_split skips it)
    if (threadIdx.x < 64u) {
      const unsigned lane = threadIdx.x;
      uint32_t cnt3[3] = {0, 0, 0};
      U mn3[3][3], mx3[3][3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if ((int)lane < K) cnt3[k] = s_cnt[k][lane][0];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          mn3[k][d] = s_min[k][lane][d][0];
          mx3[k][d] = s_max[k][lane][d][0];
        }
      }
      int best_axis;
      uint32_t best_bin, best_nl;
      eval_split<T>(K, lane, cnt3, mn3, mx3, best_axis, best_bin, best_nl);
      if (best_bin == kMedian) best_nl = (nd.r - nd.l) >> 1; // no separable centroids: object median (nanort.h:1849)
      if (lane == 0) {
        nd.axis = best_axis;
        nd.split_bin = best_bin;
        nd.nleft = best_nl;
        nd.child0 = info->child_base + 2 * a;
        chunk_left_base[chunk] = 0; // the chunk's low side starts at the node's
      }
      // this node's global slot was never touched; the slot the next level may hand out is reset as k_split would
      if (a + num_active < max_active)
        for (int k = 0; k < 3; k++) clean_bins<T>(&gbins[a + num_active], k, lane);
    }
    return;
  }
  GBins<T> *g = rep_node(num_active, nd.nchunks) ? &gbins[max_active + a * kRep + ((chunk - chunk_base[a]) & (kRep - 1u))] : &gbins[a];
  for (int i = threadIdx.x; i < 3 * kMaxBins; i += 256) {
    const int k = i / kMaxBins, b = i % kMaxBins;
    const uint32_t c = s_cnt[k][b][0];
    chunk_hist[(size_t)chunk * (3 * kMaxBins) + i] = c;
    if (c) {
      atomicAdd(&g->count[k][b], c);
      for (int d = 0; d < 3; d++) {
        atomicMin(&g->bmin[k][b][d], s_min[k][b][d][0]);
        atomicMax(&g->bmax[k][b][d], s_max[k][b][d][0]);
      }
    }
  }
}

// One wave per active node: lane == bin.  Prefix (left) and suffix (right)
// sweeps of count and AABB by shuffles, cost = nL*SA(L) + nR*SA(R) as in
// FindCutFromBinBuffer (nanort.h:1393-1422), argmin over 3 x (K-1) candidates.
template <typename T>
__global__ __launch_bounds__(64) void k_split(TopNode<T> *top, const uint32_t *__restrict__ active,
                                              GBins<T> *gbins, int kpack,
                                              const uint32_t *__restrict__ chunk_hist, uint32_t *chunk_left_base,
                                              const LevelInfo *info, uint32_t max_active) {
  const uint32_t a = blockIdx.x;
  const uint32_t num_active = info->num_active;
  if (a >= num_active) return;
  const uint32_t next_top_base = info->child_base;
  const unsigned lane = threadIdx.x;
  TopNode<T> &nd = top[active[a]];
  GBins<T> &g = gbins[a];
  typedef typename Ord<T>::U U;
  // everything this wave reads from memory is requested up front — the three axes' bins and the node's range — so that the
  // kernel pays one round trip instead of one per axis (the bins are reset right after, and stores pin later loads in place)
  const uint32_t n = nd.r - nd.l, cb = nd.chunk_base, nch = nd.nchunks;
  if (nch == 1u) return; // a node of one chunk was split by its k_bin block (which also reset slot a + num_active)
  const int K = node_bins(kpack, n);
  uint32_t cnt3[3] = {0, 0, 0};
  U mn3[3][3], mx3[3][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if ((int)lane < K) cnt3[k] = g.count[k][lane];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      mn3[k][d] = Ord<T>::highest();
      mx3[k][d] = Ord<T>::lowest();
      if ((int)lane < K) {
        mn3[k][d] = g.bmin[k][lane][d];
        mx3[k][d] = g.bmax[k][lane][d];
      }
    }
  }
  if (rep_node(num_active, nch)) { // a top level: the node's chunks added into kRep copies (the regular slot stayed clean)
    // (axis by axis: the kRep copies of an axis are requested together, folded, and cleaned only then — a store would pin the
    // loads behind it)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      uint32_t rc[kRep];
      U rmn[kRep][3], rmx[kRep][3];
#pragma unroll
      for (uint32_t r = 0; r < kRep; r++) {
        const GBins<T> &gr = gbins[max_active + a * kRep + r];
        rc[r] = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          rmn[r][d] = Ord<T>::highest();
          rmx[r][d] = Ord<T>::lowest();
        }
        if ((int)lane < K) {
          rc[r] = gr.count[k][lane];
#pragma unroll
          for (int d = 0; d < 3; d++) {
            rmn[r][d] = gr.bmin[k][lane][d];
            rmx[r][d] = gr.bmax[k][lane][d];
          }
        }
      }
#pragma unroll
      for (uint32_t r = 0; r < kRep; r++) {
        cnt3[k] += rc[r];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          mn3[k][d] = rmn[r][d] < mn3[k][d] ? rmn[r][d] : mn3[k][d];
          mx3[k][d] = rmx[r][d] > mx3[k][d] ? rmx[r][d] : mx3[k][d];
        }
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < kRep; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) clean_bins<T>(&gbins[max_active + a * kRep + r], k, lane);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    clean_bins<T>(&g, k, lane); // consumed: leave the slot clean for the next level (see clean_bins)
    if (a + num_active < max_active) clean_bins<T>(&gbins[a + num_active], k, lane);
  }
  int best_axis;
  uint32_t best_bin, best_nl;
  eval_split<T>(K, lane, cnt3, mn3, mx3, best_axis, best_bin, best_nl);
  if (best_bin == kMedian) best_nl = n >> 1; // no separable centroids: object median (nanort.h:1849)

  // per-chunk low-side counts -> exclusive prefix inside this node
  uint32_t carry = 0;
  for (uint32_t j0 = 0; j0 < nch; j0 += 64u) {
    const uint32_t j = j0 + lane;
    uint32_t left = 0;
    if (j < nch) {
      if (best_bin == kMedian) {
        const uint32_t start = j * kTile, len = (n - start < (uint32_t)kTile) ? n - start : kTile;
        left = best_nl > start ? (best_nl - start < len ? best_nl - start : len) : 0;
      } else {
        // all 64 counts of the winning axis as 16 independent 16-byte loads (one latency), masked to bins < best_bin —
        // a loop over best_bin single loads costs best_bin latencies and was most of this kernel
        const uint4 *h4 = reinterpret_cast<const uint4 *>(chunk_hist + (size_t)(cb + j) * (3 * kMaxBins) + best_axis * kMaxBins);
        uint4 hv[kMaxBins / 4];
#pragma unroll
        for (int q = 0; q < kMaxBins / 4; q++) hv[q] = h4[q];
#pragma unroll
        for (int q = 0; q < kMaxBins / 4; q++) {
          const uint32_t b0 = 4u * (uint32_t)q;
          left += (b0 < best_bin ? hv[q].x : 0u) + (b0 + 1u < best_bin ? hv[q].y : 0u) + (b0 + 2u < best_bin ? hv[q].z : 0u) +
                  (b0 + 3u < best_bin ? hv[q].w : 0u);
        }
      }
    }
    uint32_t inc = left;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off);
      if (lane >= (unsigned)off) inc += t;
    }
    if (j < nch) chunk_left_base[cb + j] = carry + inc - left;
    carry += __shfl(inc, 63);
  }
  if (lane == 0) {
    nd.axis = best_axis;
    nd.split_bin = best_bin;
    nd.nleft = best_nl;
    nd.child0 = next_top_base + 2 * a;
  }
}

// Stable partition of one chunk into the other record buffer + reduction of
// the children's AABB / centroid bounds.
template <typename T>
__global__ __launch_bounds__(256) void k_partition(const TopNode<T> *__restrict__ top,
                                                   const uint32_t *__restrict__ active,
                                                   const uint32_t *__restrict__ chunk_base, const LevelInfo *info,
                                                   const uint32_t *__restrict__ chunk_left_base,
                                                   const PrimRec<T> *__restrict__ src, PrimRec<T> *__restrict__ dst,
                                                   int kpack, BoundsAcc<T> *child_acc, uint32_t max_active) {
  typedef typename Ord<T>::U U;
  __shared__ uint32_t s_task;
  __shared__ uint32_t s_w[2][4];
  __shared__ U s_acc[2][12];
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= info->num_chunks) return; // grids are upper bounds
  const uint32_t num_active = info->num_active;
  if (tid < 64u) {
    const uint32_t t = find_task_wave(chunk_base, num_active, chunk, tid);
    if (tid == 0) s_task = t;
  }
  if (tid < 24) s_acc[tid / 12][tid % 12] = ((tid % 12) % 6 < 3) ? (U)Ord<T>::highest() : (U)Ord<T>::lowest();
  __syncthreads();
  const uint32_t a = s_task;
  const TopNode<T> &nd = top[active[a]];
  const uint32_t off_in_node = (chunk - chunk_base[a]) * kTile;
  const uint32_t begin = nd.l + off_in_node;
  const uint32_t end = (nd.r - begin < (uint32_t)kTile) ? nd.r : begin + kTile;
  const uint32_t left_base = chunk_left_base[chunk];
  const uint32_t right_base = off_in_node - left_base;
  const int axis = nd.axis;
  const uint32_t split_bin = nd.split_bin, nleft = nd.nleft;
  const T lo = axis == 0 ? nd.cmin[0] : (axis == 1 ? nd.cmin[1] : nd.cmin[2]);
  const T hi = axis == 0 ? nd.cmax[0] : (axis == 1 ? nd.cmax[1] : nd.cmax[2]);
  const T sc = bin_scale<T>(lo, hi, node_bins(kpack, nd.r - nd.l));
  const int K = node_bins(kpack, nd.r - nd.l);

  T acc_lo[2][3], acc_hi[2][3], acc_clo[2][3], acc_chi[2][3];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      acc_lo[s][d] = acc_clo[s][d] = Lim<T>::max();
      acc_hi[s][d] = acc_chi[s][d] = -Lim<T>::max();
    }

  uint32_t run_l = 0, run_r = 0;
  // (the records of the next round are requested before this round's barriers: a chunk is 8 rounds, and their loads would
  // otherwise be 8 exposed round trips — there are only about two blocks per CU to hide them)
  PrimRec<T> r_next;
  if (begin + tid < end) r_next = src[begin + tid];
  for (uint32_t p0 = begin; p0 < end; p0 += 256u) {
    const uint32_t p = p0 + tid;
    const bool valid = p < end;
    PrimRec<T> r;
    bool left = false;
    if (valid) r = r_next;
    if (p + 256u < end) r_next = src[p + 256u];
    if (valid) {
      if (split_bin == kMedian) {
        left = (p - nd.l) < nleft;
      } else {
        const T c = axis == 0 ? r.c[0] : (axis == 1 ? r.c[1] : r.c[2]);
        left = (uint32_t)bin_of<T>(c, lo, sc, K) < split_bin;
      }
    }
    const unsigned long long bl = __ballot(valid && left), br = __ballot(valid && !left);
    if (lane == 0) {
      s_w[0][w] = (uint32_t)__builtin_popcountll(bl);
      s_w[1][w] = (uint32_t)__builtin_popcountll(br);
    }
    __syncthreads();
    uint32_t pl = 0, pr = 0, tl = 0, tr = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; j++) {
      if (j < w) {
        pl += s_w[0][j];
        pr += s_w[1][j];
      }
      tl += s_w[0][j];
      tr += s_w[1][j];
    }
    if (valid) {
      const unsigned long long lt = (1ull << lane) - 1ull;
      uint32_t d;
      const int s = left ? 0 : 1;
      if (left)
        d = nd.l + left_base + run_l + pl + (uint32_t)__builtin_popcountll(bl & lt);
      else
        d = nd.l + nleft + right_base + run_r + pr + (uint32_t)__builtin_popcountll(br & lt);
      dst[d] = r;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        // select-free accumulate into side s
        if (s == 0) {
          acc_lo[0][k] = tmin(acc_lo[0][k], r.bmin[k]);
          acc_hi[0][k] = tmax(acc_hi[0][k], r.bmax[k]);
          acc_clo[0][k] = tmin(acc_clo[0][k], r.c[k]);
          acc_chi[0][k] = tmax(acc_chi[0][k], r.c[k]);
        } else {
          acc_lo[1][k] = tmin(acc_lo[1][k], r.bmin[k]);
          acc_hi[1][k] = tmax(acc_hi[1][k], r.bmax[k]);
          acc_clo[1][k] = tmin(acc_clo[1][k], r.c[k]);
          acc_chi[1][k] = tmax(acc_chi[1][k], r.c[k]);
        }
      }
    }
    run_l += tl;
    run_r += tr;
    __syncthreads();
  }
  // block reduction of the 2 x 12 child bounds, then one global atomic each
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      for (int off = 32; off > 0; off >>= 1) {
        acc_lo[s][k] = tmin(acc_lo[s][k], __shfl_xor(acc_lo[s][k], off));
        acc_hi[s][k] = tmax(acc_hi[s][k], __shfl_xor(acc_hi[s][k], off));
        acc_clo[s][k] = tmin(acc_clo[s][k], __shfl_xor(acc_clo[s][k], off));
        acc_chi[s][k] = tmax(acc_chi[s][k], __shfl_xor(acc_chi[s][k], off));
      }
      if (lane == 0) {
        atomicMin(&s_acc[s][k], Ord<T>::enc(acc_lo[s][k]));
        atomicMax(&s_acc[s][3 + k], Ord<T>::enc(acc_hi[s][k]));
        atomicMin(&s_acc[s][6 + k], Ord<T>::enc(acc_clo[s][k]));
        atomicMax(&s_acc[s][9 + k], Ord<T>::enc(acc_chi[s][k]));
      }
    }
  __syncthreads();
  if (tid < 24) {
    const int s = tid / 12, j = tid % 12;
    BoundsAcc<T> *acc = rep_node(num_active, nd.nchunks)
                            ? &child_acc[2u * max_active + 2u * (a * kRep + ((chunk - chunk_base[a]) & (kRep - 1u))) + s]
                            : &child_acc[2 * a + s];
    if (j % 6 < 3)
      atomicMin(&acc->v[j], s_acc[s][j]);
    else
      atomicMax(&acc->v[j], s_acc[s][j]);
  }
}

// A subtree task's statistics (leaves, deepest node, largest leaf) are left in fields of its own top record that only split
// nodes use, and k_layout — which visits every top record anyway — adds them up: four device-scope atomics per task on four
// neighbouring words (22 000 per 1 M-triangle build, 220 000 at 10 M, through one L2 channel at ~100 per microsecond) were a
// queue every finishing wave stood in.
template <typename T>
__device__ __forceinline__ void task_stats(TopNode<T> &task, uint32_t leaves, uint32_t deepest, uint32_t biggest_leaf) {
  task.nleft = leaves;
  task.split_bin = deepest;
  task.nchunks = biggest_leaf;
}

// ---------------------------------------------------------------------------
// subtree phase: one wave builds everything below a node of <= kSmall prims
// ---------------------------------------------------------------------------
#ifdef NRT_PROF // the one-node-per-step form lives in libnanort_hip_prof.so only: the cross-check of the row form (tests/test_gpu_build.py, tunable subtree_rows = 0)
// Pending high-side child of the per-wave subtree builder.
template <typename T>
struct SubPending {
  T bmin[3], bmax[3]; // its AABB (from the parent's bins, or reduced during the parent's median partition)
  T cmin[3], cmax[3]; // its centroid bounds (reduced during the parent's partition)
  uint16_t lo, hi, parent;
  uint16_t buf;       // which of the two permutation buffers holds [lo, hi)
  uint32_t depth;
};

// One wave per node of <= kSmall primitives: records in LDS, a 16-bit permutation ping-ponged between two
// buffers by the stable partition, LDS bin reduction (3 axes x K <= 16 bins, ds_min/ds_max on integer-ordered
// keys), lane == (axis, bin) prefix/suffix sweeps inside 16-lane groups.  The low-side child is processed next
// (so it is numbered parent + 1, pre-order); the high-side child waits on an LDS stack with its AABB and
// centroid bounds.  A node costs a chain of dependent LDS round trips, not arithmetic, so the chain is kept
// short: each lane keeps its first element (all of a node of <= 64 primitives) in registers across the binning
// and partition passes; a child's centroid bounds (and, after a median split, its AABB) are reduced in the
// parent's partition pass instead of a pass of its own; the bins are reset by the lanes that read them; wave
// reductions and the winner's broadcast go through DPP and scalar registers.  Two barriers per inner node.
template <typename T>
__global__ __launch_bounds__(64) void k_subtree(TopNode<T> *top, const uint32_t *__restrict__ small_list,
                                                const PrimRec<T> *__restrict__ recs0,
                                                const PrimRec<T> *__restrict__ recs1, int K, uint32_t min_leaf,
                                                uint32_t max_depth, typename Wire<T>::Node *scratch_nodes,
                                                uint32_t *indices, LevelInfo *info) {
  typedef typename Wire<T>::Node Node;
  typedef typename Ord<T>::U U;
#if NRT_SUBTREE_REC_LDS
  __shared__ PrimRec<T> s_rec[kHandoff];
#define NRT_SUB_REC(id_) s_rec[(id_)]
#else // records stay where they are (10 KB per subtree, contiguous: L1 / L2 hits); 10 KB less LDS per wave = more waves per CU
#define NRT_SUB_REC(id_) src[(id_)]
#endif
  __shared__ uint16_t s_perm[2][kHandoff];
  __shared__ SubPending<T> s_stack[kSubStack];
  __shared__ uint32_t s_cnt[3][kSmallBins];
  __shared__ U s_bmin[3][kSmallBins][3];
  __shared__ U s_bmax[3][kSmallBins][3];

  const unsigned lane = threadIdx.x;
  if (blockIdx.x >= info->num_small) return; // grid is an upper bound
  TopNode<T> &task = top[small_list[blockIdx.x]];
  const uint32_t L = task.l, n_all = task.r - task.l;
  const PrimRec<T> *src = (task.buf ? recs1 : recs0) + L;
  for (uint32_t i = lane; i < n_all; i += 64u) {
#if NRT_SUBTREE_REC_LDS
    s_rec[i] = src[i];
#endif
    s_perm[0][i] = (uint16_t)i;
  }
  if (lane < 3 * kSmallBins) { // bins start clean and are handed on clean by their readers
    const int k = (int)lane / kSmallBins, bq = (int)lane % kSmallBins;
    s_cnt[k][bq] = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      s_bmin[k][bq][d] = Ord<T>::highest();
      s_bmax[k][bq][d] = Ord<T>::lowest();
    }
  }
  Node *out = scratch_nodes + 2 * (size_t)L;
  uint32_t node_count = 0, leaves = 0, deepest = 0, biggest_leaf = 0;
  int sp = 0;
  const uint32_t leaf_max = min_leaf > 1u ? min_leaf : 1u;

  // current node (wave-uniform)
  uint32_t lo = 0, hi = n_all, depth = task.depth, parent = 0xFFFFu, pb = 0;
  bool is_high = false;
  T mn[3], mx[3], cmn[3], cmx[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = task.bmin[d];
    mx[d] = task.bmax[d];
    cmn[d] = task.cmin[d];
    cmx[d] = task.cmax[d];
  }
  __syncthreads();

  for (;;) {
    const uint32_t n = hi - lo;
    const uint32_t me = node_count++;
    deepest = depth > deepest ? depth : deepest;
    if (is_high && lane == 0) out[parent].data[1] = me;

    const bool leaf = n <= leaf_max || depth >= max_depth; // nanort.h:1781-1783
    // this lane's first element stays in registers for every pass over the node
    const uint32_t i_first = lo + lane;
    const bool have = i_first < hi;
    uint16_t id0 = 0;
    PrimRec<T> r0;
    if (have) {
      id0 = s_perm[pb][i_first];
      r0 = NRT_SUB_REC(id0);
    }

    Node nd;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      nd.bmin[d] = mn[d];
      nd.bmax[d] = mx[d];
    }

    bool descend = false;
    if (leaf) {
      nd.flag = 1;
      nd.axis = 0;
      nd.data[0] = n;
      nd.data[1] = L + lo;
      if (lane == 0) out[me] = nd;
      if (have) indices[L + i_first] = r0.prim;
      for (uint32_t i = i_first + 64u; i < hi; i += 64u) indices[L + i] = NRT_SUB_REC(s_perm[pb][i]).prim;
      leaves++;
      biggest_leaf = n > biggest_leaf ? n : biggest_leaf;
    } else {
      // ---- LDS bin reduction ------------------------------------------------------------------
      T sc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) sc[k] = bin_scale<T>(cmn[k], cmx[k], K);
      auto bin_one = [&](const PrimRec<T> &r) {
        U emin[3], emax[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          emin[d] = Ord<T>::enc(r.bmin[d]);
          emax[d] = Ord<T>::enc(r.bmax[d]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int b = bin_of<T>(r.c[k], cmn[k], sc[k], K);
          atomicAdd(&s_cnt[k][b], 1u);
#pragma unroll
          for (int d = 0; d < 3; d++) {
            atomicMin(&s_bmin[k][b][d], emin[d]);
            atomicMax(&s_bmax[k][b][d], emax[d]);
          }
        }
      };
      if (have) bin_one(r0);
      for (uint32_t i = i_first + 64u; i < hi; i += 64u) bin_one(NRT_SUB_REC(s_perm[pb][i]));
      __syncthreads();

      // ---- lane == (axis, bin): sweeps inside 16-lane groups, on the integer images ----------------------
      const int ax = (int)lane >> 4, bn = (int)lane & 15;
      uint32_t cnt = 0;
      U pmn[3], pmx[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        pmn[d] = Ord<T>::highest();
        pmx[d] = Ord<T>::lowest();
      }
      if (ax < 3 && bn < K) {
        cnt = s_cnt[ax][bn];
        if (cnt) {
#pragma unroll
          for (int d = 0; d < 3; d++) {
            pmn[d] = s_bmin[ax][bn][d];
            pmx[d] = s_bmax[ax][bn][d];
          }
          s_cnt[ax][bn] = 0; // read: hand the bin on clean (made visible by the barrier after the partition)
#pragma unroll
          for (int d = 0; d < 3; d++) {
            s_bmin[ax][bn][d] = Ord<T>::highest();
            s_bmax[ax][bn][d] = Ord<T>::lowest();
          }
        }
      }
      uint32_t pc = cnt, sc_n = cnt; // inclusive prefix / suffix inside the 16-lane row (DPP row shifts)
      U smn[3] = {pmn[0], pmn[1], pmn[2]}, smx[3] = {pmx[0], pmx[1], pmx[2]};
      row_prefix_e<T>(pc, pmn, pmx);
      row_suffix_e<T>(sc_n, smn, smx);
      // candidate (ax, s = bn), s in 1..K-1: low side = bins [0, s), high side = bins [s, K)
      const uint32_t nl = dpp_mov<0x111>(0u, pc);
      U lmn[3], lmx[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        lmn[d] = dpp_mov<0x111>(Ord<T>::highest(), pmn[d]);
        lmx[d] = dpp_mov<0x111>(Ord<T>::lowest(), pmx[d]);
      }
      T cost = Lim<T>::inf();
      if (ax < 3 && bn >= 1 && bn < K && nl > 0 && sc_n > 0) {
        T a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          a0[d] = Ord<T>::dec(lmn[d]);
          a1[d] = Ord<T>::dec(lmx[d]);
          b0[d] = Ord<T>::dec(smn[d]);
          b1[d] = Ord<T>::dec(smx[d]);
        }
        cost = T(nl) * half_area<T>(a0, a1) + T(sc_n) * half_area<T>(b0, b1);
      }
      if (!(cost == cost)) cost = Lim<T>::inf(); // a NaN cost never wins
      // argmin: the smallest cost, ties -> lowest lane; lane order == (axis, bin): lowest axis, then lowest bin
      const U ecost = Ord<T>::enc(cost);
      const U ebest = wave_umin<U>(ecost);
      const int who = (int)__builtin_ctzll(__ballot(ecost == ebest));
      const bool found = ebest < Ord<T>::enc(Lim<T>::inf());
      int axis = 0;
      uint32_t split_bin = kMedian, nleft = n >> 1;
      T cl[3], ch[3], rl[3], rh[3]; // children AABBs
      // a pathological chain of lopsided SAH splits could outgrow the LDS stack: past kSubStackSafe
      // pending nodes fall back to balanced object-median splits (at most log2(kSmall) more levels)
      if (found && sp < kSubStackSafe) {
        axis = who >> 4;
        split_bin = (uint32_t)who & 15u;
        nleft = lane_bcast(nl, who);
#pragma unroll
        for (int d = 0; d < 3; d++) {
          cl[d] = Ord<T>::dec(lane_bcast(lmn[d], who));
          ch[d] = Ord<T>::dec(lane_bcast(lmx[d], who));
          rl[d] = Ord<T>::dec(lane_bcast(smn[d], who));
          rh[d] = Ord<T>::dec(lane_bcast(smx[d], who));
        }
      } else {
#pragma unroll
        for (int d = 0; d < 3; d++) {
          cl[d] = rl[d] = Lim<T>::max();
          ch[d] = rh[d] = -Lim<T>::max();
        }
      }
      const bool median = split_bin == kMedian;

      // ---- stable partition of s_perm[pb][lo, hi) into s_perm[1 - pb], reducing the children's centroid bounds
      //      (and, after a median split, their AABBs) on the way ----------------------------------------------
      const bool low_leaf = nleft <= leaf_max || depth + 1 >= max_depth, high_leaf = n - nleft <= leaf_max || depth + 1 >= max_depth;
      const bool both_leaves = low_leaf && high_leaf; // the common case at the bottom: finished here, no trip through the stack
      T ccl[3], cch[3], crl[3], crh[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        ccl[d] = crl[d] = Lim<T>::max();
        cch[d] = crh[d] = -Lim<T>::max();
      }
      {
        const T clo = axis == 0 ? cmn[0] : (axis == 1 ? cmn[1] : cmn[2]);
        const T scl = axis == 0 ? sc[0] : (axis == 1 ? sc[1] : sc[2]);
        uint32_t run_l = 0, run_r = 0;
        for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {
          const uint32_t i = i0 + lane;
          const bool valid = i < hi;
          uint16_t id = id0;
          PrimRec<T> r = r0;
          if (valid && i0 != lo) {
            id = s_perm[pb][i];
            r = NRT_SUB_REC(id);
          }
          bool left = false;
          if (valid) {
            if (median) {
              left = (i - lo) < nleft;
            } else {
              const T c = axis == 0 ? r.c[0] : (axis == 1 ? r.c[1] : r.c[2]);
              left = (uint32_t)bin_of<T>(c, clo, scl, K) < split_bin;
            }
          }
          const unsigned long long bl = __ballot(valid && left), br = __ballot(valid && !left);
          const unsigned long long lt = (1ull << lane) - 1ull;
          if (valid) {
            const uint32_t d = left ? lo + run_l + (uint32_t)__builtin_popcountll(bl & lt)
                                    : lo + nleft + run_r + (uint32_t)__builtin_popcountll(br & lt);
            s_perm[1 - pb][d] = id;
            if (both_leaves || (low_leaf && left)) indices[L + d] = r.prim; // index slots of the leaves emitted below, in partition order
#pragma unroll
            for (int k = 0; k < 3; k++) {
              if (left) {
                ccl[k] = tmin(ccl[k], r.c[k]);
                cch[k] = tmax(cch[k], r.c[k]);
              } else {
                crl[k] = tmin(crl[k], r.c[k]);
                crh[k] = tmax(crh[k], r.c[k]);
              }
              if (median) {
                if (left) {
                  cl[k] = tmin(cl[k], r.bmin[k]);
                  ch[k] = tmax(ch[k], r.bmax[k]);
                } else {
                  rl[k] = tmin(rl[k], r.bmin[k]);
                  rh[k] = tmax(rh[k], r.bmax[k]);
                }
              }
            }
          }
          run_l += (uint32_t)__builtin_popcountll(bl);
          run_r += (uint32_t)__builtin_popcountll(br);
        }
      }
      // (a child that becomes a leaf needs no centroid bounds)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        if (!low_leaf) {
          ccl[d] = wave_min_u<T>(ccl[d]);
          cch[d] = wave_max_u<T>(cch[d]);
        }
        if (!high_leaf) {
          crl[d] = wave_min_u<T>(crl[d]);
          crh[d] = wave_max_u<T>(crh[d]);
        }
        if (median) {
          cl[d] = wave_min_u<T>(cl[d]);
          ch[d] = wave_max_u<T>(ch[d]);
          rl[d] = wave_min_u<T>(rl[d]);
          rh[d] = wave_max_u<T>(rh[d]);
        }
      }

      nd.flag = 0;
      nd.axis = axis;
      nd.data[0] = me + 1; // low-side child follows its parent (pre-order)
      nd.data[1] = 0;      // patched when the high-side child is emitted
      if (both_leaves) {
        // both children are leaves: emit the three nodes now (pre-order: parent, low leaf, high leaf)
        nd.data[1] = me + 2;
        if (lane == 0) {
          out[me] = nd;
          Node lf;
          lf.flag = 1;
          lf.axis = 0;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            lf.bmin[d] = cl[d];
            lf.bmax[d] = ch[d];
          }
          lf.data[0] = nleft;
          lf.data[1] = L + lo;
          out[me + 1] = lf;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            lf.bmin[d] = rl[d];
            lf.bmax[d] = rh[d];
          }
          lf.data[0] = n - nleft;
          lf.data[1] = L + lo + nleft;
          out[me + 2] = lf;
        }
        node_count += 2;
        leaves += 2;
        deepest = depth + 1 > deepest ? depth + 1 : deepest;
        const uint32_t big = nleft > n - nleft ? nleft : n - nleft;
        biggest_leaf = big > biggest_leaf ? big : biggest_leaf;
        __syncthreads(); // the reset bins are visible to the next node
      } else if (low_leaf) {
        // the low child is a leaf, the high one is not: emit the leaf (node me + 1) and continue with the high child
        // right away (it is node me + 2; the loop head patches the parent's data[1]) — no stack entry
        if (lane == 0) {
          out[me] = nd;
          Node lf;
          lf.flag = 1;
          lf.axis = 0;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            lf.bmin[d] = cl[d];
            lf.bmax[d] = ch[d];
          }
          lf.data[0] = nleft;
          lf.data[1] = L + lo;
          out[me + 1] = lf;
        }
        node_count += 1;
        leaves += 1;
        deepest = depth + 1 > deepest ? depth + 1 : deepest;
        biggest_leaf = nleft > biggest_leaf ? nleft : biggest_leaf;
        lo = lo + nleft;
        depth = depth + 1;
        parent = me;
        is_high = true;
        pb = 1u - pb;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          mn[d] = rl[d];
          mx[d] = rh[d];
          cmn[d] = crl[d];
          cmx[d] = crh[d];
        }
        descend = true;
        __syncthreads(); // the permutation and the reset bins are visible to the next node
      } else {
      if (lane == 0) {
        out[me] = nd;
        SubPending<T> &e = s_stack[sp];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          e.bmin[d] = rl[d];
          e.bmax[d] = rh[d];
          e.cmin[d] = crl[d];
          e.cmax[d] = crh[d];
        }
        e.lo = (uint16_t)(lo + nleft);
        e.hi = (uint16_t)hi;
        e.parent = (uint16_t)me;
        e.buf = (uint16_t)(1u - pb);
        e.depth = depth + 1;
      }
      sp++;
      // continue with the low side
      hi = lo + nleft;
      depth = depth + 1;
      parent = me;
      is_high = false;
      pb = 1u - pb;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        mn[d] = cl[d];
        mx[d] = ch[d];
        cmn[d] = ccl[d];
        cmx[d] = cch[d];
      }
      descend = true;
      __syncthreads(); // the permutation, the reset bins and the stack entry are visible to the next node
      }
    }
    if (!descend) {
      if (sp == 0) break;
      sp--;
      const SubPending<T> &e = s_stack[sp];
      lo = e.lo;
      hi = e.hi;
      depth = e.depth;
      parent = e.parent;
      pb = e.buf;
      is_high = true;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        mn[d] = e.bmin[d];
        mx[d] = e.bmax[d];
        cmn[d] = e.cmin[d];
        cmx[d] = e.cmax[d];
      }
    }
  }
  if (lane == 0) { // (statistics: left in the task's record, summed by k_layout — see task_stats)
    task.size = node_count;
    task_stats<T>(task, leaves, deepest, biggest_leaf);
  }
}
#undef NRT_SUB_REC
#endif // NRT_PROF

// ---------------------------------------------------------------------------
// subtree phase, row form: up to four nodes of a subtree per step, one per 16-lane row
// ---------------------------------------------------------------------------
// k_subtree above spends about 700 wave instructions on an inner node whatever its size, and three quarters of a
// subtree's inner nodes hold 16 primitives or fewer (5 to 16 of 64 lanes busy).  This form keeps the nodes that wait to be
// split on an LDS stack and takes up to FOUR of them per step, one per 16-lane DPP row: lane == primitive for the binning
// and the partition (16 at a time), lane == bin for the cut search (the three axes one after the other, the 16-lane
// prefix / suffix scans are the ones k_subtree uses), the winner's data is handed to its row by ds_bpermute.  With one or
// two nodes on the stack (the first steps of a subtree, where the nodes are large) a node gets 64 or 32 lanes instead.
// Every decision is the one k_subtree takes — same bins, cost expression, tie rule (lowest axis, then lowest bin), leaf
// rule and object-median fallback (including k_subtree's stack-depth guard, whose depth every node carries along) — and
// min / max / counts do not depend on the order they are combined in, so the tree is the same.  Nodes are created in
// step order, not in pre-order: they are written to the scratch array under their creation index (children c, c + 1
// with c odd) and the wave finishes by computing each node's pre-order index from the parent links (subtree sizes by
// walking up, then the index as the sum over the path to the root), which k_emit_small applies when it splices the
// subtree into the tree (premap).
struct RowEntry {
  uint16_t lo, hi; // range in the permutation
  uint16_t me;     // creation index of the node
  uint8_t ldepth;  // depth below the subtree's root
  uint8_t misc;    // bit 7: permutation buffer holding [lo, hi); bits 0..5: k_subtree's stack depth at this node
};
static_assert(sizeof(RowEntry) == 8, "RowEntry");
constexpr int kRowStack = kHandoff / 2; // pending nodes are disjoint ranges of at least 2 primitives

template <typename U>
__device__ __forceinline__ U row_allmin(U x) { // every lane of a 16-lane row gets the row's minimum
  x = umin_(x, dpp_mov<0xB1>((U) ~(U)0, x));
  x = umin_(x, dpp_mov<0x4E>((U) ~(U)0, x));
  x = umin_(x, dpp_mov<0x141>((U) ~(U)0, x));
  x = umin_(x, dpp_mov<0x140>((U) ~(U)0, x));
  return x;
}
template <typename U>
__device__ __forceinline__ U row_allmax(U x) {
  x = umax_(x, dpp_mov<0xB1>((U)0, x));
  x = umax_(x, dpp_mov<0x4E>((U)0, x));
  x = umax_(x, dpp_mov<0x141>((U)0, x));
  x = umax_(x, dpp_mov<0x140>((U)0, x));
  return x;
}
// (groups of 16, 32 or 64 lanes: shift = 4, 5, 6 — wave-uniform)
template <typename U>
__device__ __forceinline__ U group_allmin(U x, uint32_t shift) {
  x = row_allmin<U>(x);
  if (shift >= 5u) x = umin_(x, (U)__shfl_xor(x, 16));
  if (shift >= 6u) x = umin_(x, (U)__shfl_xor(x, 32));
  return x;
}
template <typename U>
__device__ __forceinline__ U group_allmax(U x, uint32_t shift) {
  x = row_allmax<U>(x);
  if (shift >= 5u) x = umax_(x, (U)__shfl_xor(x, 16));
  if (shift >= 6u) x = umax_(x, (U)__shfl_xor(x, 32));
  return x;
}

template <typename T>
__global__ __launch_bounds__(64) void k_subtree_rows(TopNode<T> *top, const uint32_t *__restrict__ small_list,
                                                     const PrimRec<T> *__restrict__ recs0,
                                                     const PrimRec<T> *__restrict__ recs1, int K, uint32_t min_leaf,
                                                     uint32_t max_depth, typename Wire<T>::Node *scratch_nodes,
                                                     uint16_t *premap, uint32_t *indices, LevelInfo *info) {
  typedef typename Wire<T>::Node Node;
  typedef typename Ord<T>::U U;
  __shared__ uint16_t s_perm[2][kHandoff];
  __shared__ RowEntry s_stack[kRowStack];
  __shared__ uint16_t s_parent[2 * kHandoff];
  __shared__ uint32_t s_cnt[4][3][kSmallBins];
  __shared__ U s_bmin[4][3][kSmallBins][3]; // (after the last step: the nodes' subtree sizes, 2 * kHandoff uint32)
  __shared__ U s_bmax[4][3][kSmallBins][3];
  static_assert(sizeof(U) * 4 * 3 * kSmallBins * 3 >= sizeof(uint32_t) * 2 * kHandoff, "sizes fit the bins");

  const unsigned lane = threadIdx.x;
  if (blockIdx.x >= info->num_small) return; // grid is an upper bound
  TopNode<T> &task = top[small_list[blockIdx.x]];
  const uint32_t L = task.l, n_all = task.r - task.l;
  const PrimRec<T> *src = (task.buf ? recs1 : recs0) + L;
  Node *out = scratch_nodes + 2 * (size_t)L;
  uint16_t *map = premap + 2 * (size_t)L;
  const uint32_t leaf_max = min_leaf > 1u ? min_leaf : 1u;
  const uint32_t depth0 = task.depth;

  if (n_all <= leaf_max || depth0 >= max_depth) { // the task is a leaf (nanort.h:1781-1783)
    for (uint32_t i = lane; i < n_all; i += 64u) indices[L + i] = src[i].prim;
    if (lane == 0) {
      Node nd;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        nd.bmin[d] = task.bmin[d];
        nd.bmax[d] = task.bmax[d];
      }
      nd.flag = 1;
      nd.axis = 0;
      nd.data[0] = n_all;
      nd.data[1] = L;
      out[0] = nd;
      map[0] = 0;
      task.size = 1;
      task_stats<T>(task, 1u, depth0, n_all);
    }
    return;
  }

  for (uint32_t i = lane; i < n_all; i += 64u) s_perm[0][i] = (uint16_t)i;
  for (uint32_t q = lane; q < 4u * 3u * kSmallBins; q += 64u) { // bins start clean and are handed on clean by their readers
    (&s_cnt[0][0][0])[q] = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      (&s_bmin[0][0][0][0])[3 * q + d] = Ord<T>::highest();
      (&s_bmax[0][0][0][0])[3 * q + d] = Ord<T>::lowest();
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
      out[0].bmin[d] = task.bmin[d];
      out[0].bmax[d] = task.bmax[d];
    }
    RowEntry e;
    e.lo = 0;
    e.hi = (uint16_t)n_all;
    e.me = 0;
    e.ldepth = 0;
    e.misc = 0;
    s_stack[0] = e;
    s_parent[0] = 0;
  }
  uint32_t stack_n = 1, node_count = 1;                // wave-uniform
  uint32_t leaves = 0, deepest = 0, biggest_leaf = 0;  // kept by the group leaders, combined at the end
  __syncthreads();

  for (uint32_t step = 0; stack_n > 0; step++) {
    if (step > 4u * kHandoff) { // cannot happen (every step splits at least one node, a subtree has fewer than kHandoff inner nodes)
      if (lane == 0) info->error = 1;
      break;
    }
    const uint32_t m = stack_n < 4u ? stack_n : 4u;
    const uint32_t shift = m == 1u ? 6u : (m == 2u ? 5u : 4u); // lanes per node: 64, 32 or 16
    const uint32_t G = 1u << shift, g = lane >> shift, lg = lane & (G - 1u), gbase = g << shift;
    const bool act = g < m;
    RowEntry e = s_stack[act ? stack_n - 1u - g : 0u];
    stack_n -= m;
    const uint32_t lo = act ? e.lo : 0u, hi = act ? e.hi : 0u, n = hi - lo, me = e.me;
    const uint32_t ldepth = e.ldepth, depth = depth0 + ldepth, pb = e.misc >> 7, vsp = e.misc & 63u;
    // passes over the node: G primitives at a time; the wave runs the longest group's count
    const uint32_t my_pass = (n + G - 1u) >> shift;
    uint32_t npass = (uint32_t)__builtin_amdgcn_readlane((int)my_pass, 0);
    {
      const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)my_pass, 16), p2 = (uint32_t)__builtin_amdgcn_readlane((int)my_pass, 32),
                     p3 = (uint32_t)__builtin_amdgcn_readlane((int)my_pass, 48);
      npass = npass > p1 ? npass : p1;
      npass = npass > p2 ? npass : p2;
      npass = npass > p3 ? npass : p3;
      npass = npass < (uint32_t)(kHandoff >> 4) ? npass : (uint32_t)(kHandoff >> 4); // (a range never exceeds the subtree)
    }

    // ---- this lane's first element stays in registers for every pass; the node's centroid bounds --------------------
    const uint32_t i0 = lo + lg;
    const bool have0 = i0 < hi;
    uint32_t id0 = 0;
    PrimRec<T> r0;
    if (have0) {
      id0 = s_perm[pb][i0];
      r0 = src[id0];
    }
    T cmn[3], cmx[3];
    if (step == 0) { // the subtree's root: reduced by the top phase
#pragma unroll
      for (int k = 0; k < 3; k++) {
        cmn[k] = task.cmin[k];
        cmx[k] = task.cmax[k];
      }
    } else {
      U emn[3], emx[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        emn[k] = have0 ? Ord<T>::enc(r0.c[k]) : Ord<T>::highest();
        emx[k] = have0 ? Ord<T>::enc(r0.c[k]) : Ord<T>::lowest();
      }
      for (uint32_t pass = 1; pass < npass; pass++) {
        const uint32_t i = i0 + (pass << shift);
        if (i < hi) {
          const PrimRec<T> &r = src[s_perm[pb][i]];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const U ec = Ord<T>::enc(r.c[k]);
            emn[k] = umin_(emn[k], ec);
            emx[k] = umax_(emx[k], ec);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        cmn[k] = Ord<T>::dec(group_allmin<U>(emn[k], shift));
        cmx[k] = Ord<T>::dec(group_allmax<U>(emx[k], shift));
      }
    }

    // ---- LDS bin reduction into the group's bins ---------------------------------------------------------------------
    T sc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) sc[k] = bin_scale<T>(cmn[k], cmx[k], K);
    for (uint32_t pass = 0; pass < npass; pass++) {
      const uint32_t i = i0 + (pass << shift);
      if (i < hi) {
        PrimRec<T> r = r0;
        if (pass) r = src[s_perm[pb][i]];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int b = bin_of<T>(r.c[k], cmn[k], sc[k], K);
          atomicAdd(&s_cnt[g][k][b], 1u);
#pragma unroll
          for (int d = 0; d < 3; d++) {
            atomicMin(&s_bmin[g][k][b][d], Ord<T>::enc(r.bmin[d]));
            atomicMax(&s_bmax[g][k][b][d], Ord<T>::enc(r.bmax[d]));
          }
        }
      }
    }
    __syncthreads();

    // ---- cut search: the first 16 lanes of the group, lane == bin, one axis after the other ---------------------------
    T best_cost = Lim<T>::inf();
    int axis = 0;
    uint32_t split_bin = kMedian, nleft = n >> 1;
    T cl[3], ch[3], rl[3], rh[3]; // children AABBs
    U ecl[3], ech[3], erl[3], erh[3]; // (their integer images while the axes compete)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      ecl[d] = erl[d] = Ord<T>::enc(Lim<T>::max());
      ech[d] = erh[d] = Ord<T>::enc(-Lim<T>::max());
    }
    const bool bin_lane = act && lg < (uint32_t)K && lg < 16u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      uint32_t cnt = 0;
      U pmn[3], pmx[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        pmn[d] = Ord<T>::highest();
        pmx[d] = Ord<T>::lowest();
      }
      if (bin_lane) {
        cnt = s_cnt[g][k][lg];
        if (cnt) {
#pragma unroll
          for (int d = 0; d < 3; d++) {
            pmn[d] = s_bmin[g][k][lg][d];
            pmx[d] = s_bmax[g][k][lg][d];
          }
          s_cnt[g][k][lg] = 0; // read: hand the bin on clean (made visible by the barrier that ends the step)
#pragma unroll
          for (int d = 0; d < 3; d++) {
            s_bmin[g][k][lg][d] = Ord<T>::highest();
            s_bmax[g][k][lg][d] = Ord<T>::lowest();
          }
        }
      }
      uint32_t pc = cnt, sc_n = cnt; // inclusive prefix / suffix inside the 16-lane row (DPP row shifts)
      U smn[3] = {pmn[0], pmn[1], pmn[2]}, smx[3] = {pmx[0], pmx[1], pmx[2]};
      row_prefix_e<T>(pc, pmn, pmx);
      row_suffix_e<T>(sc_n, smn, smx);
      // candidate s = bin, s in 1..K-1: low side = bins [0, s), high side = bins [s, K)
      const uint32_t nl = dpp_mov<0x111>(0u, pc);
      U lmn[3], lmx[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        lmn[d] = dpp_mov<0x111>(Ord<T>::highest(), pmn[d]);
        lmx[d] = dpp_mov<0x111>(Ord<T>::lowest(), pmx[d]);
      }
      T cost = Lim<T>::inf();
      if (bin_lane && lg >= 1u && nl > 0 && sc_n > 0) {
        T a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          a0[d] = Ord<T>::dec(lmn[d]);
          a1[d] = Ord<T>::dec(lmx[d]);
          b0[d] = Ord<T>::dec(smn[d]);
          b1[d] = Ord<T>::dec(smx[d]);
        }
        cost = T(nl) * half_area<T>(a0, a1) + T(sc_n) * half_area<T>(b0, b1);
      }
      if (!(cost == cost)) cost = Lim<T>::inf(); // a NaN cost never wins
      const U ecost = Ord<T>::enc(cost);
      const U rbest = row_allmin<U>(ecost);
      const unsigned long long hit = __ballot(ecost == rbest);
      // the group's first row holds its candidates: the row's best, ties -> lowest bin
      const U gbest = (U)__shfl(rbest, (int)gbase);
      const T c = Ord<T>::dec(gbest);
      const bool better = c < best_cost; // ties -> lowest axis
      if (__ballot(better) != 0ull) {    // (uniform: the winner's data travels only when some group wants it)
        const uint32_t who = (uint32_t)__builtin_ctz(((uint32_t)(hit >> gbase) & 0xFFFFu) | 0x10000u);
        const int from = (int)(gbase + (who & 15u));
        const uint32_t w_nl = (uint32_t)__shfl(nl, from);
        U w_lmn[3], w_lmx[3], w_smn[3], w_smx[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          w_lmn[d] = (U)__shfl(lmn[d], from);
          w_lmx[d] = (U)__shfl(lmx[d], from);
          w_smn[d] = (U)__shfl(smn[d], from);
          w_smx[d] = (U)__shfl(smx[d], from);
        }
        if (better) {
          best_cost = c;
          axis = k;
          split_bin = who;
          nleft = w_nl;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            ecl[d] = w_lmn[d];
            ech[d] = w_lmx[d];
            erl[d] = w_smn[d];
            erh[d] = w_smx[d];
          }
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
      cl[d] = Ord<T>::dec(ecl[d]);
      ch[d] = Ord<T>::dec(ech[d]);
      rl[d] = Ord<T>::dec(erl[d]);
      rh[d] = Ord<T>::dec(erh[d]);
    }
    // a pathological chain of lopsided SAH splits is cut off as in k_subtree: past kSubStackSafe pending nodes (there),
    // balanced object-median splits
    if (!(best_cost < Lim<T>::inf()) || vsp >= (uint32_t)kSubStackSafe) {
      axis = 0;
      split_bin = kMedian;
      nleft = n >> 1;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        cl[d] = rl[d] = Lim<T>::max();
        ch[d] = rh[d] = -Lim<T>::max();
      }
    }
    const bool median = split_bin == kMedian;
    const bool low_leaf = nleft <= leaf_max || depth + 1u >= max_depth, high_leaf = n - nleft <= leaf_max || depth + 1u >= max_depth;

    // ---- stable partition of s_perm[pb][lo, hi) into s_perm[1 - pb] ---------------------------------------------------
    {
      const T clo = axis == 0 ? cmn[0] : (axis == 1 ? cmn[1] : cmn[2]);
      const T scl = axis == 0 ? sc[0] : (axis == 1 ? sc[1] : sc[2]);
      const unsigned long long gmask = (G == 64u ? ~0ull : ((1ull << G) - 1ull)), lt = (1ull << lg) - 1ull;
      const bool any_median = __ballot(act && median) != 0ull;
      uint32_t run_l = 0, run_r = 0;
      for (uint32_t pass = 0; pass < npass; pass++) {
        const uint32_t i = i0 + (pass << shift);
        const bool valid = i < hi;
        uint32_t id = id0;
        PrimRec<T> r = r0;
        if (valid && pass) {
          id = s_perm[pb][i];
          r = src[id];
        }
        bool left = false;
        if (valid) {
          if (median) {
            left = (i - lo) < nleft;
          } else {
            const T c = axis == 0 ? r.c[0] : (axis == 1 ? r.c[1] : r.c[2]);
            left = (uint32_t)bin_of<T>(c, clo, scl, K) < split_bin;
          }
        }
        const unsigned long long bl = (__ballot(valid && left) >> gbase) & gmask, br = (__ballot(valid && !left) >> gbase) & gmask;
        if (valid) {
          const uint32_t d = left ? lo + run_l + (uint32_t)__builtin_popcountll(bl & lt)
                                  : lo + nleft + run_r + (uint32_t)__builtin_popcountll(br & lt);
          s_perm[1u - pb][d] = (uint16_t)id;
          if (left ? low_leaf : high_leaf) indices[L + d] = r.prim; // index slots of the leaves emitted below, in partition order
          if (median) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
              if (left) {
                cl[k] = tmin(cl[k], r.bmin[k]);
                ch[k] = tmax(ch[k], r.bmax[k]);
              } else {
                rl[k] = tmin(rl[k], r.bmin[k]);
                rh[k] = tmax(rh[k], r.bmax[k]);
              }
            }
          }
        }
        run_l += (uint32_t)__builtin_popcountll(bl);
        run_r += (uint32_t)__builtin_popcountll(br);
      }
      if (any_median) { // (uniform: the reductions run for every group, only the median ones keep the result)
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const T a = Ord<T>::dec(group_allmin<U>(Ord<T>::enc(cl[d]), shift)), b = Ord<T>::dec(group_allmax<U>(Ord<T>::enc(ch[d]), shift));
          const T c = Ord<T>::dec(group_allmin<U>(Ord<T>::enc(rl[d]), shift)), e2 = Ord<T>::dec(group_allmax<U>(Ord<T>::enc(rh[d]), shift));
          if (median) {
            cl[d] = a;
            ch[d] = b;
            rl[d] = c;
            rh[d] = e2;
          }
        }
      }
    }

    // ---- the group's first lane writes the node and its children -----------------------------------------------------
    {
      const bool lead = act && lg == 0u;
      const unsigned long long push_l = __ballot(lead && !low_leaf), push_h = __ballot(lead && !high_leaf);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (lead) {
        const uint32_t c0 = node_count + 2u * g;
        out[me].flag = 0;
        out[me].axis = axis;
        out[me].data[0] = c0;
        out[me].data[1] = c0 + 1u;
        s_parent[c0] = (uint16_t)me;
        s_parent[c0 + 1u] = (uint16_t)me;
        uint32_t slot = stack_n + (uint32_t)__builtin_popcountll(push_l & below) + (uint32_t)__builtin_popcountll(push_h & below);
        Node lf;
        lf.flag = 1;
        lf.axis = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          lf.bmin[d] = cl[d];
          lf.bmax[d] = ch[d];
        }
        lf.data[0] = nleft;
        lf.data[1] = L + lo;
        if (low_leaf) {
          out[c0] = lf;
        } else {
#pragma unroll
          for (int d = 0; d < 3; d++) {
            out[c0].bmin[d] = cl[d];
            out[c0].bmax[d] = ch[d];
          }
          RowEntry ne;
          ne.lo = (uint16_t)lo;
          ne.hi = (uint16_t)(lo + nleft);
          ne.me = (uint16_t)c0;
          ne.ldepth = (uint8_t)(ldepth + 1u);
          ne.misc = (uint8_t)(((1u - pb) << 7) | (vsp + 1u)); // k_subtree descends into the low side with the high side pending
          s_stack[slot++] = ne;
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
          lf.bmin[d] = rl[d];
          lf.bmax[d] = rh[d];
        }
        lf.data[0] = n - nleft;
        lf.data[1] = L + lo + nleft;
        if (high_leaf) {
          out[c0 + 1u] = lf;
        } else {
#pragma unroll
          for (int d = 0; d < 3; d++) {
            out[c0 + 1u].bmin[d] = rl[d];
            out[c0 + 1u].bmax[d] = rh[d];
          }
          RowEntry ne;
          ne.lo = (uint16_t)(lo + nleft);
          ne.hi = (uint16_t)hi;
          ne.me = (uint16_t)(c0 + 1u);
          ne.ldepth = (uint8_t)(ldepth + 1u);
          ne.misc = (uint8_t)(((1u - pb) << 7) | vsp);
          s_stack[slot] = ne;
        }
        if (low_leaf || high_leaf) {
          leaves += (low_leaf ? 1u : 0u) + (high_leaf ? 1u : 0u);
          deepest = depth + 1u > deepest ? depth + 1u : deepest;
          const uint32_t big = (low_leaf ? nleft : 0u) > (high_leaf ? n - nleft : 0u) ? (low_leaf ? nleft : 0u) : (high_leaf ? n - nleft : 0u);
          biggest_leaf = big > biggest_leaf ? big : biggest_leaf;
        }
      }
      stack_n += (uint32_t)__builtin_popcountll(push_l) + (uint32_t)__builtin_popcountll(push_h);
      node_count += 2u * m;
    }
    __syncthreads(); // the permutation, the reset bins, the stack and the parent links are visible to the next step
  }

  // ---- pre-order index of every node from the parent links ------------------------------------------------------------
  uint32_t *s_size = reinterpret_cast<uint32_t *>(&s_bmin[0][0][0][0]);
  const uint32_t N = node_count;
  for (uint32_t x = lane; x < N; x += 64u) s_size[x] = 1u;
  __syncthreads();
  for (uint32_t x = lane; x < N; x += 64u) {
    if (x == 0u) continue;
    uint32_t p = s_parent[x];
    for (uint32_t it = 0; it < 2u * kHandoff; it++) { // every ancestor counts this node
      atomicAdd(&s_size[p], 1u);
      if (p == 0u) break;
      p = s_parent[p];
    }
  }
  __syncthreads();
  for (uint32_t x = lane; x < N; x += 64u) {
    // pre-order: a low-side child (odd creation index) follows its parent, a high-side child follows the low side's subtree
    uint32_t acc = 0, y = x;
    for (uint32_t it = 0; it < 2u * kHandoff && y != 0u; it++) {
      acc += 1u + ((y & 1u) ? 0u : s_size[y - 1u]);
      y = s_parent[y];
    }
    map[x] = (uint16_t)acc;
  }
  // stats: the leaders' partial values
  for (int off = 32; off > 0; off >>= 1) {
    leaves += __shfl_xor(leaves, off);
    const uint32_t dd = __shfl_xor(deepest, off), bb = __shfl_xor(biggest_leaf, off);
    deepest = dd > deepest ? dd : deepest;
    biggest_leaf = bb > biggest_leaf ? bb : biggest_leaf;
  }
  if (lane == 0) {
    task.size = N;
    task_stats<T>(task, leaves, deepest, biggest_leaf);
  }
}

// ---------------------------------------------------------------------------
// relayout: sizes bottom-up, DFS pre-order top-down (single block over the
// small top array), then emission / splice
// ---------------------------------------------------------------------------
// Sizes and final (DFS pre-order) indices of the top nodes, from PARENT links — no level structure needed, so top nodes may
// have been created in any order (the level-synchronous phase and the block-resident mid phase both append to the array):
//   size(x) = sum of the contributions of x and its descendants (a subtree task contributes its node count, any other
//             top node 1): every node adds its contribution to each of its ancestors;
//   dfs(x)  = sum over the path root -> x of (1, plus the size of the low-side sibling where the path takes a high
//             side): every node walks up its parent chain once more.
// O(nodes x depth) operations and two barriers instead of two barriers per level.  Small top arrays (the usual case) keep
// sizes and parents in LDS: one block does everything (k_layout).  Larger ones run the same three steps as grid-wide
// kernels on the global arrays (k_layout_init / _sizes / _dfs).
#ifndef NRT_LAYOUT_LDS
#define NRT_LAYOUT_LDS 16384
#endif
constexpr uint32_t kLayoutLds = NRT_LAYOUT_LDS; // top arrays up to this many nodes are laid out from LDS (2 x 4 bytes per node, dynamic)
constexpr uint32_t kLayoutOwn = kLayoutLds / 1024;
template <typename T>
__device__ __forceinline__ uint32_t layout_contribution(const TopNode<T> &t) { return t.kind == KIND_SMALL ? t.size : 1u; }

// (the walk-up stops below depth kLayoutCut: the few nodes above it — on which every node's additions would pile up —
// get their sizes level by level from their children afterwards)
constexpr uint32_t kLayoutCut = 8;
template <typename T>
__global__ __launch_bounds__(1024) void k_layout(TopNode<T> *top, LevelInfo *info) {
  extern __shared__ uint32_t s_dyn[];
  __shared__ uint32_t s_upper[1u << kLayoutCut]; // global path: the nodes above the cut
  __shared__ uint32_t s_nupper;
  const uint32_t total = info->top_count;
  uint32_t *s_size = s_dyn, *s_par = s_dyn + kLayoutLds;
  uint32_t leaves = 0, branches = 0, deepest = 0, biggest = 0; // statistics of the top part + what the subtree tasks left (task_stats)
  auto count_node = [&](uint32_t kind, uint32_t depth, uint32_t prims, uint32_t size, uint32_t nleft, uint32_t split_bin, uint32_t nchunks) {
    if (kind == KIND_SPLIT) branches++;
    if (kind == KIND_LEAF) {
      leaves++;
      deepest = depth > deepest ? depth : deepest;
      biggest = prims > biggest ? prims : biggest;
    }
    if (kind == KIND_SMALL) {
      leaves += nleft;
      branches += size - nleft;
      deepest = split_bin > deepest ? split_bin : deepest;
      biggest = nchunks > biggest ? nchunks : biggest;
    }
  };
  if (total <= kLayoutLds) {
    // ONE CU reads every top record here, and a scattered 4-byte load costs its L1 the address work of a whole line: the
    // records' integer words are fetched as 8-byte pairs, once, and the statistics are taken in the same pass (a second sweep
    // over the records was half of this kernel's time)
    uint32_t own[kLayoutOwn], dep[kLayoutOwn]; // contribution and depth of node threadIdx.x + 1024 j
    uint32_t kid[kLayoutOwn];                  // its first child if it is a branch (else kNoParent)
    uint32_t small_mask = 0;                   // bit j: node j of this thread is a subtree task (its size is final)
#pragma unroll
    for (uint32_t j = 0; j < kLayoutOwn; j++) {
      const uint32_t i = threadIdx.x + 1024u * j;
      own[j] = 0;
      dep[j] = 0xFFFFFFFFu;
      kid[j] = kNoParent;
      if (i < total) {
        static_assert(offsetof(TopNode<T>, l) % 8 == 0 && sizeof(TopNode<T>) % 8 == 0 && offsetof(TopNode<T>, parent) == offsetof(TopNode<T>, l) + 52, "TopNode: 14 words from l");
        const uint2 *w = reinterpret_cast<const uint2 *>(&top[i].l);
        const uint2 lr = w[0], dk = w[1], as = w[2], nc = w[3], sd = w[4], np = w[6]; // (l, r) (depth, kind) (axis, split_bin) (nleft, child0) (size, dfs) (nchunks, parent)
        const uint32_t kind = dk.y;
        own[j] = kind == KIND_SMALL ? sd.x : 1u;
        dep[j] = dk.x;
        s_size[i] = own[j];
        s_par[i] = np.y;
        if (kind == KIND_SPLIT) kid[j] = nc.y;
        if (kind == KIND_SMALL) small_mask |= 1u << j;
        count_node(kind, dk.x, lr.y - lr.x, sd.x, nc.x, as.y, np.x);
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kLayoutOwn; j++) {
      const uint32_t i = threadIdx.x + 1024u * j;
      if (i < total && dep[j] > kLayoutCut) { // ancestors at depth dep - 1 ... kLayoutCut
        uint32_t p = s_par[i] & ~kHighChild;
        for (uint32_t d = dep[j] - 1u; d >= kLayoutCut; d--) {
          atomicAdd(&s_size[p], own[j]);
          p = s_par[p] & ~kHighChild;
        }
      }
    }
    __syncthreads();
    for (int d = (int)kLayoutCut - 1; d >= 0; d--) { // the top of the tree, level by level (at most 2^d nodes each)
#pragma unroll
      for (uint32_t j = 0; j < kLayoutOwn; j++) {
        const uint32_t i = threadIdx.x + 1024u * j;
        if (dep[j] == (uint32_t)d && kid[j] != kNoParent) s_size[i] = 1u + s_size[kid[j]] + s_size[kid[j] + 1u]; // (no memory access between the barriers)
      }
      __syncthreads();
    }
#pragma unroll
    for (uint32_t j = 0; j < kLayoutOwn; j++) {
      const uint32_t i = threadIdx.x + 1024u * j;
      if (i < total) {
        uint32_t dfs = 0;
        for (uint32_t x = i, pw = s_par[i]; (pw & ~kHighChild) != kNoParent; x = pw & ~kHighChild, pw = s_par[x])
          dfs += 1u + ((pw & kHighChild) ? s_size[x - 1u] : 0u); // (children are allocated in pairs: the low side is x - 1)
        // (size, dfs) are neighbours: one 8-byte store (a subtree task's size is final: written back as it was read)
        reinterpret_cast<uint2 *>(&top[i].l)[4] = make_uint2(((small_mask >> j) & 1u) ? own[j] : s_size[i], dfs);
      }
    }
    if (threadIdx.x == 0) info->num_nodes = s_size[0];
  } else {
    // global path, after k_layout_init / k_layout_sizes: the nodes above the cut, level by level; then k_layout_dfs
    if (threadIdx.x == 0) s_nupper = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += 1024u)
      if (top[i].depth < kLayoutCut && top[i].kind == KIND_SPLIT) s_upper[atomicAdd(&s_nupper, 1u)] = i;
    __syncthreads();
    for (int d = (int)kLayoutCut - 1; d >= 0; d--) {
      for (uint32_t k = threadIdx.x; k < s_nupper; k += 1024u) {
        TopNode<T> &t = top[s_upper[k]];
        if (t.depth == (uint32_t)d) t.size = 1u + top[t.child0].size + top[t.child0 + 1u].size;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) info->num_nodes = top[0].size;
    for (uint32_t i = threadIdx.x; i < total; i += 1024u) { // (this path's statistics: a sweep of their own)
      const TopNode<T> &t = top[i];
      count_node(t.kind, t.depth, t.r - t.l, t.size, t.nleft, t.split_bin, t.nchunks);
    }
  }
  // (one atomic per wave and counter: a thousand device-scope atomics on one word are ~10 us of a one-block kernel)
  for (int off = 32; off > 0; off >>= 1) {
    leaves += __shfl_down(leaves, off);
    branches += __shfl_down(branches, off);
    const uint32_t od = __shfl_down(deepest, off), ob = __shfl_down(biggest, off);
    deepest = od > deepest ? od : deepest;
    biggest = ob > biggest ? ob : biggest;
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (leaves) atomicAdd(&info->num_leaves, leaves);
    if (branches) atomicAdd(&info->num_branches, branches);
    if (deepest) atomicMax(&info->max_depth, deepest);
    if (biggest) atomicMax(&info->max_leaf_count, biggest);
  }
}

// The same three steps for top arrays too large for LDS: grid-wide, on the global arrays (TopNode::size is the accumulator).
template <typename T>
__global__ __launch_bounds__(256) void k_layout_init(TopNode<T> *top, const LevelInfo *info) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= info->top_count || info->top_count <= kLayoutLds) return;
  if (top[i].kind != KIND_SMALL) top[i].size = 1u;
}
template <typename T>
__global__ __launch_bounds__(256) void k_layout_sizes(TopNode<T> *top, const LevelInfo *info) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= info->top_count || info->top_count <= kLayoutLds) return;
  const uint32_t own = top[i].kind == KIND_SMALL ? top[i].size : 1u; // (a subtree task's size is final; nothing is added to it)
  const uint32_t dep = top[i].depth;
  if (dep <= kLayoutCut) return;
  uint32_t p = top[i].parent & ~kHighChild;
  for (uint32_t d = dep - 1u; d >= kLayoutCut; d--) { // ancestors at depth dep - 1 ... kLayoutCut (see k_layout)
    atomicAdd(&top[p].size, own);
    p = top[p].parent & ~kHighChild;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_layout_dfs(TopNode<T> *top, const LevelInfo *info) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= info->top_count || info->top_count <= kLayoutLds) return;
  uint32_t dfs = 0;
  for (uint32_t x = i, pw = top[i].parent; (pw & ~kHighChild) != kNoParent; x = pw & ~kHighChild, pw = top[x].parent)
    dfs += 1u + ((pw & kHighChild) ? top[x - 1u].size : 0u);
  top[i].dfs = dfs;
}

template <typename T>
__global__ __launch_bounds__(256) void k_emit_top(const TopNode<T> *__restrict__ top, const LevelInfo *info,
                                                  const PrimRec<T> *__restrict__ recs0,
                                                  const PrimRec<T> *__restrict__ recs1,
                                                  typename Wire<T>::Node *nodes, uint32_t *indices) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= info->top_count) return;
  const TopNode<T> &t = top[i];
  if (t.kind == KIND_SMALL) return;
  typename Wire<T>::Node nd;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    nd.bmin[d] = t.bmin[d];
    nd.bmax[d] = t.bmax[d];
  }
  if (t.kind == KIND_SPLIT) {
    nd.flag = 0;
    nd.axis = t.axis;
    nd.data[0] = top[t.child0].dfs;
    nd.data[1] = top[t.child0 + 1].dfs;
  } else {
    nd.flag = 1;
    nd.axis = 0;
    nd.data[0] = t.r - t.l;
    nd.data[1] = t.l;
    const PrimRec<T> *src = t.buf ? recs1 : recs0;
    for (uint32_t p = t.l; p < t.r; p++) indices[p] = src[p].prim;
  }
  nodes[t.dfs] = nd;
}

// One wave per subtree task: copy its pre-order nodes to their final place,
// rebasing child indices (cf. the reference's splice, nanort.h:2046-2058).
template <typename T>
__global__ __launch_bounds__(64) void k_emit_small(const TopNode<T> *__restrict__ top,
                                                   const uint32_t *__restrict__ small_list,
                                                   const typename Wire<T>::Node *__restrict__ scratch_nodes,
                                                   const uint16_t *__restrict__ premap, typename Wire<T>::Node *nodes,
                                                   const LevelInfo *info) {
  if (blockIdx.x >= info->num_small) return;
  const TopNode<T> &t = top[small_list[blockIdx.x]];
  const typename Wire<T>::Node *src = scratch_nodes + 2 * (size_t)t.l;
  if (premap) { // creation index -> pre-order index inside the subtree (k_subtree_rows)
    const uint16_t *map = premap + 2 * (size_t)t.l;
    for (uint32_t i = threadIdx.x; i < t.size; i += 64u) {
      typename Wire<T>::Node nd = src[i];
      if (nd.flag == 0) {
        nd.data[0] = t.dfs + map[nd.data[0]];
        nd.data[1] = t.dfs + map[nd.data[1]];
      }
      nodes[t.dfs + map[i]] = nd;
    }
    return;
  }
  for (uint32_t i = threadIdx.x; i < t.size; i += 64u) { // (k_subtree wrote its nodes in pre-order)
    typename Wire<T>::Node nd = src[i];
    if (nd.flag == 0) {
      nd.data[0] += t.dfs;
      nd.data[1] += t.dfs;
    }
    nodes[t.dfs + i] = nd;
  }
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T>
struct BuildPlan { // carve-up of the build workspace for n primitives
  size_t max_top, max_active, max_chunks;
  size_t off_recs0, off_recs1, off_scratch, off_top, off_child_acc, off_active, off_chunk_base, off_gbins,
      off_chunk_hist, off_chunk_left, off_small, off_scene, off_info, off_sort, off_premap, sort_blocks, total;
  BuildPlan(uint32_t n, size_t top_scale, bool tiny_top = false) {
    typedef typename Wire<T>::Node Node;
    max_active = (size_t)n / kHandoff + 2;
    max_chunks = (size_t)n / kTile + max_active + 1;
    max_top = top_scale * (4 * ((size_t)n / kHandoff + 1) + 64);
    if (tiny_top) max_top = 16; // test hook: the first attempt overflows, the retry path runs
    size_t o = 0;
    auto take = [&](size_t bytes) {
      const size_t at = o;
      o = align_up(o + bytes, 256);
      return at;
    };
    off_recs0 = take((size_t)n * sizeof(PrimRec<T>));
    off_recs1 = take((size_t)n * sizeof(PrimRec<T>));
    off_scratch = take(2 * (size_t)n * sizeof(Node));
    off_premap = take(2 * (size_t)n * sizeof(uint16_t));
    off_top = take(max_top * sizeof(TopNode<T>));
    off_child_acc = take(2 * (max_active + kRepNodes * kRep) * sizeof(BoundsAcc<T>)); // (+ the top levels' copies)
    off_active = take(max_active * sizeof(uint32_t));
    off_chunk_base = take(max_active * sizeof(uint32_t));
    off_gbins = take((max_active + kRepNodes * kRep) * sizeof(GBins<T>));
    off_chunk_hist = take(max_chunks * 3 * kMaxBins * sizeof(uint32_t));
    off_chunk_left = take(max_chunks * sizeof(uint32_t));
    off_small = take((max_top + 1) * sizeof(uint32_t));
    off_scene = take((1 + kSceneReplicas) * sizeof(BoundsAcc<T>));
    off_info = take(sizeof(LevelInfo));
    // Morton sort: keys/values ping-pong (4 x n u32) + digit-major block histograms
    sort_blocks = ((size_t)n + kSortTile - 1) / kSortTile;
    off_sort = take((4 * (size_t)n + 256 * sort_blocks) * sizeof(uint32_t));
    total = o;
  }
};

#define BCHK(call)                                                      \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) {                                             \
      *err = std::string(#call) + ": " + hipGetErrorString(e_);         \
      return e_;                                                        \
    }                                                                   \
  } while (0)

// ---------------------------------------------------------------------------
// Long cylinders, cut into SEGMENTS for the builder (round 5).  The cylinder example's scene is box-spanning needles (random
// end points in the scene box, examples/cylinder_primitive/main.cc:428-462): a tree over their whole boxes prunes nothing —
// every box covers a fair part of the scene (4 200 L1 look-ups per ray, 44 Mrays/s in round 4).  So the builder is handed
// one primitive per SEGMENT of a cylinder's axis, each with the tight box of its piece of the tube — the box of the two
// end points of the piece, each grown by the radius the intersector uses for the whole tube, max(r0, r1)
// (main.cc:256), plus a few ulps for the rounding of the interior end points — and the CYLINDER's id: the index array then
// names a cylinder once per segment, a leaf tests the whole cylinder (CylinderIntersector::Intersect is a pure function of
// (ray, cylinder, current t): testing a cylinder twice returns the same record or rejects it), and the closest hit of a
// cylinder lies in the box of the segment it falls on.  The segment count is fixed on the host (nrtSetCylinders: length over
// `cyl_seg_radii` tube radii, at most `cyl_split`); a cylinder of one segment keeps the reference's own box
// (CylinderGeometry::BoundingBox, main.cc:132-165: p0 -/+ r0, p1 -/+ r1).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cylinder_segments(const float *__restrict__ verts, const float *__restrict__ radii,
                                                           const uint32_t *__restrict__ seg_off, uint32_t n,
                                                           float *__restrict__ seg_verts, float *__restrict__ seg_radii,
                                                           uint32_t *__restrict__ seg_prim) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t first = seg_off[i], K = seg_off[i + 1] - first;
  float p0[3], p1[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    p0[k] = verts[6 * (size_t)i + k];
    p1[k] = verts[6 * (size_t)i + 3 + k];
  }
  const float r0 = radii[2 * (size_t)i], r1 = radii[2 * (size_t)i + 1];
  if (K <= 1u) { // unsplit: the reference's own box
#pragma unroll
    for (int k = 0; k < 3; k++) {
      seg_verts[6 * (size_t)first + k] = p0[k];
      seg_verts[6 * (size_t)first + 3 + k] = p1[k];
    }
    seg_radii[2 * (size_t)first] = r0;
    seg_radii[2 * (size_t)first + 1] = r1;
    seg_prim[first] = i;
    return;
  }
  const float rr = r0 > r1 ? r0 : r1; // std::max<float>(r0, r1), main.cc:256
  const float invK = 1.0f / (float)K;
  float a[3] = {p0[0], p0[1], p0[2]};
  // The interior end points p0 + (p1 - p0) * s are rounded: p1 - p0 alone carries an error of the order of an ulp of the
  // CYLINDER's end points, whatever the size of the interior point itself (a long cylinder spanning the origin has interior
  // points near 0 whose error is that of its far ends).  So the slack the radius carries is sized once, from the end points.
  const float mag = fmaxf(fmaxf(fabsf(p0[0]), fabsf(p0[1])), fabsf(p0[2])) + fmaxf(fmaxf(fabsf(p1[0]), fabsf(p1[1])), fabsf(p1[2])) + rr;
  const float rs = rr + 1.0e-6f * mag;
  for (uint32_t j = 0; j < K; j++) {
    float b[3];
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] = (j + 1u == K) ? p1[k] : p0[k] + (p1[k] - p0[k]) * ((float)(j + 1u) * invK);
    const size_t o = (size_t)first + j;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      seg_verts[6 * o + k] = a[k];
      seg_verts[6 * o + 3 + k] = b[k];
      a[k] = b[k];
    }
    seg_radii[2 * o] = rs;
    seg_radii[2 * o + 1] = rs;
    seg_prim[o] = i;
  }
}

hipError_t launch_cylinder_segments(const float *verts, const float *radii, const uint32_t *seg_off, uint32_t n, float *seg_verts,
                                    float *seg_radii, uint32_t *seg_prim, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_cylinder_segments, dim3((n + 255u) / 256u), dim3(256), 0, s, verts, radii, seg_off, n, seg_verts, seg_radii, seg_prim);
  return hipGetLastError();
}

// Builds into caller-owned grow-only buffers (no allocation in the steady state of a per-frame rebuild).
// The host has to learn two things from the device: that the top phase has run out of large nodes, and the tree's size
// and statistics.  Neither read-back leaves the GPU idle: the state block is copied into page-locked memory (`pinned`,
// >= kBuildPinnedBytes) behind `ev`, and the stream is kept fed while the host waits for that event — with the level's own
// kernels in the first case (they do nothing if the level turns out to be empty), with the emission kernels in the second
// (their grids are upper bounds; the node array is sized for the 2n - 1 nodes a tree over n primitives can have).
// gpu_build returns once everything is enqueued; gpu_build_result() waits for the final state block.
template <typename T>
hipError_t gpu_build(hipStream_t s, const T *d_verts, const uint32_t *d_faces, const T *d_radii, bool cylinders, const uint32_t *d_prim_map, uint32_t n,
                     uint32_t min_leaf, uint32_t max_depth, uint32_t bin_size, unsigned build_flags, DevBuf *workspace, DevBuf *nodes_buf,
                     DevBuf *indices_buf, void *pinned, hipEvent_t ev, std::string *err) {
  typedef typename Wire<T>::Node Node;
#ifdef NRT_PROF
  const bool subtree_rows = (build_flags & kBuildSubtreeDfs) == 0;
#else
  const bool subtree_rows = true; // (k_subtree is not in this library)
#endif
  const bool morton_order = (build_flags & kBuildMorton) != 0;
  static_assert(offsetof(LevelInfo, level_begin) <= kBuildPinnedBytes, "state block");
  const int K = (int)(bin_size < 2 ? 2 : (bin_size > (uint32_t)kMaxBins ? (uint32_t)kMaxBins : bin_size));
  const int Ks = K < kSmallBins ? K : kSmallBins;
  const size_t state_bytes = offsetof(LevelInfo, level_begin);
  volatile const LevelInfo *hp = (volatile const LevelInfo *)pinned;

  // (test hook, NRT_BUILD_TINY_TOP=1: the first attempt gets a 16-node top array, so every build of more than a few
  // thousand primitives overflows once and is retried — the path lopsided splits take)
  const char *tiny_env = env_overrides_allowed() ? getenv("NRT_BUILD_TINY_TOP") : nullptr;
  const bool tiny_top = tiny_env && atoi(tiny_env) != 0;
  for (size_t top_scale = 1;; top_scale *= 8) {
    const BuildPlan<T> plan(n, top_scale, tiny_top && top_scale == 1);
    BCHK(devbuf_ensure(workspace, plan.total));
    BCHK(devbuf_ensure(indices_buf, (size_t)n * sizeof(uint32_t)));
    BCHK(devbuf_ensure(nodes_buf, (2 * (size_t)n) * sizeof(Node)));
    char *base = (char *)workspace->p;
    PrimRec<T> *recs[2] = {(PrimRec<T> *)(base + plan.off_recs0), (PrimRec<T> *)(base + plan.off_recs1)};
    Node *scratch = (Node *)(base + plan.off_scratch);
    uint16_t *premap = (uint16_t *)(base + plan.off_premap);
    TopNode<T> *top = (TopNode<T> *)(base + plan.off_top);
    BoundsAcc<T> *child_acc = (BoundsAcc<T> *)(base + plan.off_child_acc);
    uint32_t *active = (uint32_t *)(base + plan.off_active);
    uint32_t *chunk_base = (uint32_t *)(base + plan.off_chunk_base);
    GBins<T> *gbins = (GBins<T> *)(base + plan.off_gbins);
    uint32_t *chunk_hist = (uint32_t *)(base + plan.off_chunk_hist);
    uint32_t *chunk_left = (uint32_t *)(base + plan.off_chunk_left);
    uint32_t *small_list = (uint32_t *)(base + plan.off_small);
    BoundsAcc<T> *scene = (BoundsAcc<T> *)(base + plan.off_scene);
    LevelInfo *info = (LevelInfo *)(base + plan.off_info);
    uint32_t *indices = (uint32_t *)indices_buf->p;

    NRT_RANGE_PUSH("build: primitive records (per-prim AABB + centroid, scene bounds)");
    hipLaunchKernelGGL((k_init_scene<T>), dim3(1 + kRepNodes * kRep), dim3(64), 0, s, scene, info, (uint32_t)plan.max_top, gbins, child_acc,
                       (uint32_t)plan.max_active);
    {
      unsigned grid = (unsigned)std::min<size_t>(((size_t)n + 255) / 256, 2048);
      hipLaunchKernelGGL((k_prim_records<T>), dim3(grid), dim3(256), 0, s, d_verts, d_faces, d_radii, cylinders, n, d_prim_map, recs[0], scene);
    }
    NRT_RANGE_POP();
    int cur = 0; // buffer holding the ranges of the nodes being split
    if (morton_order && n > 1) {
      NRT_RANGE("build: Morton keys + radix sort");
      uint32_t *keys[2] = {(uint32_t *)(base + plan.off_sort), (uint32_t *)(base + plan.off_sort) + (size_t)n};
      uint32_t *vals[2] = {keys[1] + (size_t)n, keys[1] + 2 * (size_t)n};
      uint32_t *block_hist = keys[1] + 3 * (size_t)n;
      const unsigned nb = (unsigned)plan.sort_blocks;
      hipLaunchKernelGGL((k_combine_scene<T>), dim3(1), dim3(64), 0, s, scene);
      hipLaunchKernelGGL((k_morton_keys<T>), dim3((n + 255) / 256), dim3(256), 0, s, recs[0], n, scene, keys[0], vals[0]);
      int pp = 0;
      for (int shift = 0; shift < 32; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(256), 0, s, keys[pp], n, shift, block_hist, nb);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, s, block_hist, 256u * nb);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(256), 0, s, keys[pp], vals[pp], keys[1 - pp], vals[1 - pp], n,
                           shift, block_hist, nb);
        pp = 1 - pp;
      }
      hipLaunchKernelGGL((k_gather_records<T>), dim3((n + 255) / 256), dim3(256), 0, s, recs[0], vals[pp], n, recs[1]);
      cur = 1;
    }
    const LeafRule rule = {max_depth, min_leaf > 1u ? min_leaf : 1u};
    hipLaunchKernelGGL((k_make_root<T>), dim3(1), dim3(64), 0, s, scene, n, rule, (uint32_t)cur, top, small_list,
                       info);
    BCHK(hipGetLastError());

    // ---- top phase: level after level, grids sized by upper bounds ----------------------------
    // From the level at which the nodes could first all be small, the state block written by k_level_setup is read
    // back every level; the level's kernels are enqueued before the host waits for it.
    int expect = 0;
    for (size_t m = (size_t)n / kHandoff; m > 0; m >>= 1) expect++;
    const int first_check = (n <= (uint32_t)kHandoff) ? 0 : expect + 2;
    bool overflow = false;
    uint32_t num_small = 0;
    NRT_RANGE_PUSH("build: top phase (level setup / bin / split / partition per level)");
    for (int level = 0;; level++) {
      // children of the level partitioned last (their records are in recs[cur]): inside k_level_setup while a level
      // cannot have more than kNarrowLevel active nodes, by a grid of their own beyond that
      const size_t prev_max = level == 0 ? 0 : (level - 1 < 31 ? std::min<size_t>((size_t)1 << (level - 1), plan.max_active) : plan.max_active);
      const bool wide = prev_max > kNarrowLevel;
      if (wide)
        hipLaunchKernelGGL((k_children<T>), dim3((unsigned)((prev_max + 255) / 256)), dim3(256), 0, s, top, active, child_acc,
                           (uint32_t)plan.max_active, rule, (uint32_t)cur, small_list, info);
      hipLaunchKernelGGL((k_level_setup<T>), dim3(1), dim3(1024), 0, s, top, active, chunk_base, info, child_acc,
                         (uint32_t)plan.max_active, rule, (uint32_t)cur, small_list, wide ? 0 : 1);
      const bool check = level >= first_check;
      if (check) {
        BCHK(hipMemcpyAsync(pinned, info, state_bytes, hipMemcpyDeviceToHost, s));
        BCHK(hipEventRecord(ev, s));
      }
      const size_t a_max = level < 31 ? std::min<size_t>((size_t)1 << level, plan.max_active) : plan.max_active;
      const size_t c_max = std::min<size_t>((size_t)n / kTile + a_max + 1, plan.max_chunks);
      hipLaunchKernelGGL((k_bin<T>), dim3((unsigned)c_max), dim3(256), 0, s, top, active, chunk_base, info, recs[cur],
                         K | (Ks << 8), gbins, chunk_hist, chunk_left, (uint32_t)plan.max_active);
      hipLaunchKernelGGL((k_split<T>), dim3((unsigned)a_max), dim3(64), 0, s, top, active, gbins, K | (Ks << 8), chunk_hist,
                         chunk_left, info, (uint32_t)plan.max_active);
      hipLaunchKernelGGL((k_partition<T>), dim3((unsigned)c_max), dim3(256), 0, s, top, active, chunk_base, info,
                         chunk_left, recs[cur], recs[1 - cur], K | (Ks << 8), child_acc, (uint32_t)plan.max_active);
      BCHK(hipGetLastError());
      cur = 1 - cur;
      if (check) {
        BCHK(hipEventSynchronize(ev));
        if (hp->error) {
          overflow = true;
          break;
        }
        if (hp->num_active == 0) { // (the three launches above found nothing to do)
          num_small = hp->num_small;
          break;
        }
      }
    }
    NRT_RANGE_POP();
    if (overflow) continue; // lopsided splits outgrew the top array: retry with a larger one
    NRT_RANGE("build: subtree phase + layout + emission");

    // ---- subtree phase + relayout + emission ---------------------------------------------------------
    if (num_small) {
#ifdef NRT_PROF
      if (!subtree_rows) // (the one-node-per-step form: the cross-check of the row form, tunable subtree_rows = 0 of the profiling build)
        hipLaunchKernelGGL((k_subtree<T>), dim3(num_small), dim3(64), 0, s, top, small_list, recs[0], recs[1], Ks,
                           min_leaf, max_depth, scratch, indices, info);
      else
#endif
        hipLaunchKernelGGL((k_subtree_rows<T>), dim3(num_small), dim3(64), 0, s, top, small_list, recs[0], recs[1], Ks,
                           min_leaf, max_depth, scratch, premap, indices, info);
    }
    BCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_layout<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)(2 * kLayoutLds * sizeof(uint32_t)))); // (per device: set on every build, it costs nothing)
    if (plan.max_top > kLayoutLds) { // (a top array that may not fit LDS: the grid-wide form of the same steps; no-ops if it does fit)
      const unsigned lg = (unsigned)((plan.max_top + 255) / 256);
      hipLaunchKernelGGL((k_layout_init<T>), dim3(lg), dim3(256), 0, s, top, info);
      hipLaunchKernelGGL((k_layout_sizes<T>), dim3(lg), dim3(256), 0, s, top, info);
    }
    hipLaunchKernelGGL((k_layout<T>), dim3(1), dim3(1024), 2 * kLayoutLds * sizeof(uint32_t), s, top, info);
    if (plan.max_top > kLayoutLds)
      hipLaunchKernelGGL((k_layout_dfs<T>), dim3((unsigned)((plan.max_top + 255) / 256)), dim3(256), 0, s, top, info);
    BCHK(hipMemcpyAsync(pinned, info, state_bytes, hipMemcpyDeviceToHost, s));
    BCHK(hipEventRecord(ev, s));
    Node *nodes = (Node *)nodes_buf->p;
    hipLaunchKernelGGL((k_emit_top<T>), dim3((unsigned)((plan.max_top + 255) / 256)), dim3(256), 0, s, top, info, recs[0], recs[1],
                       nodes, indices);
    if (num_small)
      hipLaunchKernelGGL((k_emit_small<T>), dim3(num_small), dim3(64), 0, s, top, small_list, scratch,
                         subtree_rows ? premap : (uint16_t *)nullptr, nodes, info);
    BCHK(hipGetLastError());
    return hipSuccess;
  }
}

// Waits for the state block of the build enqueued last with this (pinned, ev) pair.
hipError_t gpu_build_result(const void *pinned, hipEvent_t ev, BuildResult *res) {
  hipError_t e = hipEventSynchronize(ev);
  if (e != hipSuccess) return e;
  volatile const LevelInfo *hp = (volatile const LevelInfo *)pinned;
  res->num_nodes = hp->num_nodes;
  res->max_depth = hp->max_depth;
  res->num_leaves = hp->num_leaves;
  res->num_branches = hp->num_branches;
  res->max_leaf_count = hp->max_leaf_count;
  return hipSuccess;
}

template hipError_t gpu_build<float>(hipStream_t, const float *, const uint32_t *, const float *, bool, const uint32_t *, uint32_t, uint32_t,
                                     uint32_t, uint32_t, unsigned, DevBuf *, DevBuf *, DevBuf *, void *, hipEvent_t, std::string *);
template hipError_t gpu_build<double>(hipStream_t, const double *, const uint32_t *, const double *, bool, const uint32_t *, uint32_t, uint32_t,
                                      uint32_t, uint32_t, unsigned, DevBuf *, DevBuf *, DevBuf *, void *, hipEvent_t, std::string *);

} // namespace nrt

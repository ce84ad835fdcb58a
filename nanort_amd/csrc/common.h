// nanort_amd/csrc/common.h — device-side PODs and launch-argument blocks shared
// by the gfx950 kernels (traverse.hip, build.hip) and the C ABI (api.hip).
//
// The node / ray / hit records are the reference's wire formats
// (include/nanort_hip.h; reference nanort.h:474-550, 996-1005).  Everything
// else in this file is private device layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nanort_hip.h"

// roctx ranges: compiled into the PROFILING library only (-DNRT_PROF -DNRT_ROCTX, nanort_amd/csrc/Makefile), so that a
// `rocprofv3 --marker-trace --kernel-trace` run of tools/ names the build phases and every traversal launch; the product
// library carries no marker calls.  Ranges bracket the host-side enqueue of a phase (the kernels run asynchronously).
#ifdef NRT_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
namespace nrt {
struct RoctxRange {
  explicit RoctxRange(const char *msg) { roctxRangePushA(msg); }
  ~RoctxRange() { roctxRangePop(); }
};
} // namespace nrt
#define NRT_RANGE_CAT2(a, b) a##b
#define NRT_RANGE_CAT(a, b) NRT_RANGE_CAT2(a, b)
#define NRT_RANGE(msg) ::nrt::RoctxRange NRT_RANGE_CAT(nrt_range_, __LINE__)(msg)
#define NRT_RANGE_PUSH(msg) roctxRangePushA(msg)
#define NRT_RANGE_POP() roctxRangePop()
#else
#define NRT_RANGE(msg) do { } while (0)
#define NRT_RANGE_PUSH(msg) do { } while (0)
#define NRT_RANGE_POP() do { } while (0)
#endif

namespace nrt {

// NRT_<NAME> environment overrides are a debugging aid: the profiling library always honours them, the product library only
// when the process opts in with NRT_ALLOW_ENV=1 — a stray variable must not move a product context to another walk or
// parity class behind its caller's back.
inline bool env_overrides_allowed() {
#ifdef NRT_PROF
  return true;
#else
  const char *e = getenv("NRT_ALLOW_ENV");
  return e && e[0] == '1' && e[1] == 0;
#endif
}

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kTraverseBlock = 256; // 4 waves, one per SIMD
constexpr int kLdsStackDefault = 32; // per-lane stack entries kept in LDS (32 -> 32 KiB / block)
constexpr unsigned kInvalid = 0xFFFFFFFFu;
constexpr unsigned kCursorStrideWords = 1024; // per-partition work cursors 4 KiB apart (separate memory channels)
constexpr unsigned kMaxParts = 16;
constexpr size_t kBuildPinnedBytes = 1024; // page-locked block the builder's state is read back into (build.hip)

// Grow-only device buffer owned by a context.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};
inline hipError_t devbuf_ensure(DevBuf *b, size_t bytes) {
  if (bytes <= b->cap) return hipSuccess;
  if (b->p) {
    hipError_t e = hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
    if (e != hipSuccess) return e;
  }
  const size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&b->p, want);
  if (e != hipSuccess) {
    b->p = nullptr;
    return e;
  }
  b->cap = want;
  return hipSuccess;
}

template <typename T>
struct Wire;
template <>
struct Wire<float> {
  typedef nrt_ray_f32 Ray;
  typedef nrt_node_f32 Node;
  typedef nrt_hit_f32 Hit;
  typedef nrt_build_options_f32 BuildOptions;
};
template <>
struct Wire<double> {
  typedef nrt_ray_f64 Ray;
  typedef nrt_node_f64 Node;
  typedef nrt_hit_f64 Hit;
  typedef nrt_build_options_f64 BuildOptions;
};

// Leaf-ordered triangle record: slot s of the index permutation holds the three
// vertices of primitive indices[s] plus its id, so a leaf {count, first} reads
// `count` consecutive records instead of chasing indices -> faces -> vertices.
// Same values as the reference's gather (nanort.h:1065-1071), pre-resolved.
template <typename T>
struct alignas(8) LeafTri {
  T p0[3];
  T p1[3];
  T p2[3];
  uint32_t prim_id;
};
static_assert(sizeof(LeafTri<float>) == 40, "LeafTri<float>");

// Leaf-ordered sphere record (primitive kind 1: the particle primitive of
// examples/particle_primitive/main.cc:82-291).
template <typename T>
struct LeafSphere {
  T c[3];
  T r;
  uint32_t prim_id;
};
static_assert(sizeof(LeafSphere<float>) == 20, "LeafSphere<float>");

// Leaf-ordered cylinder record (primitive kind 2: examples/cylinder_primitive/main.cc:94-424).
template <typename T>
struct LeafCylinder {
  T p0[3];
  T p1[3];
  T r0, r1;
  uint32_t prim_id;
};
static_assert(sizeof(LeafCylinder<float>) == 36, "LeafCylinder<float>");
enum : int { kPrimTriangles = 0, kPrimSpheres = 1, kPrimCylinders = 2 };
static_assert(sizeof(LeafTri<double>) == 80, "LeafTri<double>");

// Private traversal layout: one record per BRANCH node holding BOTH children's boxes, so a
// step fetches one record and tests two boxes.  Records are dense, in the pre-order of the
// branch nodes (record j <-> the j-th branch of the BVHNode array); the child boxes are copied
// bit-for-bit from the BVHNode array.
// Child reference (32 bits): bit 31 clear -> inner child, value = its WideNode index;
// bit 31 set -> leaf: either PACKED {count-1 : 4 bits [30:27], first slot : 27 bits} when every
// leaf of the tree has 1..16 primitives and the index array has < 2^27 slots, or INDIRECT
// {BVHNode index of the leaf : 31 bits} (count/first are then read from that node).
template <typename T>
struct alignas(16) WideNode {
  T box0[6]; // child data[0]: bmin, bmax
  T box1[6]; // child data[1]
  uint32_t c0, c1;
  int32_t axis;
  uint32_t pad;
};
// fp64: the same record component-major — mn[k][child], mx[k][child]: a 16-byte row per plane pair, so that the walk fetches
// the near and far rows of each axis by the ray's direction sign (a per-ray byte offset, as for Wide4Node<float>:
// traverse.hip slab_pair_presel) instead of fetching all twelve doubles and selecting six pairs.
template <>
struct alignas(16) WideNode<double> {
  double mn[3][2]; // [component][child]
  double mx[3][2];
  uint32_t c0, c1;
  int32_t axis;
  uint32_t pad;
};
static_assert(sizeof(WideNode<float>) == 64, "WideNode<float>");
static_assert(sizeof(WideNode<double>) == 112, "WideNode<double>");
// Two levels of the tree in one record: the boxes of the (up to) four GRANDCHILDREN of a branch node, component-major
// so that one 16-byte load brings the same plane of all four boxes.  Slots 0,1 = the children of child data[0], slots
// 2,3 = the children of child data[1]; a child that is itself a leaf sits in the first slot of its half (its own box
// and leaf reference) and the second slot of that half is empty (c == kWide4Empty).  There is one record per branch
// node, at the same dense index as its WideNode, so any branch can be entered through either array; a walk that
// starts at record 0 and follows c[] only ever touches the records of the even-depth branches.
// axis0 = split axis of the node, axis1 / axis2 = split axes of child data[0] / data[1] (0 when that child is a leaf).
// Walking this array visits the same leaves in the same order as the binary loop (see NRT_STEP_NODE4, traverse.hip).
template <typename T>
struct alignas(16) Wide4Node {
  T bmin[3][4]; // [component][slot]
  T bmax[3][4];
  uint32_t c[4]; // child references, encoded like WideNode::c0 / c1
  int32_t axis0, axis1, axis2;
  uint32_t pad;
};
static_assert(sizeof(Wide4Node<float>) == 128, "Wide4Node<float>");
static_assert(sizeof(Wide4Node<double>) == 224, "Wide4Node<double>");
constexpr uint32_t kWide4Empty = 0xFFFFFFFFu;
#ifndef NRT_W4_LDS_STACK
#define NRT_W4_LDS_STACK 12
#endif
constexpr int kWide4LdsStack = NRT_W4_LDS_STACK; // per-lane LDS stack entries of the WIDTH = 4 variants (24 KiB per block: six blocks per CU)
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kPackedFirstBits = 27;
constexpr uint32_t kPackedFirstMask = (1u << kPackedFirstBits) - 1u;
constexpr uint32_t kPackedMaxCount = 16;

// Device-side view of a context's fp32 tree for the code inside this library that walks it in its own kernels (the
// two-level scene kernel, scene.hip).  Not part of the C ABI.
struct TreeViewF32 {
  const nrt_node_f32 *nodes;     // reference-format node array
  const uint32_t *indices;       // index permutation
  const void *wide;              // WideNode<float>[branches]
  const void *wide4;             // Wide4Node<float>[branches], or null when the tree is not walked two levels per step
  const void *prims;             // leaf-ordered primitive records (LeafTri<float> for triangle contexts)
  uint32_t num_nodes, num_indices;
  uint32_t packed_leaves, root_is_branch, tree_nested, prim_kind, tree_depth;
  uint32_t max_leaf_count;       // most records a leaf of the tree holds
  uint64_t generation;           // counts the context's rebuilds: a view is stale once the context's generation has moved on
};

// ---- two-level (instanced) traversal: one kernel for a whole ray batch (traverse.hip k_scene_trace) -------------
// Per instance, in HBM: where its tree lives and the three matrices nanosg's Node::Update derives (nanosg.h:397-437).
struct SceneInst {
  const void *wide;           // WideNode<float>[]
  const void *wide4;          // Wide4Node<float>[] (two levels per step) or null: the tree is walked one level per step
  const void *tris;           // LeafTri<float>[] in index-array order
  const nrt_node_f32 *nodes;  // reference-format nodes (leaf {count, first} when the leaf references are not packed; node 0's box)
  uint32_t packed_leaves, root_is_branch, tree_nested, pad;
  float inv_xform[4][4];      // world -> local, points
  float inv_xform33[4][4];    // world -> local, directions
  float xform[4][4];          // local -> world
  float xbmin[3], xbmax[3];   // world box
  uint32_t id, pad2;          // instance id (== its index in the by-id table)
};
static_assert(sizeof(SceneInst) == 272, "SceneInst");
// Does the ray enter an instance's world box, and over which interval?  First the BVH leaf's robust test (IntersectRayAABB,
// nanort.h:2285-2325, hit_t == ray.max_t throughout ListNodeIntersections), then NodeBBoxIntersector::Intersect
// (nanosg.h:603-639: plain reciprocal, no MaxMult, no clipping), whose near end the reference's list is sorted by.
__device__ __forceinline__ bool scene_node_interval(const nrt_ray_f32 &r, const float xbmin[3], const float xbmax[3], float &t_min_out) {
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
    const float lo = neg ? xbmax[k] : xbmin[k], hi = neg ? xbmin[k] : xbmax[k];
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
constexpr int kSceneLdsStack = 12; // per-lane stack entries of k_scene_trace kept in LDS; deeper ones go to the overflow arrays
struct SceneTraceArgs {
  const nrt_ray_f32 *rays;
  uint32_t n;
  const SceneInst *insts;
  float *list_t;              // [cap][n]: entry distances of the instances a ray enters, unsorted (k_scene_list*, or this kernel itself: scan_nodes)
  uint32_t *list_node;        // [cap][n]: their ids
  const uint32_t *count;      // [n]: list lengths (unused when scan_nodes != 0)
  uint32_t scan_nodes;        // != 0: a scene of this many (a handful of) instances — the kernel lists a ray's instances itself when it
                              // fetches the ray, by testing every world box (insts[k].xbmin/xbmax), and no listing kernel runs
  nrt_scene_hit_f32 *hits;    // [n] out
  uint8_t *mask;              // [n] out, may be null
  uint32_t *spill;            // overflow stack [spill_levels][spill_stride], may be null
  float *spill_tmin;
  uint32_t spill_stride;
  uint32_t *cursor;           // work cursors (next unclaimed ray of each partition), kCursorStrideWords apart, zero at launch
  uint32_t num_parts;         // ray partitions (<= kMaxParts)
  uint32_t refill_min;        // free lanes of a wave before it claims more rays
  uint32_t trav_min;          // lanes still walking inner nodes below which the wave turns to the waiting leaves
  uint32_t cand_min;          // lanes waiting between two instances (a local walk ended / the next candidate is due) before the wave runs
                              // that — expensive, divergent — step for them; fewer wait while `cand_busy_max` or more lanes still walk
  uint32_t cand_busy_max;
  const uint32_t *subset;     // non-null: the launch traces rays[subset[s]] for s < n (lists and counts are indexed by s, results by the ray)
};

// The single-pass scene walk (traverse.hip k_scene_walk): the top-level tree and the instances' trees walked by the same lane on
// one stack, no per-ray list.  Rays it cannot certify (see the kernel) are appended to `redo` for the listing path.
#ifndef NRT_SCENE_WALK_STACK
#define NRT_SCENE_WALK_STACK 16
#endif
constexpr int kSceneWalkLdsStack = NRT_SCENE_WALK_STACK; // per-lane stack entries kept in LDS (top-level entries below, the open instance's above)
// What opening an instance needs, one 128-byte line per instance, in the order of the top-level tree's index array (a leaf
// reference's `first` indexes the table): the twelve entries of each matrix that Matrix::MultV touches (nanosg.h:232-240:
// m[0..3][0..2]), the world box, the id, and which mesh (tree) it instantiates.
struct alignas(128) SceneOpen {
  float inv[4][3];   // inv_xform: world -> local, points
  float inv33[4][3]; // inv_xform33: world -> local, directions
  float xbmin[3], xbmax[3];
  uint32_t id, mesh;
};
static_assert(sizeof(SceneOpen) == 128, "SceneOpen");
struct SceneMesh { // per distinct mesh context: its private traversal arrays (two levels per step, packed leaf references)
  const void *wide4;
  const void *tris;
  uint32_t root_leaf;  // the tree is ONE leaf (a quad, a billboard): no records to step through — its box is tested, then its triangles
  uint32_t leaf_ref;   // ... that leaf's packed reference (count - 1, first)
  float bmin[3], bmax[3]; // ... and node 0's box
};
static_assert(sizeof(SceneMesh) == 48, "SceneMesh");
struct SceneWalkArgs {
  const nrt_ray_f32 *rays;
  uint32_t n;
  const SceneOpen *open_top;         // see SceneOpen
  const SceneMesh *meshes;
  const SceneInst *insts;            // the full instance table, by id (its xform is read when an instance was hit)
  const Wide4Node<float> *top_wide4; // the top-level tree over the instances' world boxes (root is a branch, nested, packed leaves)
  nrt_scene_hit_f32 *hits;
  uint8_t *mask;
  uint32_t *spill;
  float *spill_tmin;
  uint32_t spill_stride;
  uint32_t *cursor;
  uint32_t num_parts;
  uint32_t refill_min, trav_min, cand_min, cand_busy_max;
  uint32_t leaf_items;  // the leaf phase may hand the waiting lanes' records out over the wave (every mesh's leaves hold <= 4 records; tunable walk_leaf_items)
  uint32_t *redo;       // [n]: rays left to the listing path
  uint32_t *redo_count; // zero at launch
  unsigned long long *counters; // profiling build only (libnanort_hip_prof.so): 13 loop counters of the launch, or null
};

// Completion record of a launch slot, in page-locked host memory the device writes to: the last wave of a traversal
// launch to finish stores the launch's start / end stamps (100 MHz s_memrealtime ticks) and then its sequence number
// (system-scope release).  Whoever must know that a launch is over (a rebuild, destroy, another stream taking the slot
// over, nrtLastTraverseMs) polls `seq` — no event is recorded in the stream, so consecutive launches run back to back.
struct DoneRec {
  uint32_t seq;
  uint32_t pad;
  unsigned long long t_begin, t_end;
};
// Device-side words that go with it (one set per slot, zero / ~0 between launches).
struct DoneCount {
  uint32_t exited;   // wave groups of the current launch that have finished
  uint32_t pad;
  unsigned long long t_begin; // earliest start stamp seen (atomicMin)
  uint32_t group[8]; // waves of group g = blockIdx % 8 that have finished (eight words instead of one: same-address atomics serialise)
};

// Several independent batches walked by ONE persistent launch (nrtTraverseBatchesDevice: one tail, one completion record for all):
// the claim machinery sees one virtual ray array, the batches back to back; a lane resolves its virtual index to a batch when it
// loads the ray and when it stores the result.  The pointers are pre-offset so that they are addressed BY THE VIRTUAL INDEX.
constexpr int kMaxBatches = 8;
struct BatchPtrs {
  const void *rays_v;
  void *hits_v;
  uint8_t *mask_v; // may be null
  uint64_t pad;
};

template <typename T>
struct TraverseArgs {
  const typename Wire<T>::Node *nodes;
  const LeafTri<T> *tris;        // primitive kind 0
  const LeafSphere<T> *spheres;  // primitive kind 1 (leaf order)
  const T *centers;              // primitive kind 1: xyz per primitive id (PostTraversal)
  const LeafCylinder<T> *cylinders; // primitive kind 2 (leaf order)
  uint32_t cyl_test_cap;         // primitive kind 2: the intersector's test_cap flag
  const WideNode<T> *wide; // may be null (binary kernel only)
  const Wide4Node<T> *wide4; // may be null: two tree levels per record (the WIDTH = 4 variants)
  uint32_t packed_leaves;  // leaf references of `wide` are PACKED (see WideNode)
  uint32_t wide4_big;      // the Wide4Node array is 4 GiB or larger: the walk addresses its records with 64-bit offsets (template bit ORDER & 4)
  uint32_t wide_below_4g;  // the WideNode array is smaller than 4 GiB: the fp64 walk may address it with 32-bit byte offsets (slab_pair_presel)
  uint32_t root_is_branch; // node 0 is a branch (every tree of more than one node)
  uint32_t debug_flags;    // profiling only (env NRT_DEBUG): 1 = skip triangle tests, 2 = skip traversal
  const typename Wire<T>::Ray *rays;
  typename Wire<T>::Hit *hits; // may be null (counting pass)
  uint8_t *mask;               // may be null
  uint32_t num_rays;
  uint32_t num_batches;             // > 1: the rays are `num_batches` batches back to back (fp32 WideNode kernels only): batches[] / batch_end[] instead of rays / hits / mask
  uint32_t batch_anyhit;            // bit k: batch k is an occlusion query (its rays stop at the first primitive they accept; only the flags are written)
  uint32_t batch_end[kMaxBatches];  // virtual index one past the last ray of batch k
  BatchPtrs batches[kMaxBatches];
  uint32_t range0, range1, skip_prim; // BVHTraceOptions
  uint32_t cull_back_face;
  uint32_t any_hit;       // occlusion query (opt-in extension): a ray stops at the first primitive it accepts
  uint32_t plain_options; // the options above cannot reject any primitive of this tree (host-checked)
  uint32_t root_test;     // node 0's box must be tested before its children (an adopted tree whose child boxes may stick out)
  uint32_t leaf_items;    // two-level walk: the leaf phase hands the waiting lanes' records out over the whole wave (tunable leaf_compact; leaves of <= 4 records)
  uint32_t order4;        // two-level walk: enter the four slots of a record by entry distance instead of the binary loop's order (tunable order4)
  uint32_t *spill;        // [spill_levels][spill_stride] overflow stack, may be null
  T *spill_tmin;          // same shape, entry t_min (wide kernel)
  uint32_t spill_stride;  // == total threads of the launch
  uint32_t spill_levels;
  uint32_t *ray_cursor;              // persistent-thread work counters, one per ray partition, 4 KiB apart, zero at launch
  uint32_t *next_cursor;             // the set the NEXT launch of this slot will use: block 0 zeroes it (no memset launch)
  uint32_t num_parts;                // ray partitions (== XCDs): contiguous ranges of the ray array, one home range per XCD
  // work distribution (traverse.hip, Claim): `static_bands` bands of band_len rays + a tail
  uint32_t static_per_wave;          // rays per static slice: slice `rank` of band b = [b*band_len + rank*static_per_wave, +static_per_wave) belongs to wave `rank` without any atomic (0: no static share)
  uint32_t static_bands;             // bands
  uint32_t band_len;                 // rays per band
  uint32_t band_static;              // static part of a band = waves * static_per_wave; the rest of the band is dynamic
  uint32_t dyn_per_band;             // = band_len - band_static, a whole number of chunks
  uint32_t dyn_banded;               // = static_bands * dyn_per_band: virtual dynamic rays below this lie in the bands, the others in the tail
  uint32_t tail_begin;               // = static_bands * band_len: rays [tail_begin, num_rays) are dynamic
  uint32_t dyn_total;                // all dynamic rays = dyn_banded + num_rays - tail_begin
  uint32_t dyn_per_part;             // virtual dynamic rays per partition cursor (whole chunks; the last partition takes the rest)
  uint32_t blocks_per_part;          // gridDim.x / num_parts
  uint32_t dyn_head;                 // a batch without a static share: the first chunk of a wave is chunk `wave index` of its home range, without an atomic (tunable dyn_head)
  unsigned long long *counters;      // 4 x u64 when counting
  unsigned long long *wave_clock;    // profiling (NRT_DEBUG bit 8192): 3 x u64 per wave {start, out of rays, done}, 100 MHz ticks; else null
  uint32_t chunk;                    // rays claimed per atomic (a multiple of 32)
  uint32_t chunk_tail_pct;           // the last this-many percent of every partition's dynamic range go out in half chunks
  uint32_t refill_min;               // refill idle lanes once this many are idle (1..64)
  uint32_t trav_min;                 // leave the inner-node loop when fewer lanes than this are walking
  uint32_t leaf_min;                 // with fewer lanes than this waiting at a leaf, refill first (if a refill is due) and test triangles later
  DoneRec *done_rec;                 // completion record of this launch's slot (device-visible host memory), or null: none
  DoneCount *done_count;             // its device-side words
  uint32_t done_seq;                 // sequence number of this launch within its slot
  uint32_t done_publish;             // 1: the traversal kernel's last wave closes the record; 0: a post pass behind it does (sphere / cylinder kinds)
};

} // namespace nrt

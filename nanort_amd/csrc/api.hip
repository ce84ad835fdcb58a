// nanort_amd/csrc/api.hip — the C ABI of libnanort_hip.so (include/nanort_hip.h).
//
// One nrt_ctx == one reference BVHAccel<T> (nanort.h:698-860) resident on one
// MI355X: mesh, node array, index permutation and the leaf-ordered triangle
// records live in HBM for the lifetime of the context; the host entry points
// stage rays/hits through grow-only device buffers.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.h"
#ifdef NRT_PROF
#include "../../include/nanort_hip_prof.h"
#endif

namespace nrt {
template <typename T>
hipError_t launch_traverse(const TraverseArgs<T> &, unsigned grid, bool count, int lds_stack, hipStream_t);
template <typename T>
int traverse_blocks_per_cu(int lds_stack);
template <typename T>
hipError_t launch_traverse_wide(const TraverseArgs<T> &, unsigned grid, int lds_stack, int prim_kind, hipStream_t, const char **name_out);
template <typename T>
int traverse_wide_blocks_per_cu(int lds_stack, int prim_kind, bool wide4);
template <typename T>
hipError_t launch_gather_leaf_spheres(const uint32_t *, const T *, const T *, LeafSphere<T> *, uint32_t, hipStream_t);
template <typename T>
hipError_t launch_gather_leaf_cylinders(const uint32_t *, const T *, const T *, LeafCylinder<T> *, uint32_t, hipStream_t);
hipError_t launch_cylinder_post(const nrt_ray_f32 *, const nrt_hit_f32 *, const uint8_t *, const float *, uint32_t, void *,
                                uint8_t *, DoneRec *, DoneCount *, uint32_t, hipStream_t);
template <typename T>
hipError_t launch_make_wide(const typename Wire<T>::Node *, uint32_t, uint32_t packed, uint32_t *scratch, WideNode<T> *, Wide4Node<T> *,
                            uint32_t scramble_mod, hipStream_t);
template <typename T>
hipError_t launch_gather_leaf_tris(const uint32_t *, const uint32_t *, const T *, LeafTri<T> *,
                                   uint32_t, hipStream_t);
struct BuildResult {
  uint64_t num_nodes;
  uint32_t max_depth, num_leaves, num_branches, max_leaf_count;
};
// (build.hip) enqueues a whole build; its size and statistics arrive in `pinned` behind `ev`: gpu_build_result waits for them
template <typename T>
hipError_t gpu_build(hipStream_t s, const T *d_verts, const uint32_t *d_faces, const T *d_radii, bool cylinders, const uint32_t *d_prim_map, uint32_t num_faces,
                     uint32_t min_leaf, uint32_t max_depth, uint32_t bin_size, unsigned build_flags, // (bit 0: Morton pre-pass, bit 1: one-node-per-step subtree kernel)
                     DevBuf *workspace, DevBuf *nodes_buf, DevBuf *indices_buf, void *pinned, hipEvent_t ev, std::string *err);
hipError_t gpu_build_result(const void *pinned, hipEvent_t ev, BuildResult *res);
hipError_t launch_cylinder_segments(const float *verts, const float *radii, const uint32_t *seg_off, uint32_t n, float *seg_verts,
                                    float *seg_radii, uint32_t *seg_prim, hipStream_t s);
} // namespace nrt

using namespace nrt;

static thread_local std::string g_create_error;

struct nrt_ctx {
  int device = 0;
  hipStream_t stream = nullptr; // internal stream for host entry points / build
  std::string err;
  int prec = 0; // 0 unset, 4 = f32, 8 = f64
  int num_cus = 256;

  // mesh (tight xyz in HBM)
  int prim_kind = kPrimTriangles; // kPrimSpheres: d_verts = centres, d_radii = radii, no faces; kPrimCylinders: d_verts = 2 end points, d_radii = 2 radii per primitive
  uint32_t cyl_test_cap = 1;
  // cylinders cut into segments for the builder (build.hip k_cylinder_segments): what the tree is built over when num_segs != 0
  DevBuf b_seg_verts, b_seg_radii, b_seg_prim, b_seg_off;
  uint32_t num_segs = 0;
  int cyl_split = 32;     // most segments a cylinder is cut into (tunable "cyl_split"; 1: never; read by nrtSetCylinders)
  int cyl_seg_radii = 8;  // ... one per this many tube radii of its length (tunable "cyl_seg_radii")
  DevBuf b_verts, b_radii, b_faces; // grow-only: a per-frame SetMesh allocates nothing in the steady state
  void *d_verts = nullptr;          // == b_verts.p while primitives are set
  void *d_radii = nullptr;
  uint32_t *d_faces = nullptr;
  uint32_t num_faces = 0, num_verts = 0;

  // tree (grow-only buffers: a per-frame rebuild allocates nothing in the steady state)
  DevBuf b_nodes, b_indices, b_tris, b_wide, b_wide4, b_wide_scratch, b_build_ws;
  void *d_nodes = nullptr;       // == b_nodes.p while a tree is present
  uint32_t *d_indices = nullptr; // == b_indices.p
  void *d_tris = nullptr;        // LeafTri<T>[num_indices] == b_tris.p
  void *d_wide = nullptr;        // WideNode<T>[branches]   == b_wide.p
  void *d_wide4 = nullptr;       // Wide4Node<T>[branches]  == b_wide4.p (triangle trees only, and only when wide4 is on)
  uint32_t num_branch_records = 0; // nodes with flag == 0 in the node array (reachable or not)
  uint64_t num_nodes = 0, num_indices = 0;
  uint32_t tree_depth = 0;
  uint32_t max_leaf_count = 0, min_leaf_count = 0; // over the leaves of the current tree
  uint32_t packed_leaves = 0;
  uint32_t root_is_branch = 0; // node 0 has flag == 0
  uint32_t tree_nested = 1;    // every child box lies inside its parent's (always true of trees built here; checked for adopted ones)
  nrt_build_stats stats = {0, 0, 0, 0.f};
  uint64_t generation = 0; // bumped whenever the tree or the primitives are replaced (free_tree)

  // traversal scratch.  Every launch owns one LaunchSlot (work cursors + overflow stacks) until it
  // completes, so launches issued on different streams may overlap on the GPU: the drain tail of one
  // batch is filled by the start of the next.  Launches on one stream keep reusing one slot.
  struct LaunchSlot {
    uint32_t *d_cursor = nullptr; // two sets of ray cursors (one per partition, 4 KiB apart): a launch uses one and zeroes the other
    unsigned parity = 0;
    DevBuf spill, spill_tmin;
    DevBuf cyl_hits, cyl_bits;    // cylinder kind: compact records + {hit, cap} bits between the traversal and its post pass
    hipEvent_t done = nullptr;    // recorded after the slot's last launch (launches that do not publish a completion record)
    hipEvent_t t0 = nullptr, t1 = nullptr; // event timing of the slot's last launch (per slot: launches on different streams overlap)
    hipStream_t stream = nullptr; // stream of that launch
    bool used = false;
    // completion record (common.h, DoneRec): page-locked, written by the last wave of a launch — no event in the stream
    DoneRec *h_done = nullptr, *d_done = nullptr; // host / device address of the same record
    DoneCount *d_count = nullptr;                 // the device words that go with it
    uint32_t seq = 0;             // sequence number of the slot's last launch that publishes a record
    bool rec_pending = false;     // the slot's last launch publishes a record and nobody has seen it complete yet
    bool last_has_rec = false;    // the slot's last launch publishes a record (nrtLastTraverseMs reads its stamps)
    bool last_timed = false;      // ... or was bracketed by the t0 / t1 events
  };
  static constexpr int kSlots = 4;
  LaunchSlot slots[kSlots];
  unsigned next_victim = 0;
  std::mutex launch_mutex; // slot selection + launch (nrtTraverseBatchDevice may be called from several host threads)
  std::mutex host_mutex;   // the host-buffer traversal calls share one set of staging buffers: one at a time
  unsigned long long *d_counters = nullptr;  // 16 x u64 (counting pass / profiling instantiation only)
  DevBuf st_rays, st_hits, st_mask;
  // host entry point with page-locked caller buffers: upload / trace / download pipelined over three streams
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_tr[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  int host_pipeline = 1; // (env NRT_HOST_PIPELINE=0: one upload, one launch, one download)
  DevBuf b_wave_clock; // profiling (NRT_DEBUG bit 8192)
  uint32_t wave_clock_waves = 0;

  // traversal tuning (env NRT_LDS_STACK / NRT_REFILL_MIN / NRT_TRAV_MIN / NRT_CHUNK override)
  int lds_stack = kLdsStackDefault;
  unsigned blocks_per_cu = 0, chunk = 128, chunk_tail_pct = 0, refill_min = 44, trav_min = 16, leaf_min = 32; // (trav_min: 8 until round 3; 12-14 was the optimum of the two-level walk before its inner loop ran two rounds per trip, profiles/r03t_trav_min.txt; 16 with two rounds per trip in the fp64 walk, profiles/r03ZA)
  int f64_row_fetch = 1; // fp64 walk: fetch the plane rows of a WideNode<double> by the ray's signs (0: fetch the record and select; the path arrays of 4 GiB and more take)
  unsigned trav_min4 = 24; // the same threshold for the fp32 two-level walk, whose inner loop runs two pop + step rounds per trip (profiles/r03Z_threshold_resweep*.txt)
  unsigned num_parts = 8; // ray partitions == XCDs (env NRT_PARTS)
  unsigned debug_flags = 0;
  int subtree_rows = 1; // builder: subtree phase in row form (up to four nodes per step); 0 (profiling build only): one node per step — same tree, the cross-check (tests/test_gpu_build.py)
  int morton = 0; // Morton-order the primitive records before the build (env NRT_MORTON=1): measured +0.4 ms at 1M tris for an identical tree, so off by default
  // share of a batch handed out statically, percent (tunable static_pct).  75 until the end of round 6; 16 since: with ~400 rays per
  // wave (a 1080p wave on this grid) a wave's static share was 256 of them, and whenever the cost per ray is uneven over the image
  // (C2: 40 % sky) the waves with cheap slices drained the dynamic rest while the others were still on rays nobody could take from
  // them.  One 64-ray group per wave there (it starts every wave without an atomic) and the rest claimed in chunks:
  // C3 +2.8 %, C2 +7 %, C4 tile +2.9 %, C5 +0.9 %, a 2560x1440 view of C2's mesh +17 %; balanced frames of other sizes -3.3 ... +1 %
  // (profiles/r06y_distribution_*.txt)
  unsigned static_pct = 16;
  unsigned static_bands = 8; // ... as up to this many slices per wave, one in each band of the batch
  unsigned static_slice_groups = 2; // ... each of at least this many 64-ray groups
  unsigned max_blocks_per_cu = 0; // env NRT_BLOCKS_PER_CU caps the persistent grid
  int wide = 1, wide_stack = 10; // production path: WideNode kernel (env NRT_WIDE=0 selects the binary kernel)
  // Two tree levels per step (Wide4Node records, traverse.hip NRT_STEP_NODE4): the production walk of fp32 triangle trees
  // whose child boxes lie inside their parents'.  env NRT_WIDE4=0 goes back to one level per step.
  int wide4 = 1;
  int wide4_big_ok = 1; // ... also when the record array reaches 4 GiB (tunable wide4_big; 0: such trees walk one level per step, as until round 6)
  // two-level walk, order of a record's four slots.  0 (default since round 5): the binary loop's order — the same leaves in the
  // same order as nanort.h:2526-2548, every field of every record bit-identical to the reference on the same node array.
  // 1 (opt-in): by entry distance, +2...5 % — the closest t is the reference's except where a leaf box's entry distance rounds
  // above a hit inside it, and among primitives at exactly the same t another one may be named (contract-level parity)
  int order4 = 0;
  int leaf_compact = 1; // traverse.hip "leaf items" (round 6): when the records of all the lanes waiting at a leaf fit one trip of the wave, they are tested one per lane with the owner's ray constants and accepted by the owner in record order — records bit-identical, C3 +2 %, C4 tile +2.8 %; 0: every owner tests its own records
  int dyn_head = 1; // batches too small for a static group per wave: every wave's first chunk is its own (traverse.hip claim_init; tunable dyn_head)
  int wide_scramble = 0; // probe (tunable wide_scramble): the private node records in a pseudo-random order instead of pre-order
  unsigned wide4_blocks_per_cu = 0;
  unsigned wide_blocks_per_cu = 0, sphere_blocks_per_cu = 0;

  hipEvent_t ev_b0 = nullptr, ev_b1 = nullptr;
  hipEvent_t ev_build_state = nullptr; // the builder's state block has reached build_state
  void *build_state = nullptr;         // page-locked, kBuildPinnedBytes
  int last_timed_slot = -1; // slot of the most recent timed traversal launch (nrtLastTraverseMs)
  int launch_timing = 0;    // 1: bracket every traversal launch with timing events and follow it with a completion event (nrtSetLaunchTiming); 0: completion records
  bool have_build_time = false;
  float segment_ms = 0.f; // cylinders: wall time nrtSetCylinders spent cutting them into segments (counted into the next build's time)
  const char *last_kernel = ""; // variant of the most recent traversal launch (nrtLastKernelName)
};

static nrt_status fail(nrt_ctx *c, nrt_status st, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return st;
}

#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), NRT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

static nrt_status ensure(nrt_ctx *c, DevBuf &b, size_t bytes) {
  HIPCHK(c, devbuf_ensure(&b, bytes));
  return NRT_OK;
}

// Every call that rewrites the tree or the primitive buffers in place (nrtSetMesh / nrtSetSpheres / nrtSetCylinders /
// nrtSetTree / nrtBuild) first waits for the traversal launches still in flight on the CALLER's streams: they read
// b_nodes / b_tris / b_wide, which the rebuild reuses without reallocating (a per-frame "trace asynchronously on my
// stream, then rebuild" loop is therefore safe without a synchronisation of the caller's own).
// Event records between two kernels of a stream keep the second from starting for several microseconds each (measured:
// the three records per launch — a timing pair and the slot's `done` — held C3 at 24 us of idle time per launch).  So the
// triangle launches of the production kernel record NONE by default: the kernel's last wave publishes a completion record
// (sequence number + start / end stamps) in page-locked memory, and whoever has to wait for the launch — a rebuild, destroy,
// another stream taking the slot over, nrtLastTraverseMs — polls that record: precise, and nothing else on the device is
// waited for.  (Sphere and cylinder launches: their post pass closes the record.  The literal kernel keeps the events.)
static hipError_t wait_record(nrt_ctx::LaunchSlot &sl) {
  if (!sl.rec_pending) return hipSuccess;
  volatile uint32_t *seq = &sl.h_done->seq;
  const auto t_start = std::chrono::steady_clock::now();
  for (unsigned long spins = 0; *seq != sl.seq; spins++) {
    if (spins < 4000) continue; // (a launch lasts a fraction of a millisecond)
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30)) {
      // something is wrong (a faulted launch never writes its record): let the runtime report it
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) return e;
      if (*seq != sl.seq) return hipErrorUnknown;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  sl.rec_pending = false;
  return hipSuccess;
}

// The slot's last launch is over (host-side wait).
static hipError_t slot_done(nrt_ctx *c, nrt_ctx::LaunchSlot &sl) {
  (void)c;
  if (!sl.used) return hipSuccess;
  if (sl.last_has_rec) return wait_record(sl);
  return sl.done ? hipEventSynchronize(sl.done) : hipSuccess;
}

static hipError_t wait_for_launches(nrt_ctx *c) {
  std::lock_guard<std::mutex> lock(c->launch_mutex);
  for (nrt_ctx::LaunchSlot &sl : c->slots) {
    hipError_t e = slot_done(c, sl);
    if (e != hipSuccess) return e;
  }
  return hipStreamSynchronize(c->stream);
}

// Forget the current tree (the buffers stay allocated for the next one).
static void free_tree(nrt_ctx *c) {
  c->generation++; // whoever cached device addresses / flags of the old tree (a committed nrt_scene) can tell
  c->d_wide = nullptr;
  c->d_wide4 = nullptr;
  c->d_nodes = nullptr;
  c->d_indices = nullptr;
  c->d_tris = nullptr;
  c->num_nodes = c->num_indices = 0;
  c->tree_depth = 0;
}

static void free_mesh(nrt_ctx *c) {
  c->d_verts = nullptr; // (the buffers stay allocated for the next primitives)
  c->d_faces = nullptr;
  c->d_radii = nullptr;
  c->num_faces = c->num_verts = 0;
  c->num_segs = 0;
}

// ---------------------------------------------------------------------------
// Tunables (nrtSetTunable / nrtGetTunable; the environment variable NRT_<NAME> overrides the default at nrtCreate).
// ---------------------------------------------------------------------------
struct TunableDesc {
  const char *name;
  long long lo, hi;
  long long (*get)(const nrt_ctx *);
  void (*set)(nrt_ctx *, long long);
};
#define NRT_TUNABLE(name_, lo_, hi_, field_, type_)                                                          \
  {name_, lo_, hi_, [](const nrt_ctx *c) -> long long { return (long long)c->field_; },                      \
   [](nrt_ctx *c, long long v) { c->field_ = (type_)v; }}
static const TunableDesc kTunables[] = {
    NRT_TUNABLE("refill_min", 1, 64, refill_min, unsigned),       // idle lanes of a wave before it claims more rays
    NRT_TUNABLE("trav_min", 1, 64, trav_min, unsigned),           // lanes still walking below which the inner-node phase ends (one level per step)
    NRT_TUNABLE("trav_min4", 1, 64, trav_min4, unsigned),         // ... of the two-level walk
    NRT_TUNABLE("f64_row_fetch", 0, 1, f64_row_fetch, int),       // 0: the fp64 walk reads whole WideNode records and selects the planes (what arrays >= 4 GiB do)
    NRT_TUNABLE("leaf_min", 1, 64, leaf_min, unsigned),           // lanes at a leaf below which a due refill goes first
    NRT_TUNABLE("chunk_tail_pct", 0, 100, chunk_tail_pct, unsigned), // share of the dynamic rays handed out in half chunks (the end of a launch)
    NRT_TUNABLE("parts", 1, kMaxParts, num_parts, unsigned),      // ray partitions (== XCDs)
    NRT_TUNABLE("static_pct", 0, 100, static_pct, unsigned),      // share of a batch owned statically, percent
    NRT_TUNABLE("static_bands", 1, 64, static_bands, unsigned),   // ... in up to this many slices per wave, one per band
    NRT_TUNABLE("static_slice_groups", 1, 64, static_slice_groups, unsigned), // ... of at least this many 64-ray groups each
    NRT_TUNABLE("blocks_per_cu", 0, 8, max_blocks_per_cu, unsigned), // cap on the persistent grid (0: occupancy)
    NRT_TUNABLE("debug", 0, 0x7FFFFFFF, debug_flags, unsigned),   // profiling bit mask
    NRT_TUNABLE("morton", 0, 1, morton, int),                     // Morton pre-pass of the builder (next build)
    NRT_TUNABLE("cyl_split", 1, 64, cyl_split, int),              // cylinders: most segments one is cut into for the builder (1: never; next nrtSetCylinders)
    NRT_TUNABLE("cyl_seg_radii", 1, 1024, cyl_seg_radii, int),    // ... one segment per this many tube radii of length (next nrtSetCylinders)
#ifdef NRT_PROF
    NRT_TUNABLE("subtree_rows", 0, 1, subtree_rows, int),         // 0: the builder's one-node-per-step subtree kernel (next build; same tree) — libnanort_hip_prof.so only
#endif
    NRT_TUNABLE("wide", 0, 1, wide, int),                         // 0: the literal BVHNode loop
    NRT_TUNABLE("wide4_big", 0, 2, wide4_big_ok, int),            // ... also for record arrays of 4 GiB and more (64-bit offsets; next build / set_tree); 2: 64-bit offsets whatever the size (tests)
    NRT_TUNABLE("wide4", 0, 1, wide4, int),                       // two tree levels per step (next build / set_tree)
    NRT_TUNABLE("leaf_compact", 0, 1, leaf_compact, int),         // two-level walk, triangle trees with leaves of <= 4 records: leaf phase over items (records bit-identical)
    NRT_TUNABLE("dyn_head", 0, 1, dyn_head, int),                 // a batch without a static share: a wave's first chunk without an atomic
    NRT_TUNABLE("order4", 0, 1, order4, int),                     // two-level walk: 0 (default) = the reference's order, every field bit-identical; 1 = slots by entry distance (faster; contract-level parity at ties)
    NRT_TUNABLE("launch_timing", 0, 1, launch_timing, int),       // == nrtSetLaunchTiming
    NRT_TUNABLE("host_pipeline", 0, 1, host_pipeline, int),       // pipelined host entry point
    NRT_TUNABLE("wide_scramble", 0, 1, wide_scramble, int),       // probe: WideNode / Wide4Node records in a pseudo-random order (next build)
    {"chunk", 32, 1 << 20, [](const nrt_ctx *c) -> long long { return c->chunk; },
     [](nrt_ctx *c, long long v) { c->chunk = (unsigned)(v / 32) * 32u; }}, // rays claimed per atomic (whole and half chunks are multiples of 16)
    {"lds_stack", 16, 32, [](const nrt_ctx *c) -> long long { return c->lds_stack; },
     [](nrt_ctx *c, long long v) { if (v == 16 || v == 24 || v == 32) c->lds_stack = (int)v; }},
    {"wide_stack", 8, 16, [](const nrt_ctx *c) -> long long { return c->wide_stack; },
     [](nrt_ctx *c, long long v) { if (v == 8 || v == 10 || v == 12 || v == 16) c->wide_stack = (int)v; }},
};
#undef NRT_TUNABLE

static const TunableDesc *tunable_find(const char *name) {
  if (!name) return nullptr;
  for (const TunableDesc &d : kTunables)
    if (!strcmp(d.name, name)) return &d;
  return nullptr;
}
static bool tunable_set(nrt_ctx *c, const char *name, long long v) {
  const TunableDesc *d = tunable_find(name);
  if (!d) return false;
  d->set(c, std::min(d->hi, std::max(d->lo, v)));
  // the occupancy figures depend on the variant the tunables select: recompute on the next launch
  c->blocks_per_cu = c->wide_blocks_per_cu = c->wide4_blocks_per_cu = c->sphere_blocks_per_cu = 0;
  return true;
}

extern "C" {

const char *nrtVersion(void) { return "libnanort_hip 0.1 (gfx950, HIP)"; }

// Page-locked host memory for ray / hit buffers handed to the host entry points: the copies then run at PCIe speed
// (pageable memory is staged by the runtime at roughly half of it) — for callers that do not link HIP themselves.
nrt_status nrtHostAlloc(size_t bytes, void **out) {
  if (!out) return fail(nullptr, NRT_ERR_INVALID, "nrtHostAlloc: out == NULL");
  *out = nullptr;
  if (bytes == 0) return NRT_OK;
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(nullptr, NRT_ERR_DEVICE, "nrtHostAlloc: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return NRT_OK;
}
void nrtHostFree(void *p) {
  if (p) (void)hipHostFree(p);
}

nrt_status nrtCreate(int device, nrt_ctx **out) {
  if (!out) return fail(nullptr, NRT_ERR_INVALID, "nrtCreate: out == NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, NRT_ERR_DEVICE, "nrtCreate: no HIP device visible (%s)",
                e == hipSuccess ? "count == 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return fail(nullptr, NRT_ERR_INVALID, "nrtCreate: device %d out of range [0,%d)", device, ndev);
  nrt_ctx *c = new nrt_ctx();
  c->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&c->ev_b0)) != hipSuccess || (e = hipEventCreate(&c->ev_b1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_build_state, hipEventDisableTiming)) != hipSuccess ||
      (e = hipHostMalloc(&c->build_state, kBuildPinnedBytes, hipHostMallocDefault)) != hipSuccess ||
      (e = hipMalloc((void **)&c->d_counters, 16 * sizeof(unsigned long long))) != hipSuccess) {
    fail(nullptr, NRT_ERR_DEVICE, "nrtCreate: %s", hipGetErrorString(e));
    nrtDestroy(c);
    return NRT_ERR_DEVICE;
  }
  for (nrt_ctx::LaunchSlot &sl : c->slots) {
    if ((e = hipMalloc((void **)&sl.d_cursor, 2 * kCursorStrideWords * 4 * kMaxParts)) != hipSuccess ||
        (e = hipMemset(sl.d_cursor, 0, 2 * kCursorStrideWords * 4 * kMaxParts)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreate(&sl.t0)) != hipSuccess || (e = hipEventCreate(&sl.t1)) != hipSuccess ||
        (e = hipHostMalloc((void **)&sl.h_done, sizeof(DoneRec), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&sl.d_done, sl.h_done, 0)) != hipSuccess ||
        (e = hipMalloc((void **)&sl.d_count, sizeof(DoneCount))) != hipSuccess ||
        (e = hipMemset(sl.d_count, 0, sizeof(DoneCount))) != hipSuccess ||
        (e = hipMemset((char *)sl.d_count + offsetof(DoneCount, t_begin), 0xFF, sizeof(unsigned long long))) != hipSuccess) {
      fail(nullptr, NRT_ERR_DEVICE, "nrtCreate: %s", hipGetErrorString(e));
      nrtDestroy(c);
      return NRT_ERR_DEVICE;
    }
  }
  for (nrt_ctx::LaunchSlot &sl : c->slots) memset(sl.h_done, 0, sizeof(DoneRec));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cus = prop.multiProcessorCount;
  // Environment overrides (debugging aid): NRT_<NAME> for every tunable of nrtSetTunable, applied once at creation — in the
  // profiling library always, in the product library only when the process opts in with NRT_ALLOW_ENV=1: a stray variable in
  // a user's environment must not move the product between parity classes (order4) or walks.
  const bool allow_env = env_overrides_allowed();
  for (const TunableDesc &d : kTunables) {
    if (!allow_env) break;
    char env[64] = "NRT_";
    size_t k = 4;
    for (const char *p = d.name; *p && k + 1 < sizeof(env); p++) env[k++] = (char)toupper((unsigned char)*p);
    env[k] = 0;
    if (const char *e = getenv(env)) (void)tunable_set(c, d.name, atoll(e));
  }
  *out = c;
  return NRT_OK;
}

void nrtDestroy(nrt_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (nrt_ctx::LaunchSlot &sl : c->slots) { // launches still in flight on the caller's streams
    if (sl.h_done || sl.done) (void)slot_done(c, sl);
    if (sl.d_cursor) (void)hipFree(sl.d_cursor);
    if (sl.d_count) (void)hipFree(sl.d_count);
    if (sl.h_done) (void)hipHostFree(sl.h_done);
    if (sl.spill.p) (void)hipFree(sl.spill.p);
    if (sl.spill_tmin.p) (void)hipFree(sl.spill_tmin.p);
    if (sl.cyl_hits.p) (void)hipFree(sl.cyl_hits.p);
    if (sl.cyl_bits.p) (void)hipFree(sl.cyl_bits.p);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.t0) (void)hipEventDestroy(sl.t0);
    if (sl.t1) (void)hipEventDestroy(sl.t1);
  }
  free_tree(c);
  free_mesh(c);
  DevBuf *bufs[] = {&c->b_verts, &c->b_radii, &c->b_faces, &c->st_rays, &c->st_hits, &c->st_mask, &c->b_nodes, &c->b_indices, &c->b_tris, &c->b_wide, &c->b_wide4, &c->b_wide_scratch, &c->b_build_ws, &c->b_wave_clock, &c->b_seg_verts, &c->b_seg_radii, &c->b_seg_prim, &c->b_seg_off};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->build_state) (void)hipHostFree(c->build_state);
  hipEvent_t evs[] = {c->ev_b0, c->ev_b1, c->ev_build_state};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  for (int k = 0; k < 2; k++)
    for (hipEvent_t e : {c->ev_in[k], c->ev_tr[k], c->ev_out[k]})
      if (e) (void)hipEventDestroy(e);
  if (c->copy_in) (void)hipStreamDestroy(c->copy_in);
  if (c->copy_out) (void)hipStreamDestroy(c->copy_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *nrtLastError(const nrt_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

} // extern "C"

// ---------------------------------------------------------------------------
// mesh
// ---------------------------------------------------------------------------
template <typename T>
static nrt_status set_mesh(nrt_ctx *c, const T *vertices, size_t stride, const uint32_t *faces,
                           uint32_t num_faces) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != 0 && c->prec != (int)sizeof(T))
    return fail(c, NRT_ERR_PRECISION, "nrtSetMesh: context already holds a %s mesh",
                c->prec == 4 ? "f32" : "f64");
  if (num_faces && (!vertices || !faces)) return fail(c, NRT_ERR_INVALID, "nrtSetMesh: NULL mesh pointer");
  if (stride < 3 * sizeof(T) && num_faces)
    return fail(c, NRT_ERR_INVALID, "nrtSetMesh: vertex stride %zu < %zu", stride, 3 * sizeof(T));
  NRT_RANGE("nrtSetMesh (compaction + upload)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, wait_for_launches(c));
  free_tree(c);
  free_mesh(c);
  c->prec = (int)sizeof(T);
  c->prim_kind = kPrimTriangles;
  c->num_faces = num_faces;
  if (num_faces == 0) return NRT_OK;
  // The reference's mesh carries no vertex count (nanort.h:925-930): derive it.
  uint32_t maxv = 0;
  const size_t ni = 3 * (size_t)num_faces;
  for (size_t i = 0; i < ni; i++) maxv = faces[i] > maxv ? faces[i] : maxv;
  const uint32_t nv = maxv + 1;
  c->num_verts = nv;
  // Compact the strided vertex array (get_vertex_addr, nanort.h:467-472) to tight xyz.
  const T *src = vertices;
  std::vector<T> tight;
  if (stride != 3 * sizeof(T)) {
    tight.resize(3 * (size_t)nv);
    const unsigned char *base = reinterpret_cast<const unsigned char *>(vertices);
    for (uint32_t v = 0; v < nv; v++) {
      const T *p = reinterpret_cast<const T *>(base + (size_t)v * stride);
      tight[3 * (size_t)v + 0] = p[0];
      tight[3 * (size_t)v + 1] = p[1];
      tight[3 * (size_t)v + 2] = p[2];
    }
    src = tight.data();
  }
  nrt_status st;
  if ((st = ensure(c, c->b_verts, 3 * (size_t)nv * sizeof(T))) || (st = ensure(c, c->b_faces, ni * sizeof(uint32_t)))) return st;
  c->d_verts = c->b_verts.p;
  c->d_faces = (uint32_t *)c->b_faces.p;
  HIPCHK(c, hipMemcpy(c->d_verts, src, 3 * (size_t)nv * sizeof(T), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_faces, faces, ni * sizeof(uint32_t), hipMemcpyHostToDevice));
  return NRT_OK;
}

// Sphere primitives (SphereGeometry of examples/particle_primitive/main.cc:113-147): xyz centres, one radius each.
template <typename T>
static nrt_status set_spheres(nrt_ctx *c, const T *centers, const T *radii, uint32_t n) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != 0 && c->prec != (int)sizeof(T))
    return fail(c, NRT_ERR_PRECISION, "nrtSetSpheres: context already holds %s primitives", c->prec == 4 ? "f32" : "f64");
  if (n && (!centers || !radii)) return fail(c, NRT_ERR_INVALID, "nrtSetSpheres: NULL pointer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, wait_for_launches(c));
  free_tree(c);
  free_mesh(c);
  c->prec = (int)sizeof(T);
  c->prim_kind = kPrimSpheres;
  c->num_faces = n;
  c->num_verts = n;
  if (n == 0) return NRT_OK;
  nrt_status st;
  if ((st = ensure(c, c->b_verts, 3 * (size_t)n * sizeof(T))) || (st = ensure(c, c->b_radii, (size_t)n * sizeof(T)))) return st;
  c->d_verts = c->b_verts.p;
  c->d_radii = c->b_radii.p;
  HIPCHK(c, hipMemcpy(c->d_verts, centers, 3 * (size_t)n * sizeof(T), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_radii, radii, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
  return NRT_OK;
}

// Cylinder primitives (CylinderGeometry of examples/cylinder_primitive/main.cc:124-210): two end points and two radii each.
static nrt_status set_cylinders(nrt_ctx *c, const float *endpoints, const float *radii, uint32_t n, int test_cap) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != 0 && c->prec != 4) return fail(c, NRT_ERR_PRECISION, "nrtSetCylinders: context already holds f64 primitives");
  if (n && (!endpoints || !radii)) return fail(c, NRT_ERR_INVALID, "nrtSetCylinders: NULL pointer");
  if (n >= 0x40000000u) return fail(c, NRT_ERR_INVALID, "nrtSetCylinders: too many cylinders");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, wait_for_launches(c));
  free_tree(c);
  free_mesh(c);
  c->prec = 4;
  c->prim_kind = kPrimCylinders;
  c->cyl_test_cap = test_cap ? 1u : 0u;
  c->num_faces = n;
  c->num_verts = 2 * n;
  if (n == 0) return NRT_OK;
  nrt_status st;
  if ((st = ensure(c, c->b_verts, 6 * (size_t)n * sizeof(float))) || (st = ensure(c, c->b_radii, 2 * (size_t)n * sizeof(float)))) return st;
  c->d_verts = c->b_verts.p;
  c->d_radii = c->b_radii.p;
  HIPCHK(c, hipMemcpy(c->d_verts, endpoints, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_radii, radii, 2 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  // Segments for the builder (build.hip, k_cylinder_segments): a cylinder many radii long is handed over as several pieces
  // with their own tight boxes and the cylinder's id.  Counts on the host (the arrays are here anyway), pieces on the device.
  c->num_segs = 0;
  c->segment_ms = 0.f;
  // (zero-radius "cylinders" are boxes — the top-level tree of a scene is built over them — and cyl_split = 1 asks for the
  // example's own boxes: neither needs the counting pass or its array)
  bool any_radius = false;
  for (size_t i = 0; i < 2 * (size_t)n && !any_radius; i++) any_radius = radii[i] > 0.0f;
  if (c->cyl_split > 1 && any_radius) {
    const auto seg_t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> off((size_t)n + 1);
    uint64_t total = 0;
    for (int pass = 0; pass < 2 && total == 0; pass++) {
      // (second pass: the segment array would not fit the packed leaf references — fall back to half as many pieces at most)
      const uint32_t kmax = (uint32_t)c->cyl_split >> pass;
      uint64_t t = 0;
      for (uint32_t i = 0; i < n; i++) {
        const float *p0 = endpoints + 6 * (size_t)i, *p1 = p0 + 3;
        const float r0 = radii[2 * (size_t)i], r1 = radii[2 * (size_t)i + 1];
        const float rr = r0 > r1 ? r0 : r1;
        const double dx = (double)p1[0] - p0[0], dy = (double)p1[1] - p0[1], dz = (double)p1[2] - p0[2];
        const double len = sqrt(dx * dx + dy * dy + dz * dz);
        uint32_t k = 1;
        if (kmax > 1 && rr > 0.0f && std::isfinite(len) && std::isfinite(rr) && len > 0.0) { // (zero-radius "cylinders" are boxes: the top-level tree of a scene)
          const double want = ceil(len / ((double)c->cyl_seg_radii * (double)rr));
          k = want >= (double)kmax ? kmax : (want < 1.0 ? 1u : (uint32_t)want);
        }
        off[i] = (uint32_t)t;
        t += k;
      }
      off[n] = (uint32_t)t;
      if (t < (uint64_t)kPackedFirstMask) total = t;
    }
    if (total > (uint64_t)n) {
      if ((st = ensure(c, c->b_seg_off, ((size_t)n + 1) * sizeof(uint32_t))) || (st = ensure(c, c->b_seg_verts, 6 * (size_t)total * sizeof(float))) ||
          (st = ensure(c, c->b_seg_radii, 2 * (size_t)total * sizeof(float))) || (st = ensure(c, c->b_seg_prim, (size_t)total * sizeof(uint32_t))))
        return st;
      HIPCHK(c, hipMemcpyAsync(c->b_seg_off.p, off.data(), ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, launch_cylinder_segments((const float *)c->d_verts, (const float *)c->d_radii, (const uint32_t *)c->b_seg_off.p, n,
                                         (float *)c->b_seg_verts.p, (float *)c->b_seg_radii.p, (uint32_t *)c->b_seg_prim.p, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream)); // (`off` is pageable host memory)
      c->num_segs = (uint32_t)total;
    }
    // the segmentation is part of building this tree: its wall time is added to the build's (nrtLastBuildMs, build_secs)
    c->segment_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - seg_t0).count();
  }
  return NRT_OK;
}

// ---------------------------------------------------------------------------
// tree adoption / retrieval
// ---------------------------------------------------------------------------
// Private traversal layout of a tree, in two steps so that a build can enqueue the first while the tree's size is still
// on its way to the host: (1) leaf-ordered primitive records (needs the index array only), (2) WideNode / Wide4Node arrays.
template <typename T>
static nrt_status finish_leaf_records(nrt_ctx *c) {
  nrt_status st;
  // leaf-ordered primitive records for the traversal kernel
  if (c->prim_kind == kPrimSpheres) {
    if ((st = ensure(c, c->b_tris, std::max<size_t>(1, c->num_indices) * sizeof(LeafSphere<T>)))) return st;
    c->d_tris = c->b_tris.p;
    HIPCHK(c, launch_gather_leaf_spheres<T>(c->d_indices, (const T *)c->d_verts, (const T *)c->d_radii,
                                            (LeafSphere<T> *)c->d_tris, (uint32_t)c->num_indices, c->stream));
  } else if (c->prim_kind == kPrimCylinders) {
    if ((st = ensure(c, c->b_tris, std::max<size_t>(1, c->num_indices) * sizeof(LeafCylinder<T>)))) return st;
    c->d_tris = c->b_tris.p;
    HIPCHK(c, launch_gather_leaf_cylinders<T>(c->d_indices, (const T *)c->d_verts, (const T *)c->d_radii,
                                              (LeafCylinder<T> *)c->d_tris, (uint32_t)c->num_indices, c->stream));
  } else {
    if ((st = ensure(c, c->b_tris, std::max<size_t>(1, c->num_indices) * sizeof(LeafTri<T>)))) return st;
    c->d_tris = c->b_tris.p;
    HIPCHK(c, launch_gather_leaf_tris<T>(c->d_indices, c->d_faces, (const T *)c->d_verts,
                                         (LeafTri<T> *)c->d_tris, (uint32_t)c->num_indices, c->stream));
  }
  return NRT_OK;
}

template <typename T>
static nrt_status finish_wide(nrt_ctx *c) {
  nrt_status st;
  // one WideNode per record with flag == 0 (a loaded tree may carry unreachable ones)
  if ((st = ensure(c, c->b_wide, std::max<size_t>(1, c->num_branch_records) * sizeof(WideNode<T>)))) return st;
  c->d_wide = c->b_wide.p;
  c->packed_leaves = (c->min_leaf_count >= 1 && c->max_leaf_count <= kPackedMaxCount &&
                      c->num_indices <= (uint64_t)kPackedFirstMask) ? 1u : 0u;
  const size_t tiles = (c->num_nodes + 1023) / 1024;
  if ((st = ensure(c, c->b_wide_scratch, (tiles + c->num_nodes + 1) * sizeof(uint32_t)))) return st;
  c->d_wide4 = nullptr;
  // (every primitive kind: the step does not look at the leaves.  Below 4 GiB — 2^25 records — the walk addresses the records with
  // 32-bit byte offsets; triangle trees beyond that, ~110 M triangles and more, get the array too and 64-bit offsets)
  if (c->wide4 && sizeof(T) == 4 && (c->num_branch_records < (1ull << 25) || (c->prim_kind == kPrimTriangles && c->wide4_big_ok))) {
    if ((st = ensure(c, c->b_wide4, std::max<size_t>(1, c->num_branch_records) * sizeof(Wide4Node<T>)))) return st;
    c->d_wide4 = c->b_wide4.p;
  }
  HIPCHK(c, launch_make_wide<T>((const typename Wire<T>::Node *)c->d_nodes, (uint32_t)c->num_nodes, c->packed_leaves,
                                (uint32_t *)c->b_wide_scratch.p, (WideNode<T> *)c->d_wide, (Wide4Node<T> *)c->d_wide4,
                                c->wide_scramble ? c->num_branch_records : 0u, c->stream));
  return NRT_OK;
}

template <typename T>
static nrt_status finish_tree(nrt_ctx *c) {
  nrt_status st = finish_leaf_records<T>(c);
  return st ? st : finish_wide<T>(c);
}

template <typename T>
static nrt_status set_tree(nrt_ctx *c, const typename Wire<T>::Node *nodes, uint64_t num_nodes,
                           const uint32_t *indices, uint64_t num_indices) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != (int)sizeof(T)) return fail(c, NRT_ERR_PRECISION, "nrtSetTree: set a mesh of this precision first");
  if (!nodes || !indices || num_nodes == 0) return fail(c, NRT_ERR_INVALID, "nrtSetTree: empty tree");
  if (num_nodes >= 0xFFFFFFFFull || num_indices >= 0xFFFFFFFFull)
    return fail(c, NRT_ERR_INVALID, "nrtSetTree: tree too large");
  // Validate what the traversal loop relies on and measure depth.
  for (uint64_t i = 0; i < num_indices; i++)
    if (indices[i] >= c->num_faces)
      return fail(c, NRT_ERR_INVALID, "nrtSetTree: indices[%llu]=%u >= num_faces %u",
                  (unsigned long long)i, indices[i], c->num_faces);
  uint32_t depth = 0, max_leaf = 0, min_leaf = 0xFFFFFFFFu, nested = 1;
  {
    std::vector<std::pair<uint32_t, uint32_t> > st;
    std::vector<uint8_t> seen(num_nodes, 0);
    st.push_back(std::make_pair(0u, 0u));
    while (!st.empty()) {
      std::pair<uint32_t, uint32_t> e = st.back();
      st.pop_back();
      const typename Wire<T>::Node &n = nodes[e.first];
      if (seen[e.first]) return fail(c, NRT_ERR_INVALID, "nrtSetTree: node %u reachable twice", e.first);
      seen[e.first] = 1;
      depth = std::max(depth, e.second);
      if (n.flag == 0) {
        if (n.data[0] >= num_nodes || n.data[1] >= num_nodes)
          return fail(c, NRT_ERR_INVALID, "nrtSetTree: node %u child out of range", e.first);
        if (n.axis < 0 || n.axis > 2) return fail(c, NRT_ERR_INVALID, "nrtSetTree: node %u axis %d", e.first, n.axis);
        for (int ch = 0; ch < 2; ch++) { // do the child boxes lie inside this one?  (true of every tree a builder emits; a
          const typename Wire<T>::Node &k = nodes[n.data[ch]]; // refit or hand-edited tree may break it: see tree_nested)
          for (int ax = 0; ax < 3; ax++)
            if (!(k.bmin[ax] >= n.bmin[ax] && k.bmax[ax] <= n.bmax[ax])) nested = 0;
        }
        st.push_back(std::make_pair(n.data[1], e.second + 1));
        st.push_back(std::make_pair(n.data[0], e.second + 1));
      } else {
        if ((uint64_t)n.data[1] + n.data[0] > num_indices)
          return fail(c, NRT_ERR_INVALID, "nrtSetTree: leaf %u slots out of range", e.first);
        max_leaf = std::max(max_leaf, n.data[0]);
        min_leaf = std::min(min_leaf, n.data[0]);
      }
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, wait_for_launches(c));
  free_tree(c);
  c->num_nodes = num_nodes;
  c->num_indices = num_indices;
  c->tree_depth = depth;
  c->max_leaf_count = max_leaf;
  c->min_leaf_count = min_leaf;
  c->root_is_branch = nodes[0].flag == 0 ? 1u : 0u;
  c->tree_nested = nested;
  c->num_branch_records = 0;
  for (uint64_t i = 0; i < num_nodes; i++)
    if (nodes[i].flag == 0) c->num_branch_records++;
  nrt_status st;
  if ((st = ensure(c, c->b_nodes, num_nodes * sizeof(typename Wire<T>::Node)))) return st;
  if ((st = ensure(c, c->b_indices, std::max<uint64_t>(1, num_indices) * sizeof(uint32_t)))) return st;
  c->d_nodes = c->b_nodes.p;
  c->d_indices = (uint32_t *)c->b_indices.p;
  HIPCHK(c, hipMemcpy(c->d_nodes, nodes, num_nodes * sizeof(typename Wire<T>::Node), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_indices, indices, num_indices * sizeof(uint32_t), hipMemcpyHostToDevice));
  if ((st = finish_tree<T>(c))) return st;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NRT_OK;
}

template <typename T>
static nrt_status get_tree(nrt_ctx *c, typename Wire<T>::Node *nodes_out, uint32_t *indices_out) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != (int)sizeof(T)) return fail(c, NRT_ERR_PRECISION, "nrtGetTree: precision mismatch");
  if (!c->d_nodes) return fail(c, NRT_ERR_INVALID, "nrtGetTree: no tree");
  NRT_RANGE("nrtGetTree (read-back)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (nodes_out)
    HIPCHK(c, hipMemcpy(nodes_out, c->d_nodes, c->num_nodes * sizeof(typename Wire<T>::Node), hipMemcpyDeviceToHost));
  if (indices_out)
    HIPCHK(c, hipMemcpy(indices_out, c->d_indices, c->num_indices * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return NRT_OK;
}

template <typename T>
static nrt_status get_tree_bounds(nrt_ctx *c, T *bmin, T *bmax) {
  if (!c || !bmin || !bmax) return NRT_ERR_INVALID;
  if (c->prec != (int)sizeof(T)) return fail(c, NRT_ERR_PRECISION, "nrtGetTreeBounds: precision mismatch");
  if (!c->d_nodes) return fail(c, NRT_ERR_INVALID, "nrtGetTreeBounds: no tree");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  T box[6]; // BVHNode<T> starts with bmin[3], bmax[3] (nanort.h:498-550)
  HIPCHK(c, hipMemcpy(box, c->d_nodes, sizeof(box), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; k++) {
    bmin[k] = box[k];
    bmax[k] = box[3 + k];
  }
  return NRT_OK;
}

// ---------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------
template <typename T>
static nrt_status build(nrt_ctx *c, const typename Wire<T>::BuildOptions *opt, nrt_build_stats *stats_out,
                        uint64_t *num_nodes_out) {
  if (!c) return NRT_ERR_INVALID;
  if (c->prec != (int)sizeof(T)) return fail(c, NRT_ERR_PRECISION, "nrtBuild: set a mesh of this precision first");
  if (c->num_faces == 0) return fail(c, NRT_ERR_EMPTY, "nrtBuild: no primitives (reference Build() returns false)");
  uint32_t min_leaf = 4, max_depth = 256, bin_size = 64; // BVHBuildOptions() defaults, nanort.h:574-582
  if (opt) {
    min_leaf = opt->min_leaf_primitives;
    max_depth = opt->max_tree_depth;
    bin_size = opt->bin_size;
  }
  if (bin_size < 2) return fail(c, NRT_ERR_INVALID, "nrtBuild: bin_size must be > 1 (nanort.h:1905)");
  NRT_RANGE("nrtBuild");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, wait_for_launches(c));
  free_tree(c);
  BuildResult res;
  std::string err;
  HIPCHK(c, hipEventRecord(c->ev_b0, c->stream));
  // (cylinder contexts whose cylinders were cut into segments: the tree is built over the segments, each carrying its cylinder's id)
  const bool segs = c->prim_kind == kPrimCylinders && c->num_segs != 0 && sizeof(T) == 4;
  const uint32_t build_n = segs ? c->num_segs : c->num_faces;
  hipError_t e = gpu_build<T>(c->stream, segs ? (const T *)c->b_seg_verts.p : (const T *)c->d_verts, c->d_faces, segs ? (const T *)c->b_seg_radii.p : (const T *)c->d_radii,
                              c->prim_kind == kPrimCylinders, segs ? (const uint32_t *)c->b_seg_prim.p : nullptr, build_n, min_leaf, max_depth,
                              bin_size, (c->morton ? 1u : 0u) | (c->subtree_rows ? 0u : 2u), &c->b_build_ws, &c->b_nodes, &c->b_indices, c->build_state, c->ev_build_state, &err);
  if (e != hipSuccess) return fail(c, NRT_ERR_DEVICE, "nrtBuild: %s (%s)", err.c_str(), hipGetErrorString(e));
  // everything is enqueued; the leaf-ordered primitive records need the index array only, so they are enqueued too before
  // the host waits for the tree's size (the GPU stays busy meanwhile)
  c->d_nodes = c->b_nodes.p;
  c->d_indices = (uint32_t *)c->b_indices.p;
  c->num_indices = build_n;
  NRT_RANGE_PUSH("build: leaf-ordered primitive records");
  nrt_status fst = finish_leaf_records<T>(c);
  NRT_RANGE_POP();
  if (fst) {
    free_tree(c); // (no half-built tree is left behind: a later traversal call then reports "no tree")
    return fst;
  }
  if ((e = gpu_build_result(c->build_state, c->ev_build_state, &res)) != hipSuccess) {
    free_tree(c);
    return fail(c, NRT_ERR_DEVICE, "nrtBuild: %s", hipGetErrorString(e));
  }
  c->num_nodes = res.num_nodes;
  c->tree_depth = res.max_depth;
  c->max_leaf_count = res.max_leaf_count;
  c->num_branch_records = res.num_branches;
  c->root_is_branch = res.num_nodes > 1 ? 1u : 0u;
  c->min_leaf_count = 1; // the GPU builder never emits an empty leaf
  c->tree_nested = 1;    // a branch's box is the exact union of its children's
  NRT_RANGE_PUSH("build: WideNode / Wide4Node records");
  fst = finish_wide<T>(c); // WideNode arrays: part of the build
  NRT_RANGE_POP();
  if (fst) {
    free_tree(c);
    return fst;
  }
  HIPCHK(c, hipEventRecord(c->ev_b1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev_b1));
  c->have_build_time = true;
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev_b0, c->ev_b1));
  c->stats.max_tree_depth = res.max_depth;
  c->stats.num_leaf_nodes = res.num_leaves;
  c->stats.num_branch_nodes = res.num_branches;
  c->stats.build_secs = (ms + (segs ? c->segment_ms : 0.f)) * 1e-3f; // (cylinder trees: + the segmentation nrtSetCylinders did for this build)
  if (stats_out) *stats_out = c->stats;
  if (num_nodes_out) *num_nodes_out = c->num_nodes;
  return NRT_OK;
}

// Library-internal (scene.hip): where a built fp32 context keeps its tree on the device.  Waits for the context's own
// stream, so the arrays are complete; they stay valid until the context is rebuilt or destroyed.
nrt_status nrt_internal_tree_view(nrt_ctx *c, nrt::TreeViewF32 *out) {
  if (!c || !out) return NRT_ERR_INVALID;
  if (c->prec != 4 || !c->d_nodes || !c->d_wide) return fail(c, NRT_ERR_INVALID, "no fp32 tree on the device");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  out->nodes = (const nrt_node_f32 *)c->d_nodes;
  out->indices = c->d_indices;
  out->wide = c->d_wide;
  out->wide4 = c->d_wide4;
  out->prims = c->d_tris;
  out->num_nodes = (uint32_t)c->num_nodes;
  out->num_indices = (uint32_t)c->num_indices;
  out->packed_leaves = c->packed_leaves;
  out->root_is_branch = c->root_is_branch;
  out->tree_nested = c->tree_nested;
  out->prim_kind = (uint32_t)c->prim_kind;
  out->tree_depth = c->tree_depth;
  out->max_leaf_count = c->max_leaf_count;
  out->generation = c->generation;
  return NRT_OK;
}
uint64_t nrt_internal_generation(const nrt_ctx *c) { return c ? c->generation : 0; }
int nrt_internal_device(const nrt_ctx *c) { return c ? c->device : 0; } // group.hip

// ---------------------------------------------------------------------------
// traverse
// ---------------------------------------------------------------------------
static const nrt_trace_options kDefaultTrace = {{0u, 0x7FFFFFFFu}, 0xFFFFFFFFu, 0, {0, 0, 0}}; // nanort.h:617-623

// Several batches for one launch (nrtTraverseBatchesDevice): fp32 triangle contexts on the WideNode kernels.
template <typename T>
struct TraverseBatches {
  uint32_t nb;
  const typename Wire<T>::Ray *rays[kMaxBatches];
  typename Wire<T>::Hit *hits[kMaxBatches];
  uint8_t *mask[kMaxBatches];
  uint64_t count[kMaxBatches];
  uint32_t anyhit; // bit k: batch k is an occlusion query
};

template <typename T>
static nrt_status traverse_device(nrt_ctx *c, const typename Wire<T>::Ray *d_rays, uint64_t n,
                                  const nrt_trace_options *opt, typename Wire<T>::Hit *d_hits, uint8_t *d_mask,
                                  hipStream_t s, bool count, bool timed, void *d_cyl_hits = nullptr, bool any_hit = false,
                                  const TraverseBatches<T> *mb = nullptr) {
  if (c->prec != (int)sizeof(T)) return fail(c, NRT_ERR_PRECISION, "nrtTraverseBatch: precision mismatch");
  if (mb) { // (the caller checked that this context walks batches in one launch; n = all rays)
    d_rays = mb->rays[0];
    d_hits = mb->hits[0];
    d_mask = mb->mask[0];
  }
  if ((c->prim_kind == kPrimCylinders) != (d_cyl_hits != nullptr))
    return fail(c, NRT_ERR_INVALID, "cylinder primitives are traced with nrtTraverseBatchCylinders*_f32 (28-byte records), "
                                    "every other primitive kind with nrtTraverseBatch*");
  if (!c->d_nodes) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatch: no tree (call nrtBuild or nrtSetTree)");
  if (n == 0) return NRT_OK;
  if (!d_rays) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatch: NULL rays");
  if (n > 0x7FFFFFFFull) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatch: more than 2^31-1 rays in one call");
  if (!opt) opt = &kDefaultTrace;
#ifdef NRT_PROF
  const unsigned dbg = c->debug_flags;
#else
  const unsigned dbg = c->debug_flags & ~(32u | 64u | 8192u); // (the counting / clocked instantiations live in libnanort_hip_prof.so)
#endif
  std::lock_guard<std::mutex> lock(c->launch_mutex);
  HIPCHK(c, hipSetDevice(c->device));

  // launch slot: the one this stream used last (stream order already serialises the two launches), else
  // a fresh one, else the oldest — whose previous launch this stream then waits for on the device
  nrt_ctx::LaunchSlot *slot = nullptr;
  for (nrt_ctx::LaunchSlot &sl : c->slots)
    if (sl.used && sl.stream == s) {
      slot = &sl;
      break;
    }
  if (!slot)
    for (nrt_ctx::LaunchSlot &sl : c->slots)
      if (!sl.used) {
        slot = &sl;
        break;
      }
  if (!slot) {
    slot = &c->slots[c->next_victim];
    c->next_victim = (c->next_victim + 1) % nrt_ctx::kSlots;
    if (slot->last_has_rec)
      HIPCHK(c, wait_record(*slot)); // (the host waits: a fifth concurrent stream is the rare case)
    else
      HIPCHK(c, hipStreamWaitEvent(s, slot->done, 0));
  }

  // persistent grid: every block resident (occupancy of the chosen variant)
  const bool spheres = c->prim_kind != kPrimTriangles; // the custom primitives share one launch geometry
  if (spheres && (count || !c->d_wide))
    return fail(c, NRT_ERR_INVALID, "nrtTraverse: custom primitives run on the WideNode kernel only (no counting pass)");
  if (any_hit && (spheres || count || !c->d_wide))
    return fail(c, NRT_ERR_INVALID, "nrtOccludedBatch: occlusion queries run on the triangle WideNode kernel only");
  const bool use_wide = (c->wide || spheres || any_hit) && !count && c->d_wide;
  if (c->blocks_per_cu == 0) c->blocks_per_cu = (unsigned)traverse_blocks_per_cu<T>(c->lds_stack);
  // two levels per step: closest-hit walks of nested fp32 triangle trees, outside the profiling / splitting variants
  // prim ids are < num_faces: nothing can be rejected by these options -> the kernel variant without the id tests
  const bool plain_options = opt->prim_ids_range[0] == 0u && opt->prim_ids_range[1] >= c->num_faces && opt->skip_prim_id >= c->num_faces &&
                             !opt->cull_back_face;
  // (the profiling instantiations of the two-level walk are built for the default trace options only: with options that can
  // reject a primitive a profiled launch walks one level per step, whose profiling variant honours them)
  const bool prof_needs_w2 = (dbg & (32u | 8192u)) && !plain_options && !spheres;
  // (a record array of 4 GiB or more is walked two levels per step by the default walk of triangle trees only: closest hits in the
  // reference's order, outside the profiling instantiations)
  const bool wide4_big = c->num_branch_records >= (1ull << 25) || (c->wide4_big_ok == 2 && c->prim_kind == kPrimTriangles); // (2: forced, for the tests)
  const bool use_wide4 = use_wide && c->d_wide4 && c->wide_stack == 10 && c->tree_nested && c->root_is_branch && !prof_needs_w2 &&
                         (!wide4_big || (!spheres && !c->order4 && !(dbg & (32u | 8192u))));
  if (c->wide_blocks_per_cu == 0) c->wide_blocks_per_cu = (unsigned)traverse_wide_blocks_per_cu<T>(c->wide_stack, kPrimTriangles, false);
  if (use_wide4 && !spheres && c->wide4_blocks_per_cu == 0) c->wide4_blocks_per_cu = (unsigned)traverse_wide_blocks_per_cu<T>(kWide4LdsStack, kPrimTriangles, true);
  if (spheres && c->sphere_blocks_per_cu == 0) c->sphere_blocks_per_cu = (unsigned)traverse_wide_blocks_per_cu<T>(10, c->prim_kind, use_wide4); // (one kind and one walk per context)
  unsigned blocks_per_cu = spheres ? c->sphere_blocks_per_cu : (use_wide4 ? c->wide4_blocks_per_cu : (use_wide ? c->wide_blocks_per_cu : c->blocks_per_cu));
  if (c->max_blocks_per_cu && blocks_per_cu > c->max_blocks_per_cu) blocks_per_cu = c->max_blocks_per_cu;
  const int stack_entries = use_wide4 ? kWide4LdsStack : (spheres ? 10 : (use_wide ? c->wide_stack : c->lds_stack));
  uint64_t need_blocks = (n + kTraverseBlock - 1) / kTraverseBlock;
  unsigned grid = (unsigned)std::min<uint64_t>(need_blocks, (uint64_t)c->num_cus * blocks_per_cu);
  const unsigned parts = std::max(1u, std::min(c->num_parts, grid));
  grid = ((grid + parts - 1) / parts) * parts; // whole blocks per partition (ranks are partition-major)
  const uint32_t total_threads = grid * kTraverseBlock;
  const uint32_t total_waves = grid * (kTraverseBlock / kWave);
  // Work distribution (traverse.hip, Claim).  Static share: c->static_pct percent of the batch, in whole 64-ray groups per
  // wave, cut into up to `static_bands` slices — one at the head of each of as many equal bands of the batch; the rest of
  // each band (a whole number of chunks) and the tail behind the last band are claimed dynamically.
  // (less than one group per wave: none — a batch of fewer than ~400 rays per wave is claimed in chunks from its first ray; a forced
  // group per wave measured -2.7 % on a 1600x960 wave, profiles/r06y_distribution10.txt)
  const uint32_t static_share = (uint32_t)(((uint64_t)n * c->static_pct / 100) / total_waves / 64); // 64-ray groups per wave
  // (a slice shorter than two 64-ray groups makes the waves of an XCD drift apart over the bands within one refill, and its
  // L2 then holds several strips of the scene at once: measured on C3, 64-ray slices in 4 bands cost 3 %; two bands of 128
  // cost nothing and still take 8 % off C2, whose sky rows otherwise leave one XCD with the whole sphere: profiles/r03d_*)
  const uint32_t bands = std::max<uint32_t>(1u, std::min<uint32_t>(c->static_bands, static_share / c->static_slice_groups));
  const uint32_t static_per_wave = (static_share / bands) * 64u;
  const uint32_t band_static = static_per_wave * total_waves;
  const uint32_t dyn_per_band = static_per_wave ? (uint32_t)(((uint64_t)n / bands - band_static) / c->chunk) * c->chunk : 0u;
  const uint32_t band_len = band_static + dyn_per_band;
  // deepest possible stack: one pending sibling per level of the path — three per TWO levels when a step covers two
  const uint32_t max_entries = use_wide4 ? 3u * (c->tree_depth / 2u + 1u) + 2u : c->tree_depth + 2u;
  const uint32_t levels = max_entries > (uint32_t)stack_entries ? max_entries - stack_entries : 0;
  if (levels) { // (growing a buffer frees the old one, which waits for every launch in flight)
    nrt_status st = ensure(c, slot->spill, (size_t)levels * total_threads * sizeof(uint32_t));
    if (st) return st;
    if (use_wide) {
      st = ensure(c, slot->spill_tmin, (size_t)levels * total_threads * sizeof(T));
      if (st) return st;
    }
  }

  TraverseArgs<T> a;
  a.nodes = (const typename Wire<T>::Node *)c->d_nodes;
  a.tris = (const LeafTri<T> *)c->d_tris;
  a.spheres = (const LeafSphere<T> *)c->d_tris;
  a.centers = (const T *)c->d_verts;
  a.cylinders = (const LeafCylinder<T> *)c->d_tris;
  a.cyl_test_cap = c->cyl_test_cap;
  a.wide = (const WideNode<T> *)c->d_wide;
  a.wide4 = use_wide4 ? (const Wide4Node<T> *)c->d_wide4 : nullptr;
  a.packed_leaves = c->packed_leaves;
  a.wide4_big = (use_wide4 && wide4_big) ? 1u : 0u;
  a.wide_below_4g = (c->f64_row_fetch && (uint64_t)c->num_branch_records * sizeof(WideNode<T>) < (1ull << 32)) ? 1u : 0u;
  a.root_is_branch = c->root_is_branch;
  a.debug_flags = dbg;
  a.spill_tmin = (T *)slot->spill_tmin.p;
  a.rays = d_rays;
  a.hits = d_hits;
  a.mask = d_mask;
  if (d_cyl_hits) { // compact records + {hit, cap} bits, expanded by the post pass below
    nrt_status st = ensure(c, slot->cyl_hits, (size_t)n * sizeof(typename Wire<T>::Hit));
    if (st) return st;
    if ((st = ensure(c, slot->cyl_bits, (size_t)n))) return st;
    a.hits = (typename Wire<T>::Hit *)slot->cyl_hits.p;
    a.mask = (uint8_t *)slot->cyl_bits.p;
  }
  a.num_rays = (uint32_t)n;
  a.num_batches = 1u;
  a.batch_anyhit = 0u;
  for (int k = 0; k < kMaxBatches; k++) {
    a.batch_end[k] = (uint32_t)n;
    a.batches[k] = BatchPtrs{nullptr, nullptr, nullptr, 0};
  }
  if (mb) {
    if (!use_wide || sizeof(T) != 4 || spheres || count) return fail(c, NRT_ERR_INVALID, "internal: multi-batch launch on a context that cannot take it");
    a.num_batches = mb->nb;
    a.batch_anyhit = mb->anyhit;
    uint64_t start = 0;
    for (uint32_t k = 0; k < mb->nb; k++) { // pointers addressed by the virtual index: base - start
      a.batches[k].rays_v = mb->rays[k] - start;
      a.batches[k].hits_v = mb->hits[k] ? mb->hits[k] - start : nullptr;
      a.batches[k].mask_v = mb->mask[k] ? mb->mask[k] - start : nullptr;
      start += mb->count[k];
      a.batch_end[k] = (uint32_t)start;
    }
    for (uint32_t k = mb->nb; k < (uint32_t)kMaxBatches; k++) a.batch_end[k] = (uint32_t)start;
  }
  a.range0 = opt->prim_ids_range[0];
  a.range1 = opt->prim_ids_range[1];
  a.skip_prim = opt->skip_prim_id;
  a.cull_back_face = opt->cull_back_face ? 1u : 0u;
  a.any_hit = any_hit ? 1u : 0u;
  a.plain_options = plain_options ? 1u : 0u;
  a.root_test = c->tree_nested ? 0u : 1u;
  a.order4 = (c->order4 && use_wide4 && !spheres && !any_hit && !(dbg & (32u | 8192u))) ? 1u : 0u;
  a.leaf_items = (c->leaf_compact && use_wide4 && c->prim_kind == kPrimTriangles && c->max_leaf_count <= 4u && !(dbg & (32u | 8192u))) ? 1u : 0u;
  a.spill = (uint32_t *)slot->spill.p;
  a.spill_stride = total_threads;
  a.spill_levels = levels;
  a.ray_cursor = slot->d_cursor + (size_t)slot->parity * kCursorStrideWords * kMaxParts;
  a.next_cursor = slot->d_cursor + (size_t)(slot->parity ^ 1u) * kCursorStrideWords * kMaxParts;
  a.num_parts = parts;
  a.static_per_wave = static_per_wave;
  a.static_bands = static_per_wave ? bands : 0u;
  a.band_len = band_len;
  a.band_static = band_static;
  a.dyn_per_band = dyn_per_band;
  a.dyn_banded = a.static_bands * dyn_per_band;
  a.tail_begin = a.static_bands * band_len;
  a.dyn_total = a.dyn_banded + ((uint32_t)n - a.tail_begin);
  a.dyn_per_part = (a.dyn_total / parts / c->chunk) * c->chunk;
  a.blocks_per_part = grid / parts;
  a.dyn_head = c->dyn_head ? 1u : 0u;
  a.counters = c->d_counters;
  a.wave_clock = nullptr;
#ifdef NRT_PROF
  if (dbg & 8192u) { // profiling: per-wave time stamps of this launch (nrtDebugWaveClocks)
    nrt_status st = ensure(c, c->b_wave_clock, (size_t)total_waves * 3 * sizeof(unsigned long long));
    if (st) return st;
    a.wave_clock = (unsigned long long *)c->b_wave_clock.p;
    c->wave_clock_waves = total_waves;
  }
#endif
  a.chunk = c->chunk;
  a.chunk_tail_pct = c->chunk_tail_pct;
  a.refill_min = c->refill_min;
  a.trav_min = use_wide4 ? c->trav_min4 : c->trav_min;
  a.leaf_min = c->leaf_min;

  if (count || (dbg & 32u)) HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), s));
  // completion record instead of events: the traversal kernel is the launch's last kernel and events were not asked for
  const bool use_rec = use_wide && !count && !c->launch_timing;
  // (the sphere kind's u/v pass and the cylinder kind's normal pass run behind the traversal kernel and close the record in its place)
  const bool post_pass = (c->prim_kind == kPrimSpheres && d_hits != nullptr) || d_cyl_hits != nullptr;
  a.done_rec = use_rec ? slot->d_done : nullptr;
  a.done_count = slot->d_count;
  a.done_seq = use_rec ? slot->seq + 1u : 0u;
  a.done_publish = post_pass ? 0u : 1u;
  timed = timed && !use_rec;
  if (timed) HIPCHK(c, hipEventRecord(slot->t0, s));
  if (use_wide) {
    HIPCHK(c, launch_traverse_wide<T>(a, grid, c->wide_stack, c->prim_kind, s, &c->last_kernel));
  } else {
    HIPCHK(c, launch_traverse<T>(a, grid, count, c->lds_stack, s));
    c->last_kernel = sizeof(T) == 4 ? "nrt::k_traverse<float>" : "nrt::k_traverse<double>";
  }
  // The kernel is enqueued: the slot's state follows it NOW, before any later call of this function can fail — a launch
  // that is in flight publishes seq + 1 and flips the cursor sets whatever happens to the calls behind it.
  if (use_rec) {
    slot->seq++;
    slot->rec_pending = true;
  }
  slot->last_has_rec = use_rec;
  slot->last_timed = false;
  slot->parity ^= 1u;
  slot->stream = s;
  slot->used = true;
  if (use_rec) c->last_timed_slot = (int)(slot - c->slots);
  if (d_cyl_hits)
    HIPCHK(c, launch_cylinder_post((const nrt_ray_f32 *)d_rays, (const nrt_hit_f32 *)slot->cyl_hits.p, (const uint8_t *)slot->cyl_bits.p,
                                   (const float *)c->d_verts, (uint32_t)n, d_cyl_hits, d_mask, a.done_rec, a.done_count, a.done_seq, s));
  if (timed) {
    HIPCHK(c, hipEventRecord(slot->t1, s));
    slot->last_timed = true;
    c->last_timed_slot = (int)(slot - c->slots);
  }
  if (!use_rec) HIPCHK(c, hipEventRecord(slot->done, s));
  return NRT_OK;
}

// Is `p` page-locked host memory the device can copy from / to asynchronously (hipHostMalloc / nrtHostAlloc / registered)?
static bool is_pinned_host(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError(); // (plain malloc'd memory: "invalid value", not an error of ours)
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Host entry point.  With pageable caller buffers the copies are staged by the runtime and block the host: upload ->
// trace -> download, one after the other.  With PAGE-LOCKED buffers (nrtHostAlloc) the batch is cut into pieces and
// pipelined over three streams — piece k+1 uploads while piece k is traced and piece k-1 downloads (PCIe is full duplex) —
// so the call approaches the slower of the two copy directions instead of their sum plus the kernel.
template <typename T>
static nrt_status traverse_host(nrt_ctx *c, const typename Wire<T>::Ray *rays, uint64_t n,
                                const nrt_trace_options *opt, typename Wire<T>::Hit *hits, uint8_t *mask) {
  if (!c) return NRT_ERR_INVALID;
  if (n == 0) return NRT_OK;
  if (!rays || !hits) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatch: NULL rays/hits");
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;
  std::lock_guard<std::mutex> host_lock(c->host_mutex);
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t kPiece = 1ull << 19; // rays per pipeline stage (19 MB up, 8 MB down in fp32)
  if (n >= 2 * kPiece && c->host_pipeline && is_pinned_host(rays) && is_pinned_host(hits) && (!mask || is_pinned_host(mask))) {
    if (!c->copy_in) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
    if (!c->copy_out) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++)
      for (hipEvent_t *e : {&c->ev_in[k], &c->ev_tr[k], &c->ev_out[k]})
        if (!*e) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    nrt_status st;
    if ((st = ensure(c, c->st_rays, 2 * kPiece * sizeof(Ray))) || (st = ensure(c, c->st_hits, 2 * kPiece * sizeof(Hit))) ||
        (st = ensure(c, c->st_mask, 2 * kPiece)))
      return st;
    uint64_t piece = 0;
    for (uint64_t off = 0; off < n; off += kPiece, piece++) {
      const uint64_t m = std::min(kPiece, n - off);
      const int b = (int)(piece & 1);
      Ray *d_r = (Ray *)c->st_rays.p + (size_t)b * kPiece;
      Hit *d_h = (Hit *)c->st_hits.p + (size_t)b * kPiece;
      uint8_t *d_m = (uint8_t *)c->st_mask.p + (size_t)b * kPiece;
      // the ray slot is free once the trace that read it (two pieces ago) is done; the record slot once its download is
      if (piece >= 2) HIPCHK(c, hipStreamWaitEvent(c->copy_in, c->ev_tr[b], 0));
      HIPCHK(c, hipMemcpyAsync(d_r, rays + off, m * sizeof(Ray), hipMemcpyHostToDevice, c->copy_in));
      HIPCHK(c, hipEventRecord(c->ev_in[b], c->copy_in));
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in[b], 0));
      if (piece >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_out[b], 0));
      st = traverse_device<T>(c, d_r, m, opt, d_h, d_m, c->stream, false, true);
      if (st) { // (copies into the caller's buffers may still be in flight: let them land before the call returns)
        (void)hipStreamSynchronize(c->copy_in);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->copy_out);
        return st;
      }
      HIPCHK(c, hipEventRecord(c->ev_tr[b], c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->copy_out, c->ev_tr[b], 0));
      HIPCHK(c, hipMemcpyAsync(hits + off, d_h, m * sizeof(Hit), hipMemcpyDeviceToHost, c->copy_out));
      if (mask) HIPCHK(c, hipMemcpyAsync(mask + off, d_m, m, hipMemcpyDeviceToHost, c->copy_out));
      HIPCHK(c, hipEventRecord(c->ev_out[b], c->copy_out));
    }
    HIPCHK(c, hipStreamSynchronize(c->copy_out));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NRT_OK;
  }
  const uint64_t kMaxChunk = 1ull << 26; // rays per launch (keeps staging bounded)
  for (uint64_t off = 0; off < n; off += kMaxChunk) {
    const uint64_t m = std::min(kMaxChunk, n - off);
    nrt_status st;
    if ((st = ensure(c, c->st_rays, m * sizeof(Ray)))) return st;
    if ((st = ensure(c, c->st_hits, m * sizeof(Hit)))) return st;
    if ((st = ensure(c, c->st_mask, m))) return st;
    HIPCHK(c, hipMemcpyAsync(c->st_rays.p, rays + off, m * sizeof(Ray), hipMemcpyHostToDevice, c->stream));
    st = traverse_device<T>(c, (const Ray *)c->st_rays.p, m, opt, (Hit *)c->st_hits.p, (uint8_t *)c->st_mask.p,
                            c->stream, false, true);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(hits + off, c->st_hits.p, m * sizeof(Hit), hipMemcpyDeviceToHost, c->stream));
    if (mask) HIPCHK(c, hipMemcpyAsync(mask + off, c->st_mask.p, m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return NRT_OK;
}

// One host batch spread over several contexts — one per GPU of the node, each holding the same tree (the build is
// deterministic: replicas are bit-identical).  The batch is cut into rows of `row_len` rays (an image row; 4096 rays when 0) and
// row r goes to context r % num_ctx — interleaved, so that no device gets the empty sky or the dense ground alone.  Every
// context is driven by a host thread of its own: strided copy of its rows up (one 2-D copy), one launch, strided copies of its
// records and flags straight into the caller's arrays.  No collective: the results are host-resident and every GPU writes its
// own rows of them.  Records are exactly those of a single-context nrtTraverseBatch over the whole batch.
template <typename T>
static nrt_status traverse_share(nrt_ctx *c, uint32_t k, uint32_t num_ctx, const typename Wire<T>::Ray *rays, uint64_t n, uint64_t row_len,
                                 const nrt_trace_options *opt, typename Wire<T>::Hit *hits, uint8_t *mask) {
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;
  const uint64_t rows_total = (n + row_len - 1) / row_len;
  if (k >= rows_total) return NRT_OK;
  const uint64_t my_rows = (rows_total - k + num_ctx - 1) / num_ctx; // rows k, k + num_ctx, ...
  const uint64_t last_row = k + (my_rows - 1) * num_ctx;
  const uint64_t last_len = (last_row == rows_total - 1) ? n - last_row * row_len : row_len; // only the batch's final row can be short
  const uint64_t full_rows = last_len == row_len ? my_rows : my_rows - 1;
  const uint64_t m = full_rows * row_len + (full_rows < my_rows ? last_len : 0);
  std::lock_guard<std::mutex> host_lock(c->host_mutex);
  HIPCHK(c, hipSetDevice(c->device));
  nrt_status st;
  if ((st = ensure(c, c->st_rays, m * sizeof(Ray))) || (st = ensure(c, c->st_hits, m * sizeof(Hit))) || (st = ensure(c, c->st_mask, m))) return st;
  Ray *d_r = (Ray *)c->st_rays.p;
  Hit *d_h = (Hit *)c->st_hits.p;
  uint8_t *d_m = (uint8_t *)c->st_mask.p;
  const size_t rb = (size_t)row_len * sizeof(Ray), hb = (size_t)row_len * sizeof(Hit), mbytes = (size_t)row_len;
  if (full_rows) HIPCHK(c, hipMemcpy2DAsync(d_r, rb, rays + k * row_len, rb * num_ctx, rb, full_rows, hipMemcpyHostToDevice, c->stream));
  if (full_rows < my_rows)
    HIPCHK(c, hipMemcpyAsync(d_r + full_rows * row_len, rays + last_row * row_len, last_len * sizeof(Ray), hipMemcpyHostToDevice, c->stream));
  if ((st = traverse_device<T>(c, d_r, m, opt, d_h, d_m, c->stream, false, false))) return st;
  if (full_rows) {
    HIPCHK(c, hipMemcpy2DAsync(hits + k * row_len, hb * num_ctx, d_h, hb, hb, full_rows, hipMemcpyDeviceToHost, c->stream));
    if (mask) HIPCHK(c, hipMemcpy2DAsync(mask + k * row_len, mbytes * num_ctx, d_m, mbytes, mbytes, full_rows, hipMemcpyDeviceToHost, c->stream));
  }
  if (full_rows < my_rows) {
    HIPCHK(c, hipMemcpyAsync(hits + last_row * row_len, d_h + full_rows * row_len, last_len * sizeof(Hit), hipMemcpyDeviceToHost, c->stream));
    if (mask) HIPCHK(c, hipMemcpyAsync(mask + last_row * row_len, d_m + full_rows * row_len, last_len, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NRT_OK;
}

template <typename T>
static nrt_status traverse_multi(nrt_ctx *const *ctxs, uint32_t num_ctx, const typename Wire<T>::Ray *rays, uint64_t n, uint64_t row_len,
                                 const nrt_trace_options *opt, typename Wire<T>::Hit *hits, uint8_t *mask) {
  if (!ctxs || num_ctx == 0 || !ctxs[0]) return NRT_ERR_INVALID;
  nrt_ctx *c0 = ctxs[0];
  if (n == 0) return NRT_OK;
  if (!rays || !hits) return fail(c0, NRT_ERR_INVALID, "nrtTraverseBatchMulti: NULL rays/hits");
  for (uint32_t k = 0; k < num_ctx; k++) {
    if (!ctxs[k]) return fail(c0, NRT_ERR_INVALID, "nrtTraverseBatchMulti: context %u is NULL", k);
    for (uint32_t j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return fail(c0, NRT_ERR_INVALID, "nrtTraverseBatchMulti: context %u is listed twice", k);
    if (ctxs[k]->prec != (int)sizeof(T) || !ctxs[k]->d_nodes || ctxs[k]->prim_kind != kPrimTriangles)
      return fail(c0, NRT_ERR_INVALID, "nrtTraverseBatchMulti: context %u holds no triangle tree of this precision", k);
    if (ctxs[k]->num_nodes != c0->num_nodes || ctxs[k]->num_indices != c0->num_indices)
      return fail(c0, NRT_ERR_INVALID, "nrtTraverseBatchMulti: context %u holds another tree (%llu nodes, context 0 has %llu)", k,
                  (unsigned long long)ctxs[k]->num_nodes, (unsigned long long)c0->num_nodes);
  }
  if (row_len == 0) row_len = 4096;
  if (num_ctx == 1) return traverse_host<T>(c0, rays, n, opt, hits, mask);
  std::vector<nrt_status> st(num_ctx, NRT_OK);
  std::vector<std::thread> th;
  for (uint32_t k = 1; k < num_ctx; k++)
    th.emplace_back([&, k]() { st[k] = traverse_share<T>(ctxs[k], k, num_ctx, rays, n, row_len, opt, hits, mask); });
  st[0] = traverse_share<T>(c0, 0, num_ctx, rays, n, row_len, opt, hits, mask);
  for (std::thread &t : th) t.join();
  for (uint32_t k = 0; k < num_ctx; k++)
    if (st[k]) {
      if (k) c0->err = "context " + std::to_string(k) + ": " + ctxs[k]->err;
      return st[k];
    }
  return NRT_OK;
}

template <typename T>
static nrt_status traverse_count(nrt_ctx *c, const typename Wire<T>::Ray *d_rays, uint64_t n,
                                 const nrt_trace_options *opt, nrt_trace_counters *out) {
  if (!c || !out) return NRT_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  nrt_status st = traverse_device<T>(c, d_rays, n, opt, nullptr, nullptr, c->stream, true, false);
  if (st) return st;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long h[4];
  HIPCHK(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
  out->nodes_visited = h[0];
  out->leaves_tested = h[1];
  out->tris_tested = h[2];
  out->max_stack = h[3];
  return NRT_OK;
}

// One persistent launch over several independent batches (same trace options): the claim machinery hands out one virtual
// array, so the waves that run dry at the end of one batch carry on with the next — one tail and one completion record for
// all.  Contexts that cannot (fp64, custom primitives, the literal kernel) launch the batches one after the other on the
// same stream: the records are the same either way.
template <typename T>
static nrt_status traverse_batches_device(nrt_ctx *c, uint32_t nb, const typename Wire<T>::Ray *const *d_rays, const uint64_t *counts,
                                          const nrt_trace_options *opt, typename Wire<T>::Hit *const *d_hits,
                                          uint8_t *const *d_masks, const uint32_t *flags, hipStream_t s) {
  if (!c) return NRT_ERR_INVALID;
  if (nb == 0) return NRT_OK;
  if (!d_rays || !counts) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchesDevice: NULL argument");
  for (uint32_t k = 0; k < nb; k++)
    if (flags && (flags[k] & NRT_BATCH_OCCLUSION) && counts[k] && !(d_masks && d_masks[k]))
      return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchesDevice: occlusion batch %u has no flag array", k);
  // as nrtTraverseBatchDevice: a closest-hit batch needs its record array — an occlusion batch does not, and a call made of
  // occlusion batches only may pass no record table at all
  std::vector<typename Wire<T>::Hit *> no_hits;
  if (!d_hits) {
    no_hits.assign(nb, nullptr);
    d_hits = no_hits.data();
  }
  for (uint32_t k = 0; k < nb; k++)
    if (counts[k] && !(flags && (flags[k] & NRT_BATCH_OCCLUSION)) && !d_hits[k])
      return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchesDevice: batch %u has no hit array", k);
  TraverseBatches<T> mb;
  mb.nb = 0;
  uint64_t total = 0;
  for (uint32_t k = 0; k < nb; k++) {
    if (counts[k] == 0) continue;
    if (!d_rays[k]) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchesDevice: batch %u has no rays", k);
    total += counts[k];
  }
  if (total == 0) return NRT_OK;
  const bool one_launch = sizeof(T) == 4 && c->prec == 4 && c->prim_kind == kPrimTriangles && c->wide && c->d_wide && total <= 0x7FFFFFFFull;
  uint32_t k = 0;
  while (k < nb) { // groups of up to kMaxBatches non-empty batches per launch
    mb.nb = 0;
    mb.anyhit = 0;
    uint64_t n = 0;
    while (k < nb && mb.nb < (uint32_t)kMaxBatches) {
      if (counts[k]) {
        const bool occ = flags && (flags[k] & NRT_BATCH_OCCLUSION);
        mb.rays[mb.nb] = d_rays[k];
        mb.hits[mb.nb] = occ ? nullptr : d_hits[k]; // (an occlusion query writes the flags only)
        mb.mask[mb.nb] = d_masks ? d_masks[k] : nullptr;
        mb.count[mb.nb] = counts[k];
        if (occ) mb.anyhit |= 1u << mb.nb;
        n += counts[k];
        mb.nb++;
      }
      k++;
    }
    if (mb.nb == 0) break;
    nrt_status st;
    if (one_launch && mb.nb > 1) {
      st = traverse_device<T>(c, nullptr, n, opt, nullptr, nullptr, s, false, false, nullptr, false, &mb);
      if (st) return st;
    } else {
      for (uint32_t j = 0; j < mb.nb; j++)
        if ((st = traverse_device<T>(c, mb.rays[j], mb.count[j], opt, mb.hits[j], mb.mask[j], s, false, false, nullptr, ((mb.anyhit >> j) & 1u) != 0u)))
          return st;
    }
  }
  return NRT_OK;
}

// The same for HOST batches (nrtTraverseBatches): every batch is staged next to the others, ONE launch walks them all (one
// launch tail: a host renderer's shadow query of one depth and its path wave of the next), the records come back batch by
// batch.  Like nrtTraverseBatch the staging buffers are the context's: one host call at a time.
template <typename T>
static nrt_status traverse_batches_host(nrt_ctx *c, uint32_t nb, const typename Wire<T>::Ray *const *rays, const uint64_t *counts,
                                        const nrt_trace_options *opt, typename Wire<T>::Hit *const *hits, uint8_t *const *masks,
                                        const uint32_t *flags) {
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;
  if (!c) return NRT_ERR_INVALID;
  if (nb == 0) return NRT_OK;
  if (!rays || !counts) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatches: NULL argument");
  uint64_t total = 0, total_closest = 0; // (record staging is sized over the closest-hit batches only)
  for (uint32_t k = 0; k < nb; k++) {
    if (!counts[k]) continue;
    const bool occ = flags && (flags[k] & NRT_BATCH_OCCLUSION);
    if (!rays[k]) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatches: batch %u has no rays", k);
    if (occ && !(masks && masks[k])) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatches: occlusion batch %u has no flag array", k);
    if (!occ && !(hits && hits[k])) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatches: batch %u has no hit array", k);
    total += counts[k];
    if (!occ) total_closest += counts[k];
  }
  if (total == 0) return NRT_OK;
  if (total > (1ull << 26)) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatches: more than 2^26 rays in one call");
  std::lock_guard<std::mutex> host_lock(c->host_mutex);
  HIPCHK(c, hipSetDevice(c->device));
  nrt_status st;
  if ((st = ensure(c, c->st_rays, total * sizeof(Ray))) || (st = ensure(c, c->st_hits, std::max<uint64_t>(1, total_closest) * sizeof(Hit))) ||
      (st = ensure(c, c->st_mask, total)))
    return st;
  std::vector<const Ray *> d_r(nb, nullptr);
  std::vector<Hit *> d_h(nb, nullptr);
  std::vector<uint8_t *> d_m(nb, nullptr);
  uint64_t off = 0, off_closest = 0;
  for (uint32_t k = 0; k < nb; k++) {
    if (!counts[k]) continue;
    const bool occ = flags && (flags[k] & NRT_BATCH_OCCLUSION);
    d_r[k] = (const Ray *)c->st_rays.p + off;
    if (!occ) {
      d_h[k] = (Hit *)c->st_hits.p + off_closest;
      off_closest += counts[k];
    }
    d_m[k] = (uint8_t *)c->st_mask.p + off;
    HIPCHK(c, hipMemcpyAsync((void *)d_r[k], rays[k], counts[k] * sizeof(Ray), hipMemcpyHostToDevice, c->stream));
    off += counts[k];
  }
  st = traverse_batches_device<T>(c, nb, d_r.data(), counts, opt, d_h.data(), d_m.data(), flags, c->stream);
  if (st) {
    (void)hipStreamSynchronize(c->stream);
    return st;
  }
  for (uint32_t k = 0; k < nb; k++) {
    if (!counts[k]) continue;
    const bool occ = flags && (flags[k] & NRT_BATCH_OCCLUSION);
    if (!occ) HIPCHK(c, hipMemcpyAsync(hits[k], d_h[k], counts[k] * sizeof(Hit), hipMemcpyDeviceToHost, c->stream));
    if (masks && masks[k]) HIPCHK(c, hipMemcpyAsync(masks[k], d_m[k], counts[k], hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NRT_OK;
}

template <typename T>
static nrt_status occluded_host(nrt_ctx *c, const typename Wire<T>::Ray *rays, uint64_t n, const nrt_trace_options *opt, uint8_t *mask) {
  if (!c) return NRT_ERR_INVALID;
  if (n == 0) return NRT_OK;
  if (!rays || !mask) return fail(c, NRT_ERR_INVALID, "nrtOccludedBatch: NULL rays/mask");
  typedef typename Wire<T>::Ray Ray;
  std::lock_guard<std::mutex> host_lock(c->host_mutex);
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t kMaxChunk = 1ull << 26;
  for (uint64_t off = 0; off < n; off += kMaxChunk) {
    const uint64_t m = std::min(kMaxChunk, n - off);
    nrt_status st;
    if ((st = ensure(c, c->st_rays, m * sizeof(Ray)))) return st;
    if ((st = ensure(c, c->st_mask, m))) return st;
    HIPCHK(c, hipMemcpyAsync(c->st_rays.p, rays + off, m * sizeof(Ray), hipMemcpyHostToDevice, c->stream));
    st = traverse_device<T>(c, (const Ray *)c->st_rays.p, m, opt, nullptr, (uint8_t *)c->st_mask.p, c->stream, false, true, nullptr, true);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(mask + off, c->st_mask.p, m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return NRT_OK;
}

extern "C" {

nrt_status nrtSetMesh_f32(nrt_ctx *c, const float *v, size_t stride, const uint32_t *f, uint32_t nf) {
  return set_mesh<float>(c, v, stride, f, nf);
}
nrt_status nrtSetMesh_f64(nrt_ctx *c, const double *v, size_t stride, const uint32_t *f, uint32_t nf) {
  return set_mesh<double>(c, v, stride, f, nf);
}

nrt_status nrtSetCylinders_f32(nrt_ctx *c, const float *endpoints, const float *radii, uint32_t n, int test_cap) {
  return set_cylinders(c, endpoints, radii, n, test_cap);
}

nrt_status nrtTraverseBatchCylindersDevice_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o,
                                               nrt_cyl_hit_f32 *h, uint8_t *m, void *s) {
  if (!c) return NRT_ERR_INVALID;
  if (n && !h) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchCylindersDevice: NULL hits");
  return traverse_device<float>(c, r, n, o, nullptr, m, (hipStream_t)s, false, true, h);
}

nrt_status nrtTraverseBatchCylinders_f32(nrt_ctx *c, const nrt_ray_f32 *rays, uint64_t n, const nrt_trace_options *opt,
                                         nrt_cyl_hit_f32 *hits, uint8_t *mask) {
  if (!c) return NRT_ERR_INVALID;
  if (n == 0) return NRT_OK;
  if (!rays || !hits) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchCylinders: NULL rays/hits");
  std::lock_guard<std::mutex> host_lock(c->host_mutex);
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t kMaxChunk = 1ull << 26;
  for (uint64_t off = 0; off < n; off += kMaxChunk) {
    const uint64_t m = std::min(kMaxChunk, n - off);
    nrt_status st;
    if ((st = ensure(c, c->st_rays, m * sizeof(nrt_ray_f32)))) return st;
    if ((st = ensure(c, c->st_hits, m * sizeof(nrt_cyl_hit_f32)))) return st;
    if ((st = ensure(c, c->st_mask, m))) return st;
    HIPCHK(c, hipMemcpyAsync(c->st_rays.p, rays + off, m * sizeof(nrt_ray_f32), hipMemcpyHostToDevice, c->stream));
    st = traverse_device<float>(c, (const nrt_ray_f32 *)c->st_rays.p, m, opt, nullptr, (uint8_t *)c->st_mask.p, c->stream, false,
                                true, c->st_hits.p);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(hits + off, c->st_hits.p, m * sizeof(nrt_cyl_hit_f32), hipMemcpyDeviceToHost, c->stream));
    if (mask) HIPCHK(c, hipMemcpyAsync(mask + off, c->st_mask.p, m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return NRT_OK;
}

nrt_status nrtOccludedBatch_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o, uint8_t *m) {
  return occluded_host<float>(c, r, n, o, m);
}
nrt_status nrtOccludedBatch_f64(nrt_ctx *c, const nrt_ray_f64 *r, uint64_t n, const nrt_trace_options *o, uint8_t *m) {
  return occluded_host<double>(c, r, n, o, m);
}
nrt_status nrtOccludedBatchDevice_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o, uint8_t *m, void *s) {
  if (!c) return NRT_ERR_INVALID;
  if (n && !m) return fail(c, NRT_ERR_INVALID, "nrtOccludedBatchDevice: NULL mask");
  return traverse_device<float>(c, r, n, o, nullptr, m, (hipStream_t)s, false, true, nullptr, true);
}
nrt_status nrtOccludedBatchDevice_f64(nrt_ctx *c, const nrt_ray_f64 *r, uint64_t n, const nrt_trace_options *o, uint8_t *m, void *s) {
  if (!c) return NRT_ERR_INVALID;
  if (n && !m) return fail(c, NRT_ERR_INVALID, "nrtOccludedBatchDevice: NULL mask");
  return traverse_device<double>(c, r, n, o, nullptr, m, (hipStream_t)s, false, true, nullptr, true);
}

nrt_status nrtSetSpheres_f32(nrt_ctx *c, const float *centers, const float *radii, uint32_t n) {
  return set_spheres<float>(c, centers, radii, n);
}

nrt_status nrtBuild_f32(nrt_ctx *c, const nrt_build_options_f32 *o, nrt_build_stats *st, uint64_t *nn) {
  return build<float>(c, o, st, nn);
}
nrt_status nrtBuild_f64(nrt_ctx *c, const nrt_build_options_f64 *o, nrt_build_stats *st, uint64_t *nn) {
  return build<double>(c, o, st, nn);
}

nrt_status nrtGetTree_f32(nrt_ctx *c, nrt_node_f32 *n, uint32_t *i) { return get_tree<float>(c, n, i); }
nrt_status nrtGetTree_f64(nrt_ctx *c, nrt_node_f64 *n, uint32_t *i) { return get_tree<double>(c, n, i); }

nrt_status nrtGetTreeBounds_f32(nrt_ctx *c, float *lo, float *hi) { return get_tree_bounds<float>(c, lo, hi); }
nrt_status nrtGetTreeBounds_f64(nrt_ctx *c, double *lo, double *hi) { return get_tree_bounds<double>(c, lo, hi); }

nrt_status nrtTreeSize(nrt_ctx *c, uint64_t *nn, uint64_t *ni) {
  if (!c) return NRT_ERR_INVALID;
  if (nn) *nn = c->num_nodes;
  if (ni) *ni = c->num_indices;
  return NRT_OK;
}

nrt_status nrtSetTree_f32(nrt_ctx *c, const nrt_node_f32 *n, uint64_t nn, const uint32_t *i, uint64_t ni) {
  return set_tree<float>(c, n, nn, i, ni);
}
nrt_status nrtSetTree_f64(nrt_ctx *c, const nrt_node_f64 *n, uint64_t nn, const uint32_t *i, uint64_t ni) {
  return set_tree<double>(c, n, nn, i, ni);
}

nrt_status nrtTraverseBatch_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o,
                                nrt_hit_f32 *h, uint8_t *m) {
  return traverse_host<float>(c, r, n, o, h, m);
}
nrt_status nrtTraverseBatch_f64(nrt_ctx *c, const nrt_ray_f64 *r, uint64_t n, const nrt_trace_options *o,
                                nrt_hit_f64 *h, uint8_t *m) {
  return traverse_host<double>(c, r, n, o, h, m);
}

nrt_status nrtTraverseBatchDevice_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o,
                                      nrt_hit_f32 *h, uint8_t *m, void *s) {
  if (!c) return NRT_ERR_INVALID;
  if (n && !h) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchDevice: NULL hits");
  return traverse_device<float>(c, r, n, o, h, m, (hipStream_t)s, false, true);
}
nrt_status nrtTraverseBatchDevice_f64(nrt_ctx *c, const nrt_ray_f64 *r, uint64_t n, const nrt_trace_options *o,
                                      nrt_hit_f64 *h, uint8_t *m, void *s) {
  if (!c) return NRT_ERR_INVALID;
  if (n && !h) return fail(c, NRT_ERR_INVALID, "nrtTraverseBatchDevice: NULL hits");
  return traverse_device<double>(c, r, n, o, h, m, (hipStream_t)s, false, true);
}

nrt_status nrtTraverseBatchMulti_f32(nrt_ctx *const *ctxs, uint32_t num_ctx, const nrt_ray_f32 *r, uint64_t n, uint64_t row_len,
                                     const nrt_trace_options *o, nrt_hit_f32 *h, uint8_t *m) {
  return traverse_multi<float>(ctxs, num_ctx, r, n, row_len, o, h, m);
}
nrt_status nrtTraverseBatchMulti_f64(nrt_ctx *const *ctxs, uint32_t num_ctx, const nrt_ray_f64 *r, uint64_t n, uint64_t row_len,
                                     const nrt_trace_options *o, nrt_hit_f64 *h, uint8_t *m) {
  return traverse_multi<double>(ctxs, num_ctx, r, n, row_len, o, h, m);
}
int nrtDeviceCount(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
nrt_status nrtTraverseBatchesDevice_f32(nrt_ctx *c, uint32_t nb, const nrt_ray_f32 *const *r, const uint64_t *n, const nrt_trace_options *o,
                                        nrt_hit_f32 *const *h, uint8_t *const *m, const uint32_t *fl, void *s) {
  return traverse_batches_device<float>(c, nb, r, n, o, h, m, fl, (hipStream_t)s);
}
nrt_status nrtTraverseBatchesDevice_f64(nrt_ctx *c, uint32_t nb, const nrt_ray_f64 *const *r, const uint64_t *n, const nrt_trace_options *o,
                                        nrt_hit_f64 *const *h, uint8_t *const *m, const uint32_t *fl, void *s) {
  return traverse_batches_device<double>(c, nb, r, n, o, h, m, fl, (hipStream_t)s);
}
nrt_status nrtTraverseBatches_f32(nrt_ctx *c, uint32_t nb, const nrt_ray_f32 *const *r, const uint64_t *n, const nrt_trace_options *o,
                                  nrt_hit_f32 *const *h, uint8_t *const *m, const uint32_t *fl) {
  return traverse_batches_host<float>(c, nb, r, n, o, h, m, fl);
}
nrt_status nrtTraverseBatches_f64(nrt_ctx *c, uint32_t nb, const nrt_ray_f64 *const *r, const uint64_t *n, const nrt_trace_options *o,
                                  nrt_hit_f64 *const *h, uint8_t *const *m, const uint32_t *fl) {
  return traverse_batches_host<double>(c, nb, r, n, o, h, m, fl);
}
nrt_status nrtTraverseCountDevice_f32(nrt_ctx *c, const nrt_ray_f32 *r, uint64_t n, const nrt_trace_options *o,
                                      nrt_trace_counters *out) {
  return traverse_count<float>(c, r, n, o, out);
}
nrt_status nrtTraverseCountDevice_f64(nrt_ctx *c, const nrt_ray_f64 *r, uint64_t n, const nrt_trace_options *o,
                                      nrt_trace_counters *out) {
  return traverse_count<double>(c, r, n, o, out);
}

// on = 1: events around every traversal launch (a timing pair nrtLastTraverseMs reads, the slot's completion event) — the
// cross-check of the default, which records no event at all: the kernel's last wave publishes a completion record with its
// own start / end stamps (see slot_done / wait_record for who waits on what).
nrt_status nrtSetLaunchTiming(nrt_ctx *c, int on) {
  if (!c) return NRT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->launch_mutex);
  c->launch_timing = on ? 1 : 0;
  return NRT_OK;
}

// Traversal / build tunables by name (the table above).  Values are clamped to the tunable's range; a tunable that shapes
// the private tree layout (wide4, wide_scramble, morton) takes effect with the next nrtBuild / nrtSetTree.
nrt_status nrtSetTunable(nrt_ctx *c, const char *name, long long value) {
  if (!c) return NRT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->launch_mutex);
  if (!tunable_set(c, name, value)) return fail(c, NRT_ERR_INVALID, "nrtSetTunable: unknown tunable '%s'", name ? name : "(null)");
  return NRT_OK;
}
nrt_status nrtGetTunable(nrt_ctx *c, const char *name, long long *value_out) {
  if (!c || !value_out) return NRT_ERR_INVALID;
  const TunableDesc *d = tunable_find(name);
  if (!d) return fail(c, NRT_ERR_INVALID, "nrtGetTunable: unknown tunable '%s'", name ? name : "(null)");
  std::lock_guard<std::mutex> lock(c->launch_mutex);
  *value_out = d->get(c);
  return NRT_OK;
}

float nrtLastTraverseMs(nrt_ctx *c) {
  if (!c) return -1.f;
  hipEvent_t t0, t1;
  {
    std::unique_lock<std::mutex> lock(c->launch_mutex);
    if (c->last_timed_slot < 0) return -1.f;
    nrt_ctx::LaunchSlot &sl = c->slots[c->last_timed_slot];
    if (sl.last_has_rec) { // the kernel's own stamps: first block started -> last wave finished (100 MHz realtime ticks)
      if (wait_record(sl) != hipSuccess) return -1.f;
      const unsigned long long b = sl.h_done->t_begin, e = sl.h_done->t_end;
      // The record says that every wave has stopped reading; the hit records' non-temporal stores are not fenced by it
      // (traverse.hip, done_end).  This call is documented as the caller's synchronisation point, so it also drains the
      // launch's stream (the stamps are taken: the reported time is unaffected).
      const hipStream_t s = sl.stream;
      lock.unlock();
      if (hipStreamSynchronize(s) != hipSuccess) return -1.f;
      return e >= b ? (float)((double)(e - b) * 1e-5) : -1.f;
    }
    if (!sl.last_timed) return -1.f;
    t0 = sl.t0;
    t1 = sl.t1;
  }
  if (hipEventSynchronize(t1) != hipSuccess) return -1.f;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, t0, t1) != hipSuccess) return -1.f;
  return ms;
}

// Name of the traversal kernel variant the most recent launch of this context used (as rocprofv3 prints it, without
// the argument list) — bench.py reports it instead of guessing.
const char *nrtLastKernelName(const nrt_ctx *c) { return c ? c->last_kernel : ""; }

#ifdef NRT_PROF // libnanort_hip_prof.so only (include/nanort_hip_prof.h)
// Profiling aid (not part of the public header): loop-occupancy counters of the last launch made with
// NRT_DEBUG bit 32 set.  out[0..6] = it1, act1, trav1, it2, act2, refills, refilled.
int nrtDebugCounters(nrt_ctx *c, unsigned long long *out, int cap) {
  if (!c || !out || cap < 16) return 1; // (sixteen counters are written)
  if (hipStreamSynchronize(c->stream) != hipSuccess) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpy(out, c->d_counters, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}

// Profiling aid (not part of the public header): with NRT_DEBUG bit 8192 every wave of a traversal launch records when
// it started, ran out of rays and finished (100 MHz realtime ticks).  Copies up to `cap` records (3 x u64 each) of the
// last launch; returns the number of waves, or -1.
long nrtDebugWaveClocks(nrt_ctx *c, unsigned long long *out, long cap) {
  if (!c || !out || !c->b_wave_clock.p) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const long n = std::min<long>(cap, (long)c->wave_clock_waves);
  if (hipMemcpy(out, c->b_wave_clock.p, (size_t)n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)c->wave_clock_waves;
}

#endif

float nrtLastBuildMs(nrt_ctx *c) {
  if (!c || !c->have_build_time) return -1.f;
  return c->stats.build_secs * 1e3f;
}

} // extern "C"

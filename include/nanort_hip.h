/* include/nanort_hip.h — C ABI of libnanort_hip.so (MI355X / gfx950 backend).
 *
 * The drop-in boundary for the hot path of lighttransport/nanort:
 * BVHAccel<T>::Build() + BVHAccel<T>::Traverse() for the built-in triangle
 * plugin.  The reference has no FFI of its own — its boundary is a C++
 * template API in one header — so the entry points below are what the inline
 * code of include/nanort.h (this repo's API-compatible header) binds when
 * NANORT_USE_HIP_BACKEND is defined.  Each entry point cites the reference
 * interface it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++ / torch types;
 *   - PODs are layout-identical to the reference's (static_asserts below);
 *   - nrt_status 0 == the reference's `true`; non-zero == `false`, with a
 *     human-readable reason from nrtLastError() (the reference has no error
 *     channel beyond bool/assert, nanort.h:1905-1909);
 *   - "host" entry points take host pointers in/out and own all device
 *     memory; "Device" entry points take device pointers (HBM-resident
 *     buffers, e.g. torch tensors) and a hipStream_t passed as void*;
 *   - one nrt_ctx per BVHAccel object; a context is bound to one GPU.  The
 *     primitive / build / tree calls are not re-entrant on one context (like
 *     BVHAccel::Build, nanort.h:1892).  The traversal calls may be issued
 *     from several host threads: the host-buffer ones share the context's
 *     staging buffers and are served one at a time; the Device ones run
 *     concurrently, on several streams at once, and launches on different
 *     streams overlap on the GPU (each launch owns its scratch until it
 *     completes).  Distinct contexts may be used from distinct host threads.
 *
 * Beyond the triangle path: nrtSetSpheres_f32 /
 * nrtSetCylinders_f32 (the reference's two custom-primitive examples as
 * device primitives) and nrtScene* (NanoSG's instanced two-level traversal).
 */
#ifndef NANORT_HIP_H_
#define NANORT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRT_API __attribute__((visibility("default")))

typedef int nrt_status;
enum {
  NRT_OK = 0,
  NRT_ERR_INVALID = 1,  /* bad argument / call order                  */
  NRT_ERR_EMPTY = 2,    /* n == 0: the reference's Build() == false   */
  NRT_ERR_DEVICE = 3,   /* a HIP call failed (see nrtLastError)       */
  NRT_ERR_PRECISION = 4 /* _f32 call on an _f64 context or vice versa */
};

typedef struct nrt_ctx nrt_ctx;

/* ---- PODs: wire formats shared with the reference ----------------------- */

/* nanort::Ray<float> / Ray<double> — nanort.h:474-496. */
typedef struct {
  float org[3];
  float dir[3];
  float min_t;
  float max_t;
  uint32_t type;
} nrt_ray_f32;
typedef struct {
  double org[3];
  double dir[3];
  double min_t;
  double max_t;
  uint32_t type;
  uint32_t _pad;
} nrt_ray_f64;

/* nanort::BVHNode<T> — nanort.h:498-550.
 * leaf: flag=1, data={count, first slot in indices}; branch: flag=0,
 * axis in {0,1,2}, data={low-side child, high-side child}. */
typedef struct {
  float bmin[3];
  float bmax[3];
  int32_t flag;
  int32_t axis;
  uint32_t data[2];
} nrt_node_f32;
typedef struct {
  double bmin[3];
  double bmax[3];
  int32_t flag;
  int32_t axis;
  uint32_t data[2];
} nrt_node_f64;

/* nanort::TriangleIntersection<T> — nanort.h:996-1005.
 * The batched entry points write EVERY record: on a miss the record is
 * {u=0, v=0, t=ray.max_t, prim_id=0xFFFFFFFF} and hit_mask[i]=0 (the
 * reference leaves the caller's struct untouched on a miss, nanort.h:1206;
 * include/nanort.h's TraverseBatch restores that behaviour on the host). */
typedef struct {
  float u;
  float v;
  float t;
  uint32_t prim_id;
} nrt_hit_f32;
typedef struct {
  double u;
  double v;
  double t;
  uint32_t prim_id;
  uint32_t _pad;
} nrt_hit_f64;

/* nanort::BVHBuildOptions<T> — nanort.h:559-583. */
typedef struct {
  float cost_t_aabb;
  uint32_t min_leaf_primitives;
  uint32_t max_tree_depth;
  uint32_t bin_size;
  uint32_t shallow_depth;
  uint32_t min_primitives_for_parallel_build;
  uint8_t cache_bbox;
  uint8_t pad[3];
} nrt_build_options_f32;
typedef struct {
  double cost_t_aabb;
  uint32_t min_leaf_primitives;
  uint32_t max_tree_depth;
  uint32_t bin_size;
  uint32_t shallow_depth;
  uint32_t min_primitives_for_parallel_build;
  uint8_t cache_bbox;
  uint8_t pad[3];
} nrt_build_options_f64;

/* nanort::BVHBuildStatistics — nanort.h:586-599.  build_secs IS filled here
 * (device time of the build; the reference declares it but never writes it). */
typedef struct {
  uint32_t max_tree_depth;
  uint32_t num_leaf_nodes;
  uint32_t num_branch_nodes;
  float build_secs;
} nrt_build_stats;

/* nanort::BVHTraceOptions — nanort.h:604-624. */
typedef struct {
  uint32_t prim_ids_range[2]; /* half-open [r0, r1)            nanort.h:1055 */
  uint32_t skip_prim_id;      /* 0xFFFFFFFF = none             nanort.h:1061 */
  uint8_t cull_back_face;     /*                               nanort.h:1111 */
  uint8_t pad[3];
} nrt_trace_options;

/* Work counters of one batched traversal, in the units SURVEY.md §8(d)
 * defines the algorithmic bytes on (counted on the reference's traversal
 * order over the node array actually traversed). */
typedef struct {
  uint64_t nodes_visited; /* stack pops == BVHNode fetches + slab tests */
  uint64_t leaves_tested;
  uint64_t tris_tested;   /* Intersect() calls                          */
  uint64_t max_stack;     /* deepest stack (entries) any ray needed     */
} nrt_trace_counters;

#ifdef __cplusplus
static_assert(sizeof(nrt_ray_f32) == 36 && sizeof(nrt_ray_f64) == 72, "Ray layout");
static_assert(sizeof(nrt_node_f32) == 40 && sizeof(nrt_node_f64) == 64, "BVHNode layout");
static_assert(sizeof(nrt_hit_f32) == 16 && sizeof(nrt_hit_f64) == 32, "TriangleIntersection layout");
static_assert(sizeof(nrt_build_options_f32) == 28 && sizeof(nrt_build_options_f64) == 32, "BVHBuildOptions layout");
static_assert(sizeof(nrt_build_stats) == 16 && sizeof(nrt_trace_options) == 16, "stats/options layout");
#endif

/* ---- context ------------------------------------------------------------ */

/* One context == one nanort::BVHAccel<T> (nanort.h:698-860) living on GPU
 * `device`.  Precision is fixed by the first nrtSetMesh_* call. */
NRT_API nrt_status nrtCreate(int device, nrt_ctx **out);
NRT_API void nrtDestroy(nrt_ctx *ctx);
/* Reason of the last failure on this context (or of the last failed
 * nrtCreate when ctx == NULL).  Never NULL. */
NRT_API const char *nrtLastError(const nrt_ctx *ctx);
/* Library / device identification, e.g. "libnanort_hip gfx950 ...". */
NRT_API const char *nrtVersion(void);

/* Page-locked (pinned) host memory, for callers that do not link HIP themselves: ray / hit arrays allocated here make
 * the copies inside the host entry points (nrtTraverseBatch_*, nrtSceneTraverseBatch_f32, ...) run at PCIe speed instead of
 * being staged through the runtime's bounce buffers.  No reference counterpart.  nrtHostFree(NULL) is a no-op. */
NRT_API nrt_status nrtHostAlloc(size_t bytes, void **out);
NRT_API void nrtHostFree(void *p);

/* ---- mesh: replaces the TriangleMesh / TriangleSAHPred / TriangleIntersector
 * constructors (nanort.h:866-873, 925-930, 1032-1039) ----------------------
 * vertices are read through `vertex_stride_bytes` exactly like
 * get_vertex_addr (nanort.h:467-472); faces are tight 3 x u32.  The mesh
 * carries no vertex count in the reference, so it is derived as
 * max(faces)+1.  The data is copied to HBM; the caller's arrays are not
 * referenced after the call returns. */
NRT_API nrt_status nrtSetMesh_f32(nrt_ctx *ctx, const float *vertices, size_t vertex_stride_bytes,
                                  const uint32_t *faces, uint32_t num_faces);
NRT_API nrt_status nrtSetMesh_f64(nrt_ctx *ctx, const double *vertices, size_t vertex_stride_bytes,
                                  const uint32_t *faces, uint32_t num_faces);

/* ---- occlusion queries: an OPT-IN EXTENSION with no reference counterpart ----------------------------------
 * The reference answers "is anything in the way?" with a closest-hit Traverse (CheckForOccluder,
 * examples/path_tracer/main.cc:675-701).  These entry points return only the hit flag — mask_out[i] is exactly what
 * nrtTraverseBatch* would put in hit_mask_out[i], for every ray and option — but a ray stops at the first primitive it
 * accepts instead of looking for the nearest one.  Triangle contexts only. */
NRT_API nrt_status nrtOccludedBatch_f32(nrt_ctx *ctx, const nrt_ray_f32 *rays, uint64_t num_rays,
                                        const nrt_trace_options *options, uint8_t *mask_out);
NRT_API nrt_status nrtOccludedBatch_f64(nrt_ctx *ctx, const nrt_ray_f64 *rays, uint64_t num_rays,
                                        const nrt_trace_options *options, uint8_t *mask_out);
NRT_API nrt_status nrtOccludedBatchDevice_f32(nrt_ctx *ctx, const nrt_ray_f32 *d_rays, uint64_t num_rays,
                                              const nrt_trace_options *options, uint8_t *d_mask_out, void *hip_stream);
NRT_API nrt_status nrtOccludedBatchDevice_f64(nrt_ctx *ctx, const nrt_ray_f64 *d_rays, uint64_t num_rays,
                                              const nrt_trace_options *options, uint8_t *d_mask_out, void *hip_stream);

/* ---- sphere primitives: replaces the SpherePred / SphereGeometry / SphereIntersector constructors of the
 * reference's custom-primitive example (examples/particle_primitive/main.cc:82-147, 161-166) -------------
 * `centers` holds xyz per sphere (tight), `radii` one radius per sphere.  After this call nrtBuild_f32 builds
 * over the spheres' boxes (centre +- radius, SAH position = the centre) and nrtTraverseBatch*_f32 runs that
 * example's intersector: nearest root of the quadratic with the reference's acceptance rules (main.cc:174-236),
 * hit record {u, v, t, prim_id} with u, v the spherical coordinates of the hit normal (main.cc:262-277).
 * BVHTraceOptions: prim_ids_range is honoured; skip_prim_id and cull_back_face do not exist in that intersector
 * and are ignored.  Replaces the mesh of the context (one primitive kind per context). */
NRT_API nrt_status nrtSetSpheres_f32(nrt_ctx *ctx, const float *centers, const float *radii, uint32_t num_spheres);

/* ---- cylinder primitives: replaces the CylinderPred / CylinderGeometry / CylinderIntersector constructors of the
 * reference's second custom-primitive example (examples/cylinder_primitive/main.cc:94-232) ------------------
 * `endpoints` holds two xyz points per cylinder (tight), `radii` two radii per cylinder (the example's intersector
 * uses the larger of the two for the whole cylinder), `test_cap` is that intersector's constructor flag.
 * nrtBuild_f32 then builds over the example's boxes (union of end point +- radius, SAH position = the midpoint)
 * and nrtTraverseBatchCylinders*_f32 runs its intersector (main.cc:237-343: two cap planes, then the side via
 * solve2e :61-90) and PostTraversal (:367-418).  Hit record: the example's CylinderIntersection (:213-224) —
 * {u, v, normal[3], t, prim_id}; `t` is the intersector's hit distance (the example itself never stores it).
 * A miss leaves {0, 0, (0,0,0), ray.max_t, 0xFFFFFFFF}.  Of BVHTraceOptions only prim_ids_range exists there.
 * SEGMENTS (tunables "cyl_split", default 32, and "cyl_seg_radii", default 8; read here): a cylinder many radii long — the
 * example's own scene is box-spanning needles, over whose whole boxes a tree prunes nothing — is handed to the builder as
 * up to cyl_split pieces of its axis, one per cyl_seg_radii tube radii of length, each with the tight box of its piece
 * (end points of the piece +- max(r0, r1), the radius the intersector uses) and the CYLINDER's id.  The index array of the
 * built tree then names a cylinder once per segment (nrtTreeSize gives its length), a leaf tests the whole cylinder, and
 * the intersector — a pure function of (ray, cylinder, current t) — returns the same record or rejects when a cylinder is
 * tested again.  The example's scene at 1920x1080: 44 -> 2 100 Mrays/s.  The reference's Traverse over the same arrays gives
 * the same records (the restated example walks them in the tests).  cyl_split = 1: the example's own whole-cylinder boxes. */
typedef struct nrt_cyl_hit_f32 {
  float u, v;
  float normal[3];
  float t;
  uint32_t prim_id;
} nrt_cyl_hit_f32;
NRT_API nrt_status nrtSetCylinders_f32(nrt_ctx *ctx, const float *endpoints, const float *radii, uint32_t num_cylinders,
                                       int test_cap);
NRT_API nrt_status nrtTraverseBatchCylinders_f32(nrt_ctx *ctx, const nrt_ray_f32 *rays, uint64_t num_rays,
                                                 const nrt_trace_options *options, nrt_cyl_hit_f32 *hits_out,
                                                 uint8_t *hit_mask_out);
/* Same with HBM-resident buffers, asynchronous on `hip_stream` (a hipStream_t). */
NRT_API nrt_status nrtTraverseBatchCylindersDevice_f32(nrt_ctx *ctx, const nrt_ray_f32 *d_rays, uint64_t num_rays,
                                                       const nrt_trace_options *options, nrt_cyl_hit_f32 *d_hits_out,
                                                       uint8_t *d_hit_mask_out, void *hip_stream);

/* ---- build: replaces BVHAccel<T>::Build (nanort.h:716-718, 1892-2149) ----
 * Binned-SAH construction on the GPU over the mesh set above.  Honours
 * min_leaf_primitives and max_tree_depth (the reference's leaf rule,
 * nanort.h:1781-1783) and bin_size — with two caps the reference does not
 * have (nanort.h:574-582, 1314-1367 take any bin_size): at most 64 bins for a
 * node of more than 256 primitives (one lane of a wave per bin) and at most
 * 16 bins for a node of 256 primitives or fewer (the one-wave-per-subtree
 * phase: one 16-lane row per axis).  A larger bin_size is clamped, not
 * rejected; hit records never depend on it.  shallow_depth,
 * min_primitives_for_parallel_build, cache_bbox and cost_t_aabb are CPU
 * scheduling knobs (or dead, nanort.h:574) and are ignored.  options == NULL
 * means BVHBuildOptions<T>() defaults.  Emits the reference's node format
 * and invariants (root = node 0; indices a permutation of [0, n) — for
 * cylinders cut into segments, nrtSetCylinders_f32, every id once per
 * segment —; DFS pre-order, left child = parent + 1).  NRT_ERR_EMPTY iff num_faces == 0. */
NRT_API nrt_status nrtBuild_f32(nrt_ctx *ctx, const nrt_build_options_f32 *options,
                                nrt_build_stats *stats_out, uint64_t *num_nodes_out);
NRT_API nrt_status nrtBuild_f64(nrt_ctx *ctx, const nrt_build_options_f64 *options,
                                nrt_build_stats *stats_out, uint64_t *num_nodes_out);

/* Copy the built tree to the host: feeds BVHAccel::nodes_ / indices_ so
 * GetNodes/GetIndices/BoundingBox/Dump and the per-ray CPU Traverse keep
 * working (nanort.h:786-806, 2164-2217).  nodes_out holds num_nodes records,
 * indices_out holds num_faces entries. */
NRT_API nrt_status nrtGetTree_f32(nrt_ctx *ctx, nrt_node_f32 *nodes_out, uint32_t *indices_out);
NRT_API nrt_status nrtGetTree_f64(nrt_ctx *ctx, nrt_node_f64 *nodes_out, uint32_t *indices_out);
NRT_API nrt_status nrtTreeSize(nrt_ctx *ctx, uint64_t *num_nodes_out, uint64_t *num_indices_out);
/* The root node's box alone (24 / 48 bytes instead of the whole array): what BVHAccel::BoundingBox returns (nanort.h:786-799)
 * while the header leaves the tree on the device until the host asks for it (include/nanort.h: EnsureHostTree). */
NRT_API nrt_status nrtGetTreeBounds_f32(nrt_ctx *ctx, float bmin_out[3], float bmax_out[3]);
NRT_API nrt_status nrtGetTreeBounds_f64(nrt_ctx *ctx, double bmin_out[3], double bmax_out[3]);

/* Adopt a tree built elsewhere: BVHAccel<T>::Load (nanort.h:2219-2275) or a
 * CPU Build().  Validates child / slot bounds and measures the depth. */
NRT_API nrt_status nrtSetTree_f32(nrt_ctx *ctx, const nrt_node_f32 *nodes, uint64_t num_nodes,
                                  const uint32_t *indices, uint64_t num_indices);
NRT_API nrt_status nrtSetTree_f64(nrt_ctx *ctx, const nrt_node_f64 *nodes, uint64_t num_nodes,
                                  const uint32_t *indices, uint64_t num_indices);

/* ---- traverse: replaces N calls of BVHAccel<T>::Traverse with a
 * TriangleIntersector (nanort.h:757-759, 2487-2556, 1014-1229) -------------
 * Closest hit per ray, same arithmetic as the reference (no contraction,
 * fp64 edge fallback, MaxMult slab test).  options == NULL means
 * BVHTraceOptions() defaults.  hit_mask_out may be NULL. */
NRT_API nrt_status nrtTraverseBatch_f32(nrt_ctx *ctx, const nrt_ray_f32 *rays, uint64_t num_rays,
                                        const nrt_trace_options *options, nrt_hit_f32 *hits_out,
                                        uint8_t *hit_mask_out);
NRT_API nrt_status nrtTraverseBatch_f64(nrt_ctx *ctx, const nrt_ray_f64 *rays, uint64_t num_rays,
                                        const nrt_trace_options *options, nrt_hit_f64 *hits_out,
                                        uint8_t *hit_mask_out);

/* Same, on HBM-resident buffers, asynchronously on `hip_stream`
 * (a hipStream_t; NULL = the default stream).  No host synchronisation. */
NRT_API nrt_status nrtTraverseBatchDevice_f32(nrt_ctx *ctx, const nrt_ray_f32 *d_rays,
                                              uint64_t num_rays, const nrt_trace_options *options,
                                              nrt_hit_f32 *d_hits_out, uint8_t *d_hit_mask_out,
                                              void *hip_stream);
NRT_API nrt_status nrtTraverseBatchDevice_f64(nrt_ctx *ctx, const nrt_ray_f64 *d_rays,
                                              uint64_t num_rays, const nrt_trace_options *options,
                                              nrt_hit_f64 *d_hits_out, uint8_t *d_hit_mask_out,
                                              void *hip_stream);

/* Several independent HBM-resident batches in ONE launch (same trace options; arrays of `num_batches` device pointers and
 * counts; d_masks or any of its entries may be NULL; batch_flags may be NULL).  The persistent kernel hands out the batches as
 * one virtual ray array, so the waves that run dry at the end of one batch carry on with the next: one launch tail instead of
 * `num_batches` (a renderer's shadow and bounce waves, or two tiles, without a second stream).  A batch flagged
 * NRT_BATCH_OCCLUSION is an occlusion query (nrtOccludedBatchDevice's contract: its d_masks entry receives the flags, its
 * d_hits entry is ignored).  Records are exactly those of separate nrtTraverseBatchDevice / nrtOccludedBatchDevice calls.
 * fp64, sphere and cylinder contexts launch the batches one after the other.  Asynchronous on `hip_stream` like
 * nrtTraverseBatchDevice.  (No reference counterpart: nanort.h traces one ray per call, :2487-2556.) */
#define NRT_BATCH_OCCLUSION 1u
NRT_API nrt_status nrtTraverseBatchesDevice_f32(nrt_ctx *ctx, uint32_t num_batches, const nrt_ray_f32 *const *d_rays,
                                                const uint64_t *num_rays, const nrt_trace_options *options,
                                                nrt_hit_f32 *const *d_hits_out, uint8_t *const *d_masks_out,
                                                const uint32_t *batch_flags, void *hip_stream);
NRT_API nrt_status nrtTraverseBatchesDevice_f64(nrt_ctx *ctx, uint32_t num_batches, const nrt_ray_f64 *const *d_rays,
                                                const uint64_t *num_rays, const nrt_trace_options *options,
                                                nrt_hit_f64 *const *d_hits_out, uint8_t *const *d_masks_out,
                                                const uint32_t *batch_flags, void *hip_stream);

/* The same for HOST batches: every batch is uploaded next to the others, ONE launch walks them all, the records come back batch
 * by batch — what a host-shaded wavefront renderer submits together (the shadow query of one depth and the path wave of the
 * next), with one launch tail instead of `num_batches`.  Host pointers in and out, synchronous; at most 2^26 rays per call.
 * An occlusion batch needs its masks_out entry; a closest-hit batch its hits_out entry (masks_out[k] may be NULL).  Unlike the
 * reference's Traverse (nanort.h:1205-1211) — and like nrtTraverseBatch — every record of a closest-hit batch is written: a
 * miss stores {0, 0, ray.max_t, 0xFFFFFFFF}.  Records are exactly those of separate nrtTraverseBatch / nrtOccludedBatch calls. */
NRT_API nrt_status nrtTraverseBatches_f32(nrt_ctx *ctx, uint32_t num_batches, const nrt_ray_f32 *const *rays, const uint64_t *num_rays,
                                          const nrt_trace_options *options, nrt_hit_f32 *const *hits_out, uint8_t *const *masks_out,
                                          const uint32_t *batch_flags);
NRT_API nrt_status nrtTraverseBatches_f64(nrt_ctx *ctx, uint32_t num_batches, const nrt_ray_f64 *const *rays, const uint64_t *num_rays,
                                          const nrt_trace_options *options, nrt_hit_f64 *const *hits_out, uint8_t *const *masks_out,
                                          const uint32_t *batch_flags);

/* One HOST batch spread over several contexts — typically one per GPU of the node (nrtDeviceCount), each holding the same
 * tree: nrtBuild is deterministic, so building the same mesh on every context gives bit-identical replicas (or nrtSetTree the
 * same arrays).  The batch is cut into rows of `row_len` rays (an image row; 0 = 4096) and row r is traced by context
 * r % num_ctx — the interleaved image-tile split of SURVEY.md §8(e); each context is driven by its own host thread and its
 * GPU copies its rows of the records (and flags, when hit_mask_out is not NULL) straight into the caller's arrays, which are
 * written in full like nrtTraverseBatch's.  Records are exactly those of nrtTraverseBatch on one context.  Page-locked caller
 * buffers (nrtHostAlloc) let the copies of different GPUs overlap at PCIe speed.  Errors: NRT_ERR_INVALID when a context is
 * NULL, listed twice, of another precision, without a triangle tree or with another tree than context 0's.
 * (No reference counterpart: the reference's only parallel loop is the example's OpenMP row loop,
 * examples/path_tracer/main.cc:785-806.) */
NRT_API nrt_status nrtTraverseBatchMulti_f32(nrt_ctx *const *ctxs, uint32_t num_ctx, const nrt_ray_f32 *rays, uint64_t num_rays,
                                             uint64_t row_len, const nrt_trace_options *options, nrt_hit_f32 *hits_out,
                                             uint8_t *hit_mask_out);
NRT_API nrt_status nrtTraverseBatchMulti_f64(nrt_ctx *const *ctxs, uint32_t num_ctx, const nrt_ray_f64 *rays, uint64_t num_rays,
                                             uint64_t row_len, const nrt_trace_options *options, nrt_hit_f64 *hits_out,
                                             uint8_t *hit_mask_out);
/* HIP devices visible to this process (0 when there is none or the runtime cannot be initialised). */
NRT_API int nrtDeviceCount(void);

/* ---- multi-GPU with DEVICE-RESIDENT rays: tiles traced where they live, hit records gathered to one GPU by RCCL ----------
 * replicated BVH, image-tile split, RCCL gather of the 16 / 32-byte records over xGMI.  (No reference
 * counterpart: examples/path_tracer/main.cc:785-806 is the reference's only parallel loop.)  A frame of `total_rays` rays is
 * cut into rows of `row_len` rays (0 = 4096); tile t of N owns the interleaved rows t, t + N, t + 2N, ... in that order
 * (nrtGroupTileRays gives its size) — the split of nrtTraverseBatchMulti.  Every tile belongs to one context holding a replica
 * of the tree (nrtBuild is deterministic: the same mesh gives bit-identical replicas).
 *   nrtGroupCreate        ONE process drives all N contexts, normally one per GPU (several on one GPU work: their records are
 *                         read in place); one RCCL rank per distinct device.
 *   nrtGroupCreateRanked  one process per GPU: this process owns tile `rank` of `nranks`; `unique_id` = the 128 bytes rank 0
 *                         got from nrtGroupUniqueId, handed round by the host program.
 * RCCL is bound at run time (librccl.so.1 — the process's own copy when a framework has loaded one); when it cannot be, a
 * single-process group moves the records with hipMemcpyPeerAsync instead and nrtGroupLastError(group) says so.
 * nrtGroupTraverseGather_*: d_rays[k] / counts[k] describe this process's k-th tile (device pointers on that tile's GPU,
 * complete before the call).  Every tile is traced on a stream of the group (nrtTraverseBatchDevice_* semantics: every record
 * written, a miss = {0, 0, max_t, ~0}); tiles on other GPUs than the root tile's send their records (ncclSend / ncclRecv in one
 * group call); a kernel on the root GPU writes them to d_frame_hits[total_rays] (and flags to d_frame_mask, optional) in FRAME
 * order.  Only the process owning `root_tile` passes frame pointers (memory of the root tile's GPU); the others pass NULL.  The
 * call returns once everything is enqueued: nrtGroupSynchronize before reading the frame or reusing the ray buffers.  Frames
 * are byte-identical to nrtTraverseBatchDevice_* over the whole ray array on one context.
 * Tunables: "transport" (0 RCCL, 1 peer copies; single-process groups), "self_send" (1: the root tile's own records take the
 * send / receive path too — lets a one-GPU box execute the exchange). */
typedef struct nrt_group nrt_group;
NRT_API nrt_status nrtGroupUniqueId(void *id_out, size_t bytes /* >= 128 */);
NRT_API nrt_status nrtGroupCreate(nrt_ctx *const *ctxs, uint32_t num_ctx, nrt_group **out);
NRT_API nrt_status nrtGroupCreateRanked(nrt_ctx *ctx, const void *unique_id, int rank, int nranks, nrt_group **out);
NRT_API void nrtGroupDestroy(nrt_group *group);
NRT_API const char *nrtGroupLastError(const nrt_group *group); /* NULL: the calling thread's last creation error */
NRT_API nrt_status nrtGroupSetTunable(nrt_group *group, const char *name, long long value);
NRT_API nrt_status nrtGroupInfo(const nrt_group *group, uint32_t *num_tiles_out, uint32_t *num_local_out, int *nranks_out, int *rccl_bound_out);
NRT_API uint64_t nrtGroupTileRays(uint64_t total_rays, uint64_t row_len, uint32_t tile, uint32_t num_tiles);
NRT_API nrt_status nrtGroupTraverseGather_f32(nrt_group *group, const nrt_ray_f32 *const *d_rays, const uint64_t *counts, uint64_t total_rays,
                                              uint64_t row_len, const nrt_trace_options *options, uint32_t root_tile, nrt_hit_f32 *d_frame_hits,
                                              uint8_t *d_frame_mask);
NRT_API nrt_status nrtGroupTraverseGather_f64(nrt_group *group, const nrt_ray_f64 *const *d_rays, const uint64_t *counts, uint64_t total_rays,
                                              uint64_t row_len, const nrt_trace_options *options, uint32_t root_tile, nrt_hit_f64 *d_frame_hits,
                                              uint8_t *d_frame_mask);
/* Ragged waves (secondary rays: every tile has its own number of them) — TILE-MAJOR gather: tile t traces counts[k] <= slot_rays
 * rays and the root receives the tile's whole slot at d_tiles_hits + t * slot_rays (and d_tiles_mask + t * slot_rays, optional);
 * records past a tile's count are undefined.  No frame order to restore: RCCL receives straight into the caller's array. */
NRT_API nrt_status nrtGroupTraverseGatherTiles_f32(nrt_group *group, const nrt_ray_f32 *const *d_rays, const uint64_t *counts, uint64_t slot_rays,
                                                   const nrt_trace_options *options, uint32_t root_tile, nrt_hit_f32 *d_tiles_hits,
                                                   uint8_t *d_tiles_mask);
NRT_API nrt_status nrtGroupTraverseGatherTiles_f64(nrt_group *group, const nrt_ray_f64 *const *d_rays, const uint64_t *counts, uint64_t slot_rays,
                                                   const nrt_trace_options *options, uint32_t root_tile, nrt_hit_f64 *d_tiles_hits,
                                                   uint8_t *d_tiles_mask);
NRT_API nrt_status nrtGroupSynchronize(nrt_group *group);
/* bytes the last gather moved by RCCL / by peer copies / read in place on the root GPU */
NRT_API nrt_status nrtGroupLastTraffic(const nrt_group *group, uint64_t *bytes_rccl, uint64_t *bytes_peer, uint64_t *bytes_in_place);

/* Measurement aid: run the batch once on HBM-resident rays with the work
 * counters on (synchronous; results are not written).  Used by bench.py to
 * turn kernel time into algorithmic bytes per SURVEY.md §8(d). */
NRT_API nrt_status nrtTraverseCountDevice_f32(nrt_ctx *ctx, const nrt_ray_f32 *d_rays,
                                              uint64_t num_rays, const nrt_trace_options *options,
                                              nrt_trace_counters *counters_out);
NRT_API nrt_status nrtTraverseCountDevice_f64(nrt_ctx *ctx, const nrt_ray_f64 *d_rays,
                                              uint64_t num_rays, const nrt_trace_options *options,
                                              nrt_trace_counters *counters_out);

/* Device time (ms) of the most recent traversal launch (the kernel's own start / end stamps, or HIP events on the
 * launch stream: see nrtSetLaunchTiming) / build (HIP events) on this context; < 0 if none has completed.
 * Synchronises with that work: when it returns, the launch's stream has drained (the records have landed). */
NRT_API float nrtLastTraverseMs(nrt_ctx *ctx);
NRT_API float nrtLastBuildMs(nrt_ctx *ctx);
/* By default a traversal launch records NO event in the caller's stream (an event record between two kernels of a stream
 * keeps the second from starting for ~8 us; three per launch cost C3 6.5 % of a step): the kernel's last wave publishes a
 * completion record — sequence number, start and end stamps — in page-locked memory.  nrtLastTraverseMs reads those stamps
 * (first block started -> last wave finished), and whatever must wait for launches in flight (nrtBuild / nrtSetMesh /
 * nrtSetTree / nrtDestroy, a fifth stream launching concurrently on one context) waits for exactly those launches by polling
 * their records.  on = 1 brackets every launch with a pair of timing events and follows it with a completion event instead
 * (nrtLastTraverseMs then reports the event time, dispatch included) — a cross-check, not the fast path.  A launch made while
 * its stream is being CAPTURED into a graph does not run, so its record never completes: capture traversal launches only on
 * contexts that are not rebuilt or destroyed before the graph has been launched and has finished (or use on = 1 while capturing).  (For the sphere
 * and cylinder kinds the record is closed by their post pass; the literal BVHNode kernel always uses events.)
 * (No reference counterpart.) */
NRT_API nrt_status nrtSetLaunchTiming(nrt_ctx *ctx, int on);
/* Traversal / build tunables by name (no reference counterpart; the library's defaults are the measured optimum on MI355X):
 *   which walk        "wide" (0: the literal BVHNode loop), "wide4" (two tree levels per step; next build / set_tree),
 *                     "order4" (0, the default: a record's four slots in the binary loop's order, nanort.h:2538-2543 — the same
 *                     leaves in the same order as the reference, every field of every record bit-identical to it on the same
 *                     node array; 1, opt-in, 2...5 % faster: slots entered by entry distance — another leaf sequence, so among
 *                     primitives at EXACTLY the same t another one may be named (prim_id / u / v), and where a leaf box's entry
 *                     distance rounds above the distance of a hit inside it the box may be culled under another hit and t itself
 *                     differ by its last bits: contract-level parity, SURVEY.md §8d, not the bit-exact class), "f64_row_fetch"
 *   scheduling        "refill_min", "trav_min", "trav_min4", "leaf_min", "chunk", "chunk_tail_pct", "parts",
 *                     "static_pct", "static_bands", "static_slice_groups", "blocks_per_cu", "lds_stack", "wide_stack"
 *   builder           "morton" (Morton pre-pass); libnanort_hip_prof.so only: "subtree_rows" (0: the one-node-per-step subtree kernel; same tree)
 *   launches / host   "launch_timing" (== nrtSetLaunchTiming), "host_pipeline"
 *   probes            "debug" (bit mask: 1 / 2 skip triangle tests / traversal, 4 plain ray loads; the profiling bits 32 / 64 / 8192
 *                     act in libnanort_hip_prof.so only), "wide_scramble" (layout probe; next build)
 * Values are clamped to the tunable's range; an unknown name is NRT_ERR_INVALID.  Tunables that shape the private tree layout
 * ("wide4", "wide_scramble", "morton") take effect with the next nrtBuild / nrtSetTree.  The environment variable
 * NRT_<NAME> (upper case) overrides a default at nrtCreate ONLY when the process also sets NRT_ALLOW_ENV=1 (libnanort_hip_prof.so:
 * always) — a debugging aid; a stray variable cannot move a product context to another walk.  Programs use these calls. */
NRT_API nrt_status nrtSetTunable(nrt_ctx *ctx, const char *name, long long value);
NRT_API nrt_status nrtGetTunable(nrt_ctx *ctx, const char *name, long long *value_out);
/* Name of the traversal kernel variant the most recent traversal launch of this context used, spelled as
 * rocprofv3 prints it without the argument list (static storage; "" before the first launch).  bench.py
 * reports it in `roofline.kernel` and matches the counter rows of its PMC passes against it. */
NRT_API const char *nrtLastKernelName(const nrt_ctx *ctx);
/* (The profiling entry points — loop-occupancy counters, per-wave time stamps — are not part of this library: they live in
 * libnanort_hip_prof.so, declared in nanort_hip_prof.h, together with the counting / clocked kernel instantiations.) */

/* ---- two-level scenes (instancing): replaces nanosg::Scene<float, M> -----------------------
 * examples/nanosg/nanosg.h — AddNode :682, Commit :700-760 (per-node world AABB / inverse
 * transforms, nanosg.h:397-437), Traverse :773-870 on top of BVHAccel::ListNodeIntersections
 * (nanort.h:781-784, 2608-2692).  Semantics kept, including the reference's quirks: the 64 nearest
 * node boxes by entry distance are considered (kMaxIntersections), front to back with the early
 * cull `t_nearest < t_min`; the local ray carries the default [0, FLT_MAX] interval and default
 * trace options (the world ray's interval and the cull flag never reach the per-node Traverse);
 * the reported t is the WORLD distance |xform(P_local) - org| and a node replaces the current hit
 * only when strictly nearer.  Nodes entered at exactly the same distance are visited in node order (the
 * reference: in the order its std::priority_queue pops them), which only shows when several COINCIDENT
 * instances produce the same hit record: either may be named in node_id.  A node is a built nrt_ctx (f32) plus nanosg's T[4][4] local transform
 * (row 3 = translation, nanosg.h:232-240); the mesh contexts must outlive the scene, and nrtSceneCommit caches where their trees
 * live: after nrtBuild / nrtSetMesh / nrtSetTree on a mesh context the scene has to be committed again (until then
 * nrtSceneTraverseBatch* fail with NRT_ERR_INVALID instead of walking the old buffers). */
typedef struct nrt_scene nrt_scene;
typedef struct {
  float t;
  float u;
  float v;
  uint32_t prim_id;
  uint32_t node_id;
} nrt_scene_hit_f32; /* the fields of nanosg::Intersection<float> the traversal fills; miss: t = ray.max_t, ids = 0xFFFFFFFF */

NRT_API nrt_status nrtSceneCreate(int device, nrt_scene **out);
NRT_API void nrtSceneDestroy(nrt_scene *scene);
NRT_API const char *nrtSceneLastError(const nrt_scene *scene);
NRT_API nrt_status nrtSceneAddNode_f32(nrt_scene *scene, nrt_ctx *built_mesh, const float local_xform[16],
                                       uint32_t *node_id_out);
NRT_API nrt_status nrtSceneCommit(nrt_scene *scene);
/* The matrices Node::Update derives for node `node_id` (nanosg.h:397-437), valid after nrtSceneCommit: xform,
 * inv_xform, inv_xform33, inv_transpose_xform33 — 16 floats each (nanosg's row-major T[4][4]), 64 in all. */
NRT_API nrt_status nrtSceneNodeState_f32(nrt_scene *scene, uint32_t node_id, float out[64]);
/* Scene::GetBoundingBox (nanosg.h:761-769): union of the nodes' world boxes, valid after nrtSceneCommit. */
NRT_API nrt_status nrtSceneBounds_f32(nrt_scene *scene, float bmin[3], float bmax[3]);
NRT_API nrt_status nrtSceneTraverseBatch_f32(nrt_scene *scene, const nrt_ray_f32 *rays, uint64_t num_rays,
                                             nrt_scene_hit_f32 *hits_out, uint8_t *hit_mask_out);

/* The same traversal with rays and results resident in HBM (device pointers; d_mask_out may be NULL): no PCIe traffic.
 * On the scene's own stream: scenes of 64 nodes or more — ONE launch of the single-pass walk (top-level tree and instance trees on
 * one per-lane stack, no per-ray list), a 4-byte read-back, and only if the walk handed rays over, the two launches below on
 * those; scenes of at most 8 nodes — ONE launch (the trace kernel tests every world box itself as it fetches a ray); 9 to 63
 * nodes — two launches (the listing over the top-level BVH, then one trace kernel for the whole batch).  The call
 * is SYNCHRONOUS (the scene owns the per-ray scratch), and the caller makes sure `d_rays` is complete before calling. */
NRT_API nrt_status nrtSceneTraverseBatchDevice_f32(nrt_scene *scene, const nrt_ray_f32 *d_rays, uint64_t num_rays,
                                                   nrt_scene_hit_f32 *d_hits_out, uint8_t *d_mask_out);
/* Scheduling knobs of the scene kernels by name (they never change a result): "single_pass" (1: scenes of 64 nodes or more
 * are traced by the single-pass walk — top-level tree and instance trees on one stack, no per-ray list; rays it cannot certify
 * are re-done by the listing path; 2: every scene of two nodes or more; 0: listing + trace for every ray), "trav_min", "refill_min"
 * (listing path), "walk_trav_min", "walk_refill_min" (the walk's), "cand_min",
 * "cand_busy_max" (lane-count thresholds of the phases), "prune_min" (instance count from which the listing prunes beyond a
 * full list), "walk_min" (instance count from which single_pass = 1 uses the walk; 64), "scan_max" (scenes of at most this many
 * nodes get no top-level tree: every world box is tested per ray; 8, at most 64; takes effect at the next nrtSceneCommit),
 * "fuse_scan" (1: such a scene is listed inside the trace kernel, one launch; 0: by a listing launch of its own).  After a batch of which the walk had to
 * hand more than a quarter ("walk_backoff_pct", 25) to the listing path (direction vectors far shorter than 1, where the reference's cull compares a
 * distance with a parameter) the next 15 calls use the listing path directly. */
NRT_API nrt_status nrtSceneSetTunable(nrt_scene *scene, const char *name, int value);
/* How many rays of the last nrtSceneTraverseBatch* call the single-pass walk handed to the listing path. */
NRT_API uint64_t nrtSceneLastRedone(const nrt_scene *scene);
/* Which path the last nrtSceneTraverseBatch* call took: 1 = the single-pass walk (plus the listing path for the rays it handed
 * over), 0 = the listing path alone (small scenes, single_pass = 0, a mesh whose tree the walk cannot step through, back-off). */
NRT_API int nrtSceneLastPath(const nrt_scene *scene);

#ifdef __cplusplus
}
#endif
#endif /* NANORT_HIP_H_ */

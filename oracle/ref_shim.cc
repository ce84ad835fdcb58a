// oracle/ref_shim.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A C-ABI wrapper around the UNMODIFIED reference header, which is included
// from where it lies (-I/root/reference) and never copied into this repo.
// The recipe in oracle/Makefile compiles this file into
// oracle/_ref/libnanort_ref.so (git-ignored; it travels to the GPU box as a
// prebuilt binary because /root/reference does not exist there).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load it, and only as the checker / the timed CPU baseline ("kind":
// "reference").  Nothing under nanort_amd/ may depend on it.
//
// What it exposes is exactly the reference's public path:
//   BVHAccel<T>::Build      nanort.h:1892-2149   (serial or OpenMP 2-phase)
//   BVHAccel<T>::Traverse   nanort.h:2487-2556   (one call per ray, OpenMP
//                           row loop as in examples/path_tracer/main.cc:801-804)
//   BVHAccel<T>::Load(FILE*) nanort.h:2253-2275  (to make the reference walk a
//                           node array produced elsewhere, e.g. by the GPU)
//   GetNodes/GetIndices/GetStatistics  nanort.h:725,786-787
//
// Compile-time configuration mirrors examples/path_tracer/Makefile.omp
// (-O3 -fopenmp, no NANORT_USE_CPP11_FEATURE => the `v < 0` sign rule of
// vsafe_inverse, nanort.h:442-461) plus NANORT_ENABLE_PARALLEL_BUILD and
// NANORT_ENABLE_SERIALIZATION.

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "nanort.h"  // the reference, via -I/root/reference

namespace {

template <typename T>
struct RefAccel {
  const T *vertices;
  const unsigned int *faces;
  size_t stride;
  unsigned int num_faces;
  nanort::BVHAccel<T> accel;
};

struct ShimBuildOptions {  // field-for-field nanort::BVHBuildOptions minus cost_t_aabb
  uint32_t min_leaf_primitives;
  uint32_t max_tree_depth;
  uint32_t bin_size;
  uint32_t shallow_depth;
  uint32_t min_primitives_for_parallel_build;
  uint32_t cache_bbox;
};

struct ShimStats {
  uint32_t max_tree_depth;
  uint32_t num_leaf_nodes;
  uint32_t num_branch_nodes;
  float build_secs;  // measured by the shim (the reference never fills it)
};

struct ShimTraceOptions {  // == nanort::BVHTraceOptions, 16 bytes
  uint32_t prim_ids_range[2];
  uint32_t skip_prim_id;
  uint8_t cull_back_face;
  uint8_t pad[3];
};

template <typename T>
void *Create(const T *v, size_t stride, const uint32_t *f, uint32_t nf) {
  RefAccel<T> *a = new RefAccel<T>();
  a->vertices = v;
  a->faces = f;
  a->stride = stride;
  a->num_faces = nf;
  return a;
}

template <typename T>
int Build(void *h, const ShimBuildOptions *o, ShimStats *st, int num_threads) {
  RefAccel<T> *a = static_cast<RefAccel<T> *>(h);
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
  nanort::BVHBuildOptions<T> opt;
  if (o) {
    opt.min_leaf_primitives = o->min_leaf_primitives;
    opt.max_tree_depth = o->max_tree_depth;
    opt.bin_size = o->bin_size;
    opt.shallow_depth = o->shallow_depth;
    opt.min_primitives_for_parallel_build = o->min_primitives_for_parallel_build;
    opt.cache_bbox = o->cache_bbox != 0;
  }
  nanort::TriangleMesh<T> mesh(a->vertices, a->faces, a->stride);
  nanort::TriangleSAHPred<T> pred(a->vertices, a->faces, a->stride);
  auto t0 = std::chrono::steady_clock::now();
  bool ok = a->accel.Build(a->num_faces, mesh, pred, opt);
  auto t1 = std::chrono::steady_clock::now();
  if (st) {
    nanort::BVHBuildStatistics s = a->accel.GetStatistics();
    st->max_tree_depth = s.max_tree_depth;
    st->num_leaf_nodes = s.num_leaf_nodes;
    st->num_branch_nodes = s.num_branch_nodes;
    st->build_secs = std::chrono::duration<float>(t1 - t0).count();
  }
  return ok ? 1 : 0;
}

template <typename T>
int LoadTree(void *h, const void *nodes, uint64_t num_nodes,
             const uint32_t *indices, uint64_t num_indices) {
  RefAccel<T> *a = static_cast<RefAccel<T> *>(h);
  // Serialise into the reference's Dump format (nanort.h:2197-2216) in memory
  // and let the reference's own Load(FILE*) read it back.
  size_t bytes = 2 * sizeof(size_t) + num_nodes * sizeof(nanort::BVHNode<T>) +
                 num_indices * sizeof(unsigned int);
  std::vector<unsigned char> buf(bytes);
  unsigned char *p = buf.data();
  size_t nn = num_nodes, ni = num_indices;
  memcpy(p, &nn, sizeof(size_t));
  p += sizeof(size_t);
  memcpy(p, nodes, nn * sizeof(nanort::BVHNode<T>));
  p += nn * sizeof(nanort::BVHNode<T>);
  memcpy(p, &ni, sizeof(size_t));
  p += sizeof(size_t);
  memcpy(p, indices, ni * sizeof(unsigned int));
  FILE *fp = fmemopen(buf.data(), bytes, "rb");
  if (!fp) return 0;
  bool ok = a->accel.Load(fp);
  fclose(fp);
  return ok ? 1 : 0;
}

template <typename T>
double TraverseBatch(void *h, const nanort::Ray<T> *rays, uint64_t n,
                     const ShimTraceOptions *o,
                     nanort::TriangleIntersection<T> *hits, uint8_t *mask,
                     int num_threads, int chunk) {
  RefAccel<T> *a = static_cast<RefAccel<T> *>(h);
  nanort::BVHTraceOptions opt;
  if (o) memcpy(&opt, o, sizeof(opt));
  if (chunk < 1) chunk = 1;
  const int64_t num_chunks = (int64_t)((n + (uint64_t)chunk - 1) / (uint64_t)chunk);
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
  auto t0 = std::chrono::steady_clock::now();
  // One "row" (chunk of consecutive rays) per dynamic work item, as the
  // reference's own render loop does (examples/path_tracer/main.cc:801-804).
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int64_t c = 0; c < num_chunks; c++) {
    uint64_t b = (uint64_t)c * (uint64_t)chunk;
    uint64_t e = b + (uint64_t)chunk;
    if (e > n) e = n;
    for (uint64_t i = b; i < e; i++) {
      // One intersector per ray: its scratch is mutable (nanort.h:1220-1228).
      nanort::TriangleIntersector<T, nanort::TriangleIntersection<T> > isector(
          a->vertices, a->faces, a->stride);
      nanort::TriangleIntersection<T> isect;
      memset(&isect, 0, sizeof(isect));
      isect.prim_id = 0xFFFFFFFFu;
      isect.t = rays[i].max_t;
      bool hit = a->accel.Traverse(rays[i], isector, &isect, opt);
      hits[i] = isect;
      if (mask) mask[i] = hit ? 1 : 0;
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // namespace

extern "C" {

int ref_sizeof(int what) {
  switch (what) {
    case 0: return (int)sizeof(nanort::Ray<float>);
    case 1: return (int)sizeof(nanort::Ray<double>);
    case 2: return (int)sizeof(nanort::BVHNode<float>);
    case 3: return (int)sizeof(nanort::BVHNode<double>);
    case 4: return (int)sizeof(nanort::TriangleIntersection<float>);
    case 5: return (int)sizeof(nanort::TriangleIntersection<double>);
    case 6: return (int)sizeof(nanort::BVHBuildOptions<float>);
    case 7: return (int)sizeof(nanort::BVHBuildOptions<double>);
    case 8: return (int)sizeof(nanort::BVHTraceOptions);
    case 9: return (int)sizeof(nanort::BVHBuildStatistics);
  }
  return -1;
}

int ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

#define REF_INSTANTIATE(SUF, T)                                                         \
  void *ref_create_##SUF(const T *v, size_t stride, const uint32_t *f, uint32_t nf) {   \
    return Create<T>(v, stride, f, nf);                                                 \
  }                                                                                     \
  void ref_destroy_##SUF(void *h) { delete static_cast<RefAccel<T> *>(h); }             \
  int ref_build_##SUF(void *h, const ShimBuildOptions *o, ShimStats *st, int threads) { \
    return Build<T>(h, o, st, threads);                                                 \
  }                                                                                     \
  uint64_t ref_num_nodes_##SUF(void *h) {                                               \
    return static_cast<RefAccel<T> *>(h)->accel.GetNodes().size();                      \
  }                                                                                     \
  uint64_t ref_num_indices_##SUF(void *h) {                                             \
    return static_cast<RefAccel<T> *>(h)->accel.GetIndices().size();                    \
  }                                                                                     \
  void ref_get_tree_##SUF(void *h, void *nodes_out, uint32_t *indices_out) {            \
    RefAccel<T> *a = static_cast<RefAccel<T> *>(h);                                     \
    const std::vector<nanort::BVHNode<T> > &n = a->accel.GetNodes();                    \
    const std::vector<unsigned int> &ix = a->accel.GetIndices();                        \
    if (nodes_out && !n.empty())                                                        \
      memcpy(nodes_out, &n[0], n.size() * sizeof(nanort::BVHNode<T>));                  \
    if (indices_out && !ix.empty())                                                     \
      memcpy(indices_out, &ix[0], ix.size() * sizeof(unsigned int));                    \
  }                                                                                     \
  int ref_load_tree_##SUF(void *h, const void *nodes, uint64_t nn, const uint32_t *ix,  \
                          uint64_t ni) {                                                \
    return LoadTree<T>(h, nodes, nn, ix, ni);                                           \
  }                                                                                     \
  void ref_bounding_box_##SUF(void *h, T *bmin, T *bmax) {                              \
    static_cast<RefAccel<T> *>(h)->accel.BoundingBox(bmin, bmax);                       \
  }                                                                                     \
  double ref_traverse_##SUF(void *h, const void *rays, uint64_t n,                      \
                            const ShimTraceOptions *o, void *hits, uint8_t *mask,       \
                            int num_threads, int chunk) {                               \
    return TraverseBatch<T>(h, static_cast<const nanort::Ray<T> *>(rays), n, o,         \
                            static_cast<nanort::TriangleIntersection<T> *>(hits), mask, \
                            num_threads, chunk);                                        \
  }

REF_INSTANTIATE(f32, float)
REF_INSTANTIATE(f64, double)

}  // extern "C"

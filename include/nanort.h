// include/nanort.h — header-only host side of the MI355X ray-tracing kernel.
//
// API-compatible with lighttransport/nanort's `nanort.h` for the hot path
// (BVHAccel<T>::Build / Traverse with Ray, BVHNode, TriangleMesh,
// TriangleSAHPred, TriangleIntersector<>, TriangleIntersection and the option
// structs), so the reference's examples (path_tracer, objrender,
// double_precision, the regression program, the custom-primitive demos)
// compile against it unchanged.  It is written from scratch: only names,
// argument meaning, struct layouts and result semantics follow the reference
// (cited as `ref nanort.h:LINE`).
//
// Two execution paths:
//
//   * Generic host path — any Prim / Pred / Intersector that satisfies the
//     reference's concepts (ref nanort.h:716-718, 757-759).  Needed for custom
//     primitives and for the per-ray Traverse() the reference API exposes.
//
//   * NANORT_USE_HIP_BACKEND — when Build() is called with the built-in
//     TriangleMesh<T> + TriangleSAHPred<T>, construction runs on the GPU through
//     the C ABI of libnanort_hip.so (include/nanort_hip.h): the node array and
//     index permutation stay on the device and are copied into nodes_/indices_
//     by the first host access, so GetNodes(), Dump(), BoundingBox() and the
//     per-ray Traverse() keep working while an application that only calls
//     TraverseBatch() never pays the read-back.  The one
//     API addition, TraverseBatch(), intersects N rays in one GPU launch; it
//     does not exist without the backend (there is no CPU stand-in for it).
//     Link with -lnanort_hip.
//
// Macros accepted for source compatibility: NANORT_USE_CPP11_FEATURE,
// NANORT_ENABLE_PARALLEL_BUILD (with OpenMP or NANORT_USE_CPP11_FEATURE the generic
// host build of user primitives runs in parallel and yields the same tree as the serial
// one; the built-in primitive types build on the GPU), NANORT_ENABLE_SERIALIZATION.
#ifndef NANORT_H_
#define NANORT_H_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <queue>
#include <string>
#include <type_traits>
#include <vector>

#ifdef NANORT_USE_HIP_BACKEND
#include <mutex>

#include "nanort_hip.h"
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#define kNANORT_MAX_STACK_DEPTH (512)
#define kNANORT_MIN_PRIMITIVES_FOR_PARALLEL_BUILD (1024 * 8)
#define kNANORT_SHALLOW_DEPTH (4)
#ifdef NANORT_USE_CPP11_FEATURE
// the reference's header pulls these in under this macro and its examples rely on that
#include <atomic>
#include <mutex>
#include <thread>
#define kNANORT_MAX_THREADS (256)
#ifndef NANORT_ENABLE_PARALLEL_BUILD
#define NANORT_ENABLE_PARALLEL_BUILD
#endif
#endif

namespace nanort {

namespace detail {

// Worker threads for the host-side loops of this header (the parallel host build, the hit scatter of TraverseBatch):
// what the process may actually use — the cgroup CPU quota when there is one (an OpenMP default of "every hardware
// thread" inside a quota'd container collapses, and starves the HIP runtime's own threads), else the OpenMP / hardware
// thread count — clamped to [1, 64].
inline unsigned int HostThreadsUncached() {
  unsigned int n = 1;
#if defined(_OPENMP)
  n = static_cast<unsigned int>(std::max(1, omp_get_max_threads()));
#elif defined(NANORT_USE_CPP11_FEATURE)
  n = std::max(1u, std::thread::hardware_concurrency());
#endif
  long quota = -1, period = 0;
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[32];
    if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
    std::fclose(f);
  } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
    std::fclose(g);
    if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(h, "%ld", &period) != 1) period = 0;
      std::fclose(h);
    }
  }
  if (quota > 0 && period > 0 && static_cast<unsigned long>(quota / period) >= 1 && static_cast<unsigned long>(quota / period) < n)
    n = static_cast<unsigned int>(quota / period);
  return std::min(64u, std::max(1u, n));
}
inline unsigned int HostThreads() {
  static const unsigned int cached = HostThreadsUncached();  // (function-local static: initialised once, thread-safely)
  return cached;
}

// f(i) for i in [0, n) on up to `threads` workers, dynamically scheduled; serial when the header was compiled without
// OpenMP and without NANORT_USE_CPP11_FEATURE (the reference's two ways of building in parallel, nanort.h:2018-2117).
template <class F>
inline void ParallelFor(unsigned int n, unsigned int threads, const F &f) {
  if (threads > n) threads = n;
#if defined(_OPENMP)
  if (threads > 1) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long long i = 0; i < static_cast<long long>(n); i++) f(static_cast<unsigned int>(i));
    return;
  }
#elif defined(NANORT_USE_CPP11_FEATURE)
  if (threads > 1) {
    std::atomic<unsigned int> next(0);
    std::vector<std::thread> pool;
    for (unsigned int t = 0; t < threads; t++)
      pool.push_back(std::thread([&]() {
        for (unsigned int i = next++; i < n; i = next++) f(i);
      }));
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    return;
  }
#endif
  for (unsigned int i = 0; i < n; i++) f(i);
}

}  // namespace detail


typedef enum {
  RAY_TYPE_NONE = 0x0,
  RAY_TYPE_PRIMARY = 0x1,
  RAY_TYPE_SECONDARY = 0x2,
  RAY_TYPE_DIFFUSE = 0x4,
  RAY_TYPE_REFLECTION = 0x8,
  RAY_TYPE_REFRACTION = 0x10
} RayType;

// ---------------------------------------------------------------------------
// Small vector with the access idiom the reference's two-level traversal uses
// (`(*v)->clear()`, `v->size()`, `v[i]`; ref nanort.h:134-317, 784).  Capacity
// is reserved up front so the common case never reallocates.
// ---------------------------------------------------------------------------
template <typename TElem, size_t stack_capacity>
class StackVector {
 public:
  typedef std::vector<TElem> ContainerType;
  StackVector() { items_.reserve(stack_capacity); }
  StackVector(const StackVector &rhs) : items_(rhs.items_) { items_.reserve(stack_capacity); }
  StackVector &operator=(const StackVector &rhs) {
    items_ = rhs.items_;
    return *this;
  }
  ContainerType &container() { return items_; }
  const ContainerType &container() const { return items_; }
  ContainerType *operator->() { return &items_; }
  const ContainerType *operator->() const { return &items_; }
  TElem &operator[](size_t i) { return items_[i]; }
  const TElem &operator[](size_t i) const { return items_[i]; }

 private:
  ContainerType items_;
};

// ---------------------------------------------------------------------------
// 3-vector and helpers (ref nanort.h:321-472)
// ---------------------------------------------------------------------------
template <typename T = float>
class real3 {
 public:
  real3() {}
  real3(T s) { v[0] = v[1] = v[2] = s; }
  real3(T a, T b, T c) {
    v[0] = a;
    v[1] = b;
    v[2] = c;
  }
  explicit real3(const T *p) {
    v[0] = p[0];
    v[1] = p[1];
    v[2] = p[2];
  }
  T x() const { return v[0]; }
  T y() const { return v[1]; }
  T z() const { return v[2]; }
  real3 operator*(T s) const { return real3(v[0] * s, v[1] * s, v[2] * s); }
  real3 operator*(const real3 &o) const { return real3(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
  real3 operator/(const real3 &o) const { return real3(v[0] / o.v[0], v[1] / o.v[1], v[2] / o.v[2]); }
  real3 operator+(const real3 &o) const { return real3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  real3 operator-(const real3 &o) const { return real3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  real3 operator-() const { return real3(-v[0], -v[1], -v[2]); }
  real3 &operator+=(const real3 &o) {
    v[0] += o.v[0];
    v[1] += o.v[1];
    v[2] += o.v[2];
    return *this;
  }
  T operator[](int i) const { return v[i]; }
  T &operator[](int i) { return v[i]; }

  T v[3];
};

template <typename T>
inline real3<T> operator*(T s, const real3<T> &a) {
  return real3<T>(a[0] * s, a[1] * s, a[2] * s);
}
template <typename T>
inline real3<T> vneg(const real3<T> &a) {
  return real3<T>(-a[0], -a[1], -a[2]);
}
template <typename T>
inline T vdot(const real3<T> a, const real3<T> b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <typename T>
inline T vlength(const real3<T> &a) {
  return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}
template <typename T>
inline real3<T> vnormalize(const real3<T> &a) {
  real3<T> r = a;
  const T len = vlength(a);
  if (std::fabs(len) > std::numeric_limits<T>::epsilon()) {
    const T inv = static_cast<T>(1.0) / len;
    r[0] *= inv;
    r[1] *= inv;
    r[2] *= inv;
  }
  return r;
}
template <typename T>
inline real3<T> vcross(const real3<T> a, const real3<T> b) {
  return real3<T>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// Reciprocal that maps near-zero components to a signed infinity
// (ref nanort.h:414-465; both sign rules of the reference are kept).
template <typename T>
inline real3<T> vsafe_inverse(const real3<T> d) {
  real3<T> r;
  for (int k = 0; k < 3; k++) {
    if (std::fabs(d[k]) < std::numeric_limits<T>::epsilon()) {
#ifdef NANORT_USE_CPP11_FEATURE
      r[k] = std::numeric_limits<T>::infinity() * std::copysign(static_cast<T>(1), d[k]);
#else
      r[k] = std::numeric_limits<T>::infinity() * ((d[k] < static_cast<T>(0)) ? static_cast<T>(-1) : static_cast<T>(1));
#endif
    } else {
      r[k] = static_cast<T>(1.0) / d[k];
    }
  }
  return r;
}

template <typename real>
inline const real *get_vertex_addr(const real *p, const size_t idx, const size_t stride_bytes) {
  return reinterpret_cast<const real *>(reinterpret_cast<const unsigned char *>(p) + idx * stride_bytes);
}

// Comparison-based min/max that drop a NaN first argument (ref nanort.h:1236-1243).
template <class T>
const T &safemin(const T &a, const T &b) {
  return (a < b) ? a : b;
}
template <class T>
const T &safemax(const T &a, const T &b) {
  return (a > b) ? a : b;
}

// ---------------------------------------------------------------------------
// Wire-format PODs (layouts pinned below; ref nanort.h:474-639)
// ---------------------------------------------------------------------------
template <typename T = float>
class Ray {
 public:
  Ray() : min_t(static_cast<T>(0.0)), max_t(std::numeric_limits<T>::max()), type(RAY_TYPE_NONE) {
    org[0] = org[1] = org[2] = static_cast<T>(0.0);
    dir[0] = dir[1] = static_cast<T>(0.0);
    dir[2] = static_cast<T>(-1.0);
  }
  T org[3];
  T dir[3];
  T min_t;
  T max_t;
  unsigned int type;
};

template <typename T = float>
class BVHNode {
 public:
  BVHNode() {}
  T bmin[3];
  T bmax[3];
  int flag;  // 1 = leaf, 0 = branch
  int axis;
  // leaf: {primitive count, first slot in the index array}; branch: {low child, high child}
  unsigned int data[2];
};

template <class H>
class IntersectComparator {
 public:
  bool operator()(const H &a, const H &b) const { return a.t < b.t; }
};

template <typename T = float>
struct BVHBuildOptions {
  T cost_t_aabb;
  unsigned int min_leaf_primitives;
  unsigned int max_tree_depth;
  unsigned int bin_size;
  unsigned int shallow_depth;
  unsigned int min_primitives_for_parallel_build;
  bool cache_bbox;
  unsigned char pad[3];
  BVHBuildOptions()
      : cost_t_aabb(static_cast<T>(0.2)),
        min_leaf_primitives(4),
        max_tree_depth(256),
        bin_size(64),
        shallow_depth(kNANORT_SHALLOW_DEPTH),
        min_primitives_for_parallel_build(kNANORT_MIN_PRIMITIVES_FOR_PARALLEL_BUILD),
        cache_bbox(false) {
    pad[0] = pad[1] = pad[2] = 0;
  }
};

class BVHBuildStatistics {
 public:
  unsigned int max_tree_depth;
  unsigned int num_leaf_nodes;
  unsigned int num_branch_nodes;
  float build_secs;
  BVHBuildStatistics() : max_tree_depth(0), num_leaf_nodes(0), num_branch_nodes(0), build_secs(0.0f) {}
};

class BVHTraceOptions {
 public:
  unsigned int prim_ids_range[2];  // half-open [first, last)
  unsigned int skip_prim_id;       // 0xFFFFFFFF: skip nothing
  bool cull_back_face;
  unsigned char pad[3];
  BVHTraceOptions() {
    prim_ids_range[0] = 0;
    prim_ids_range[1] = 0x7FFFFFFF;
    skip_prim_id = static_cast<unsigned int>(-1);
    cull_back_face = false;
    pad[0] = pad[1] = pad[2] = 0;
  }
};

template <typename T>
class BBox {
 public:
  real3<T> bmin;
  real3<T> bmax;
  BBox() {
    bmin[0] = bmin[1] = bmin[2] = std::numeric_limits<T>::max();
    bmax[0] = bmax[1] = bmax[2] = -std::numeric_limits<T>::max();
  }
};

template <typename T>
class NodeHit {
 public:
  NodeHit()
      : t_min(std::numeric_limits<T>::max()), t_max(-std::numeric_limits<T>::max()), node_id(static_cast<unsigned int>(-1)) {}
  T t_min;
  T t_max;
  unsigned int node_id;
};

template <typename T>
class NodeHitComparator {
 public:
  inline bool operator()(const NodeHit<T> &a, const NodeHit<T> &b) { return a.t_min < b.t_min; }
};

// ---------------------------------------------------------------------------
// Built-in triangle plugin
// ---------------------------------------------------------------------------

// Partition predicate (ref nanort.h:863-919): sum of the three vertex
// coordinates on `axis` against 3*pos.
template <typename T = float>
class TriangleSAHPred {
 public:
  TriangleSAHPred(const T *vertices, const unsigned int *faces, size_t vertex_stride_bytes)
      : axis_(0), pos_(static_cast<T>(0.0)), vertices_(vertices), faces_(faces), vertex_stride_bytes_(vertex_stride_bytes) {}
  TriangleSAHPred(const TriangleSAHPred<T> &o)
      : axis_(o.axis_), pos_(o.pos_), vertices_(o.vertices_), faces_(o.faces_), vertex_stride_bytes_(o.vertex_stride_bytes_) {}
  TriangleSAHPred<T> &operator=(const TriangleSAHPred<T> &o) {
    axis_ = o.axis_;
    pos_ = o.pos_;
    vertices_ = o.vertices_;
    faces_ = o.faces_;
    vertex_stride_bytes_ = o.vertex_stride_bytes_;
    return *this;
  }
  void Set(int axis, T pos) const {
    axis_ = axis;
    pos_ = pos;
  }
  bool operator()(unsigned int i) const {
    const int a = axis_;
    const T s = get_vertex_addr<T>(vertices_, faces_[3 * i + 0], vertex_stride_bytes_)[a] +
                get_vertex_addr<T>(vertices_, faces_[3 * i + 1], vertex_stride_bytes_)[a] +
                get_vertex_addr<T>(vertices_, faces_[3 * i + 2], vertex_stride_bytes_)[a];
    return s < pos_ * static_cast<T>(3.0);
  }
  const T *GetVertices() const { return vertices_; }
  const unsigned int *GetFaces() const { return faces_; }
  size_t GetVertexStrideBytes() const { return vertex_stride_bytes_; }

 private:
  mutable int axis_;
  mutable T pos_;
  const T *vertices_;
  const unsigned int *faces_;
  size_t vertex_stride_bytes_;
};

// Primitive accessor (ref nanort.h:922-991).  Carries no vertex count.
template <typename T = float>
class TriangleMesh {
 public:
  TriangleMesh(const T *vertices, const unsigned int *faces, const size_t vertex_stride_bytes)
      : vertices_(vertices), faces_(faces), vertex_stride_bytes_(vertex_stride_bytes) {}

  void BoundingBox(real3<T> *bmin, real3<T> *bmax, unsigned int prim_index) const {
    const T *p = get_vertex_addr<T>(vertices_, faces_[3 * prim_index], vertex_stride_bytes_);
    for (int k = 0; k < 3; k++) (*bmin)[k] = (*bmax)[k] = p[k];
    for (unsigned int c = 1; c < 3; c++) {
      p = get_vertex_addr<T>(vertices_, faces_[3 * prim_index + c], vertex_stride_bytes_);
      for (int k = 0; k < 3; k++) {
        (*bmin)[k] = std::min((*bmin)[k], p[k]);
        (*bmax)[k] = std::max((*bmax)[k], p[k]);
      }
    }
  }

  void BoundingBoxAndCenter(real3<T> *bmin, real3<T> *bmax, real3<T> *center, unsigned int prim_index) const {
    const real3<T> a(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 0], vertex_stride_bytes_));
    const real3<T> b(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 1], vertex_stride_bytes_));
    const real3<T> c(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 2], vertex_stride_bytes_));
    for (int k = 0; k < 3; k++) {
      (*bmin)[k] = std::min(a[k], std::min(b[k], c[k]));
      (*bmax)[k] = std::max(a[k], std::max(b[k], c[k]));
    }
    *center = (a + b + c) * (T(1) / T(3));
  }

  const T *GetVertices() const { return vertices_; }
  const unsigned int *GetFaces() const { return faces_; }
  size_t GetVertexStrideBytes() const { return vertex_stride_bytes_; }

  const T *vertices_;
  const unsigned int *faces_;
  const size_t vertex_stride_bytes_;
};

template <typename T = float>
class TriangleIntersection {
 public:
  T u;
  T v;
  T t;
  unsigned int prim_id;
};

// Watertight ray/triangle test (Woop, Benthin, Wald 2013), same operation
// order and tie rules as the reference's intersector (ref nanort.h:1014-1229):
// edge functions recomputed from double products when one of them is exactly
// zero; a candidate replaces the current hit when its t is <= the best t and
// >= the ray's min_t.
template <typename T = float, class H = TriangleIntersection<T> >
class TriangleIntersector {
 public:
  template <class M>
  TriangleIntersector(const M &m) : vertices_(m.GetVertices()), faces_(m.GetFaces()), vertex_stride_bytes_(m.GetVertexStrideBytes()) {}
  template <class M>
  TriangleIntersector(const M *m) : vertices_(m->GetVertices()), faces_(m->GetFaces()), vertex_stride_bytes_(m->GetVertexStrideBytes()) {}
  TriangleIntersector(const T *vertices, const unsigned int *faces, const size_t vertex_stride_bytes)
      : vertices_(vertices), faces_(faces), vertex_stride_bytes_(vertex_stride_bytes) {}

  typedef struct {
    T Sx, Sy, Sz;
    int kx, ky, kz;
  } RayCoeff;

  bool Intersect(T *t_inout, const unsigned int prim_index) const {
    if (prim_index < trace_options_.prim_ids_range[0] || prim_index >= trace_options_.prim_ids_range[1]) return false;
    if (prim_index == trace_options_.skip_prim_id) return false;

    const real3<T> A = real3<T>(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 0], vertex_stride_bytes_)) - ray_org_;
    const real3<T> B = real3<T>(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 1], vertex_stride_bytes_)) - ray_org_;
    const real3<T> C = real3<T>(get_vertex_addr<T>(vertices_, faces_[3 * prim_index + 2], vertex_stride_bytes_)) - ray_org_;
    const int kx = ray_coeff_.kx, ky = ray_coeff_.ky, kz = ray_coeff_.kz;

    const T Ax = A[kx] - ray_coeff_.Sx * A[kz], Ay = A[ky] - ray_coeff_.Sy * A[kz];
    const T Bx = B[kx] - ray_coeff_.Sx * B[kz], By = B[ky] - ray_coeff_.Sy * B[kz];
    const T Cx = C[kx] - ray_coeff_.Sx * C[kz], Cy = C[ky] - ray_coeff_.Sy * C[kz];

    T U = Cx * By - Cy * Bx;
    T V = Ax * Cy - Ay * Cx;
    T W = Bx * Ay - By * Ax;
    const T zero = static_cast<T>(0.0);
    if (U == zero || V == zero || W == zero) {
      U = static_cast<T>(static_cast<double>(Cx) * static_cast<double>(By) - static_cast<double>(Cy) * static_cast<double>(Bx));
      V = static_cast<T>(static_cast<double>(Ax) * static_cast<double>(Cy) - static_cast<double>(Ay) * static_cast<double>(Cx));
      W = static_cast<T>(static_cast<double>(Bx) * static_cast<double>(Ay) - static_cast<double>(By) * static_cast<double>(Ax));
    }
    if (U < zero || V < zero || W < zero) {
      if (trace_options_.cull_back_face || U > zero || V > zero || W > zero) return false;
    }
    const T det = U + V + W;
    if (det == zero) return false;

    const T Az = ray_coeff_.Sz * A[kz], Bz = ray_coeff_.Sz * B[kz], Cz = ray_coeff_.Sz * C[kz];
    const T D = U * Az + V * Bz + W * Cz;
    const T rcp_det = static_cast<T>(1.0) / det;
    const T tt = D * rcp_det;
    if (tt > (*t_inout)) return false;
    if (tt < t_min_) return false;
    (*t_inout) = tt;
    u_ = V * rcp_det;
    v_ = W * rcp_det;
    return true;
  }

  T GetT() const { return t_; }

  void Update(T t, unsigned int prim_idx) const {
    t_ = t;
    prim_id_ = prim_idx;
  }

  void PrepareTraversal(const Ray<T> &ray, const BVHTraceOptions &trace_options) const {
    ray_org_ = real3<T>(ray.org[0], ray.org[1], ray.org[2]);
    int kz = 0;
    T longest = std::fabs(ray.dir[0]);
    for (int k = 1; k < 3; k++) {
      if (longest < std::fabs(ray.dir[k])) {  // strict: ties keep the lower axis
        kz = k;
        longest = std::fabs(ray.dir[k]);
      }
    }
    int kx = (kz + 1) % 3, ky = (kz + 2) % 3;
    if (ray.dir[kz] < static_cast<T>(0.0)) std::swap(kx, ky);  // keep the winding
    ray_coeff_.kx = kx;
    ray_coeff_.ky = ky;
    ray_coeff_.kz = kz;
    ray_coeff_.Sx = ray.dir[kx] / ray.dir[kz];
    ray_coeff_.Sy = ray.dir[ky] / ray.dir[kz];
    ray_coeff_.Sz = static_cast<T>(1.0) / ray.dir[kz];
    trace_options_ = trace_options;
    t_min_ = ray.min_t;
    u_ = v_ = static_cast<T>(0.0);
  }

  void PostTraversal(const Ray<T> &ray, bool hit, H *isect) const {
    (void)ray;
    if (hit && isect) {
      isect->t = t_;
      isect->u = u_;
      isect->v = v_;
      isect->prim_id = prim_id_;
    }
  }

 private:
  const T *vertices_;
  const unsigned int *faces_;
  const size_t vertex_stride_bytes_;
  mutable real3<T> ray_org_;
  mutable RayCoeff ray_coeff_;
  mutable BVHTraceOptions trace_options_;
  mutable T t_min_;
  mutable T t_;
  mutable T u_;
  mutable T v_;
  mutable unsigned int prim_id_;
};

// Robust slab test (Ize 2013), t_max terms widened by MaxMult
// (ref nanort.h:2278-2370).
namespace detail {
template <typename T>
struct MaxMult;
template <>
struct MaxMult<float> {
  static float value() { return 1.00000024f; }
};
template <>
struct MaxMult<double> {
  static double value() { return 1.0000000000000004; }
};
}  // namespace detail

template <typename T>
inline bool IntersectRayAABB(T *tminOut, T *tmaxOut, T min_t, T max_t, const T bmin[3], const T bmax[3], real3<T> ray_org,
                             real3<T> ray_inv_dir, int ray_dir_sign[3]) {
  T lo = min_t, hi = max_t;
  const T widen = detail::MaxMult<T>::value();
  for (int k = 0; k < 3; k++) {
    const T near_plane = ray_dir_sign[k] ? bmax[k] : bmin[k];
    const T far_plane = ray_dir_sign[k] ? bmin[k] : bmax[k];
    const T t_near = (near_plane - ray_org[k]) * ray_inv_dir[k];
    const T t_far = (far_plane - ray_org[k]) * ray_inv_dir[k] * widen;
    lo = safemax(t_near, lo);
    hi = safemin(t_far, hi);
  }
  if (lo <= hi) {
    *tminOut = lo;
    *tmaxOut = hi;
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------
// Built-in sphere ("particle") primitive.  Not in the reference header: the reference ships it as user code in
// examples/particle_primitive/main.cc (SpherePred :82-108, SphereGeometry :113-147, SphereIntersection :149-159,
// SphereIntersector :161-291).  Same concepts, same arithmetic, same names — an application drops its local
// copies and uses these; with NANORT_USE_HIP_BACKEND, Build() over (SphereGeometry, SpherePred) and
// TraverseBatch() with SphereIntersection run on the GPU (nrtSetSpheres_f32).  fp32, like the example.
// ---------------------------------------------------------------------------
class SpherePred {
 public:
  explicit SpherePred(const float *centers) : axis_(0), pos_(0.0f), centers_(centers) {}
  void Set(int axis, float pos) const {
    axis_ = axis;
    pos_ = pos;
  }
  bool operator()(unsigned int i) const { return centers_[3 * i + axis_] < pos_; }

 private:
  mutable int axis_;
  mutable float pos_;
  const float *centers_;
};

class SphereGeometry {
 public:
  SphereGeometry(const float *centers, const float *radii) : centers_(centers), radii_(radii) {}
  void BoundingBox(real3<float> *bmin, real3<float> *bmax, unsigned int i) const {
    for (int k = 0; k < 3; k++) {
      (*bmin)[k] = centers_[3 * i + k] - radii_[i];
      (*bmax)[k] = centers_[3 * i + k] + radii_[i];
    }
  }
  void BoundingBoxAndCenter(real3<float> *bmin, real3<float> *bmax, real3<float> *center, unsigned int i) const {
    BoundingBox(bmin, bmax, i);
    for (int k = 0; k < 3; k++) (*center)[k] = centers_[3 * i + k];
  }
  const float *GetCenters() const { return centers_; }
  const float *GetRadii() const { return radii_; }

 private:
  const float *centers_;
  const float *radii_;
};

class SphereIntersection {
 public:
  SphereIntersection() : u(0.0f), v(0.0f), t(std::numeric_limits<float>::max()), prim_id(static_cast<unsigned int>(-1)) {}
  float u, v;  // spherical coordinates of the hit normal, both in [0, 1]
  float t;
  unsigned int prim_id;
};

template <class H = SphereIntersection>
class SphereIntersector {
 public:
  SphereIntersector(const float *centers, const float *radii) : centers_(centers), radii_(radii), t_(0.0f), prim_id_(0) {}

  // Nearest root of |org + t dir - c|^2 = r^2 that lies in front of the origin, accepted iff <= *t_inout.
  bool Intersect(float *t_inout, unsigned int i) const {
    if (i < opts_.prim_ids_range[0] || i >= opts_.prim_ids_range[1]) return false;
    const real3<float> oc = org_ - real3<float>(&centers_[3 * i]);
    const float a = vdot(dir_, dir_);
    const float b = 2.0f * vdot(dir_, oc);
    const float c = vdot(oc, oc) - radii_[i] * radii_[i];
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float t0, t1;
    if (std::fabs(disc) < std::numeric_limits<float>::epsilon()) {
      t0 = t1 = -0.5f * (b / a);
    } else {
      const float root = std::sqrt(disc);
      const float q = (b < 0) ? (-b - root) / 2.0f : (-b + root) / 2.0f;
      t0 = q / a;
      t1 = c / q;
    }
    if (t0 > t1) std::swap(t0, t1);
    if (t1 < 0) return false;
    const float t = (t0 < 0) ? t1 : t0;
    if (t > *t_inout) return false;
    *t_inout = t;
    return true;
  }
  float GetT() const { return t_; }
  void Update(float t, unsigned int i) const {
    t_ = t;
    prim_id_ = i;
  }
  void PrepareTraversal(const Ray<float> &ray, const BVHTraceOptions &options) const {
    org_ = real3<float>(ray.org);
    dir_ = real3<float>(ray.dir);
    opts_ = options;
  }
  void PostTraversal(const Ray<float> &, bool hit, H *isect) const {
    if (!hit) return;
    const real3<float> n = vnormalize((org_ + t_ * dir_) - real3<float>(&centers_[3 * prim_id_]));
    const double pi = 3.14159265358979323846;
    isect->t = t_;
    isect->prim_id = prim_id_;
    isect->u = float(std::atan2(double(n[0]), double(n[2])) + pi) * 0.5f * float(1.0 / pi);
    isect->v = float(std::acos(double(n[1])) / pi);
  }

 private:
  const float *centers_;
  const float *radii_;
  mutable real3<float> org_, dir_;
  mutable BVHTraceOptions opts_;
  mutable float t_;
  mutable unsigned int prim_id_;
};

// ---------------------------------------------------------------------------
// Built-in cylinder primitive: the reference's second custom-primitive example, examples/cylinder_primitive/main.cc
// (solve2e :61-90, CylinderPred :94-120, CylinderGeometry :124-210, CylinderIntersection :213-224,
// CylinderIntersector :226-424) — same concepts, same arithmetic.  Two end points and two radii per cylinder (the
// intersector uses the larger radius for the whole cylinder).  With NANORT_USE_HIP_BACKEND, Build() over
// (CylinderGeometry, CylinderPred) and TraverseBatch() with CylinderIntersection run on the GPU.
// One difference: PostTraversal also stores isect->t (the example leaves it untouched).
// ---------------------------------------------------------------------------
class CylinderPred {
 public:
  explicit CylinderPred(const float *endpoints) : axis_(0), pos_(0.0f), endpoints_(endpoints) {}
  void Set(int axis, float pos) const {
    axis_ = axis;
    pos_ = pos;
  }
  bool operator()(unsigned int i) const { return (endpoints_[6 * i + axis_] + endpoints_[6 * i + 3 + axis_]) / 2.0f < pos_; }

 private:
  mutable int axis_;
  mutable float pos_;
  const float *endpoints_;
};

class CylinderGeometry {
 public:
  CylinderGeometry(const float *endpoints, const float *radii) : endpoints_(endpoints), radii_(radii) {}
  void BoundingBox(real3<float> *bmin, real3<float> *bmax, unsigned int i) const {
    for (int k = 0; k < 3; k++) {
      const float a = endpoints_[6 * i + k], b = endpoints_[6 * i + 3 + k];
      (*bmin)[k] = std::min(b - radii_[2 * i + 1], a - radii_[2 * i]);
      (*bmax)[k] = std::max(b + radii_[2 * i + 1], a + radii_[2 * i]);
    }
  }
  void BoundingBoxAndCenter(real3<float> *bmin, real3<float> *bmax, real3<float> *center, unsigned int i) const {
    BoundingBox(bmin, bmax, i);
    for (int k = 0; k < 3; k++) (*center)[k] = (endpoints_[6 * i + k] + endpoints_[6 * i + 3 + k]) / 2.0f;
  }
  const float *GetEndpoints() const { return endpoints_; }
  const float *GetRadii() const { return radii_; }

 private:
  const float *endpoints_;
  const float *radii_;
};

class CylinderIntersection {
 public:
  CylinderIntersection() : u(0.0f), v(0.0f), normal(0.0f), t(std::numeric_limits<float>::max()), prim_id(static_cast<unsigned int>(-1)) {}
  float u;  // distance from the axis on a cap hit, 0 on the side
  float v;  // 0 / 1 on the first / second cap, the axis parameter in [0, 1] on the side
  real3<float> normal;
  float t;
  unsigned int prim_id;
};

template <class H = CylinderIntersection>
class CylinderIntersector {
 public:
  CylinderIntersector(const float *endpoints, const float *radii, bool test_cap = true)
      : endpoints_(endpoints), radii_(radii), test_cap_(test_cap), t_(0.0f), prim_id_(0), hit_cap_(false), u_param_(0.0f), v_param_(0.0f) {}

  bool Intersect(float *t_inout, unsigned int i) const {
    if (i < opts_.prim_ids_range[0] || i >= opts_.prim_ids_range[1]) return false;
    const float eps = 1.0e-6f;
    const real3<float> p0(&endpoints_[6 * i]), p1(&endpoints_[6 * i + 3]);
    const float tmax = *t_inout;
    const float rr = std::max<float>(radii_[2 * i], radii_[2 * i + 1]);
    const real3<float> d = p1 - p0, m = org_ - p0;
    const float md = vdot(m, d), nd = vdot(dir_, d), dd = vdot(d, d);
    bool hit_cap = false;
    float cap_t = std::numeric_limits<float>::max();
    if (test_cap_) {  // the two end planes first
      const real3<float> n0 = vnormalize(p0 - p1), n1 = vneg(n0), rd = vnormalize(dir_);
      if (std::fabs(vdot(dir_, n0)) > eps) {
        const float d0 = -vdot(p0, n0), d1 = -vdot(p1, n1);
        const float t0 = -(vdot(org_, n0) + d0) / vdot(rd, n0);
        const float t1 = -(vdot(org_, n1) + d1) / vdot(rd, n1);
        const real3<float> q0 = org_ + t0 * rd, q1 = org_ + t1 * rd;
        const float r0sq = vdot(q0 - p0, q0 - p0), r1sq = vdot(q1 - p1, q1 - p1);
        if (t0 > 0.0 && t0 < tmax && (r0sq < rr * rr)) {
          hit_cap_ = hit_cap = true;
          cap_t = t0;
          *t_inout = cap_t;
          u_param_ = std::sqrt(r0sq);
          v_param_ = 0;
        }
        if (t1 > 0.0 && t1 < tmax && t1 < cap_t && (r1sq < rr * rr)) {
          hit_cap_ = hit_cap = true;
          cap_t = t1;
          *t_inout = cap_t;
          u_param_ = std::sqrt(r1sq);
          v_param_ = 1.0;
        }
      }
    }
    if (md <= 0.0 && nd <= 0.0) return hit_cap;  // origin behind the first cap, pointing away
    if (md >= dd && nd >= 0.0) return hit_cap;   // origin beyond the second cap, pointing away
    const float nn = vdot(dir_, dir_), mn = vdot(m, dir_);
    const float A = dd * nn - nd * nd;
    const float k = vdot(m, m) - rr * rr;
    const float C = dd * k - md * md;
    const float B = dd * mn - nd * md;
    float root = 0.0f;
    if (SmallerRoot(&root, A, B, C)) {
      const float t = root;
      if (0 <= t && t <= tmax && t <= cap_t) {
        float s = md + t * nd;
        s /= dd;
        if (0 <= s && s <= 1) {
          hit_cap_ = false;
          *t_inout = t;
          u_param_ = 0;
          v_param_ = s;
          return true;
        }
      }
    }
    return hit_cap;
  }
  float GetT() const { return t_; }
  void Update(float t, unsigned int i) const {
    t_ = t;
    prim_id_ = i;
  }
  void PrepareTraversal(const Ray<float> &ray, const BVHTraceOptions &options) const {
    org_ = real3<float>(ray.org);
    dir_ = real3<float>(ray.dir);
    opts_ = options;
  }
  void PostTraversal(const Ray<float> &, bool hit, H *isect) const {
    if (!hit) return;
    const real3<float> p0(&endpoints_[6 * prim_id_]), p1(&endpoints_[6 * prim_id_ + 3]);
    const real3<float> axis_point = p0 + real3<float>(v_param_, v_param_, v_param_) * (p1 - p0);
    const real3<float> position = org_ + t_ * dir_;
    real3<float> n;
    if (hit_cap_) {
      const real3<float> mid = 0.5f * (p1 - p0) + p0;
      n = vnormalize(p1 - p0);
      if (!(vdot(position - mid, n) > 0.0)) n = vneg(n);
    } else {
      n = vnormalize(position - axis_point);
    }
    isect->u = u_param_;
    isect->v = v_param_;
    isect->normal = n;
    isect->t = t_;
    isect->prim_id = prim_id_;
  }

 private:
  // The smaller root of A x^2 + 2 B x + C = 0 in the example's formulation (its solve2e); false when there is none.
  static bool SmallerRoot(float *root, float A, float B, float C) {
    if (std::fabs(A) <= 1.0e-6f) {
      *root = -C / B;
      return true;
    }
    const float D = B * B - A * C;
    if (D < 0) return false;
    if (D == 0) {
      *root = -B / A;
      return true;
    }
    float x1 = (std::fabs(B) + std::sqrt(D)) / A;
    if (B >= 0.0) x1 = -x1;
    const float x2 = C / (A * x1);
    *root = (x1 > x2) ? x2 : x1;
    return true;
  }

  const float *endpoints_;
  const float *radii_;
  const bool test_cap_;
  mutable real3<float> org_, dir_;
  mutable BVHTraceOptions opts_;
  mutable float t_;
  mutable unsigned int prim_id_;
  mutable bool hit_cap_;
  mutable float u_param_, v_param_;

 public:
  bool TestsCaps() const { return test_cap_; }
};

// ---------------------------------------------------------------------------
// BVHAccel
// ---------------------------------------------------------------------------
namespace detail {
template <class A, class B>
struct same_type {
  static const bool value = false;
};
template <class A>
struct same_type<A, A> {
  static const bool value = true;
};
struct generic_tag {};
struct triangle_tag {};
struct sphere_tag {};
struct cylinder_tag {};
template <typename T, class Prim, class Pred>
struct build_tag {
#ifdef NANORT_USE_HIP_BACKEND
  typedef typename std::conditional<
      same_type<Prim, TriangleMesh<T> >::value && same_type<Pred, TriangleSAHPred<T> >::value, triangle_tag,
      typename std::conditional<
          same_type<T, float>::value && same_type<Prim, SphereGeometry>::value && same_type<Pred, SpherePred>::value, sphere_tag,
          typename std::conditional<same_type<T, float>::value && same_type<Prim, CylinderGeometry>::value && same_type<Pred, CylinderPred>::value,
                                    cylinder_tag, generic_tag>::type>::type>::type type;
#else
  typedef generic_tag type;
#endif
};

#ifdef NANORT_USE_HIP_BACKEND
// float/double -> the matching C-ABI entry points
template <typename T>
struct HipApi;
template <>
struct HipApi<float> {
  typedef nrt_ray_f32 RayPod;
  typedef nrt_node_f32 NodePod;
  typedef nrt_hit_f32 HitPod;
  typedef nrt_build_options_f32 BuildPod;
  static nrt_status SetMesh(nrt_ctx *c, const float *v, size_t s, const unsigned int *f, unsigned int n) { return nrtSetMesh_f32(c, v, s, f, n); }
  static nrt_status Build(nrt_ctx *c, const BuildPod *o, nrt_build_stats *st, uint64_t *nn) { return nrtBuild_f32(c, o, st, nn); }
  static nrt_status GetTree(nrt_ctx *c, NodePod *n, uint32_t *i) { return nrtGetTree_f32(c, n, i); }
  static nrt_status TreeBounds(nrt_ctx *c, float *lo, float *hi) { return nrtGetTreeBounds_f32(c, lo, hi); }
  static nrt_status SetTree(nrt_ctx *c, const NodePod *n, uint64_t nn, const uint32_t *i, uint64_t ni) { return nrtSetTree_f32(c, n, nn, i, ni); }
  static nrt_status Traverse(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, HitPod *h, uint8_t *m) {
    return nrtTraverseBatch_f32(c, r, n, o, h, m);
  }
  static nrt_status TraverseDevice(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, HitPod *h, uint8_t *m, void *s) {
    return nrtTraverseBatchDevice_f32(c, r, n, o, h, m, s);
  }
  static nrt_status TraverseBatches(nrt_ctx *c, uint32_t nb, const RayPod *const *r, const uint64_t *n, const nrt_trace_options *o, HitPod *const *h,
                                    uint8_t *const *m, const uint32_t *fl, void *s) {
    return nrtTraverseBatchesDevice_f32(c, nb, r, n, o, h, m, fl, s);
  }
  static nrt_status TraverseBatchesHost(nrt_ctx *c, uint32_t nb, const RayPod *const *r, const uint64_t *n, const nrt_trace_options *o, HitPod *const *h,
                                        uint8_t *const *m, const uint32_t *fl) {
    return nrtTraverseBatches_f32(c, nb, r, n, o, h, m, fl);
  }
  static nrt_status TraverseMulti(nrt_ctx *const *cs, uint32_t nc, const RayPod *r, uint64_t n, uint64_t row, const nrt_trace_options *o, HitPod *h, uint8_t *m) {
    return nrtTraverseBatchMulti_f32(cs, nc, r, n, row, o, h, m);
  }
  static nrt_status Occluded(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, uint8_t *m) { return nrtOccludedBatch_f32(c, r, n, o, m); }
  static nrt_status OccludedDevice(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, uint8_t *m, void *s) {
    return nrtOccludedBatchDevice_f32(c, r, n, o, m, s);
  }
};
template <>
struct HipApi<double> {
  typedef nrt_ray_f64 RayPod;
  typedef nrt_node_f64 NodePod;
  typedef nrt_hit_f64 HitPod;
  typedef nrt_build_options_f64 BuildPod;
  static nrt_status SetMesh(nrt_ctx *c, const double *v, size_t s, const unsigned int *f, unsigned int n) { return nrtSetMesh_f64(c, v, s, f, n); }
  static nrt_status Build(nrt_ctx *c, const BuildPod *o, nrt_build_stats *st, uint64_t *nn) { return nrtBuild_f64(c, o, st, nn); }
  static nrt_status GetTree(nrt_ctx *c, NodePod *n, uint32_t *i) { return nrtGetTree_f64(c, n, i); }
  static nrt_status TreeBounds(nrt_ctx *c, double *lo, double *hi) { return nrtGetTreeBounds_f64(c, lo, hi); }
  static nrt_status SetTree(nrt_ctx *c, const NodePod *n, uint64_t nn, const uint32_t *i, uint64_t ni) { return nrtSetTree_f64(c, n, nn, i, ni); }
  static nrt_status Traverse(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, HitPod *h, uint8_t *m) {
    return nrtTraverseBatch_f64(c, r, n, o, h, m);
  }
  static nrt_status TraverseDevice(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, HitPod *h, uint8_t *m, void *s) {
    return nrtTraverseBatchDevice_f64(c, r, n, o, h, m, s);
  }
  static nrt_status TraverseBatches(nrt_ctx *c, uint32_t nb, const RayPod *const *r, const uint64_t *n, const nrt_trace_options *o, HitPod *const *h,
                                    uint8_t *const *m, const uint32_t *fl, void *s) {
    return nrtTraverseBatchesDevice_f64(c, nb, r, n, o, h, m, fl, s);
  }
  static nrt_status TraverseBatchesHost(nrt_ctx *c, uint32_t nb, const RayPod *const *r, const uint64_t *n, const nrt_trace_options *o, HitPod *const *h,
                                        uint8_t *const *m, const uint32_t *fl) {
    return nrtTraverseBatches_f64(c, nb, r, n, o, h, m, fl);
  }
  static nrt_status TraverseMulti(nrt_ctx *const *cs, uint32_t nc, const RayPod *r, uint64_t n, uint64_t row, const nrt_trace_options *o, HitPod *h, uint8_t *m) {
    return nrtTraverseBatchMulti_f64(cs, nc, r, n, row, o, h, m);
  }
  static nrt_status Occluded(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, uint8_t *m) { return nrtOccludedBatch_f64(c, r, n, o, m); }
  static nrt_status OccludedDevice(nrt_ctx *c, const RayPod *r, uint64_t n, const nrt_trace_options *o, uint8_t *m, void *s) {
    return nrtOccludedBatchDevice_f64(c, r, n, o, m, s);
  }
};
struct CtxDeleter {
  void operator()(nrt_ctx *c) const { nrtDestroy(c); }
};
// serialises the on-demand read-back of GPU-built trees (BVHAccel::EnsureHostTree); one per process is plenty
inline std::mutex &HostTreeMutex() {
  static std::mutex m;
  return m;
}
#endif
}  // namespace detail

template <typename T>
class BVHAccel {
 public:
  BVHAccel() : pad0_(0) { (void)pad0_; }
  ~BVHAccel() {}
#ifdef NANORT_USE_HIP_BACKEND
  // A GPU Build() leaves the tree on the device: nodes_ / indices_ are fetched by the first host access (GetNodes(),
  // GetIndices(), Traverse(), ListNodeIntersections(), Dump(), Debug()).  A copy shares the device context (shared_ptr), so
  // it takes the host arrays with it: copying materialises them first — the copy then never depends on what the original's
  // context holds later.
  BVHAccel(const BVHAccel &o) : pad0_(0) { CopyFrom(o); }
  BVHAccel &operator=(const BVHAccel &o) {
    if (this != &o) CopyFrom(o);
    return *this;
  }
#endif

  // Build a BVH over `num_primitives` primitives (ref nanort.h:716-718).
  // Returns false iff num_primitives == 0.
  template <class Prim, class Pred>
  bool Build(const unsigned int num_primitives, const Prim &p, const Pred &pred, const BVHBuildOptions<T> &options = BVHBuildOptions<T>()) {
    return BuildImpl(num_primitives, p, pred, options, typename detail::build_tag<T, Prim, Pred>::type());
  }

  BVHBuildStatistics GetStatistics() const { return stats_; }

#if defined(NANORT_ENABLE_SERIALIZATION)
  // Raw host-endian dump: size_t count, nodes, size_t count, indices (ref nanort.h:2164-2276).
  bool Dump(const char *filename) const {
    FILE *fp = fopen(filename, "wb");
    if (!fp) return false;
    const bool ok = Dump(fp);
    fclose(fp);
    return ok;
  }
  bool Dump(FILE *fp) const {
    EnsureHostTree();
    const size_t nn = nodes_.size(), ni = indices_.size();
    if (nn == 0) return false;
    bool ok = fwrite(&nn, sizeof(size_t), 1, fp) == 1;
    ok = ok && fwrite(&nodes_[0], sizeof(BVHNode<T>), nn, fp) == nn;
    ok = ok && fwrite(&ni, sizeof(size_t), 1, fp) == 1;
    ok = ok && (ni == 0 || fwrite(&indices_[0], sizeof(unsigned int), ni, fp) == ni);
    return ok;
  }
  bool Load(const char *filename) {
    FILE *fp = fopen(filename, "rb");
    if (!fp) return false;
    const bool ok = Load(fp);
    fclose(fp);
    return ok;
  }
  bool Load(FILE *fp) {
#ifdef NANORT_USE_HIP_BACKEND
    DropPendingHostTree();
#endif
    size_t nn = 0, ni = 0;
    if (fread(&nn, sizeof(size_t), 1, fp) != 1 || nn == 0) return false;
    nodes_.resize(nn);
    if (fread(&nodes_[0], sizeof(BVHNode<T>), nn, fp) != nn) return false;
    if (fread(&ni, sizeof(size_t), 1, fp) != 1) return false;
    indices_.resize(ni);
    if (ni && fread(&indices_[0], sizeof(unsigned int), ni, fp) != ni) return false;
#ifdef NANORT_USE_HIP_BACKEND
    device_tree_stale_ = true;  // re-uploaded lazily by TraverseBatch once a mesh is known
#endif
    return true;
  }
#endif

  void Debug() {
    EnsureHostTree();
    for (size_t i = 0; i < indices_.size(); i++) printf("index[%d] = %d\n", int(i), int(indices_[i]));
    for (size_t i = 0; i < nodes_.size(); i++) {
      printf("node[%d] : bmin %f, %f, %f, bmax %f, %f, %f\n", int(i), double(nodes_[i].bmin[0]), double(nodes_[i].bmin[1]),
             double(nodes_[i].bmin[2]), double(nodes_[i].bmax[0]), double(nodes_[i].bmax[1]), double(nodes_[i].bmax[2]));
    }
  }

  // Closest hit along one ray (ref nanort.h:757-759, 2487-2556).
  template <class I, class H>
  bool Traverse(const Ray<T> &ray, const I &intersector, H *isect, const BVHTraceOptions &options = BVHTraceOptions()) const {
    EnsureHostTree();
    unsigned int todo[kNANORT_MAX_STACK_DEPTH];
    int top = 0;
    todo[0] = 0;

    T best_t = ray.max_t;
    intersector.Update(best_t, static_cast<unsigned int>(-1));
    intersector.PrepareTraversal(ray, options);

    int dir_sign[3];
    real3<T> dir;
    for (int k = 0; k < 3; k++) {
      dir_sign[k] = ray.dir[k] < static_cast<T>(0.0) ? 1 : 0;
      dir[k] = ray.dir[k];
    }
    const real3<T> inv_dir = vsafe_inverse(dir);
    const real3<T> org(ray.org[0], ray.org[1], ray.org[2]);

    T box_t0, box_t1;
    while (top >= 0) {
      const BVHNode<T> &node = nodes_[todo[top--]];
      if (!IntersectRayAABB(&box_t0, &box_t1, ray.min_t, best_t, node.bmin, node.bmax, org, inv_dir, dir_sign)) continue;
      if (node.flag == 0) {
        const int near_side = dir_sign[node.axis];
        todo[++top] = node.data[1 - near_side];  // far child waits
        todo[++top] = node.data[near_side];      // near child is visited next
        assert(top < kNANORT_MAX_STACK_DEPTH);
      } else {
        T t = intersector.GetT();
        bool any = false;
        for (unsigned int i = 0; i < node.data[0]; i++) {
          const unsigned int prim = indices_[node.data[1] + i];
          T cand = t;
          if (intersector.Intersect(&cand, prim)) {
            t = cand;
            intersector.Update(t, prim);
            any = true;
          }
        }
        if (any) best_t = intersector.GetT();
      }
    }
    const bool hit = intersector.GetT() < ray.max_t;  // strict
    intersector.PostTraversal(ray, hit, isect);
    return hit;
  }

  // The K nearest leaf primitives' [t_min, t_max] intervals along the ray, front to back, for
  // two-level traversal (ref nanort.h:781-784, 2558-2692).  `I` is the *interval* intersector
  // concept: PrepareTraversal(ray), Intersect(&t_min, &t_max, prim).
  template <class I>
  bool ListNodeIntersections(const Ray<T> &ray, int max_intersections, const I &intersector,
                             StackVector<NodeHit<T>, 128> *hits) const {
    EnsureHostTree();
    std::priority_queue<NodeHit<T>, std::vector<NodeHit<T> >, NodeHitComparator<T> > farthest_first;
    (*hits)->clear();
    unsigned int todo[kNANORT_MAX_STACK_DEPTH];
    int top = 0;
    todo[0] = 0;
    int dir_sign[3];
    real3<T> dir;
    for (int k = 0; k < 3; k++) {
      dir_sign[k] = ray.dir[k] < static_cast<T>(0.0) ? 1 : 0;
      dir[k] = ray.dir[k];
    }
    const real3<T> inv_dir = vsafe_inverse(dir);
    const real3<T> org(ray.org[0], ray.org[1], ray.org[2]);
    T box_t0, box_t1;
    while (top >= 0) {
      const BVHNode<T> &node = nodes_[todo[top--]];
      if (!IntersectRayAABB(&box_t0, &box_t1, ray.min_t, ray.max_t, node.bmin, node.bmax, org, inv_dir, dir_sign)) continue;
      if (node.flag == 0) {
        const int near_side = dir_sign[node.axis];
        todo[++top] = node.data[1 - near_side];
        todo[++top] = node.data[near_side];
      } else {
        intersector.PrepareTraversal(ray);
        for (unsigned int i = 0; i < node.data[0]; i++) {
          const unsigned int prim = indices_[node.data[1] + i];
          NodeHit<T> h;
          if (!intersector.Intersect(&h.t_min, &h.t_max, prim)) continue;
          h.node_id = prim;
          if (farthest_first.size() < static_cast<size_t>(max_intersections)) {
            farthest_first.push(h);
          } else if (h.t_min < farthest_first.top().t_min) {
            farthest_first.pop();
            farthest_first.push(h);
          }
        }
      }
    }
    if (farthest_first.empty()) return false;
    const size_t n = farthest_first.size();
    (*hits)->resize(n);
    for (size_t i = 0; i < n; i++) {
      (*hits)[n - i - 1] = farthest_first.top();
      farthest_first.pop();
    }
    return true;
  }

#ifdef NANORT_USE_HIP_BACKEND
  // Closest hit for each of `num_rays` rays in one GPU launch.  isects[i] is written only when
  // ray i hits (exactly like N calls of Traverse()); hit_out[i] (optional) receives 1 / 0.
  // Requires a tree built by Build() with the built-in triangle types (or Load() after such a
  // Build set the mesh).  Returns false and leaves the outputs untouched on a backend error
  // (LastBackendError() tells why).
  bool TraverseBatch(const Ray<T> *rays, size_t num_rays, TriangleIntersection<T> *isects, unsigned char *hit_out = NULL,
                     const BVHTraceOptions &options = BVHTraceOptions()) const {
    return TraverseBatchImpl(rays, num_rays, isects, hit_out, options);
  }
  // The same launch for callers that keep their ray waves in HBM (a wavefront renderer whose shading runs on the GPU):
  // `d_rays`, `d_isects`, `d_hit` are device pointers, the call is asynchronous on `hip_stream` (a hipStream_t).
  // Unlike the host variant every record is written: a miss stores {u = 0, v = 0, t = ray.max_t, prim_id = 0xFFFFFFFF}.
  // Launches on different streams may overlap on the GPU.
  bool TraverseBatchDevice(const Ray<T> *d_rays, size_t num_rays, TriangleIntersection<T> *d_isects, unsigned char *d_hit,
                           void *hip_stream, const BVHTraceOptions &options = BVHTraceOptions()) const {
    if (!ctx_ || device_tree_stale_ || device_prim_kind_ != 0) {
      backend_error_ = "TraverseBatchDevice: no triangle tree on the GPU (Build() with the built-in triangle types, or TraverseBatch() once after Load())";
      return false;
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (DeviceLaunch(ctx_.get(), d_rays, num_rays, &o, d_isects, d_hit, hip_stream) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    return true;
  }
  // Several independent device-resident waves in ONE launch (nrtTraverseBatchesDevice): `occlusion[k]` != 0 makes wave k an
  // occlusion query (only d_hit[k] is written, d_isects[k] may be NULL).  One launch tail for all the waves — a renderer's
  // shadow query and next path wave, without a second stream.  Records equal those of separate calls.  `occlusion` may be NULL.
  bool TraverseBatchesDevice(size_t num_waves, const Ray<T> *const *d_rays, const size_t *num_rays, TriangleIntersection<T> *const *d_isects,
                             unsigned char *const *d_hit, const unsigned char *occlusion, void *hip_stream,
                             const BVHTraceOptions &options = BVHTraceOptions()) const {
    typedef detail::HipApi<T> Api;
    if (!ctx_ || device_tree_stale_ || device_prim_kind_ != 0) {
      backend_error_ = "TraverseBatchesDevice: no triangle tree on the GPU";
      return false;
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    std::vector<const typename Api::RayPod *> r(num_waves);
    std::vector<typename Api::HitPod *> h(num_waves);
    std::vector<uint8_t *> m(num_waves);
    std::vector<uint64_t> n(num_waves);
    std::vector<uint32_t> fl(num_waves);
    for (size_t k = 0; k < num_waves; k++) {
      r[k] = reinterpret_cast<const typename Api::RayPod *>(d_rays[k]);
      h[k] = reinterpret_cast<typename Api::HitPod *>(d_isects ? d_isects[k] : NULL);
      m[k] = d_hit ? d_hit[k] : NULL;
      n[k] = num_rays[k];
      fl[k] = (occlusion && occlusion[k]) ? NRT_BATCH_OCCLUSION : 0u;
    }
    if (Api::TraverseBatches(ctx_.get(), static_cast<uint32_t>(num_waves), r.data(), n.data(), &o, h.data(), m.data(), fl.data(), hip_stream) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    return true;
  }
  // Several independent HOST waves in ONE launch (nrtTraverseBatches): what a host-shaded wavefront renderer has ready at the
  // same time — the shadow query of one depth and the path wave of the next — uploaded together, walked by one persistent
  // launch (one launch tail instead of one per wave), downloaded together.  `occlusion[k]` != 0 makes wave k an occlusion
  // query: only hit_out[k] (required) is written.  For the other waves isects[k][i] is written only on a hit, like
  // TraverseBatch(); hit_out[k] may be NULL.  Records and flags are exactly those of separate TraverseBatch() /
  // OccludedBatch() calls.  `occlusion` may be NULL.
  bool TraverseBatches(size_t num_waves, const Ray<T> *const *rays, const size_t *num_rays, TriangleIntersection<T> *const *isects,
                       unsigned char *const *hit_out, const unsigned char *occlusion, const BVHTraceOptions &options = BVHTraceOptions()) const {
    typedef detail::HipApi<T> Api;
    typedef typename Api::HitPod HitPod;
    if (!TraverseBatchImpl(static_cast<const Ray<T> *>(NULL), 0, static_cast<TriangleIntersection<T> *>(NULL), NULL, options)) return false;  // (context, primitive kind, a tree adopted by Load())
    size_t total = 0;
    for (size_t k = 0; k < num_waves; k++) total += num_rays[k];
    if (total == 0) return true;
    HitPod *tmp = static_cast<HitPod *>(StageEnsure(&stage_hits_, &stage_hits_cap_, total * sizeof(HitPod)));
    unsigned char *tmask = static_cast<unsigned char *>(StageEnsure(&stage_mask_, &stage_mask_cap_, total));
    if (!tmp || !tmask) {
      backend_error_ = "TraverseBatches: out of host memory for the staging buffers";
      return false;
    }
    std::vector<const typename Api::RayPod *> r(num_waves);
    std::vector<HitPod *> h(num_waves);
    std::vector<uint8_t *> m(num_waves);
    std::vector<uint64_t> n(num_waves);
    std::vector<uint32_t> fl(num_waves);
    size_t off = 0;
    for (size_t k = 0; k < num_waves; k++) {
      const bool occ = occlusion && occlusion[k];
      if (num_rays[k] && occ && !(hit_out && hit_out[k])) {
        backend_error_ = "TraverseBatches: an occlusion wave needs its hit_out array";
        return false;
      }
      r[k] = reinterpret_cast<const typename Api::RayPod *>(rays[k]);
      h[k] = occ ? NULL : tmp + off;
      m[k] = (hit_out && hit_out[k]) ? hit_out[k] : tmask + off;
      n[k] = num_rays[k];
      fl[k] = occ ? NRT_BATCH_OCCLUSION : 0u;
      off += num_rays[k];
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (Api::TraverseBatchesHost(ctx_.get(), static_cast<uint32_t>(num_waves), r.data(), n.data(), &o, h.data(), m.data(), fl.data()) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    for (size_t k = 0; k < num_waves; k++) {
      if (fl[k] || !isects || !isects[k]) continue;
      const HitPod *src = h[k];
      const uint8_t *mk = m[k];
      TriangleIntersection<T> *dst = isects[k];
      for (size_t i = 0; i < num_rays[k]; i++)
        if (mk[i]) std::memcpy(static_cast<void *>(&dst[i]), &src[i], sizeof(HitPod));
    }
    return true;
  }
  // Opt-in extension without a reference counterpart: occlusion queries.  occluded_out[i] is exactly what
  // TraverseBatch() would report in hit_out[i], but a ray stops at the first primitive it accepts (shadow rays).
  bool OccludedBatch(const Ray<T> *rays, size_t num_rays, unsigned char *occluded_out, const BVHTraceOptions &options = BVHTraceOptions()) const {
    typedef detail::HipApi<T> Api;
    if (!ctx_ || device_tree_stale_ || device_prim_kind_ != 0) {
      backend_error_ = "OccludedBatch: no triangle tree on the GPU (Build() with the built-in triangle types, or TraverseBatch() once after Load())";
      return false;
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (Api::Occluded(ctx_.get(), reinterpret_cast<const typename Api::RayPod *>(rays), num_rays, &o, occluded_out) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    return true;
  }
  // ... and with device pointers, asynchronous on `hip_stream` (see TraverseBatchDevice).
  bool OccludedBatchDevice(const Ray<T> *d_rays, size_t num_rays, unsigned char *d_occluded, void *hip_stream,
                           const BVHTraceOptions &options = BVHTraceOptions()) const {
    typedef detail::HipApi<T> Api;
    if (!ctx_ || device_tree_stale_ || device_prim_kind_ != 0) {
      backend_error_ = "OccludedBatchDevice: no triangle tree on the GPU";
      return false;
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (Api::OccludedDevice(ctx_.get(), reinterpret_cast<const typename Api::RayPod *>(d_rays), num_rays, &o, d_occluded, hip_stream) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    return true;
  }
  // Same for a tree built over the built-in sphere primitive (SphereGeometry + SpherePred).
  bool TraverseBatch(const Ray<T> *rays, size_t num_rays, SphereIntersection *isects, unsigned char *hit_out = NULL,
                     const BVHTraceOptions &options = BVHTraceOptions()) const {
    static_assert(detail::same_type<T, float>::value, "the sphere primitive is fp32");
    return TraverseBatchImpl(rays, num_rays, isects, hit_out, options);
  }
  // Same for a tree built over the built-in cylinder primitive (CylinderGeometry + CylinderPred); `test_cap` is the
  // CylinderIntersector constructor flag.
  bool TraverseBatch(const Ray<T> *rays, size_t num_rays, CylinderIntersection *isects, unsigned char *hit_out = NULL,
                     const BVHTraceOptions &options = BVHTraceOptions(), bool test_cap = true) const {
    static_assert(detail::same_type<T, float>::value, "the cylinder primitive is fp32");
    static_assert(sizeof(CylinderIntersection) == sizeof(nrt_cyl_hit_f32), "CylinderIntersection layout");
    if (!ctx_ || !cyl_endpoints_ || device_prim_kind_ != 2) {
      backend_error_ = "TraverseBatch(CylinderIntersection*): Build() with CylinderGeometry/CylinderPred first";
      return false;
    }
    if (test_cap != cyl_test_cap_ || device_tree_stale_) {  // the flag lives with the primitives on the device
      if (nodes_.empty() || nrtSetCylinders_f32(ctx_.get(), cyl_endpoints_, cyl_radii_, cyl_count_, test_cap ? 1 : 0) != NRT_OK ||
          nrtSetTree_f32(ctx_.get(), reinterpret_cast<const nrt_node_f32 *>(&nodes_[0]), nodes_.size(), &indices_[0], indices_.size()) != NRT_OK) {
        backend_error_ = nodes_.empty() ? "TraverseBatch: empty tree" : nrtLastError(ctx_.get());
        return false;
      }
      cyl_test_cap_ = test_cap;
      device_tree_stale_ = false;
    }
    if (num_rays == 0) return true;
    std::vector<CylinderIntersection> tmp(num_rays);
    std::vector<unsigned char> mask(num_rays);
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (nrtTraverseBatchCylinders_f32(ctx_.get(), reinterpret_cast<const nrt_ray_f32 *>(rays), num_rays, &o,
                                      reinterpret_cast<nrt_cyl_hit_f32 *>(&tmp[0]), &mask[0]) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    for (size_t i = 0; i < num_rays; i++) {
      if (mask[i]) isects[i] = tmp[i];
      if (hit_out) hit_out[i] = mask[i];
    }
    return true;
  }
  const std::string &LastBackendError() const { return backend_error_; }
  // The C-ABI context behind this accel (NULL before a GPU Build()): what nrtSceneAddNode_f32 takes (include/nanosg_hip.h).
  nrt_ctx *HipContext() const { return device_tree_stale_ ? NULL : ctx_.get(); }
  // Multi-GPU (environment variable NANORT_HIP_DEVICES = "all" or a list such as "0,1,2,3"; read by Build()): Build() makes a
  // replica of the tree on every listed device (the build is deterministic: the replicas are bit-identical) and
  // TraverseBatch() splits a host batch over them in interleaved rows of this many rays (default 4096; a renderer passes its
  // image width).  Records are the single-device ones.
  size_t NumHipDevices() const { return ctx_ ? 1 + peers_.size() : 0; }
  void SetTraverseBatchRowLength(size_t rays_per_row) { batch_row_len_ = rays_per_row; }

 private:
  static nrt_status DeviceLaunch(nrt_ctx *c, const Ray<T> *r, size_t n, const nrt_trace_options *o, TriangleIntersection<T> *h,
                                 unsigned char *m, void *stream) {
    typedef detail::HipApi<T> Api;
    return Api::TraverseDevice(c, reinterpret_cast<const typename Api::RayPod *>(r), n, o, reinterpret_cast<typename Api::HitPod *>(h), m, stream);
  }
  template <class Hit>
  bool TraverseBatchImpl(const Ray<T> *rays, size_t num_rays, Hit *isects, unsigned char *hit_out, const BVHTraceOptions &options) const {
    typedef detail::HipApi<T> Api;
    static_assert(sizeof(Ray<T>) == sizeof(typename Api::RayPod), "Ray layout");
    static_assert(sizeof(Hit) == sizeof(typename Api::HitPod), "intersection record layout");
    static_assert(sizeof(BVHTraceOptions) == sizeof(nrt_trace_options), "BVHTraceOptions layout");
    if (!ctx_) {
      backend_error_ = "TraverseBatch: no GPU context (Build() with TriangleMesh/TriangleSAHPred first)";
      return false;
    }
    const int want_kind = detail::same_type<Hit, TriangleIntersection<T> >::value ? 0 : 1;
    if (device_prim_kind_ != want_kind) {  // records of one primitive kind must not be read as another's
      backend_error_ = want_kind == 0 ? "TraverseBatch(TriangleIntersection*): this accel was not built over a TriangleMesh"
                                      : "TraverseBatch(SphereIntersection*): this accel was not built over a SphereGeometry";
      return false;
    }
    if (device_tree_stale_) {
      if (nodes_.empty() || Api::SetTree(ctx_.get(), reinterpret_cast<const typename Api::NodePod *>(&nodes_[0]), nodes_.size(),
                                         indices_.empty() ? NULL : &indices_[0], indices_.size()) != NRT_OK) {
        backend_error_ = nodes_.empty() ? "TraverseBatch: empty tree" : nrtLastError(ctx_.get());
        return false;
      }
      for (size_t k = 0; k < peers_.size(); k++) // (the replicas on the other devices adopt the same arrays)
        if (Api::SetTree(peers_[k].get(), reinterpret_cast<const typename Api::NodePod *>(&nodes_[0]), nodes_.size(),
                         indices_.empty() ? NULL : &indices_[0], indices_.size()) != NRT_OK) {
          backend_error_ = nrtLastError(peers_[k].get());
          return false;
        }
      device_tree_stale_ = false;
    }
    if (num_rays == 0) return true;
    // Grow-only byte staging owned by the accel, page-locked when the backend can provide it (the device-to-host copy
    // then runs at PCIe speed; PODs: no per-element construction, no fresh pages to fault in on every wave); the caller's
    // mask array is used directly when there is one.
    typedef typename Api::HitPod HitPod;
    HitPod *tmp = static_cast<HitPod *>(StageEnsure(&stage_hits_, &stage_hits_cap_, num_rays * sizeof(HitPod)));
    unsigned char *mask = hit_out ? hit_out : static_cast<unsigned char *>(StageEnsure(&stage_mask_, &stage_mask_cap_, num_rays));
    if (!tmp || !mask) {
      backend_error_ = "TraverseBatch: out of host memory for the staging buffers";
      return false;
    }
    nrt_trace_options o;
    std::memcpy(&o, &options, sizeof(o));
    if (!peers_.empty() && want_kind == 0) {
      // NANORT_HIP_DEVICES: the batch is split row-interleaved over the replicas, one per device (nrtTraverseBatchMulti)
      std::vector<nrt_ctx *> cs(1, ctx_.get());
      for (size_t k = 0; k < peers_.size(); k++) cs.push_back(peers_[k].get());
      if (Api::TraverseMulti(&cs[0], static_cast<uint32_t>(cs.size()), reinterpret_cast<const typename Api::RayPod *>(rays), num_rays, batch_row_len_,
                             &o, reinterpret_cast<HitPod *>(tmp), mask) != NRT_OK) {
        backend_error_ = nrtLastError(ctx_.get());
        return false;
      }
    } else if (Api::Traverse(ctx_.get(), reinterpret_cast<const typename Api::RayPod *>(rays), num_rays, &o, tmp, mask) != NRT_OK) {
      backend_error_ = nrtLastError(ctx_.get());
      return false;
    }
    // isects[i] is written only on a hit, like Traverse().  (At most half of what the process may use, detail::HostThreads():
    // an OpenMP default of "all hardware threads" inside a CPU-quota'd container starves the HIP runtime's own threads —
    // measured 10x slower.)
#ifdef _OPENMP
    const int scatter_threads = static_cast<int>(std::min(8u, std::max(1u, detail::HostThreads() / 2)));
#pragma omp parallel for schedule(static) num_threads(scatter_threads) if (num_rays > (1u << 18))
#endif
    for (long long i = 0; i < static_cast<long long>(num_rays); i++)
      if (mask[i]) std::memcpy(static_cast<void *>(&isects[i]), &tmp[i], sizeof(HitPod));
    return true;
  }

 public:
#endif

  const std::vector<BVHNode<T> > &GetNodes() const {
    EnsureHostTree();
    return nodes_;
  }
  const std::vector<unsigned int> &GetIndices() const {
    EnsureHostTree();
    return indices_;
  }

  void BoundingBox(T bmin[3], T bmax[3]) const {
#ifdef NANORT_USE_HIP_BACKEND
    if (HostTreePending()) {  // (the root's box came back with the build: no need for the whole array)
      for (int k = 0; k < 3; k++) {
        bmin[k] = root_bmin_[k];
        bmax[k] = root_bmax_[k];
      }
      return;
    }
#endif
    for (int k = 0; k < 3; k++) {
      bmin[k] = nodes_.empty() ? std::numeric_limits<T>::max() : nodes_[0].bmin[k];
      bmax[k] = nodes_.empty() ? -std::numeric_limits<T>::max() : nodes_[0].bmax[k];
    }
  }

  bool IsValid() const { return HostTreePending() || nodes_.size() > 0; }

#ifdef NANORT_USE_HIP_BACKEND
  // The host copy of a GPU-built tree, on demand (thread-safe: Traverse() is const and may be called from many threads).
  void EnsureHostTree() const {
    if (!__atomic_load_n(&host_tree_pending_, __ATOMIC_ACQUIRE)) return;
    std::lock_guard<std::mutex> lock(detail::HostTreeMutex());
    if (!host_tree_pending_) return;
    typedef detail::HipApi<T> Api;
    nodes_.resize(static_cast<size_t>(pending_nodes_));
    indices_.resize(static_cast<size_t>(pending_indices_));
    if (!ctx_ || Api::GetTree(ctx_.get(), reinterpret_cast<typename Api::NodePod *>(&nodes_[0]), indices_.empty() ? NULL : &indices_[0]) != NRT_OK) {
      backend_error_ = ctx_ ? nrtLastError(ctx_.get()) : "host tree requested without a GPU context";
      fprintf(stderr, "[nanort] reading the tree back from the GPU failed: %s\n", backend_error_.c_str());
      nodes_.clear();
      indices_.clear();
    }
    __atomic_store_n(&host_tree_pending_, false, __ATOMIC_RELEASE);
  }
  bool HostTreePending() const { return __atomic_load_n(&host_tree_pending_, __ATOMIC_ACQUIRE); }
  // Set NANORT_HIP_EAGER_READBACK=1 (environment) to make Build() fetch the arrays itself, as rounds 1-5 did.
#else
  void EnsureHostTree() const {}
  bool HostTreePending() const { return false; }
#endif

 private:
  // ---- generic host builder: binned SAH over all three axes, iterative, pre-order ----
  struct Pending {
    unsigned int lo, hi, depth, parent;
    bool is_high_child;
  };
  struct HostBin {
    BBox<T> box;
    size_t count;
    HostBin() : count(0) {}
  };

  static T HalfArea(const real3<T> &mn, const real3<T> &mx) {
    const real3<T> e = mx - mn;
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
  }

  // Bounding box of the primitives in indices_[lo, hi) (`threads` > 1: in chunks; min / max do not depend on the order).
  template <class Prim>
  void RangeBounds(const Prim &prim, unsigned int lo, unsigned int hi, unsigned int threads, real3<T> *out_min, real3<T> *out_max) const {
    const unsigned int count = hi - lo;
    const unsigned int chunks = (threads > 1 && count >= 16384u) ? threads * 4u : 1u;
    std::vector<BBox<T> > part(chunks);
    const unsigned int *idx = &indices_[0];
    detail::ParallelFor(chunks, threads, [&](unsigned int c) {
      const unsigned int b = lo + static_cast<unsigned int>((static_cast<unsigned long long>(count) * c) / chunks);
      const unsigned int e = lo + static_cast<unsigned int>((static_cast<unsigned long long>(count) * (c + 1)) / chunks);
      BBox<T> acc;
      for (unsigned int i = b; i < e; i++) {
        real3<T> a, bb;
        prim.BoundingBox(&a, &bb, idx[i]);
        for (int k = 0; k < 3; k++) {
          acc.bmin[k] = std::min(acc.bmin[k], a[k]);
          acc.bmax[k] = std::max(acc.bmax[k], bb[k]);
        }
      }
      part[c] = acc;
    });
    BBox<T> all = part[0];
    for (unsigned int c = 1; c < chunks; c++)
      for (int k = 0; k < 3; k++) {
        all.bmin[k] = std::min(all.bmin[k], part[c].bmin[k]);
        all.bmax[k] = std::max(all.bmax[k], part[c].bmax[k]);
      }
    *out_min = all.bmin;
    *out_max = all.bmax;
  }

  // One node of the generic host builder over indices_[lo, hi): its box, the leaf decision, or the binned-SAH split over
  // x, y and z followed by the partition of the range (pred is this thread's own copy: Set() mutates it).
  // Returns true for a leaf; else *mid and *axis describe the split.
  template <class Prim, class Pred>
  bool SplitRange(const Prim &prim, Pred &pred, unsigned int lo, unsigned int hi, unsigned int depth, unsigned int threads,
                  std::vector<HostBin> &bins, std::vector<BBox<T> > &sweep, BVHNode<T> *node, unsigned int *mid_out, int *axis_out) {
    const unsigned int K = options_.bin_size;
    real3<T> nmin, nmax;
    RangeBounds(prim, lo, hi, threads, &nmin, &nmax);
    for (int k = 0; k < 3; k++) {
      node->bmin[k] = nmin[k];
      node->bmax[k] = nmax[k];
    }
    const unsigned int count = hi - lo;
    if (count <= options_.min_leaf_primitives || depth >= options_.max_tree_depth || count < 2) {
      node->flag = 1;
      node->axis = 0;
      node->data[0] = count;
      node->data[1] = lo;
      return true;
    }
    // bin the centres over the node's box on x, y and z (`threads` > 1: per-chunk bins merged in chunk order — counts
    // add and boxes min / max, so the merged bins are the serial ones)
    real3<T> scale;
    for (int k = 0; k < 3; k++) {
      const T ext = nmax[k] - nmin[k];
      scale[k] = ext > static_cast<T>(0.0) ? static_cast<T>(K) / ext : static_cast<T>(0.0);
    }
    const unsigned int chunks = (threads > 1 && count >= 16384u) ? threads * 4u : 1u;
    std::vector<HostBin> chunk_bins;
    if (chunks > 1) chunk_bins.assign(static_cast<size_t>(chunks) * 3 * K, HostBin());
    std::fill(bins.begin(), bins.end(), HostBin());
    const unsigned int *idx = &indices_[0];
    detail::ParallelFor(chunks, threads, [&](unsigned int c) {
      HostBin *dst = chunks > 1 ? &chunk_bins[static_cast<size_t>(c) * 3 * K] : &bins[0];
      const unsigned int b = lo + static_cast<unsigned int>((static_cast<unsigned long long>(count) * c) / chunks);
      const unsigned int e = lo + static_cast<unsigned int>((static_cast<unsigned long long>(count) * (c + 1)) / chunks);
      for (unsigned int i = b; i < e; i++) {
        real3<T> a, bb, cc;
        prim.BoundingBoxAndCenter(&a, &bb, &cc, idx[i]);
        for (int k = 0; k < 3; k++) {
          const long q = static_cast<long>((cc[k] - nmin[k]) * scale[k]);
          const unsigned int slot = static_cast<unsigned int>(std::min<long>(static_cast<long>(K) - 1, std::max<long>(0, q)));
          HostBin &hb = dst[static_cast<size_t>(k) * K + slot];
          hb.count++;
          for (int d = 0; d < 3; d++) {
            hb.box.bmin[d] = std::min(hb.box.bmin[d], a[d]);
            hb.box.bmax[d] = std::max(hb.box.bmax[d], bb[d]);
          }
        }
      }
    });
    for (unsigned int c = 0; chunks > 1 && c < chunks; c++)
      for (size_t j = 0; j < static_cast<size_t>(3) * K; j++) {
        const HostBin &src = chunk_bins[static_cast<size_t>(c) * 3 * K + j];
        HostBin &hb = bins[j];
        hb.count += src.count;
        for (int d = 0; d < 3; d++) {
          hb.box.bmin[d] = std::min(hb.box.bmin[d], src.box.bmin[d]);
          hb.box.bmax[d] = std::max(hb.box.bmax[d], src.box.bmax[d]);
        }
      }
    // two sweeps per axis; candidate s splits bins [0,s) | [s,K)
    T axis_cost[3], axis_cut[3];
    for (int k = 0; k < 3; k++) {
      axis_cost[k] = std::numeric_limits<T>::infinity();
      axis_cut[k] = nmin[k] + (nmax[k] - nmin[k]) * static_cast<T>(0.5);
      BBox<T> acc;
      for (unsigned int s2 = K; s2-- > 1;) {  // suffix boxes
        const HostBin &hb = bins[static_cast<size_t>(k) * K + s2];
        for (int d = 0; d < 3; d++) {
          acc.bmin[d] = std::min(acc.bmin[d], hb.box.bmin[d]);
          acc.bmax[d] = std::max(acc.bmax[d], hb.box.bmax[d]);
        }
        sweep[s2] = acc;
      }
      size_t left_n = 0;
      BBox<T> left;
      for (unsigned int s2 = 1; s2 < K; s2++) {
        const HostBin &hb = bins[static_cast<size_t>(k) * K + s2 - 1];
        left_n += hb.count;
        for (int d = 0; d < 3; d++) {
          left.bmin[d] = std::min(left.bmin[d], hb.box.bmin[d]);
          left.bmax[d] = std::max(left.bmax[d], hb.box.bmax[d]);
        }
        const size_t right_n = count - left_n;
        if (left_n == 0 || right_n == 0) continue;
        const T c = static_cast<T>(left_n) * HalfArea(left.bmin, left.bmax) + static_cast<T>(right_n) * HalfArea(sweep[s2].bmin, sweep[s2].bmax);
        if (c < axis_cost[k]) {
          axis_cost[k] = c;
          axis_cut[k] = nmin[k] + (nmax[k] - nmin[k]) * (static_cast<T>(s2) / static_cast<T>(K));
        }
      }
    }
    int order[3] = {0, 1, 2};
    if (axis_cost[order[1]] < axis_cost[order[0]]) std::swap(order[0], order[1]);
    if (axis_cost[order[2]] < axis_cost[order[1]]) std::swap(order[1], order[2]);
    if (axis_cost[order[1]] < axis_cost[order[0]]) std::swap(order[0], order[1]);

    unsigned int mid = lo;
    int axis = order[0];
    for (int attempt = 0; attempt < 3; attempt++) {
      axis = order[attempt];
      pred.Set(axis, axis_cut[axis]);
      unsigned int *first = &indices_[lo];
      unsigned int *split = std::partition(first, first + count, pred);
      mid = lo + static_cast<unsigned int>(split - first);
      if (mid != lo && mid != hi) break;
      mid = lo + (count >> 1);  // object median when the predicate cannot separate them
    }
    node->flag = 0;
    node->axis = axis;
    node->data[0] = 0;
    node->data[1] = 0;
    *mid_out = mid;
    *axis_out = axis;
    return false;
  }

  // A subtree to be built by a worker: its range, depth, and the stub that stands for it among the top nodes.
  struct SubtreeTask {
    unsigned int lo, hi, depth, stub;
    std::vector<BVHNode<T> > nodes;  // local pre-order, child indices relative to nodes[0]
    unsigned int max_depth, leaves, branches;
  };

  // Pre-order build of indices_[lo, hi) into `out` (child indices local to `out`).  With `tasks`: ranges that reach
  // `stop_depth` with at least `stop_count` primitives are not descended into — a stub (flag 2) is emitted and the range
  // recorded instead.
  template <class Prim, class Pred>
  void BuildRange(const Prim &prim, Pred &pred, unsigned int lo0, unsigned int hi0, unsigned int depth0, unsigned int threads,
                  std::vector<BVHNode<T> > &out, unsigned int *max_depth, unsigned int *leaves, unsigned int *branches,
                  std::vector<SubtreeTask> *tasks, unsigned int stop_depth, unsigned int stop_count) {
    const unsigned int K = options_.bin_size;
    std::vector<HostBin> bins(3 * static_cast<size_t>(K));
    std::vector<BBox<T> > sweep(K);
    std::vector<Pending> todo;
    Pending root = {lo0, hi0, depth0, 0, false};
    todo.push_back(root);
    while (!todo.empty()) {
      const Pending cur = todo.back();
      todo.pop_back();
      const unsigned int me = static_cast<unsigned int>(out.size());
      if (me != 0 && cur.is_high_child) out[cur.parent].data[1] = me;
      if (tasks && cur.depth >= stop_depth && cur.hi - cur.lo >= stop_count && me != 0) {
        BVHNode<T> stub;
        stub.flag = 2;
        stub.axis = 0;
        stub.data[0] = static_cast<unsigned int>(tasks->size());
        stub.data[1] = 0;
        out.push_back(stub);
        SubtreeTask t;
        t.lo = cur.lo;
        t.hi = cur.hi;
        t.depth = cur.depth;
        t.stub = me;
        t.max_depth = t.leaves = t.branches = 0;
        tasks->push_back(t);
        continue;
      }
      *max_depth = std::max(*max_depth, cur.depth);
      BVHNode<T> node;
      unsigned int mid = 0;
      int axis = 0;
      if (SplitRange(prim, pred, cur.lo, cur.hi, cur.depth, threads, bins, sweep, &node, &mid, &axis)) {
        out.push_back(node);
        (*leaves)++;
        continue;
      }
      node.data[0] = me + 1;
      out.push_back(node);
      (*branches)++;
      Pending high = {mid, cur.hi, cur.depth + 1, me, true};
      Pending low = {cur.lo, mid, cur.depth + 1, me, false};
      todo.push_back(high);
      todo.push_back(low);
    }
  }

  // BVHAccel::Build for user primitives (nanort.h:1892-2149): binned SAH over x, y and z, pre-order node array.
  // Parallel (NANORT_ENABLE_PARALLEL_BUILD with OpenMP, or NANORT_USE_CPP11_FEATURE; n >=
  // min_primitives_for_parallel_build) in the reference's shape — a shallow top tree, then one worker per subtree,
  // then the splice (nanort.h:2018-2117) — with the top levels' box and bin passes chunked over the workers as well.
  // The tree does not depend on the number of threads: it is the serial one, node for node.
  template <class Prim, class Pred>
  bool BuildImpl(unsigned int n, const Prim &prim, const Pred &pred, const BVHBuildOptions<T> &options, detail::generic_tag) {
    options_ = options;
    stats_ = BVHBuildStatistics();
#ifdef NANORT_USE_HIP_BACKEND
    DropPendingHostTree();
#endif
    nodes_.clear();
    indices_.clear();
    assert(options_.bin_size > 1);
    if (n == 0) return false;
    indices_.resize(n);
    for (unsigned int i = 0; i < n; i++) indices_[i] = i;

    unsigned int threads = 1;
#if defined(NANORT_ENABLE_PARALLEL_BUILD) && (defined(_OPENMP) || defined(NANORT_USE_CPP11_FEATURE))
    if (n >= options_.min_primitives_for_parallel_build) threads = detail::HostThreads();
#endif
    Pred top_pred = pred;
    if (threads <= 1) {
      BuildRange(prim, top_pred, 0, n, 0, 1, nodes_, &stats_.max_tree_depth, &stats_.num_leaf_nodes, &stats_.num_branch_nodes,
                 static_cast<std::vector<SubtreeTask> *>(0), 0, 0);
    } else {
      // deep enough for a few subtrees per worker (at least the reference's shallow_depth)
      unsigned int stop_depth = std::max(1u, options_.shallow_depth);
      while ((1u << stop_depth) < 4u * threads && stop_depth < 12u) stop_depth++;
      std::vector<BVHNode<T> > top;
      std::vector<SubtreeTask> tasks;
      BuildRange(prim, top_pred, 0, n, 0, threads, top, &stats_.max_tree_depth, &stats_.num_leaf_nodes, &stats_.num_branch_nodes, &tasks,
                 stop_depth, 1024u);
      detail::ParallelFor(static_cast<unsigned int>(tasks.size()), threads, [&](unsigned int k) {
        SubtreeTask &t = tasks[k];
        Pred local_pred = pred;
        BuildRange(prim, local_pred, t.lo, t.hi, t.depth, 1, t.nodes, &t.max_depth, &t.leaves, &t.branches,
                   static_cast<std::vector<SubtreeTask> *>(0), 0, 0);
      });
      // splice: where each top node and each subtree lands in the final pre-order array
      std::vector<unsigned int> final_of(top.size());
      unsigned int at = 0;
      for (size_t i = 0; i < top.size(); i++) {
        final_of[i] = at;
        at += top[i].flag == 2 ? static_cast<unsigned int>(tasks[top[i].data[0]].nodes.size()) : 1u;
      }
      nodes_.resize(at);
      for (size_t i = 0; i < top.size(); i++) {
        if (top[i].flag == 2) {
          const SubtreeTask &t = tasks[top[i].data[0]];
          const unsigned int base = final_of[i];
          for (size_t j = 0; j < t.nodes.size(); j++) {
            BVHNode<T> nd = t.nodes[j];
            if (nd.flag == 0) {
              nd.data[0] += base;
              nd.data[1] += base;
            }
            nodes_[base + j] = nd;
          }
          stats_.max_tree_depth = std::max(stats_.max_tree_depth, t.max_depth);
          stats_.num_leaf_nodes += t.leaves;
          stats_.num_branch_nodes += t.branches;
        } else {
          BVHNode<T> nd = top[i];
          if (nd.flag == 0) {
            nd.data[0] = final_of[nd.data[0]];
            nd.data[1] = final_of[nd.data[1]];
          }
          nodes_[final_of[i]] = nd;
        }
      }
    }
#ifdef NANORT_USE_HIP_BACKEND
    // user primitives live on the host only: a device context left over from an earlier built-in Build() holds other
    // primitives and must not be traced against this tree
    ctx_.reset();
    peers_.clear();
    device_tree_stale_ = false;
    device_prim_kind_ = -1;
#endif
    return true;
  }

#ifdef NANORT_USE_HIP_BACKEND
  // Built-in primitive types: construction on the GPU through the C ABI.
  bool BuildImpl(unsigned int n, const TriangleMesh<T> &mesh, const TriangleSAHPred<T> &pred, const BVHBuildOptions<T> &options,
                 detail::triangle_tag) {
    (void)pred;
    typedef detail::HipApi<T> Api;
    return HipBuild(n, options, 0, [&](nrt_ctx *c) { return Api::SetMesh(c, mesh.GetVertices(), mesh.GetVertexStrideBytes(), mesh.GetFaces(), n); });
  }
  bool BuildImpl(unsigned int n, const SphereGeometry &geom, const SpherePred &pred, const BVHBuildOptions<T> &options,
                 detail::sphere_tag) {
    (void)pred;
    return HipBuild(n, options, 1, [&](nrt_ctx *c) { return nrtSetSpheres_f32(c, geom.GetCenters(), geom.GetRadii(), n); });
  }

  // (the intersector's test_cap flag is a traversal-time property: TraverseBatch() re-sends the primitives if it differs)
  bool BuildImpl(unsigned int n, const CylinderGeometry &geom, const CylinderPred &pred, const BVHBuildOptions<T> &options,
                 detail::cylinder_tag) {
    (void)pred;
    cyl_endpoints_ = geom.GetEndpoints();
    cyl_radii_ = geom.GetRadii();
    cyl_count_ = n;
    cyl_test_cap_ = true;
    return HipBuild(n, options, 2, [&](nrt_ctx *c) { return nrtSetCylinders_f32(c, geom.GetEndpoints(), geom.GetRadii(), n, 1); });
  }

  template <class SetPrims>
  bool HipBuild(unsigned int n, const BVHBuildOptions<T> &options, int prim_kind, SetPrims set_prims) {
    device_prim_kind_ = -1;
    typedef detail::HipApi<T> Api;
    static_assert(sizeof(BVHNode<T>) == sizeof(typename Api::NodePod), "BVHNode layout");
    static_assert(sizeof(BVHBuildOptions<T>) == sizeof(typename Api::BuildPod), "BVHBuildOptions layout");
    static_assert(sizeof(BVHBuildStatistics) == sizeof(nrt_build_stats), "BVHBuildStatistics layout");
    options_ = options;
    stats_ = BVHBuildStatistics();
    DropPendingHostTree();
    nodes_.clear();
    indices_.clear();
    assert(options_.bin_size > 1);
    if (n == 0) return false;
    if (!ctx_) {
      // devices: NANORT_HIP_DEVICES ("all" or a comma-separated list; the first is the primary) or the one of NANORT_HIP_DEVICE
      std::vector<int> devices;
      if (const char *list = std::getenv("NANORT_HIP_DEVICES")) {
        if (std::strcmp(list, "all") == 0) {
          for (int d = 0; d < nrtDeviceCount(); d++) devices.push_back(d);
        } else {
          for (const char *p = list; *p;) {
            devices.push_back(std::atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
          }
        }
      }
      if (devices.empty()) {
        int device = 0;
        if (const char *env = std::getenv("NANORT_HIP_DEVICE")) device = std::atoi(env);
        devices.push_back(device);
      }
      for (size_t k = 0; k < devices.size(); k++) {
        nrt_ctx *raw = NULL;
        if (nrtCreate(devices[k], &raw) != NRT_OK) {
          backend_error_ = nrtLastError(NULL);
          if (k == 0) {
            fprintf(stderr, "[nanort] HIP backend unavailable: %s\n", backend_error_.c_str());
            return false;
          }
          fprintf(stderr, "[nanort] HIP device %d unavailable (%s): continuing with %zu device(s)\n", devices[k], backend_error_.c_str(), k);
          break;
        }
        if (k == 0)
          ctx_ = std::shared_ptr<nrt_ctx>(raw, detail::CtxDeleter());
        else
          peers_.push_back(std::shared_ptr<nrt_ctx>(raw, detail::CtxDeleter()));
      }
    }
    nrt_ctx *c = ctx_.get();
    typename Api::BuildPod o;
    std::memcpy(&o, &options, sizeof(o));
    nrt_build_stats st;
    uint64_t num_nodes = 0;
    if (set_prims(c) != NRT_OK || Api::Build(c, &o, &st, &num_nodes) != NRT_OK) {
      backend_error_ = nrtLastError(c);
      fprintf(stderr, "[nanort] HIP build failed: %s\n", backend_error_.c_str());
      return false;
    }
    // (the index array names a primitive once per leaf slot: n entries, except for cylinders the library cut into segments for
    // its builder, which are named once per segment — the reference's Traverse takes such a tree as it is)
    uint64_t tree_nodes = num_nodes, tree_indices = n;
    if (nrtTreeSize(c, &tree_nodes, &tree_indices) != NRT_OK) tree_indices = n;
    // The arrays stay on the device until the host asks for them (EnsureHostTree): an application that only calls
    // TraverseBatch() never pays the 27 MB read-back of a 1 M-triangle tree.  The root's box comes back now (24 / 48 bytes).
    pending_nodes_ = num_nodes;
    pending_indices_ = tree_indices;
    if (Api::TreeBounds(c, root_bmin_, root_bmax_) != NRT_OK) {
      backend_error_ = nrtLastError(c);
      return false;
    }
    __atomic_store_n(&host_tree_pending_, true, __ATOMIC_RELEASE);
    static const bool eager = std::getenv("NANORT_HIP_EAGER_READBACK") != NULL && std::atoi(std::getenv("NANORT_HIP_EAGER_READBACK")) != 0;
    if (eager) {
      EnsureHostTree();
      if (nodes_.empty()) return false;
    }
    stats_.max_tree_depth = st.max_tree_depth;
    stats_.num_leaf_nodes = st.num_leaf_nodes;
    stats_.num_branch_nodes = st.num_branch_nodes;
    stats_.build_secs = st.build_secs;
    // replicas on the other devices: the same primitives, the same (deterministic) build
    for (size_t k = 0; k < peers_.size(); k++) {
      nrt_build_stats pst;
      uint64_t pn = 0;
      if (set_prims(peers_[k].get()) != NRT_OK || Api::Build(peers_[k].get(), &o, &pst, &pn) != NRT_OK || pn != num_nodes) {
        backend_error_ = nrtLastError(peers_[k].get());
        fprintf(stderr, "[nanort] HIP build of replica %zu failed (%s): tracing on one device\n", k + 1, backend_error_.c_str());
        peers_.clear();
        break;
      }
    }
    device_tree_stale_ = false;
    device_prim_kind_ = prim_kind;
    return true;
  }
#endif

#ifdef NANORT_USE_HIP_BACKEND
  mutable std::vector<BVHNode<T> > nodes_;  // (mutable: filled by EnsureHostTree() on the first host access after a GPU build)
  mutable std::vector<unsigned int> indices_;
#else
  std::vector<BVHNode<T> > nodes_;
  std::vector<unsigned int> indices_;
#endif
  BVHBuildOptions<T> options_;
  BVHBuildStatistics stats_;
  unsigned int pad0_;
#ifdef NANORT_USE_HIP_BACKEND
  void DropPendingHostTree() { __atomic_store_n(&host_tree_pending_, false, __ATOMIC_RELEASE); }
  void CopyFrom(const BVHAccel &o) {
    o.EnsureHostTree();
    nodes_ = o.nodes_;
    indices_ = o.indices_;
    options_ = o.options_;
    stats_ = o.stats_;
    ctx_ = o.ctx_;
    peers_ = o.peers_;
    batch_row_len_ = o.batch_row_len_;
    device_tree_stale_ = o.device_tree_stale_;
    device_prim_kind_ = o.device_prim_kind_;
    cyl_endpoints_ = o.cyl_endpoints_;
    cyl_radii_ = o.cyl_radii_;
    cyl_count_ = o.cyl_count_;
    cyl_test_cap_ = o.cyl_test_cap_;
    host_tree_pending_ = false;
    pending_nodes_ = pending_indices_ = 0;
    for (int k = 0; k < 3; k++) {
      root_bmin_[k] = o.root_bmin_[k];
      root_bmax_[k] = o.root_bmax_[k];
    }
    backend_error_ = o.backend_error_;
    // (the staging buffers stay with their owner: they are scratch, grown on the first TraverseBatch())
  }
  mutable bool host_tree_pending_ = false;  // the tree of the last GPU Build() has not been copied to nodes_ / indices_ yet
  uint64_t pending_nodes_ = 0, pending_indices_ = 0;
  T root_bmin_[3] = {T(0), T(0), T(0)}, root_bmax_[3] = {T(0), T(0), T(0)};
  std::shared_ptr<nrt_ctx> ctx_;
  std::vector<std::shared_ptr<nrt_ctx> > peers_;  // replicas on the other devices of NANORT_HIP_DEVICES
  size_t batch_row_len_ = 0;                      // rays per interleaved row of a multi-device TraverseBatch (0: 4096)
  mutable bool device_tree_stale_ = false;
  int device_prim_kind_ = -1;  // what the device context was built over: 0 triangles, 1 spheres, 2 cylinders, -1 nothing usable
  const float *cyl_endpoints_ = NULL;  // cylinder primitive: what Build() was given
  const float *cyl_radii_ = NULL;
  unsigned int cyl_count_ = 0;
  mutable bool cyl_test_cap_ = true;
  // TraverseBatch staging (grow-only): pinned through nrtHostAlloc, plain malloc if that fails
  mutable std::shared_ptr<void> stage_hits_, stage_mask_;
  mutable size_t stage_hits_cap_ = 0, stage_mask_cap_ = 0;
  static void StageFreePinned(void *p) { nrtHostFree(p); }
  static void *StageEnsure(std::shared_ptr<void> *buf, size_t *cap, size_t bytes) {
    if (bytes <= *cap && buf->get()) return buf->get();
    const size_t want = bytes + bytes / 4;
    void *p = NULL;
    buf->reset();
    *cap = 0;
    if (nrtHostAlloc(want, &p) == NRT_OK && p) {
      buf->reset(p, StageFreePinned);
    } else {
      p = std::malloc(want);
      if (!p) return NULL;
      buf->reset(p, std::free);
    }
    *cap = want;
    return p;
  }
  mutable std::string backend_error_;
#endif
};

// Layouts shared with the device code and the reference.
static_assert(sizeof(Ray<float>) == 36 && sizeof(Ray<double>) == 72, "Ray layout");
static_assert(sizeof(BVHNode<float>) == 40 && sizeof(BVHNode<double>) == 64, "BVHNode layout");
static_assert(sizeof(TriangleIntersection<float>) == 16 && sizeof(TriangleIntersection<double>) == 32, "hit layout");
static_assert(sizeof(BVHBuildOptions<float>) == 28 && sizeof(BVHBuildOptions<double>) == 32, "build options layout");
static_assert(sizeof(BVHTraceOptions) == 16 && sizeof(BVHBuildStatistics) == 16, "trace options / stats layout");

}  // namespace nanort

#endif  // NANORT_H_

// nanort_amd/csrc/traverse.hip — batched closest-hit traversal for gfx950.
//
// Replaces N calls of the reference's BVHAccel<T>::Traverse (nanort.h:2487-2556)
// with its TriangleIntersector (nanort.h:1014-1229) by one persistent-threads
// kernel: one ray per lane, per-lane stack in LDS (global spill beyond
// kLdsStack entries), rays claimed in chunks through one atomic per wave and
// handed to idle lanes by ballot rank.
//
// The arithmetic is the reference's, operation for operation (this file is
// compiled with -ffp-contract=off; IEEE division; denormals kept):
//   vsafe_inverse            nanort.h:442-461  (the `v < 0` sign rule)
//   IntersectRayAABB         nanort.h:2285-2370 (MaxMult 1.00000024f / 1.0000000000000004)
//   PrepareTraversal         nanort.h:1163-1201 (kz = argmax |dir|, strict <)
//   Intersect (watertight)   nanort.h:1054-1150 (fp64 edge fallback, tie rules)
//   Traverse / TestLeafNode  nanort.h:2526-2556, 2374-2407 (near child first,
//                            hit iff t_best < ray.max_t)
#include "common.h"

namespace nrt {

template <typename T>
struct Const;
template <>
struct Const<float> {
  static __device__ __forceinline__ float eps() { return 1.1920928955078125e-07f; }
  static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
  static __device__ __forceinline__ float maxmult() { return 1.00000024f; }
  static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
};
template <>
struct Const<double> {
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
  static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
  static __device__ __forceinline__ double maxmult() { return 1.0000000000000004; }
  static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
};

template <typename T>
__device__ __forceinline__ T sel3(T a0, T a1, T a2, int k) {
  return k == 0 ? a0 : (k == 1 ? a1 : a2);
}

// vsafe_inverse, non-C++11 arm (nanort.h:442-461).
template <typename T>
__device__ __forceinline__ T safe_inverse(T v) {
  if (Const<T>::abs(v) < Const<T>::eps()) {
    T sgn = (v < T(0)) ? T(-1) : T(1);
    return Const<T>::inf() * sgn;
  }
  return T(1.0) / v;
}

// Per-lane traversal state (all registers).
template <typename T>
struct Lane {
  T org[3];
  T inv[3];
  T min_t, max_t, hit_t; // hit_t == intersector t_ == best so far
  T Sx, Sy, Sz;
  T u, v;
  uint32_t prim;
  int kx, ky, kz;
  int sign[3];
};

template <typename T>
__device__ __forceinline__ void lane_init(Lane<T> &L, const typename Wire<T>::Ray &r) {
  T d0 = r.dir[0], d1 = r.dir[1], d2 = r.dir[2];
  L.org[0] = r.org[0];
  L.org[1] = r.org[1];
  L.org[2] = r.org[2];
  L.min_t = r.min_t;
  L.max_t = r.max_t;
  L.hit_t = r.max_t; // nanort.h:2494, 2501
  L.prim = kInvalid;
  L.u = T(0);
  L.v = T(0);
  // PrepareTraversal (nanort.h:1170-1193)
  int kz = 0;
  T a = Const<T>::abs(d0);
  if (a < Const<T>::abs(d1)) {
    kz = 1;
    a = Const<T>::abs(d1);
  }
  if (a < Const<T>::abs(d2)) {
    kz = 2;
    a = Const<T>::abs(d2);
  }
  int kx = kz + 1;
  if (kx == 3) kx = 0;
  int ky = kx + 1;
  if (ky == 3) ky = 0;
  T dz = sel3(d0, d1, d2, kz);
  if (dz < T(0)) {
    int t = kx;
    kx = ky;
    ky = t;
  }
  L.kx = kx;
  L.ky = ky;
  L.kz = kz;
  L.Sx = sel3(d0, d1, d2, kx) / dz;
  L.Sy = sel3(d0, d1, d2, ky) / dz;
  L.Sz = T(1.0) / dz;
  // Traverse prologue (nanort.h:2505-2516)
  L.sign[0] = d0 < T(0) ? 1 : 0;
  L.sign[1] = d1 < T(0) ? 1 : 0;
  L.sign[2] = d2 < T(0) ? 1 : 0;
  L.inv[0] = safe_inverse<T>(d0);
  L.inv[1] = safe_inverse<T>(d1);
  L.inv[2] = safe_inverse<T>(d2);
}

// IntersectRayAABB (nanort.h:2285-2370); safemin/safemax (nanort.h:1236-1243).
template <typename T>
__device__ __forceinline__ bool slab_test(const Lane<T> &L, const T bmin[3], const T bmax[3]) {
  const T mm = Const<T>::maxmult();
  T tmin = L.min_t, tmax = L.hit_t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T lo = L.sign[k] ? bmax[k] : bmin[k];
    const T hi = L.sign[k] ? bmin[k] : bmax[k];
    const T t0 = (lo - L.org[k]) * L.inv[k];
    const T t1 = (hi - L.org[k]) * L.inv[k] * mm;
    tmin = (t0 > tmin) ? t0 : tmin; // safemax(t0, tmin)
    tmax = (t1 < tmax) ? t1 : tmax; // safemin(t1, tmax)
  }
  return tmin <= tmax;
}

// TriangleIntersector::Intersect (nanort.h:1054-1150) against one leaf record.
template <typename T>
__device__ __forceinline__ void tri_test(Lane<T> &L, const LeafTri<T> &tri, uint32_t range0,
                                         uint32_t range1, uint32_t skip, bool cull) {
  const uint32_t prim = tri.prim_id;
  if (prim < range0 || prim >= range1) return;
  if (prim == skip) return;
  const T A0 = tri.p0[0] - L.org[0], A1 = tri.p0[1] - L.org[1], A2 = tri.p0[2] - L.org[2];
  const T B0 = tri.p1[0] - L.org[0], B1 = tri.p1[1] - L.org[1], B2 = tri.p1[2] - L.org[2];
  const T C0 = tri.p2[0] - L.org[0], C1 = tri.p2[1] - L.org[1], C2 = tri.p2[2] - L.org[2];
  const T Akz = sel3(A0, A1, A2, L.kz), Bkz = sel3(B0, B1, B2, L.kz), Ckz = sel3(C0, C1, C2, L.kz);
  const T Ax = sel3(A0, A1, A2, L.kx) - L.Sx * Akz;
  const T Ay = sel3(A0, A1, A2, L.ky) - L.Sy * Akz;
  const T Bx = sel3(B0, B1, B2, L.kx) - L.Sx * Bkz;
  const T By = sel3(B0, B1, B2, L.ky) - L.Sy * Bkz;
  const T Cx = sel3(C0, C1, C2, L.kx) - L.Sx * Ckz;
  const T Cy = sel3(C0, C1, C2, L.ky) - L.Sy * Ckz;
  T U = Cx * By - Cy * Bx;
  T V = Ax * Cy - Ay * Cx;
  T W = Bx * Ay - By * Ax;
  if (U == T(0) || V == T(0) || W == T(0)) { // nanort.h:1094-1107
    const double CxBy = double(Cx) * double(By), CyBx = double(Cy) * double(Bx);
    const double AxCy = double(Ax) * double(Cy), AyCx = double(Ay) * double(Cx);
    const double BxAy = double(Bx) * double(Ay), ByAx = double(By) * double(Ax);
    U = T(CxBy - CyBx);
    V = T(AxCy - AyCx);
    W = T(BxAy - ByAx);
  }
  if (U < T(0) || V < T(0) || W < T(0)) { // nanort.h:1109-1116
    if (cull || (U > T(0) || V > T(0) || W > T(0))) return;
  }
  const T det = U + V + W;
  if (det == T(0)) return;
  const T Az = L.Sz * Akz, Bz = L.Sz * Bkz, Cz = L.Sz * Ckz;
  const T D = U * Az + V * Bz + W * Cz;
  const T rcp = T(1.0) / det;
  const T tt = D * rcp;
  if (tt > L.hit_t) return; // equality accepted (nanort.h:1133)
  if (tt < L.min_t) return; // nanort.h:1137
  L.hit_t = tt;
  L.u = V * rcp;
  L.v = W * rcp;
  L.prim = prim;
}

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

template <typename T, bool COUNT>
__global__ __launch_bounds__(kTraverseBlock) void k_traverse(const TraverseArgs<T> a) {
  // [depth][thread]: a wave's 64 lanes hit 64 consecutive dwords -> conflict-free.
  __shared__ uint32_t s_stack[kLdsStack][kTraverseBlock];

  typedef typename Wire<T>::Node Node;
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;

  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned gslot = blockIdx.x * kTraverseBlock + tid;
  const bool cull = a.cull_back_face != 0;

  Lane<T> L;
  uint32_t rid = kInvalid; // ray this lane is working on
  uint32_t cur = 0;        // node to visit next
  int sp = 0;              // entries on this lane's stack

  // wave-uniform claimed range [chunk_next, chunk_end)
  uint32_t chunk_next = 0, chunk_end = 0;
  bool exhausted = false;

  unsigned long long c_nodes = 0, c_leaves = 0, c_tris = 0, c_stack = 0;

  for (;;) {
    // ---- hand new rays to idle lanes (ballot rank inside the wave's chunk) ----
    unsigned long long idle = __ballot(rid == kInvalid);
    if (idle != 0ull) {
      while (idle != 0ull && !exhausted) {
        if (chunk_next == chunk_end) {
          uint32_t base = 0;
          if (lane == (unsigned)__builtin_ctzll(idle)) base = atomicAdd(a.ray_cursor, a.chunk);
          base = __builtin_amdgcn_readfirstlane(__shfl(base, __builtin_ctzll(idle)));
          if (base >= a.num_rays) {
            exhausted = true;
            break;
          }
          chunk_next = base;
          chunk_end = (a.num_rays - base < a.chunk) ? a.num_rays : base + a.chunk;
        }
        const unsigned want = (unsigned)__builtin_popcountll(idle);
        const unsigned avail = chunk_end - chunk_next;
        const unsigned take = want < avail ? want : avail;
        const unsigned rank = (unsigned)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
        if (rid == kInvalid && rank < take) {
          rid = chunk_next + rank;
          const Ray r = a.rays[rid];
          lane_init<T>(L, r);
          cur = 0;
          sp = 0;
        }
        chunk_next += take;
        idle = __ballot(rid == kInvalid);
      }
      if (exhausted && idle == ~0ull) break; // nothing left anywhere in this wave
    }

    // ---- one traversal step for every live lane --------------------------------
    if (rid != kInvalid) {
      const Node nd = a.nodes[cur];
      if (COUNT) c_nodes++;
      bool descend = false;
      if (slab_test<T>(L, nd.bmin, nd.bmax)) {
        if (nd.flag == 0) {
          const int near = sel3(L.sign[0], L.sign[1], L.sign[2], nd.axis);
          const uint32_t far_child = near ? nd.data[0] : nd.data[1];
          cur = near ? nd.data[1] : nd.data[0];
          // push far; near stays in `cur` (it would be popped next anyway: nanort.h:2542-2543)
          if (sp < kLdsStack) {
            s_stack[sp][tid] = far_child;
          } else {
            a.spill[(size_t)(sp - kLdsStack) * a.spill_stride + gslot] = far_child;
          }
          sp++;
          if (COUNT) {
            // the reference holds near+far on its stack at this point
            unsigned long long need = (unsigned long long)sp + 1ull;
            c_stack = need > c_stack ? need : c_stack;
          }
          descend = true;
        } else {
          const uint32_t cnt = nd.data[0], first = nd.data[1];
          if (COUNT) c_leaves++;
          for (uint32_t i = 0; i < cnt; i++) {
            const LeafTri<T> tri = a.tris[first + i];
            if (COUNT) c_tris++;
            tri_test<T>(L, tri, a.range0, a.range1, a.skip_prim, cull);
          }
        }
      }
      if (!descend) {
        if (sp == 0) {
          // PostTraversal (nanort.h:1205-1211) with the strict final predicate (:2552)
          const bool hit = L.hit_t < L.max_t;
          if (a.hits) {
            Hit h;
            h.u = hit ? L.u : T(0);
            h.v = hit ? L.v : T(0);
            h.t = hit ? L.hit_t : L.max_t;
            h.prim_id = hit ? L.prim : kInvalid;
            a.hits[rid] = h;
          }
          if (a.mask) a.mask[rid] = hit ? 1 : 0;
          rid = kInvalid;
        } else {
          sp--;
          if (sp < kLdsStack) {
            cur = s_stack[sp][tid];
          } else {
            cur = a.spill[(size_t)(sp - kLdsStack) * a.spill_stride + gslot];
          }
        }
      }
    }
  }

  if (COUNT) {
    // wave reduction, then one atomic per wave per counter
    for (int off = 32; off > 0; off >>= 1) {
      c_nodes += __shfl_xor(c_nodes, off);
      c_leaves += __shfl_xor(c_leaves, off);
      c_tris += __shfl_xor(c_tris, off);
      unsigned long long o = __shfl_xor(c_stack, off);
      c_stack = o > c_stack ? o : c_stack;
    }
    if (lane == 0) {
      atomicAdd(&a.counters[0], c_nodes);
      atomicAdd(&a.counters[1], c_leaves);
      atomicAdd(&a.counters[2], c_tris);
      atomicMax(&a.counters[3], c_stack);
    }
  }
}

// Leaf-ordered triangle records from (indices, faces, tight vertices).
template <typename T>
__global__ __launch_bounds__(256) void k_gather_leaf_tris(const uint32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ faces,
                                                          const T *__restrict__ verts,
                                                          LeafTri<T> *__restrict__ out, uint32_t n) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = indices[s];
  const uint32_t f0 = faces[3 * (size_t)prim + 0], f1 = faces[3 * (size_t)prim + 1],
                 f2 = faces[3 * (size_t)prim + 2];
  LeafTri<T> t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    t.p0[k] = verts[3 * (size_t)f0 + k];
    t.p1[k] = verts[3 * (size_t)f1 + k];
    t.p2[k] = verts[3 * (size_t)f2 + k];
  }
  t.prim_id = prim;
  out[s] = t;
}

// ---- host-side launchers (called from api.hip) ------------------------------

template <typename T>
hipError_t launch_traverse(const TraverseArgs<T> &args, unsigned grid, bool count, hipStream_t s) {
  if (count) {
    hipLaunchKernelGGL((k_traverse<T, true>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  } else {
    hipLaunchKernelGGL((k_traverse<T, false>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_gather_leaf_tris(const uint32_t *indices, const uint32_t *faces, const T *verts,
                                   LeafTri<T> *out, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_gather_leaf_tris<T>), dim3((n + 255u) / 256u), dim3(256), 0, s, indices,
                     faces, verts, out, n);
  return hipGetLastError();
}

template hipError_t launch_traverse<float>(const TraverseArgs<float> &, unsigned, bool, hipStream_t);
template hipError_t launch_traverse<double>(const TraverseArgs<double> &, unsigned, bool, hipStream_t);
template hipError_t launch_gather_leaf_tris<float>(const uint32_t *, const uint32_t *, const float *,
                                                   LeafTri<float> *, uint32_t, hipStream_t);
template hipError_t launch_gather_leaf_tris<double>(const uint32_t *, const uint32_t *,
                                                    const double *, LeafTri<double> *, uint32_t,
                                                    hipStream_t);

} // namespace nrt

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

#include <algorithm>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>

namespace nrt {

template <typename T>
struct Const;
template <>
struct Const<float> {
  static __device__ __forceinline__ float eps() { return 1.1920928955078125e-07f; }
  static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
  static __device__ __forceinline__ float maxmult() { return 1.00000024f; }
  static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
  static __device__ __forceinline__ float fmax(float a, float b) { return __builtin_fmaxf(a, b); }
  static __device__ __forceinline__ float fmin(float a, float b) { return __builtin_fminf(a, b); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
  static __device__ __forceinline__ float fltmax() { return 3.402823466e+38f; }
};
template <>
struct Const<double> {
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
  static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
  static __device__ __forceinline__ double maxmult() { return 1.0000000000000004; }
  static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
  static __device__ __forceinline__ double fmax(double a, double b) { return __builtin_fmax(a, b); }
  static __device__ __forceinline__ double fmin(double a, double b) { return __builtin_fmin(a, b); }
  static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
  static __device__ __forceinline__ double fltmax() { return 3.402823466e+38; } // the example is fp32 only
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
  // (scalars, not arrays, and no two floats of one kind next to each other: the vectoriser otherwise merges neighbours
  // into overlapping vector accesses that pin parts of the lane state in scratch memory)
  T org0, inv0, org1, inv1, org2, inv2;
  __device__ __forceinline__ T org(int k) const { return k == 0 ? org0 : (k == 1 ? org1 : org2); }
  __device__ __forceinline__ T inv(int k) const { return k == 0 ? inv0 : (k == 1 ? inv1 : inv2); }
  T min_t, max_t, hit_t; // hit_t == intersector t_ == best so far
  T d0, d1, d2;          // ray direction (sphere / cylinder kinds; dead otherwise)
  uint32_t cap;          // cylinder kind: hit_cap_ of the accepted hit (u, v hold u_param_, v_param_)
  // (floats and integers alternate on purpose: as neighbours, Sx Sy Sz u v get merged into overlapping two- and
  // four-float vector accesses by the vectoriser, which then pins all five in scratch memory instead of registers)
  T Sx;
  uint32_t pk; // (dir < 0 per axis) << 0..2 | kx << 3 | ky << 5 | kz << 7: six small integers in one register (the kernel sits near
               // the 80-register edge of six waves per SIMD; the loops turn the fields into lane masks once, on entry).  The signs
               // are the lowest bits so that sign(axis) is ONE bit-field extract at `axis` — three instructions fewer per step
               // than with the signs above the axes (round 6)
  T Sy;
  uint32_t prim;
  T Sz;
  uint32_t so0; // 48 if dir[k] < 0 else 0: byte offset of the ray's NEAR plane row inside a Wide4Node (bmax rows sit 48 bytes after bmin rows)
  T u;
  uint32_t so1;
  T v;
  uint32_t so2;
  __device__ __forceinline__ int kx() const { return (int)((pk >> 3) & 3u); }
  __device__ __forceinline__ int ky() const { return (int)((pk >> 5) & 3u); }
  __device__ __forceinline__ int kz() const { return (int)((pk >> 7) & 3u); }
  __device__ __forceinline__ int sign(int k) const { return (int)((pk >> k) & 1u); }
};

template <typename T>
__device__ __forceinline__ void lane_init(Lane<T> &L, const typename Wire<T>::Ray &r) {
  T d0 = r.dir[0], d1 = r.dir[1], d2 = r.dir[2];
  L.d0 = d0;
  L.d1 = d1;
  L.d2 = d2;
  L.org0 = r.org[0];
  L.org1 = r.org[1];
  L.org2 = r.org[2];
  L.min_t = r.min_t;
  L.max_t = r.max_t;
  L.hit_t = r.max_t; // nanort.h:2494, 2501
  L.prim = kInvalid;
  L.cap = 0u;
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
  uint32_t pk = ((uint32_t)kx << 3) | ((uint32_t)ky << 5) | ((uint32_t)kz << 7);
  L.Sx = sel3(d0, d1, d2, kx) / dz;
  L.Sy = sel3(d0, d1, d2, ky) / dz;
  L.Sz = T(1.0) / dz;
  // Traverse prologue (nanort.h:2505-2516)
  pk |= (d0 < T(0) ? 1u : 0u) | (d1 < T(0) ? 2u : 0u) | (d2 < T(0) ? 4u : 0u);
  L.pk = pk;
  L.so0 = d0 < T(0) ? 48u : 0u;
  L.so1 = d1 < T(0) ? 48u : 0u;
  L.so2 = d2 < T(0) ? 48u : 0u;
  L.inv0 = safe_inverse<T>(d0);
  L.inv1 = safe_inverse<T>(d1);
  L.inv2 = safe_inverse<T>(d2);
}

// IntersectRayAABB (nanort.h:2285-2370); safemin/safemax (nanort.h:1236-1243).
template <typename T>
__device__ __forceinline__ bool slab_test(const Lane<T> &L, const T bmin[3], const T bmax[3]) {
  const T mm = Const<T>::maxmult();
  T tmin = L.min_t, tmax = L.hit_t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int sg = L.sign(k);
    const T lo = sg ? bmax[k] : bmin[k];
    const T hi = sg ? bmin[k] : bmax[k];
    const T t0 = (lo - L.org(k)) * L.inv(k);
    const T t1 = (hi - L.org(k)) * L.inv(k) * mm;
    // safemax(t0, tmin) / safemin(t1, tmax) (nanort.h:1236-1243): a NaN first operand is dropped and the
    // running value is never NaN, which is exactly maxNum/minNum (v_max_f32 / v_min_f32); the only
    // difference, the sign of a zero result, cannot change `tmin <= tmax`.
    tmin = Const<T>::fmax(t0, tmin);
    tmax = Const<T>::fmin(t1, tmax);
  }
  return tmin <= tmax;
}

// TriangleIntersector::Intersect (nanort.h:1054-1150) against one leaf record.
// Written as one running predicate with select-style updates (the reference's early returns
// in the same order): all loads of the record are issued together, and the lane state stays
// in the same registers on every path.
template <typename T, bool PLAIN = false>
__device__ __forceinline__ void tri_test(Lane<T> &L, const LeafTri<T> &tri, bool active, uint32_t range0,
                                         uint32_t range1, uint32_t skip, bool cull) {
  const uint32_t prim = tri.prim_id;
  bool ok = PLAIN ? active : (active & (prim >= range0) & (prim < range1) & (prim != skip)); // nanort.h:2387-2395
  if (PLAIN) cull = false;
  const T A0 = tri.p0[0] - L.org0, A1 = tri.p0[1] - L.org1, A2 = tri.p0[2] - L.org2;
  const T B0 = tri.p1[0] - L.org0, B1 = tri.p1[1] - L.org1, B2 = tri.p1[2] - L.org2;
  const T C0 = tri.p2[0] - L.org0, C1 = tri.p2[1] - L.org1, C2 = tri.p2[2] - L.org2;
  const T Akz = sel3(A0, A1, A2, L.kz()), Bkz = sel3(B0, B1, B2, L.kz()), Ckz = sel3(C0, C1, C2, L.kz());
  const T Ax = sel3(A0, A1, A2, L.kx()) - L.Sx * Akz;
  const T Ay = sel3(A0, A1, A2, L.ky()) - L.Sy * Akz;
  const T Bx = sel3(B0, B1, B2, L.kx()) - L.Sx * Bkz;
  const T By = sel3(B0, B1, B2, L.ky()) - L.Sy * Bkz;
  const T Cx = sel3(C0, C1, C2, L.kx()) - L.Sx * Ckz;
  const T Cy = sel3(C0, C1, C2, L.ky()) - L.Sy * Ckz;
  T U = Cx * By - Cy * Bx;
  T V = Ax * Cy - Ay * Cx;
  T W = Bx * Ay - By * Ax;
  if (ok && (U == T(0) || V == T(0) || W == T(0))) { // nanort.h:1094-1107 (rare: a wave-level branch)
    const double CxBy = double(Cx) * double(By), CyBx = double(Cy) * double(Bx);
    const double AxCy = double(Ax) * double(Cy), AyCx = double(Ay) * double(Cx);
    const double BxAy = double(Bx) * double(Ay), ByAx = double(By) * double(Ax);
    U = T(CxBy - CyBx);
    V = T(AxCy - AyCx);
    W = T(BxAy - ByAx);
  }
  const bool neg = (U < T(0)) | (V < T(0)) | (W < T(0)); // nanort.h:1109-1116
  const bool pos = (U > T(0)) | (V > T(0)) | (W > T(0));
  ok = ok & !(neg & (cull | pos));
  const T det = U + V + W;
  ok = ok & !(det == T(0));
  if (ok) { // skipped by the whole wave when no lane got this far
    const T Az = L.Sz * Akz, Bz = L.Sz * Bkz, Cz = L.Sz * Ckz;
    const T D = U * Az + V * Bz + W * Cz;
    const T rcp = T(1.0) / det;
    const T tt = D * rcp;
    // `if (tt > t) return; if (tt < min_t) return;` — equality (and NaN) accepted (nanort.h:1133-1139)
    const bool acc = !(tt > L.hit_t) & !(tt < L.min_t);
    const T uu = V * rcp, vv = W * rcp;
    L.hit_t = acc ? tt : L.hit_t;
    L.u = acc ? uu : L.u;
    L.v = acc ? vv : L.v;
    L.prim = acc ? prim : L.prim;
  }
}

// SphereIntersector::Intersect (examples/particle_primitive/main.cc:161-236) against one leaf record: the
// quadratic in the reference's own operation order (vdot = (x*x + y*y) + z*z, nanort.h:410-412), IEEE sqrt and
// divisions, no contraction.  No min_t test and no skip_prim_id in that intersector; equality with the best t
// is accepted (`if (t > *t_inout) return false`).
template <typename T>
__device__ __forceinline__ void sphere_test(Lane<T> &L, const LeafSphere<T> &sp, bool active, uint32_t range0,
                                            uint32_t range1) {
  const uint32_t prim = sp.prim_id;
  bool ok = active & (prim >= range0) & (prim < range1);
  const T oc0 = L.org0 - sp.c[0], oc1 = L.org1 - sp.c[1], oc2 = L.org2 - sp.c[2];
  const T a = (L.d0 * L.d0 + L.d1 * L.d1) + L.d2 * L.d2;
  const T b = T(2.0) * ((L.d0 * oc0 + L.d1 * oc1) + L.d2 * oc2);
  const T c = ((oc0 * oc0 + oc1 * oc1) + oc2 * oc2) - sp.r * sp.r;
  const T disc = b * b - T(4.0) * a * c;
  ok = ok & !(disc < T(0));
  if (ok) {
    T t0, t1;
    if (Const<T>::abs(disc) < Const<T>::eps()) {
      t0 = t1 = T(-0.5) * (b / a);
    } else {
      const T ds = Const<T>::sqrt(disc);
      const T q = (b < T(0)) ? (-b - ds) / T(2.0) : (-b + ds) / T(2.0);
      t0 = q / a;
      t1 = c / q;
    }
    if (t0 > t1) {
      const T tmp = t0;
      t0 = t1;
      t1 = tmp;
    }
    const T t = (t0 < T(0)) ? t1 : t0;
    const bool acc = ok & !(t1 < T(0)) & !(t > L.hit_t);
    L.hit_t = acc ? t : L.hit_t;
    L.prim = acc ? prim : L.prim;
  }
}

// CylinderIntersector::Intersect (examples/cylinder_primitive/main.cc:237-343) with solve2e (:61-90) against one leaf
// record, as one running predicate (the reference's early returns in the same order, every comparison in the
// reference's own form so that NaNs take the same side).  The intersector's mutable members hit_cap_, u_param_,
// v_param_ are the lane's cap, u, v: as there, they change exactly when the primitive is accepted.
template <typename T>
__device__ __forceinline__ void cyl_normalize(const T a[3], T o[3]) { // vnormalize, nanort.h:383-398
  const T len = Const<T>::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
  o[0] = a[0];
  o[1] = a[1];
  o[2] = a[2];
  if (Const<T>::abs(len) > Const<T>::eps()) {
    const T inv_len = T(1.0) / len;
    o[0] *= inv_len;
    o[1] *= inv_len;
    o[2] *= inv_len;
  }
}
template <typename T>
__device__ __forceinline__ T cyl_dot(const T a[3], const T b[3]) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

template <typename T>
__device__ __forceinline__ void cylinder_test(Lane<T> &L, const LeafCylinder<T> &cy, bool active, uint32_t range0,
                                              uint32_t range1, bool test_cap) {
  const uint32_t prim = cy.prim_id;
  const bool ok = active & (prim >= range0) & (prim < range1);
  const T kEPS = T(1.0e-6f);
  const T org[3] = {L.org0, L.org1, L.org2}, dir[3] = {L.d0, L.d1, L.d2};
  const T tmax = L.hit_t;
  const T rr = (cy.r0 < cy.r1) ? cy.r1 : cy.r0; // std::max(r0, r1)
  T d[3], m[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    d[k] = cy.p1[k] - cy.p0[k];
    m[k] = org[k] - cy.p0[k];
  }
  const T md = cyl_dot(m, d), nd = cyl_dot(dir, d), dd = cyl_dot(d, d);
  bool hitCap = false;
  T capT = Const<T>::fltmax();
  T t_new = tmax, u_new = L.u, v_new = L.v;
  uint32_t cap_new = L.cap;
  if (test_cap) {
    T t01[3], dN0[3], dN1[3], rd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) t01[k] = cy.p0[k] - cy.p1[k];
    cyl_normalize<T>(t01, dN0);
#pragma unroll
    for (int k = 0; k < 3; k++) dN1[k] = -dN0[k];
    cyl_normalize<T>(dir, rd);
    const bool facing = Const<T>::abs(cyl_dot(dir, dN0)) > kEPS;
    const T p0D = -cyl_dot(cy.p0, dN0), p1D = -cyl_dot(cy.p1, dN1);
    const T p0T = -(cyl_dot(org, dN0) + p0D) / cyl_dot(rd, dN0);
    const T p1T = -(cyl_dot(org, dN1) + p1D) / cyl_dot(rd, dN1);
    T e0[3], e1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      e0[k] = (org[k] + rd[k] * p0T) - cy.p0[k];
      e1[k] = (org[k] + rd[k] * p1T) - cy.p1[k];
    }
    const T qp0Sqr = cyl_dot(e0, e0), qp1Sqr = cyl_dot(e1, e1);
    const bool c0 = facing & (p0T > T(0)) & (p0T < tmax) & (qp0Sqr < rr * rr);
    hitCap = c0;
    capT = c0 ? p0T : capT;
    t_new = c0 ? p0T : t_new;
    u_new = c0 ? Const<T>::sqrt(qp0Sqr) : u_new;
    v_new = c0 ? T(0) : v_new;
    const bool c1 = facing & (p1T > T(0)) & (p1T < tmax) & (p1T < capT) & (qp1Sqr < rr * rr);
    hitCap = hitCap | c1;
    capT = c1 ? p1T : capT;
    t_new = c1 ? p1T : t_new;
    u_new = c1 ? Const<T>::sqrt(qp1Sqr) : u_new;
    v_new = c1 ? T(1.0) : v_new;
    cap_new = hitCap ? 1u : cap_new;
  }
  bool accept = hitCap;
  const bool outside = ((md <= T(0)) & (nd <= T(0))) | ((md >= dd) & (nd >= T(0)));
  {
    const T nn = cyl_dot(dir, dir), mn = cyl_dot(m, dir);
    const T A = dd * nn - nd * nd;
    const T kk = cyl_dot(m, m) - rr * rr;
    const T C = dd * kk - md * md;
    const T B = dd * mn - nd * md;
    // solve2e: the smaller root (root[0]) and whether there is one
    T root;
    bool have;
    if (Const<T>::abs(A) <= kEPS) {
      root = -C / B;
      have = true;
    } else {
      const T D = B * B - A * C;
      if (D < T(0)) {
        root = T(0);
        have = false;
      } else if (D == T(0)) {
        root = -B / A;
        have = true;
      } else {
        T x1 = (Const<T>::abs(B) + Const<T>::sqrt(D)) / A;
        if (B >= T(0)) x1 = -x1;
        const T x2 = C / (A * x1);
        root = (x1 > x2) ? x2 : x1;
        have = true;
      }
    }
    const T t = root;
    T sv = md + t * nd;
    sv = sv / dd;
    const bool side = !outside & have & (T(0) <= t) & (t <= tmax) & (t <= capT) & (T(0) <= sv) & (sv <= T(1));
    accept = accept | side;
    t_new = side ? t : t_new;
    u_new = side ? T(0) : u_new;
    v_new = side ? sv : v_new;
    cap_new = side ? 0u : cap_new;
  }
  accept = accept & ok;
  L.hit_t = accept ? t_new : L.hit_t;
  L.u = accept ? u_new : L.u;
  L.v = accept ? v_new : L.v;
  L.cap = accept ? cap_new : L.cap;
  L.prim = accept ? prim : L.prim;
}

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Which batch a virtual ray index lies in (multi-batch launches: common.h BatchPtrs; the ends are wave-uniform scalars).
template <typename T>
__device__ __forceinline__ uint32_t batch_of(const TraverseArgs<T> &a, uint32_t rid) {
  uint32_t b = 0;
#pragma unroll
  for (int k = 0; k + 1 < kMaxBatches; k++) b += ((uint32_t)(k + 1) < a.num_batches && rid >= a.batch_end[k]) ? 1u : 0u;
  return b;
}
// ... and the batch table copied from the kernel arguments into LDS once per block, so that lanes can index it by their own batch.
#define NRT_BATCH_TABLE_SETUP()                                                                        \
  __shared__ BatchPtrs s_tbl[sizeof(T) == 4 ? kMaxBatches : 1];                                        \
  const bool multi = sizeof(T) == 4 && a.num_batches > 1u; /* (wave-uniform) */                        \
  if (multi) {                                                                                         \
    _Pragma("unroll") for (int k_ = 0; k_ < kMaxBatches; k_++)                                         \
      if (threadIdx.x == (unsigned)k_) s_tbl[sizeof(T) == 4 ? k_ : 0] = a.batches[k_];                 \
    __syncthreads();                                                                                   \
  }

// Work distribution of the persistent kernels.
//  * The batch is cut into `static_bands` equal bands (+ a short tail).  The first part of every band is handed out
//    STATICALLY, without any atomic: slice `rank` of band b, rays [b * band_len + rank * static_per_wave, +static_per_wave),
//    belongs to wave `rank`.  Ranks are XCD-major (the dispatcher places block b on XCD b % 8 — used for L2 affinity only,
//    never for correctness), so at any moment the waves of an XCD walk neighbouring slices of one band and its L2 keeps one
//    part of the tree.  Every wave samples every band: an image whose cost per ray varies from region to region (C2: 40 %
//    sky) does not leave one XCD with the expensive rows.
//  * The rest of every band (and the tail) is claimed DYNAMICALLY, `chunk` rays per atomicAdd.  These rays form one virtual
//    array (band 0's dynamic part, band 1's, ..., the tail) that is cut into `num_parts` ranges with one cursor each (4 KiB
//    apart); a wave drains its home range first and then steals from the others.  Band parts and ranges are whole chunks, so a
//    chunk never straddles two bands.  Because the dynamic rays come from all over the batch too (round 2: the last quarter
//    of the array — for a camera wave the bottom of the image), what is left to balance the end of a launch is a sample of the
//    whole batch, not its cheapest corner.
//    (Device-scope atomics on one word saturate near 100 per microsecond on this part, hence the static share and the modest
//    chunk count.  Round 2 re-measured static share 0-75 %, chunks of 16-128 rays, claims issued one chunk ahead of need:
//    nothing beats 75 % / 128; profiles/r02d_scheduling_sweep.txt.  Round 6, with finer steps: the static share is 16 % — ONE
//    64-ray group per wave at 1080p, enough to start every wave without an atomic — because whatever a wave owns nobody can take
//    from it when the cost per ray is uneven over the image: C2 +7 %, C3 +2.8 %, profiles/r06y_distribution_static_share.txt.)
//  * A batch too small for a static group per wave has no static share at all; its waves then own the FIRST chunk of their home
//    range (chunk `wave index`, no atomic: claim_init) and the cursors count from the range's wave count.
struct Claim {
  uint32_t next, end; // claimed, not yet handed out: [next, end)
  uint32_t part, tried;
  uint32_t rank, band; // static share: this wave's rank, the band its current slice lies in
  bool exhausted;
};

// Chunk `idx` of cursor range `part`: where it starts in the virtual array of dynamic rays and how many rays it holds (false: past
// the range's end).  The cursor counts CHUNKS: the first `main_chunks` are whole ones, the rest of the range goes out in half chunks
// (tunable chunk_tail_pct) — the last rays of a launch in finer portions; half chunks subdivide whole ones, so no chunk straddles two
// bands.
template <typename T>
__device__ __forceinline__ bool chunk_of(const TraverseArgs<T> &a, uint32_t part, uint32_t idx, uint32_t &v, uint32_t &cnt) {
  const uint32_t lo = part * a.dyn_per_part; // range of this part in the virtual array of dynamic rays
  const uint32_t len = (part + 1u == a.num_parts) ? a.dyn_total - lo : a.dyn_per_part;
  const uint32_t main_chunks = (uint32_t)(((unsigned long long)(len / a.chunk) * (100u - a.chunk_tail_pct)) / 100u), half = a.chunk >> 1;
  const uint32_t base = idx < main_chunks ? idx * a.chunk : main_chunks * a.chunk + (idx - main_chunks) * half;
  const uint32_t want = idx < main_chunks ? a.chunk : half;
  if (!(idx < 0x1000000u && base < len)) return false;
  v = lo + base;
  cnt = (len - base < want) ? len - base : want;
  return true;
}
// ... and where those rays lie in the batch
template <typename T>
__device__ __forceinline__ uint32_t dyn_to_real(const TraverseArgs<T> &a, uint32_t v) {
  if (v < a.dyn_banded) { // inside band b's dynamic part
    const uint32_t b = v / a.dyn_per_band;
    return b * a.band_len + a.band_static + (v - b * a.dyn_per_band);
  }
  return a.tail_begin + (v - a.dyn_banded); // the tail behind the last band
}

template <typename T>
__device__ __forceinline__ void claim_init(const TraverseArgs<T> &a, Claim &c) {
  const uint32_t part = blockIdx.x % a.num_parts;
  // (readfirstlane: the compiler cannot see that threadIdx.x / 64 is the same in every lane; told so, it keeps the whole
  // claim state in scalar registers)
  const uint32_t rank = (part * a.blocks_per_part + blockIdx.x / a.num_parts) * (kTraverseBlock / kWave) +
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  c.rank = rank;
  c.band = 0;
  c.next = rank * a.static_per_wave;
  c.end = c.next + a.static_per_wave; // (static_per_wave == 0: nothing is owned statically)
  c.part = part;
  c.tried = 0;
  c.exhausted = false;
  if (a.dyn_head != 0u && a.static_per_wave == 0u) {
    // A batch too small for a static group per wave (fewer than ~400 rays per wave at the default share): no static share, but no
    // start-up burst on the cursors either — chunk `wi` of the wave's HOME range belongs to wave `wi` of that partition without an
    // atomic (the cursors then count from the partition's wave count: claim_chunk), so the launch starts inside every XCD's own
    // strip of the batch and everything after a wave's first chunk is balanced dynamically.  (C3's mesh at 1600x960 +3 %, C2 +2.5 %
    // through its 1.24 M-ray bounce wave: profiles/r06z_dyn_head_small.txt.)
    const uint32_t wi = rank - part * a.blocks_per_part * (uint32_t)(kTraverseBlock / kWave);
    uint32_t v, cnt;
    if (chunk_of<T>(a, part, wi, v, cnt)) {
      c.next = dyn_to_real<T>(a, v);
      c.end = c.next + cnt;
    } else {
      c.next = c.end = 0u;
    }
  }
}

// All lanes of the wave call this (uniform control flow); `leader` is any active lane index.
template <typename T>
__device__ __forceinline__ bool claim_chunk(const TraverseArgs<T> &a, Claim &c, unsigned lane, int leader) {
  if (a.static_per_wave != 0u && c.band + 1u < a.static_bands) { // the wave's slice of the next band (no atomic)
    c.band++;
    c.next = c.band * a.band_len + c.rank * a.static_per_wave;
    c.end = c.next + a.static_per_wave;
    return true;
  }
  while (c.tried < a.num_parts) {
    uint32_t idx = 0;
    if (lane == (unsigned)leader) idx = atomicAdd(a.ray_cursor + kCursorStrideWords * c.part, 1u);
    idx = __builtin_amdgcn_readfirstlane(__shfl(idx, leader));
    if (a.dyn_head != 0u && a.static_per_wave == 0u) idx += a.blocks_per_part * (uint32_t)(kTraverseBlock / kWave); // (the first chunks of every range have owners: claim_init)
    uint32_t v, cnt;
    if (chunk_of<T>(a, c.part, idx, v, cnt)) {
      c.next = dyn_to_real<T>(a, v);
      c.end = c.next + cnt;
      return true;
    }
    c.part = (c.part + 1 == a.num_parts) ? 0 : c.part + 1;
    c.tried++;
  }
  c.exhausted = true;
  return false;
}

// Streaming accesses (each ray is read once, each hit written once): keep them out of the
// way of the tree data in L2 with the non-temporal hint.
template <typename T>
__device__ __forceinline__ typename Wire<T>::Ray load_ray_nt(const typename Wire<T>::Ray *p) {
  typename Wire<T>::Ray r;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(p);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
  for (unsigned k = 0; k < sizeof(r) / 4; k++) dst[k] = __builtin_nontemporal_load(src + k);
  return r;
}
template <typename T>
__device__ __forceinline__ void store_hit_nt(typename Wire<T>::Hit *p, const typename Wire<T>::Hit &h) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *src = reinterpret_cast<const u32x4 *>(&h);
  u32x4 *dst = reinterpret_cast<u32x4 *>(p);
#pragma unroll
  for (unsigned k = 0; k < sizeof(h) / 16; k++) __builtin_nontemporal_store(src[k], dst + k);
}

// Completion record of a launch (common.h, DoneRec): no event is recorded in the stream for it.  Start — one thread of each
// of the first eight blocks stamps the time; end — every wave counts itself out of its block's group (blockIdx % 8: eight
// counters, so that the exit atomics of ~5000 waves do not queue on one word), a group's last wave counts the group out,
// and the last group publishes the two stamps and then the launch's sequence number to the page-locked record.  No LDS
// (the fp64 variants use all of it for their stacks), no barrier, nothing fenced: a waiter learns that every wave has
// stopped READING the tree and the slot's scratch (what a rebuild or the slot's next launch must know), not that the hit
// records have landed — for that the caller synchronises its stream as usual.  Every word a later launch depends on is
// handed on by an atomic (performed at the memory side, visible to every XCD), not left dirty in one XCD's L2.
template <typename T>
__device__ __forceinline__ void done_begin(const TraverseArgs<T> &a) {
  if (a.done_rec != nullptr && threadIdx.x == 0u && blockIdx.x < 8u)
    atomicMin(&a.done_count->t_begin, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
// The same hand-off at the end of an ordinary (non-persistent) grid — the post passes of the sphere and cylinder kinds, which
// then close the launch's record in place of the traversal kernel: every block counts itself out once all its threads are
// past their reads.
__device__ __forceinline__ void done_end_blocks(DoneRec *rec, DoneCount *cnt, uint32_t seq) {
  if (rec == nullptr) return; // (uniform)
  __syncthreads();
  if (threadIdx.x != 0u) return;
  const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u, g = blockIdx.x % 8u;
  const uint32_t group_blocks = (gridDim.x - g + 7u) / 8u;
  if (atomicAdd(&cnt->group[g], 1u) != group_blocks - 1u) return;
  (void)atomicExch(&cnt->group[g], 0u);
  if (atomicAdd(&cnt->exited, 1u) != groups - 1u) return;
  (void)atomicExch(&cnt->exited, 0u);
  const unsigned long long t0 = atomicExch(&cnt->t_begin, ~0ull);
  __hip_atomic_store(&rec->t_begin, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&rec->t_end, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&rec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T>
__device__ __forceinline__ void done_end(const TraverseArgs<T> &a, unsigned lane) {
  if (a.done_rec == nullptr || !a.done_publish || lane != 0u) return; // (done_publish == 0: a post pass closes the record)
  const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u, g = blockIdx.x % 8u;
  const uint32_t group_waves = ((gridDim.x - g + 7u) / 8u) * (uint32_t)(kTraverseBlock / kWave);
  DoneCount *cnt = a.done_count;
  if (atomicAdd(&cnt->group[g], 1u) != group_waves - 1u) return; // not the group's last wave
  (void)atomicExch(&cnt->group[g], 0u); // (handed on clean to the slot's next launch)
  if (atomicAdd(&cnt->exited, 1u) != groups - 1u) return; // not the launch's last group
  (void)atomicExch(&cnt->exited, 0u);
  const unsigned long long t0 = atomicExch(&cnt->t_begin, ~0ull);
  DoneRec *r = a.done_rec;
  __hip_atomic_store(&r->t_begin, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&r->t_end, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&r->seq, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Lane states of the while-while loop.
enum : int { LANE_IDLE = 0, LANE_TRAV = 1, LANE_LEAF = 2 };

template <typename T, bool COUNT, int STACK>
__global__ __launch_bounds__(kTraverseBlock) void k_traverse(const TraverseArgs<T> a) {
  // [depth][thread]: a wave's 64 lanes hit 64 consecutive dwords -> conflict-free.
  __shared__ uint32_t s_stack[STACK][kTraverseBlock];

  typedef typename Wire<T>::Node Node;
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;

  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned gslot = blockIdx.x * kTraverseBlock + tid;
  const bool cull = a.cull_back_face != 0;

  Lane<T> L;
  uint32_t rid = kInvalid; // ray this lane is working on
  uint32_t cur = 0;        // node to visit next (LANE_TRAV)
  uint32_t leaf_first = 0, leaf_cnt = 0; // pending leaf (LANE_LEAF)
  int state = LANE_IDLE;
  int sp = 0; // entries on this lane's stack

  Claim ck; // wave-uniform claimed range
  claim_init<T>(a, ck);
  if (blockIdx.x == 0 && threadIdx.x < kMaxParts) a.next_cursor[kCursorStrideWords * threadIdx.x] = 0u;

  unsigned long long c_nodes = 0, c_leaves = 0, c_tris = 0, c_stack = 0;

  // Pop the next node, or finish the ray when the stack is empty:
  // PostTraversal (nanort.h:1205-1211) with the strict final predicate (:2552).
  // (A macro with select-style updates: every state variable is assigned on both
  // paths, which keeps them all in registers.)
#define NRT_POP_OR_FINISH()                                                              \
  do {                                                                                   \
    const bool fin_ = (sp == 0);                                                         \
    if (fin_) {                                                                          \
      const bool hit_ = L.hit_t < L.max_t;                                               \
      if (a.hits) {                                                                      \
        Hit h_;                                                                          \
        h_.u = hit_ ? L.u : T(0);                                                        \
        h_.v = hit_ ? L.v : T(0);                                                        \
        h_.t = hit_ ? L.hit_t : L.max_t;                                                 \
        h_.prim_id = hit_ ? L.prim : kInvalid;                                           \
        store_hit_nt<T>(a.hits + rid, h_);                                               \
      }                                                                                  \
      if (a.mask) a.mask[rid] = hit_ ? 1 : 0;                                            \
    }                                                                                    \
    uint32_t popped_ = cur;                                                              \
    if (!fin_) {                                                                         \
      const int sp1_ = sp - 1;                                                           \
      if (sp1_ < STACK) {                                                                \
        popped_ = s_stack[sp1_][tid];                                                    \
      } else {                                                                           \
        popped_ = a.spill[(size_t)(sp1_ - STACK) * a.spill_stride + gslot];              \
      }                                                                                  \
    }                                                                                    \
    cur = popped_;                                                                       \
    sp = fin_ ? sp : sp - 1;                                                             \
    rid = fin_ ? kInvalid : rid;                                                         \
    state = fin_ ? LANE_IDLE : LANE_TRAV;                                                \
  } while (0)

  for (;;) {
    // ---- hand new rays to idle lanes (ballot rank inside the wave's chunk) ----
    unsigned long long idle = __ballot(state == LANE_IDLE);
    if (!ck.exhausted && (unsigned)__builtin_popcountll(idle) >= a.refill_min) {
      while (idle != 0ull && !ck.exhausted) {
        if (ck.next == ck.end && !claim_chunk<T>(a, ck, lane, __builtin_ctzll(idle))) break;
        const unsigned want = (unsigned)__builtin_popcountll(idle);
        const unsigned avail = ck.end - ck.next;
        const unsigned take = want < avail ? want : avail;
        const unsigned rank = (unsigned)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
        if (state == LANE_IDLE && rank < take) {
          rid = ck.next + rank;
          const Ray r = (a.debug_flags & 4u) ? a.rays[rid] : load_ray_nt<T>(a.rays + rid);
          lane_init<T>(L, r);
          cur = 0;
          sp = 0;
          state = LANE_TRAV;
          if (COUNT) c_stack = c_stack > 1ull ? c_stack : 1ull;
        }
        ck.next += take;
        idle = __ballot(state == LANE_IDLE);
      }
    }
    if (idle == ~0ull) {
      if (ck.exhausted) break; // nothing left anywhere in this wave
      continue;             // (cannot happen: refill_min <= 64)
    }

    // ---- phase 1: inner nodes, until this lane reaches a leaf or finishes -----------
    while (state == LANE_TRAV) {
      const Node nd = a.nodes[cur];
      if (COUNT) c_nodes++;
      if (slab_test<T>(L, nd.bmin, nd.bmax)) {
        if (nd.flag == 0) {
          const int near = L.sign(nd.axis);
          const uint32_t far_child = near ? nd.data[0] : nd.data[1];
          cur = near ? nd.data[1] : nd.data[0];
          // push far; near stays in `cur` (it would be popped next anyway: nanort.h:2542-2543)
          if (sp < STACK) {
            s_stack[sp][tid] = far_child;
          } else {
            a.spill[(size_t)(sp - STACK) * a.spill_stride + gslot] = far_child;
          }
          sp++;
          if (COUNT) {
            // the reference holds near+far on its stack at this point
            const unsigned long long need = (unsigned long long)sp + 1ull;
            c_stack = need > c_stack ? need : c_stack;
          }
        } else {
          leaf_cnt = nd.data[0];
          leaf_first = nd.data[1];
          state = LANE_LEAF;
          if (COUNT) c_leaves++;
        }
      } else {
        NRT_POP_OR_FINISH();
      }
      // leave early when too few lanes are still walking inner nodes
      if ((unsigned)__builtin_popcountll(__ballot(state == LANE_TRAV)) < a.trav_min) break;
    }

    // ---- phase 2: lanes holding a leaf test its triangles together ------------------
    if (__ballot(state == LANE_LEAF) != 0ull) {
      const uint32_t cnt = state == LANE_LEAF ? leaf_cnt : 0u;
      for (uint32_t i = 0; __ballot(i < cnt) != 0ull; i++) {
        if (i < cnt) {
          const LeafTri<T> tri = a.tris[leaf_first + i];
          if (COUNT) c_tris++;
          tri_test<T>(L, tri, true, a.range0, a.range1, a.skip_prim, cull);
        }
      }
      if (state == LANE_LEAF) NRT_POP_OR_FINISH();
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
#undef NRT_POP_OR_FINISH
}

// ---------------------------------------------------------------------------
// Production kernel: same traversal ORDER and same culling decisions as the
// binary loop above (hence the same hit records, ties included), but each step
// fetches one WideNode and tests both children.  A child that passes its slab
// test is entered at once (near first) or pushed with its t_min; a popped entry
// is entered iff t_min <= hit_t, which is exactly the reference's test at pop
// time, because its slab test factors into (t_min <= t_max of the planes) — which
// does not depend on hit_t and was established at push time — and
// (t_min <= hit_t) (nanort.h:2315-2318: hit_t is the innermost operand of the
// safemin chain).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool slab_test_tmin(const Lane<T> &L, const T box[6], T &tmin_out) {
  const T mm = Const<T>::maxmult();
  T tmin = L.min_t, tmax = L.hit_t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int sg = L.sign(k);
    const T lo = sg ? box[3 + k] : box[k];
    const T hi = sg ? box[k] : box[3 + k];
    const T t0 = (lo - L.org(k)) * L.inv(k);
    const T t1 = (hi - L.org(k)) * L.inv(k) * mm;
    tmin = Const<T>::fmax(t0, tmin); // see slab_test
    tmax = Const<T>::fmin(t1, tmax);
  }
  tmin_out = tmin;
  return tmin <= tmax;
}

enum : int { W_IDLE = 0, W_TRAV = 1, W_LEAF = 2, W_POP = 3 };

// SphereIntersector::PostTraversal (examples/particle_primitive/main.cc:262-277) as a pass over the finished
// hit records (the double-precision atan2/acos would otherwise cost the traversal kernel half its occupancy):
// u, v = spherical coordinates of the unit normal at the hit point; atan2/acos in double as there (device
// libm agrees with glibc to the last place or so of the double, i.e. to ~1 ulp of the float result).
template <typename T>
__global__ __launch_bounds__(256) void k_sphere_uv(const typename Wire<T>::Ray *__restrict__ rays,
                                                   typename Wire<T>::Hit *__restrict__ hits,
                                                   const T *__restrict__ centers, uint32_t n, DoneRec *done_rec,
                                                   DoneCount *done_count, uint32_t done_seq) {
  // (grid-stride: a bounded number of blocks, so that the completion hand-off at the end — one returning atomic per block —
  // is paid a couple of thousand times, not once per 256 rays)
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
  typename Wire<T>::Hit h = hits[i];
  if (h.prim_id == kInvalid) continue;
  const typename Wire<T>::Ray r = rays[i];
  const double kPi = 3.14159265358979323846;
  const T h0 = r.org[0] + h.t * r.dir[0], h1 = r.org[1] + h.t * r.dir[1], h2 = r.org[2] + h.t * r.dir[2];
  T n0 = h0 - centers[3 * (size_t)h.prim_id + 0], n1 = h1 - centers[3 * (size_t)h.prim_id + 1],
    n2 = h2 - centers[3 * (size_t)h.prim_id + 2];
  const T len = Const<T>::sqrt((n0 * n0 + n1 * n1) + n2 * n2); // vnormalize (nanort.h:383-398)
  if (Const<T>::abs(len) > Const<T>::eps()) {
    const T inv_len = T(1.0) / len;
    n0 *= inv_len;
    n1 *= inv_len;
    n2 *= inv_len;
  }
  h.u = T(float(atan2(double(n0), double(n2)) + kPi) * 0.5f * float(1.0 / kPi));
  h.v = T(float(acos(double(n1)) / kPi));
  hits[i] = h;
  }
  done_end_blocks(done_rec, done_count, done_seq);
}

// CylinderIntersector::PostTraversal (examples/cylinder_primitive/main.cc:367-418) as a pass over the finished compact
// records {u_param, v_param, t, prim} + mask {bit 0 hit, bit 1 hit_cap_}: the surface normal, into the caller's
// 28-byte records {u, v, normal[3], t, prim_id} (t is the intersector's t; the example never writes isect->t) and
// 0/1 mask.  `verts` holds the two end points of every cylinder (2 x xyz).  A miss writes {0, 0, 0, max_t, ~0}.
struct CylHit32 {
  float u, v, normal[3], t;
  uint32_t prim_id;
};
static_assert(sizeof(CylHit32) == 28, "nrt_cyl_hit_f32");

__global__ __launch_bounds__(256) void k_cylinder_post(const Wire<float>::Ray *__restrict__ rays,
                                                       const Wire<float>::Hit *__restrict__ compact,
                                                       const uint8_t *__restrict__ bits, const float *__restrict__ verts,
                                                       uint32_t n, CylHit32 *__restrict__ out, uint8_t *__restrict__ mask,
                                                       DoneRec *done_rec, DoneCount *done_count, uint32_t done_seq) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { // (grid-stride: see k_sphere_uv)
  const Wire<float>::Hit h = compact[i];
  const uint8_t b = bits[i];
  CylHit32 o;
  if (b & 1u) {
    const Wire<float>::Ray r = rays[i];
    const float *p0 = verts + 3 * (size_t)(2 * h.prim_id), *p1 = p0 + 3;
    float d01[3], pos[3], nrm[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      d01[k] = p1[k] - p0[k];
      pos[k] = r.org[k] + r.dir[k] * h.t;
    }
    if (b & 2u) { // a cap: +-axis, whichever faces the hit point from the cylinder's middle
      float pc[3];
      cyl_normalize<float>(d01, nrm);
#pragma unroll
      for (int k = 0; k < 3; k++) pc[k] = pos[k] - (d01[k] * 0.5f + p0[k]);
      if (!(cyl_dot<float>(pc, nrm) > 0.0f)) {
        nrm[0] = -nrm[0];
        nrm[1] = -nrm[1];
        nrm[2] = -nrm[2];
      }
    } else { // the side: away from the axis point at parameter v
      float pc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) pc[k] = pos[k] - (p0[k] + h.v * d01[k]);
      cyl_normalize<float>(pc, nrm);
    }
    o.u = h.u;
    o.v = h.v;
    o.normal[0] = nrm[0];
    o.normal[1] = nrm[1];
    o.normal[2] = nrm[2];
    o.t = h.t;
    o.prim_id = h.prim_id;
  } else {
    o.u = o.v = 0.0f;
    o.normal[0] = o.normal[1] = o.normal[2] = 0.0f;
    o.t = h.t; // the kernel's miss record carries max_t
    o.prim_id = kInvalid;
  }
  out[i] = o;
  if (mask) mask[i] = b & 1u;
  }
  done_end_blocks(done_rec, done_count, done_seq);
}

// Both child boxes of one WideNode at once.  For fp32 the two boxes ride in the two halves of
// 64-bit register pairs, so the subtract / multiply chain issues as v_pk_add_f32 / v_pk_mul_f32
// (one VALU slot for two IEEE operations: same operations, same rounding, half the issue slots).
template <typename T>
struct SlabPair {
  bool h0, h1;
  T tm0, tm1;
};

__device__ __forceinline__ SlabPair<float> slab_pair(const Lane<float> &L, const WideNode<float> &w) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 mm = {Const<float>::maxmult(), Const<float>::maxmult()};
  float tmin0 = L.min_t, tmin1 = L.min_t, tmax0 = L.hit_t, tmax1 = L.hit_t;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int sg = L.sign(k);
    const f2 lo = {sg ? w.box0[3 + k] : w.box0[k], sg ? w.box1[3 + k] : w.box1[k]};
    const f2 hi = {sg ? w.box0[k] : w.box0[3 + k], sg ? w.box1[k] : w.box1[3 + k]};
    const f2 o = {L.org(k), L.org(k)};
    const f2 iv = {L.inv(k), L.inv(k)};
    const f2 t0 = (lo - o) * iv;
    const f2 t1 = ((hi - o) * iv) * mm;
    tmin0 = Const<float>::fmax(t0.x, tmin0); // see slab_test
    tmin1 = Const<float>::fmax(t0.y, tmin1);
    tmax0 = Const<float>::fmin(t1.x, tmax0);
    tmax1 = Const<float>::fmin(t1.y, tmax1);
  }
  SlabPair<float> r;
  r.h0 = tmin0 <= tmax0;
  r.h1 = tmin1 <= tmax1;
  r.tm0 = tmin0;
  r.tm1 = tmin1;
  return r;
}

__device__ __forceinline__ SlabPair<double> slab_pair(const Lane<double> &L, const WideNode<double> &w) {
  SlabPair<double> r;
  const double b0[6] = {w.mn[0][0], w.mn[1][0], w.mn[2][0], w.mx[0][0], w.mx[1][0], w.mx[2][0]};
  const double b1[6] = {w.mn[0][1], w.mn[1][1], w.mn[2][1], w.mx[0][1], w.mx[1][1], w.mx[2][1]};
  r.h0 = slab_test_tmin<double>(L, b0, r.tm0);
  r.h1 = slab_test_tmin<double>(L, b1, r.tm1);
  return r;
}
// The same two tests with the near / far rows of each axis fetched by the ray's direction signs (Lane::so0..2: 0 or 48,
// the distance between the mn and mx rows of a WideNode<double>): the same values into slab_test_tmin's arithmetic,
// twelve 64-bit selects per step fewer.  `rec` = byte offset of the record (32 bits: arrays below 4 GiB).
struct WideTail { // bytes 96..111 of a WideNode<double>
  uint32_t c0, c1;
  int32_t axis;
  uint32_t pad;
};
__device__ __forceinline__ SlabPair<double> slab_pair_presel(const Lane<double> &L, const char *base, uint32_t rec) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const double mm = Const<double>::maxmult();
  double tmin0 = L.min_t, tmin1 = L.min_t, tmax0 = L.hit_t, tmax1 = L.hit_t;
  const uint32_t rec48 = rec + 48u;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t so = k == 0 ? L.so0 : (k == 1 ? L.so1 : L.so2);
    const d2 lo = *reinterpret_cast<const d2 *>(base + (size_t)(rec + so) + 16 * k);
    const d2 hi = *reinterpret_cast<const d2 *>(base + (size_t)(rec48 - so) + 16 * k);
    const double t00 = (lo.x - L.org(k)) * L.inv(k), t01 = (lo.y - L.org(k)) * L.inv(k);
    const double t10 = (hi.x - L.org(k)) * L.inv(k) * mm, t11 = (hi.y - L.org(k)) * L.inv(k) * mm;
    tmin0 = Const<double>::fmax(t00, tmin0); // see slab_test
    tmin1 = Const<double>::fmax(t01, tmin1);
    tmax0 = Const<double>::fmin(t10, tmax0);
    tmax1 = Const<double>::fmin(t11, tmax1);
  }
  SlabPair<double> r;
  r.h0 = tmin0 <= tmax0;
  r.h1 = tmin1 <= tmax1;
  r.tm0 = tmin0;
  r.tm1 = tmin1;
  return r;
}

// The four boxes of one Wide4Node.  fp32: slots (0,1) and (2,3) ride in register pairs as in slab_pair.
template <typename T>
struct Slab4 {
  bool h[4];
  T tm[4];
};

__device__ __forceinline__ Slab4<float> slab4(const Lane<float> &L, const Wide4Node<float> &w) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 mm = {Const<float>::maxmult(), Const<float>::maxmult()};
  float tmin[4] = {L.min_t, L.min_t, L.min_t, L.min_t}, tmax[4] = {L.hit_t, L.hit_t, L.hit_t, L.hit_t};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int sg = L.sign(k);
    const f2 o = {L.org(k), L.org(k)};
    const f2 iv = {L.inv(k), L.inv(k)};
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const f2 lo = {sg ? w.bmax[k][2 * p] : w.bmin[k][2 * p], sg ? w.bmax[k][2 * p + 1] : w.bmin[k][2 * p + 1]};
      const f2 hi = {sg ? w.bmin[k][2 * p] : w.bmax[k][2 * p], sg ? w.bmin[k][2 * p + 1] : w.bmax[k][2 * p + 1]};
      const f2 t0 = (lo - o) * iv;
      const f2 t1 = ((hi - o) * iv) * mm;
      tmin[2 * p] = Const<float>::fmax(t0.x, tmin[2 * p]); // see slab_test
      tmin[2 * p + 1] = Const<float>::fmax(t0.y, tmin[2 * p + 1]);
      tmax[2 * p] = Const<float>::fmin(t1.x, tmax[2 * p]);
      tmax[2 * p + 1] = Const<float>::fmin(t1.y, tmax[2 * p + 1]);
    }
  }
  Slab4<float> r;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    r.h[j] = tmin[j] <= tmax[j];
    r.tm[j] = tmin[j];
  }
  return r;
}

// The same four tests with the near / far plane rows fetched straight from where the ray's direction signs say they are
// (a per-ray byte offset inside the record: Lane::so0..2) instead of fetching both rows of every axis and selecting per
// value: the same values reach the same arithmetic, 24 selects per step do not.  `rec` = byte offset of the record in the
// array (32 bits: arrays below 4 GiB).
struct Wide4Tail { // bytes 96..127 of a Wide4Node<float>
  uint32_t c[4];
  int32_t axis0, axis1, axis2;
  uint32_t pad;
};
// (OFF = uint32_t for arrays below 4 GiB — scalar base + 32-bit lane offset —, uint64_t for larger ones: template bit ORDER & 4)
template <typename OFF>
__device__ __forceinline__ Slab4<float> slab4_presel(const Lane<float> &L, const char *base, OFF rec) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f2 mm = {Const<float>::maxmult(), Const<float>::maxmult()};
  float tmin[4] = {L.min_t, L.min_t, L.min_t, L.min_t}, tmax[4] = {L.hit_t, L.hit_t, L.hit_t, L.hit_t};
  const OFF rec48 = rec + (OFF)48u;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t so = k == 0 ? L.so0 : (k == 1 ? L.so1 : L.so2);
    const f4 lo4 = *reinterpret_cast<const f4 *>(base + (size_t)(rec + so) + 16 * k);   // bmin[k][0..3] or bmax[k][0..3]
    const f4 hi4 = *reinterpret_cast<const f4 *>(base + (size_t)(rec48 - so) + 16 * k); // the other row
    const f2 o = {L.org(k), L.org(k)};
    const f2 iv = {L.inv(k), L.inv(k)};
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const f2 lo = {p ? lo4.z : lo4.x, p ? lo4.w : lo4.y};
      const f2 hi = {p ? hi4.z : hi4.x, p ? hi4.w : hi4.y};
      const f2 t0 = (lo - o) * iv;
      const f2 t1 = ((hi - o) * iv) * mm;
      tmin[2 * p] = Const<float>::fmax(t0.x, tmin[2 * p]); // see slab_test
      tmin[2 * p + 1] = Const<float>::fmax(t0.y, tmin[2 * p + 1]);
      tmax[2 * p] = Const<float>::fmin(t1.x, tmax[2 * p]);
      tmax[2 * p + 1] = Const<float>::fmin(t1.y, tmax[2 * p + 1]);
    }
  }
  Slab4<float> r;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    r.h[j] = tmin[j] <= tmax[j];
    r.tm[j] = tmin[j];
  }
  return r;
}

__device__ __forceinline__ Slab4<double> slab4(const Lane<double> &L, const Wide4Node<double> &w) {
  Slab4<double> r;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double box[6] = {w.bmin[0][j], w.bmin[1][j], w.bmin[2][j], w.bmax[0][j], w.bmax[1][j], w.bmax[2][j]};
    r.h[j] = slab_test_tmin<double>(L, box, r.tm[j]);
  }
  return r;
}

// One stack entry: child reference + its t_min (the t_min is kept as raw bits next to the reference so
// that one LDS access moves both).
template <typename T>
struct StackEntry;
template <>
struct StackEntry<float> {
  typedef uint2 type;
  static __device__ __forceinline__ type make(uint32_t ref, float tm) { return make_uint2(ref, __float_as_uint(tm)); }
  static __device__ __forceinline__ uint32_t ref(const type &e) { return e.x; }
  static __device__ __forceinline__ float tmin(const type &e) { return __uint_as_float(e.y); }
};
template <>
struct StackEntry<double> {
  typedef uint4 type; // {ref, pad, t_min lo, t_min hi}
  static __device__ __forceinline__ type make(uint32_t ref, double tm) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(tm);
    return make_uint4(ref, 0u, (uint32_t)b, (uint32_t)(b >> 32));
  }
  static __device__ __forceinline__ uint32_t ref(const type &e) { return e.x; }
  static __device__ __forceinline__ double tmin(const type &e) {
    return __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
  }
};

// The two per-lane moves of the WideNode walk, shared by k_traverse_wide and the two-level kernel k_scene_trace.  They
// expand inside a kernel that has these names in scope: L (Lane<T>), cur, state, sp, tid, gslot, s_stack[STACK][block],
// SE = StackEntry<T>, `a` with .spill / .spill_tmin / .spill_stride.

// One stack pop (a lane in W_POP): the entry is entered iff its t_min still beats the hit distance — the reference's
// slab test at pop time (see the comment above the kernel); an empty stack finishes the ray.
#define NRT_POP_ENTRY()                                                                                \
do {                                                                                                 \
  int s1_ = sp - 1;                                                                                  \
  s1_ = s1_ < 0 ? 0 : s1_;                                                                           \
  const int sl_ = s1_ > STACK - 1 ? STACK - 1 : s1_; /* (max then min: one v_med3_i32) */            \
  typename SE::type e_ = s_stack[sl_][tid];                                                          \
  if (s1_ >= STACK) { /* rare: the entry lives in the global overflow stack */                       \
    const size_t o_ = (size_t)(s1_ - STACK) * a.spill_stride + gslot;                                \
    e_ = SE::make(a.spill[o_], a.spill_tmin[o_]);                                                    \
  }                                                                                                  \
  const bool fin_ = (sp == 0);                        /* empty stack: the ray is done */             \
  const bool enter_ = !fin_ & (SE::tmin(e_) <= L.hit_t);                                             \
  const uint32_t ref_ = SE::ref(e_);                                                                 \
  sp = s1_;                                                                                          \
  cur = enter_ ? (ref_ & ~kLeafBit) : cur;                                                           \
  state = fin_ ? W_IDLE : (enter_ ? ((ref_ & kLeafBit) ? W_LEAF : W_TRAV) : W_POP);                  \
} while (0)

// One step of a lane in W_TRAV over the WideNode record `w_`: both child boxes tested, the far child of two hits
// pushed with its t_min, the near one (or the only one) entered.
#define NRT_STEP_NODE(w_)                                                                              \
do {                                                                                                 \
  const SlabPair<T> sl2_ = slab_pair(L, (w_));                                                       \
  NRT_STEP_NODE_SL(sl2_, (w_));                                                                      \
} while (0)
/* (sl_: the two box tests; w_: anything with c0, c1, axis) */
#define NRT_STEP_NODE_SL(sl_, w_)                                                                      \
do {                                                                                                 \
  const bool near1_ = L.sign((w_).axis) != 0; /* near child = data[dir_sign[axis]] (nanort.h:2538) */ \
  const bool both_ = sl_.h0 & sl_.h1, any_ = sl_.h0 | sl_.h1;                                        \
  if (both_) { /* the far child waits with its t_min */                                              \
    const uint32_t rf_ = near1_ ? (w_).c0 : (w_).c1;                                                 \
    const T tf_ = near1_ ? sl_.tm0 : sl_.tm1;                                                        \
    if (sp < STACK) {                                                                                \
      s_stack[sp][tid] = SE::make(rf_, tf_);                                                         \
    } else {                                                                                         \
      const size_t o_ = (size_t)(sp - STACK) * a.spill_stride + gslot;                               \
      a.spill[o_] = rf_;                                                                             \
      a.spill_tmin[o_] = tf_;                                                                        \
    }                                                                                                \
    sp++;                                                                                            \
  }                                                                                                  \
  /* both hit: the near one; one hit: that one */                                                    \
  const bool go1_ = both_ ? near1_ : sl_.h1;                                                         \
  const uint32_t next_ = go1_ ? (w_).c1 : (w_).c0;                                                   \
  cur = any_ ? (next_ & ~kLeafBit) : cur;                                                            \
  state = any_ ? ((next_ & kLeafBit) ? W_LEAF : W_TRAV) : W_POP;                                     \
} while (0)


// The same step over a Wide4Node record: the four grandchild boxes tested at once.  Why this visits the same leaves in
// the same order as the binary loop (nanort.h:2526-2548) on a tree whose child boxes lie inside their parents':
//  * order — the binary loop enters near child (by the node's axis) before far child, and inside each child its near
//    grandchild (by the child's axis) before its far one; the four slots are ranked by exactly that rule, the first hit
//    in rank order is entered and the others are pushed in reverse rank order, so they pop in rank order;
//  * culling — the binary loop would also test the child's own box; a ray that hits a grandchild's box within
//    [min_t, hit_t] hits the child's box within it too (the grandchild's box lies inside, the slab arithmetic is
//    monotone), and a ray that misses the child's box misses both grandchildren, so skipping that test changes nothing;
//  * hit distance — a grandchild of the FAR child is tested here with the hit distance of this moment instead of the
//    later one the binary loop would use; every pushed entry is re-tested against the current hit distance when it is
//    popped (NRT_POP_ENTRY), which is the binary loop's decision.
// The half of a leaf child has one slot (the other is marked empty and never hit).
#define NRT_STEP_NODE4(w_)                                                                             \
do {                                                                                                 \
  const Slab4<T> sl4_ = slab4(L, (w_));                                                              \
  NRT_STEP_NODE4_SL(sl4_, (w_));                                                                     \
} while (0)
/* (sl_: the four box tests of the record; w_: anything with c[4], axis0, axis1, axis2) */
#define NRT_STEP_NODE4_SL(sl_, w_)                                                                     \
do {                                                                                                 \
  const bool sA_ = L.sign((w_).axis1) != 0, sB_ = L.sign((w_).axis2) != 0, s0_ = L.sign((w_).axis0) != 0; \
  const bool v1_ = (w_).c[1] != kWide4Empty, v3_ = (w_).c[3] != kWide4Empty;                         \
  const bool h0_ = sl_.h[0], h1_ = sl_.h[1] & v1_, h2_ = sl_.h[2], h3_ = sl_.h[3] & v3_;             \
  /* inside each half: near slot first */                                                            \
  const bool an_ = sA_ ? h1_ : h0_, af_ = sA_ ? h0_ : h1_, bn_ = sB_ ? h3_ : h2_, bf_ = sB_ ? h2_ : h3_; \
  const uint32_t ran_ = sA_ ? (w_).c[1] : (w_).c[0], raf_ = sA_ ? (w_).c[0] : (w_).c[1];             \
  const uint32_t rbn_ = sB_ ? (w_).c[3] : (w_).c[2], rbf_ = sB_ ? (w_).c[2] : (w_).c[3];             \
  const T tan_ = sA_ ? sl_.tm[1] : sl_.tm[0], taf_ = sA_ ? sl_.tm[0] : sl_.tm[1];                    \
  const T tbn_ = sB_ ? sl_.tm[3] : sl_.tm[2], tbf_ = sB_ ? sl_.tm[2] : sl_.tm[3];                    \
  /* the near half (by the node's own axis) first: ranks 0..3 */                                     \
  const bool q0_ = s0_ ? bn_ : an_, q1_ = s0_ ? bf_ : af_, q2_ = s0_ ? an_ : bn_, q3_ = s0_ ? af_ : bf_; \
  const uint32_t r0_ = s0_ ? rbn_ : ran_, r1_ = s0_ ? rbf_ : raf_, r2_ = s0_ ? ran_ : rbn_, r3_ = s0_ ? raf_ : rbf_; \
  const T t1_ = s0_ ? tbf_ : taf_, t2_ = s0_ ? tan_ : tbn_, t3_ = s0_ ? taf_ : tbf_;                 \
  const bool p3_ = q3_ & (q0_ | q1_ | q2_), p2_ = q2_ & (q0_ | q1_), p1_ = q1_ & q0_;                \
  if (NRT_W4_FAST_PUSH && __ballot(sp > STACK - 3) == 0ull) {                                         \
    /* every stepping lane has room for three entries in LDS: the candidates are stored unconditionally, in push order, \
       and the stack pointer moves past the ones that count — three stores and three adds instead of three guarded    \
       blocks (compare, branch, room check, store, add) that nearly always run because SOME lane needs them */         \
    s_stack[sp][tid] = SE::make(r3_, t3_);                                                           \
    sp += p3_ ? 1 : 0;                                                                               \
    s_stack[sp][tid] = SE::make(r2_, t2_);                                                           \
    sp += p2_ ? 1 : 0;                                                                               \
    s_stack[sp][tid] = SE::make(r1_, t1_);                                                           \
    sp += p1_ ? 1 : 0;                                                                               \
  } else {                                                                                           \
    NRT_PUSH_IF(p3_, r3_, t3_);                                                                      \
    NRT_PUSH_IF(p2_, r2_, t2_);                                                                      \
    NRT_PUSH_IF(p1_, r1_, t1_);                                                                      \
  }                                                                                                  \
  const bool any_ = q0_ | q1_ | q2_ | q3_;                                                           \
  const uint32_t next_ = q0_ ? r0_ : (q1_ ? r1_ : (q2_ ? r2_ : r3_));                                \
  cur = any_ ? (next_ & ~kLeafBit) : cur;                                                            \
  state = any_ ? ((next_ & kLeafBit) ? W_LEAF : W_TRAV) : W_POP;                                     \
} while (0)

#define NRT_PUSH_IF(cond_, ref_, tm_)                                                                  \
do {                                                                                                 \
  if (cond_) {                                                                                       \
    if (sp < STACK) {                                                                                \
      s_stack[sp][tid] = SE::make((ref_), (tm_));                                                    \
    } else {                                                                                         \
      const size_t o_ = (size_t)(sp - STACK) * a.spill_stride + gslot;                               \
      a.spill[o_] = (ref_);                                                                          \
      a.spill_tmin[o_] = (tm_);                                                                      \
    }                                                                                                \
    sp++;                                                                                            \
  }                                                                                                  \
} while (0)

// The Wide4Node step with the four slots entered in order of their ENTRY DISTANCE (template parameter ORDER = 1; tunable
// `order4`) instead of the binary loop's axis-sign order.  What this keeps and what it gives up:
//  * culling is unchanged — a slot is entered iff its slab test passes now and, when it was pushed, iff its t_min still
//    beats the hit distance at pop time — so the walk tests a subset of the primitives the reference can reach and every
//    primitive whose box chain the final hit distance does not cull: the closest hit's t (and u, v, prim_id of the record
//    that realises it) are the reference's, except that among several primitives at EXACTLY the same t the survivor is
//    the one this order tests last, not the one the reference's order tests last (prim_id may differ at
//    exact-t ties only; tests/helpers.py:assert_hits_match re-verifies every such ray);
//  * the LEAF SEQUENCE is no longer the reference's, so records are not bit-identical to the same-tree oracle at ties.
// Keys: a hit slot sorts by min(t_min, FLT_MAX) (a hit can carry t_min = +inf only while the hit distance is +inf: the
// clamped key then still passes the pop test), a missed or empty slot by +inf — so hits always precede misses.  Five
// compare-exchanges (a 4-input sorting network) order {key, reference}; the first hit is entered, the others are pushed
// farthest first.
#define NRT_CSWAP4(ka_, ra_, kb_, rb_)                                                                 \
do {                                                                                                 \
  const bool sw_ = kb_ < ka_;                                                                        \
  const T klo_ = Const<T>::fmin(ka_, kb_), khi_ = Const<T>::fmax(ka_, kb_);                          \
  const uint32_t rlo_ = sw_ ? rb_ : ra_, rhi_ = sw_ ? ra_ : rb_;                                     \
  ka_ = klo_; kb_ = khi_; ra_ = rlo_; rb_ = rhi_;                                                    \
} while (0)
#define NRT_STEP_NODE4_DIST(sl_, w_)                                                                   \
do {                                                                                                 \
  const T inf_ = Const<T>::inf(), big_ = Const<T>::fltmax();                                         \
  const bool h0_ = sl_.h[0], h1_ = sl_.h[1] & ((w_).c[1] != kWide4Empty);                            \
  const bool h2_ = sl_.h[2], h3_ = sl_.h[3] & ((w_).c[3] != kWide4Empty);                            \
  T k0_ = h0_ ? Const<T>::fmin(sl_.tm[0], big_) : inf_, k1_ = h1_ ? Const<T>::fmin(sl_.tm[1], big_) : inf_; \
  T k2_ = h2_ ? Const<T>::fmin(sl_.tm[2], big_) : inf_, k3_ = h3_ ? Const<T>::fmin(sl_.tm[3], big_) : inf_; \
  uint32_t r0_ = (w_).c[0], r1_ = (w_).c[1], r2_ = (w_).c[2], r3_ = (w_).c[3];                       \
  NRT_CSWAP4(k0_, r0_, k1_, r1_);                                                                    \
  NRT_CSWAP4(k2_, r2_, k3_, r3_);                                                                    \
  NRT_CSWAP4(k0_, r0_, k2_, r2_);                                                                    \
  NRT_CSWAP4(k1_, r1_, k3_, r3_);                                                                    \
  NRT_CSWAP4(k1_, r1_, k2_, r2_);                                                                    \
  const bool any_ = k0_ < inf_, p1_ = k1_ < inf_, p2_ = k2_ < inf_, p3_ = k3_ < inf_;                \
  if (__ballot(sp > STACK - 3) == 0ull) {                                                             \
    s_stack[sp][tid] = SE::make(r3_, k3_);                                                           \
    sp += p3_ ? 1 : 0;                                                                               \
    s_stack[sp][tid] = SE::make(r2_, k2_);                                                           \
    sp += p2_ ? 1 : 0;                                                                               \
    s_stack[sp][tid] = SE::make(r1_, k1_);                                                           \
    sp += p1_ ? 1 : 0;                                                                               \
  } else {                                                                                           \
    NRT_PUSH_IF(p3_, r3_, k3_);                                                                      \
    NRT_PUSH_IF(p2_, r2_, k2_);                                                                      \
    NRT_PUSH_IF(p1_, r1_, k1_);                                                                      \
  }                                                                                                  \
  cur = any_ ? (r0_ & ~kLeafBit) : cur;                                                              \
  state = any_ ? ((r0_ & kLeafBit) ? W_LEAF : W_TRAV) : W_POP;                                       \
} while (0)

// PLAIN: the launch uses trace options that cannot reject a primitive (full prim_ids_range, no skip_prim_id, no
// back-face culling — the reference's defaults): the three id comparisons per triangle test are compiled out.
//
// (Round 2 also carried a SPLIT variant here — exact work splitting in the drain of a launch: a busy lane handed the oldest
// entry of its stack to an idle lane, results folded per ray by (smallest t, latest segment) — and a per-lane drain loop.
// Both were exact and soaked; both were measured to lose (the variant's bulk loop ran 9 % slower than production through
// register pressure and the extra checks, 13 % per step in all: profiles/r02c_split_*.txt, r03g_split_asis.txt) and were
// removed in round 3.  What a launch waits for at its end is the dependent chain of its few longest rays
// (tools/drain_probe.py, tools/tail_first_probe.py); DESIGN.md 3.1 and 10 keep the numbers.)
#ifndef NRT_SCENE_WALK_WAVES
#define NRT_SCENE_WALK_WAVES 4 // waves per SIMD k_scene_walk is compiled for (128 registers; 5 -> 96 registers spills 30 of them and measured slower)
#endif
#ifndef NRT_SCENE_P1_UNROLL
#define NRT_SCENE_P1_UNROLL 2 // ... of the two-level scene kernel
#endif
#ifndef NRT_W2_F64_P1_UNROLL
#define NRT_W2_F64_P1_UNROLL 2 // ... of the fp64 one-level walk
#endif
#ifndef NRT_W4_P1_UNROLL
#define NRT_W4_P1_UNROLL 2 // fp32 two-level walk: pop + step rounds per trip of the inner-node loop
#endif
#ifndef NRT_W4_FAST_PUSH
#define NRT_W4_FAST_PUSH 1 // two-level step: unconditional stores of the three push candidates when every lane has room in LDS
#endif
#ifndef NRT_W4_PRESEL
#define NRT_W4_PRESEL 1 // fp32 two-level walk: the near / far plane rows are fetched by the ray's signs (slab4_presel) instead of selected per value
#endif
#ifndef NRT_W2_F64_PRESEL
#define NRT_W2_F64_PRESEL 1 // fp64 one-level walk: the plane rows of both children are fetched by the ray's signs (slab_pair_presel)
#endif
#ifndef NRT_W4_TRI_UNROLL
#define NRT_W4_TRI_UNROLL 2 // triangle records fetched per trip of the leaf loop in the WIDTH = 4 variants (1 or 2)
#endif
#ifndef NRT_SPHERE_UNROLL
#define NRT_SPHERE_UNROLL 2 // sphere records fetched per trip of the leaf loop (sphere kind)
#endif
#ifndef NRT_W2_TRI_UNROLL
#define NRT_W2_TRI_UNROLL 2 // ... and in the one-level variants (fp32 and fp64)
#endif
#ifndef NRT_W4_WAVES
#define NRT_W4_WAVES 1 // minimum waves per SIMD asked of the WIDTH = 4 variants (1: whatever the register allocation gives)
#endif
// CLOCK: profiling instantiation that stamps when each wave starts, runs dry and finishes (tools/drain_probe.py).
// WIDTH: 2 = one WideNode (two boxes) per step, 4 = one Wide4Node (two levels, four boxes) per step.
// Leaf items (k_traverse_wide with ORDER bit 1, k_scene_walk): the records of all the lanes of a wave that wait at a leaf, tested
// ONE PER LANE in one trip when they are at most 64 together.  `cnt` (0 for a lane that does not wait, at most 4) and `first` (the
// lane's first record) describe the leaves; record k of owner o becomes item base(o) + k (prefix sums of the counts by three
// ballots), item j is tested by lane j with its OWNER's ray constants (org, Sx Sy Sz and the packed axes, fetched by ds_bpermute;
// the record's address and the owner's lane through two small LDS arrays), and every owner then takes its items' results IN RECORD
// ORDER through the reference's own accept rule (`tt > t` / `tt < min_t` reject, equality and NaN accepted, nanort.h:1133-1139).
// The same tests on the same operands (tri_test's operations, one for one), accepted in the same sequence: the lane state after
// the leaf is bit for bit what the owner's own loop leaves (tests/test_gpu_leaf_items.py, tests/test_gpu_scene.py).  Returns false
// — nothing done — when the records do not fit one trip: the caller's owner loop then runs.  Wave-uniform control flow only.
// (SLOT: how an item names its record in LDS — a 32-bit index into `base` (k_traverse_wide: one array per launch; 4 bytes per item
// keep the block's LDS below a sixth of the CU's) or the record's address (k_scene_walk: every owner's mesh has its own array))
template <bool PLAIN, typename SLOT>
__device__ __forceinline__ bool leaf_items_one_trip(Lane<float> &L, uint32_t cnt, const LeafTri<float> *base, SLOT first, unsigned lane, volatile uint32_t *own_,
                                                    volatile SLOT *rec_, uint32_t range0, uint32_t range1, uint32_t skip_prim, bool cull) {
  typedef float T;
  const unsigned long long b0_ = __ballot((cnt & 1u) != 0u), b1_ = __ballot((cnt & 2u) != 0u), b2_ = __ballot((cnt & 4u) != 0u);
  const uint32_t items_ = (uint32_t)__builtin_popcountll(b0_) + 2u * (uint32_t)__builtin_popcountll(b1_) + 4u * (uint32_t)__builtin_popcountll(b2_);
  if (items_ > (uint32_t)kWave || items_ == 0u) return false;
  const unsigned long long below_ = (1ull << lane) - 1ull;
  const uint32_t base_ = (uint32_t)__builtin_popcountll(b0_ & below_) + 2u * (uint32_t)__builtin_popcountll(b1_ & below_) +
                         4u * (uint32_t)__builtin_popcountll(b2_ & below_);
  const uint32_t kmax_ = b2_ ? 4u : ((b1_ & b0_) ? 3u : (b1_ ? 2u : 1u)); // most records any lane holds (wave-uniform)
#pragma unroll
  for (uint32_t k_ = 0; k_ < 4u; k_++)
    if (k_ < cnt) {
      own_[base_ + k_] = lane; // (a word per item: sub-word stores of neighbouring lanes into one word would queue)
      rec_[base_ + k_] = first + k_;
    }
  // (LDS operations of one wave are performed in issue order and the accesses are volatile: the reads below see the writes)
  const bool item_ = lane < items_;
  const uint32_t me_ = item_ ? lane : 0u; // (a lane without an item repeats item 0 — there is one: the callers come here with a lane waiting — and drops the result)
  const SLOT sv_ = rec_[me_];
  const LeafTri<float> *slot_;
  if constexpr (sizeof(SLOT) == 4)
    slot_ = base + sv_;
  else
    slot_ = sv_;
  const int oa_ = (int)(own_[me_] << 2); // the owner lane's byte address for ds_bpermute
  const LeafTri<T> tri = *slot_; // (issued before the constants are fetched: the two latencies overlap)
  const float o0 = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.org0))), o1 = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.org1))),
              o2 = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.org2)));
  const float sx = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.Sx))), sy = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.Sy))),
              sz = __int_as_float(__builtin_amdgcn_ds_bpermute(oa_, __float_as_int(L.Sz)));
  const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute(oa_, (int)L.pk);
  const int ikx = (int)((pk >> 3) & 3u), iky = (int)((pk >> 5) & 3u), ikz = (int)((pk >> 7) & 3u);
  // TriangleIntersector::Intersect (nanort.h:1054-1150) up to the hit distance: tri_test's own operations on the owner's constants
  const uint32_t prim_i = tri.prim_id;
  bool ok = PLAIN ? item_ : (item_ & (prim_i >= range0) & (prim_i < range1) & (prim_i != skip_prim));
  const bool cull_i = PLAIN ? false : cull;
  const T A0 = tri.p0[0] - o0, A1 = tri.p0[1] - o1, A2 = tri.p0[2] - o2;
  const T B0 = tri.p1[0] - o0, B1 = tri.p1[1] - o1, B2 = tri.p1[2] - o2;
  const T C0 = tri.p2[0] - o0, C1 = tri.p2[1] - o1, C2 = tri.p2[2] - o2;
  const T Akz = sel3(A0, A1, A2, ikz), Bkz = sel3(B0, B1, B2, ikz), Ckz = sel3(C0, C1, C2, ikz);
  const T Ax = sel3(A0, A1, A2, ikx) - sx * Akz;
  const T Ay = sel3(A0, A1, A2, iky) - sy * Akz;
  const T Bx = sel3(B0, B1, B2, ikx) - sx * Bkz;
  const T By = sel3(B0, B1, B2, iky) - sy * Bkz;
  const T Cx = sel3(C0, C1, C2, ikx) - sx * Ckz;
  const T Cy = sel3(C0, C1, C2, iky) - sy * Ckz;
  T U = Cx * By - Cy * Bx;
  T V = Ax * Cy - Ay * Cx;
  T W = Bx * Ay - By * Ax;
  if (ok && (U == T(0) || V == T(0) || W == T(0))) { // nanort.h:1094-1107
    const double CxBy = double(Cx) * double(By), CyBx = double(Cy) * double(Bx);
    const double AxCy = double(Ax) * double(Cy), AyCx = double(Ay) * double(Cx);
    const double BxAy = double(Bx) * double(Ay), ByAx = double(By) * double(Ax);
    U = T(CxBy - CyBx);
    V = T(AxCy - AyCx);
    W = T(BxAy - ByAx);
  }
  const bool neg = (U < T(0)) | (V < T(0)) | (W < T(0));
  const bool pos = (U > T(0)) | (V > T(0)) | (W > T(0));
  ok = ok & !(neg & (cull_i | pos));
  const T det = U + V + W;
  ok = ok & !(det == T(0));
  T tt_i = T(0), uu_i = T(0), vv_i = T(0);
  if (ok) {
    const T Az = sz * Akz, Bz = sz * Bkz, Cz = sz * Ckz;
    const T D = U * Az + V * Bz + W * Cz;
    const T rcp = T(1.0) / det;
    tt_i = D * rcp;
    uu_i = V * rcp;
    vv_i = W * rcp;
  }
  // ... and back to the owners, record by record
  const unsigned long long okm_ = __ballot(ok);
  bool got_ = false;
  uint32_t win_ = 0;
  for (uint32_t k2_ = 0; k2_ < kmax_; k2_++) {
    const uint32_t src_ = (base_ + k2_) & 63u;
    const T ttk = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_ << 2), __float_as_int(tt_i)));
    const bool okk = (k2_ < cnt) & (((okm_ >> src_) & 1ull) != 0ull);
    const bool acc = okk & !(ttk > L.hit_t) & !(ttk < L.min_t); // nanort.h:1133-1139: equality and NaN accepted
    L.hit_t = acc ? ttk : L.hit_t;
    win_ = acc ? src_ : win_;
    got_ = got_ | acc;
  }
  if (__ballot(got_) != 0ull) {
    const int wa_ = (int)(win_ << 2);
    const T uw = __int_as_float(__builtin_amdgcn_ds_bpermute(wa_, __float_as_int(uu_i)));
    const T vw = __int_as_float(__builtin_amdgcn_ds_bpermute(wa_, __float_as_int(vv_i)));
    const uint32_t pw = (uint32_t)__builtin_amdgcn_ds_bpermute(wa_, (int)prim_i);
    L.u = got_ ? uw : L.u;
    L.v = got_ ? vw : L.v;
    L.prim = got_ ? pw : L.prim;
  }
  return true;
}

// ORDER (WIDTH = 4 only), two bits: bit 0: 0 = the four slots in the binary loop's order (same leaf sequence as the reference), 1 = by
// entry distance.  Bit 1 (tunable leaf_compact): the LEAF PHASE hands the records of all the lanes waiting at a leaf out over the
// whole wave, one record per lane and trip (see "leaf items" below) — same tests on the same operands, accepted in the same order.
template <typename T, int STACK, bool STATS, int KIND, bool PLAIN = false, bool CLOCK = false, int WIDTH = 2, int ORDER = 0>
__global__ __launch_bounds__(kTraverseBlock, (WIDTH == 4 && sizeof(T) == 4) ? NRT_W4_WAVES : 1) void k_traverse_wide(const TraverseArgs<T> a) {
  static_assert(WIDTH == 2 || WIDTH == 4, "one or two tree levels per step");
  static_assert(ORDER == 0 || (WIDTH == 4 && sizeof(T) == 4), "distance order / leaf items are variants of the fp32 two-level step");
  static_assert((ORDER & 2) == 0 || (KIND == kPrimTriangles && !STATS), "leaf items: triangle records, production instantiations");
  static_assert((ORDER & 4) == 0 || ((ORDER & 1) == 0 && KIND == kPrimTriangles && !STATS && !CLOCK), "records addressed by 64-bit offsets: the default walk of triangle trees");
  constexpr bool LEAFC = (ORDER & 2) != 0;
  constexpr bool BIG = (ORDER & 4) != 0; // a Wide4Node array of 4 GiB or more (trees beyond ~110 M triangles): 64-bit record offsets
  typedef typename std::conditional<BIG, uint64_t, uint32_t>::type RecOff;
  // leaf items (LEAFC): which lane owns the record a lane tests, and which of the owner's records it is (lane | k << 6)
  __shared__ uint32_t s_item_owner[LEAFC ? kTraverseBlock / kWave : 1][LEAFC ? kWave : 1];
  __shared__ uint32_t s_item_rec[LEAFC ? kTraverseBlock / kWave : 1][LEAFC ? kWave : 1]; // ... and which record of a.tris that is
  typedef StackEntry<T> SE;
  __shared__ typename SE::type s_stack[STACK][kTraverseBlock];

  typedef typename Wire<T>::Node Node;
  typedef typename Wire<T>::Ray Ray;
  typedef typename Wire<T>::Hit Hit;

  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned gslot = blockIdx.x * kTraverseBlock + tid;
  const bool cull = a.cull_back_face != 0;

  NRT_BATCH_TABLE_SETUP();
  Lane<T> L;
  uint32_t rid = kInvalid; // ray whose result this lane holds (W_IDLE with rid valid: finished, not yet written)
  uint32_t cur = 0;        // W_TRAV: WideNode index; W_LEAF: leaf reference without the leaf bit
  int state = W_IDLE;
  int sp = 0;
  // (profiling, NRT_DEBUG bit 8192: when did this wave start, run out of rays, finish — 100 MHz realtime ticks)
  const bool clocked = CLOCK && a.wave_clock != nullptr;
  unsigned long long clk_begin = 0ull, clk_dry = 0ull;
  if (clocked) clk_begin = __builtin_amdgcn_s_memrealtime();
  Claim ck;
  claim_init<T>(a, ck);
  // (an atomic store: performed at the memory side, so that the slot's next launch sees it wherever and whenever it runs)
  if (blockIdx.x == 0 && threadIdx.x < kMaxParts)
    __hip_atomic_store(a.next_cursor + kCursorStrideWords * threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  done_begin<T>(a);
  // STATS (profiling instantiation only): wave-level loop occupancy
  unsigned long long st_it1 = 0, st_act1 = 0, st_idle2 = 0, st_it2 = 0, st_act2 = 0, st_refills = 0, st_refilled = 0, st_entries2 = 0;
  uint32_t st_steps = 0, st_tris = 0; // per ray; with debug flag 64 they replace u, v of the hit record
  // ... and where the wave's time goes (shader-clock ticks): refill (claim, result stores, ray loads, lane set-up), inner-node phase, leaf phase
  unsigned long long st_t_refill = 0, st_t_p1 = 0, st_t_p2 = 0, st_act2b = 0, st_stamp = 0;
  // ... and the same occupancy counts restricted to the STEADY part of the launch (rays still to be handed out), so that the
  // drain (every wave finishing what it holds) can be told from the steady state: counters[12..15]
  unsigned long long st_it1_s = 0, st_act1_s = 0, st_it2_s = 0, st_act2_s = 0;

  // PostTraversal (nanort.h:1205-1211) with the strict final predicate (:2552).  Finished lanes keep
  // their result in registers until the lane is refilled (or the wave runs out of rays), so the
  // stores are issued by many lanes at once instead of by the odd lane of every loop iteration.
#define NRT_WRITE_RESULT()                              \
  do {                                                  \
    const bool hit_ = L.hit_t < L.max_t;                \
    Hit h_;                                             \
    h_.u = hit_ ? L.u : T(0);                           \
    h_.v = hit_ ? L.v : T(0);                           \
    if (STATS && (a.debug_flags & 64u)) {               \
      h_.u = T(st_steps);                               \
      h_.v = T(st_tris);                                \
    }                                                   \
    h_.t = hit_ ? L.hit_t : L.max_t;                    \
    h_.prim_id = hit_ ? L.prim : kInvalid;              \
    Hit *hp_ = a.hits;                                  \
    uint8_t *mp_ = a.mask;                              \
    if (multi) {                                        \
      const BatchPtrs bp_ = s_tbl[batch_of<T>(a, rid)]; \
      hp_ = (Hit *)bp_.hits_v;                          \
      mp_ = bp_.mask_v;                                 \
    }                                                   \
    if (hp_) store_hit_nt<T>(hp_ + rid, h_);            \
    if (mp_) mp_[rid] = hit_ ? (KIND == kPrimCylinders ? (uint8_t)(1u | (L.cap << 1)) : (uint8_t)1) : (uint8_t)0; \
  } while (0)

  // One primitive record (slot `slot_` of the leaf-ordered arrays) against a lane's ray; `act_` false -> no effect.
#define NRT_TEST_PRIM(slot_, act_)                                                                     \
  do {                                                                                                 \
    if (KIND == kPrimSpheres) {                                                                        \
      const LeafSphere<T> sp_ = a.spheres[(slot_)];                                                    \
      sphere_test<T>(L, sp_, (act_), a.range0, a.range1);                                              \
    } else if (KIND == kPrimCylinders) {                                                               \
      const LeafCylinder<T> cy_ = a.cylinders[(slot_)];                                                \
      cylinder_test<T>(L, cy_, (act_), a.range0, a.range1, a.cyl_test_cap != 0u);                      \
    } else {                                                                                           \
      const LeafTri<T> tri_ = a.tris[(slot_)];                                                         \
      if (PLAIN)                                                                                       \
        tri_test<T, true>(L, tri_, (act_), 0u, 0u, 0u, false);                                         \
      else                                                                                             \
        tri_test<T>(L, tri_, (act_), a.range0, a.range1, a.skip_prim, cull);                           \
    }                                                                                                  \
  } while (0)

  for (;;) {
    // ---- refill idle lanes ---------------------------------------------------------
    if (STATS) st_stamp = __builtin_amdgcn_s_memtime();
    unsigned long long idle = __ballot(state == W_IDLE);
    if (!ck.exhausted && (unsigned)__builtin_popcountll(idle) >= a.refill_min) {
      // lanes that still hold a result and lanes that never had a ray are both W_IDLE
      unsigned long long fresh = idle;
      while (fresh != 0ull && !ck.exhausted) {
        if (ck.next == ck.end && !claim_chunk<T>(a, ck, lane, __builtin_ctzll(fresh))) break;
        const unsigned want = (unsigned)__builtin_popcountll(fresh);
        const unsigned avail = ck.end - ck.next;
        const unsigned take = want < avail ? want : avail;
        const unsigned rank = (unsigned)__builtin_popcountll(fresh & ((1ull << lane) - 1ull));
        if (state == W_IDLE && rank < take) {
          if (rid != kInvalid) NRT_WRITE_RESULT();
          rid = ck.next + rank;
          const Ray *rp_ = a.rays;
          uint32_t bq_ = 0u; // this ray belongs to an occlusion-query batch
          if (multi) {
            const uint32_t b_ = batch_of<T>(a, rid);
            rp_ = (const Ray *)s_tbl[b_].rays_v;
            bq_ = (a.batch_anyhit >> b_) & 1u;
          }
          const Ray r = (a.debug_flags & 4u) ? rp_[rid] : load_ray_nt<T>(rp_ + rid);
          lane_init<T>(L, r);
          L.pk |= bq_ << 9;
          sp = 0;
          if (STATS) st_steps = st_tris = 0;
          // The reference pops and tests the root first (nanort.h:2526-2533).  For a branch root that test is implied by
          // the first step: a ray that misses the root's box misses both children's boxes (each lies inside it and the
          // slab arithmetic is monotone), so the step on record 0 ends in W_POP with an empty stack — the same miss.
          if (a.root_is_branch && !a.root_test) {
            cur = 0u;
            state = W_TRAV;
          } else if (a.root_is_branch) { // adopted tree whose child boxes may stick out of node 0's box: test it, as the reference does
            const Node root = a.nodes[0];
            cur = 0u;
            state = slab_test<T>(L, root.bmin, root.bmax) ? W_TRAV : W_POP;
          } else { // single-leaf tree: test the root box, then its primitives
            const Node root = a.nodes[0];
            const bool root_hit = slab_test<T>(L, root.bmin, root.bmax);
            cur = a.packed_leaves ? (((root.data[0] - 1u) << kPackedFirstBits) | root.data[1]) : 0u;
            state = root_hit ? W_LEAF : W_POP; // W_POP with sp == 0 finishes the ray
          }
          if (a.debug_flags & 2u) state = W_POP;
        }
        if (STATS) {
          st_refills++;
          st_refilled += take;
        }
        ck.next += take;
        fresh = __ballot(state == W_IDLE);
      }
      idle = __ballot(state == W_IDLE);
    }
    if (clocked && ck.exhausted && clk_dry == 0ull) clk_dry = __builtin_amdgcn_s_memrealtime();
    if (idle == ~0ull) {
      if (ck.exhausted) break;
      continue;
    }

    // ---- phase 1: inner nodes / stack pops ---------------------------------------------
    // (lanes that leave the loop are invisible to its ballots: the wave counts those waiting at a leaf itself)
    if (STATS) {
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();
      st_t_refill += now_ - st_stamp;
      st_stamp = now_;
    }
    unsigned n_wait_leaf = (unsigned)__builtin_popcountll(__ballot(state == W_LEAF));
    while (state == W_TRAV || state == W_POP) {
      if (STATS) {
        st_it1++;
        st_act1 += (unsigned)__builtin_popcountll(__ballot(true));
        if (!ck.exhausted) {
          st_it1_s++;
          st_act1_s += (unsigned)__builtin_popcountll(__ballot(true));
        }
      }
      // a lane that must pop does so first and, if the popped entry survives, steps into it in the same iteration
#pragma unroll
      for (int u_ = 0; u_ < ((WIDTH == 4 && sizeof(T) == 4 && !STATS) ? NRT_W4_P1_UNROLL : ((WIDTH == 2 && sizeof(T) == 8 && !STATS) ? NRT_W2_F64_P1_UNROLL : 1)); u_++) { // (several pop + step rounds per trip: the loop's own bookkeeping — two ballots, the exit test — runs once per trip)
      if (state == W_POP) NRT_POP_ENTRY();
      if (state == W_TRAV) {
        if (STATS) st_steps++;
        if constexpr (WIDTH == 4) {
#if NRT_W4_PRESEL
          if constexpr (sizeof(T) == 4) {
            const char *wb_ = reinterpret_cast<const char *>(a.wide4);
            const RecOff rec_ = (RecOff)cur << 7; // (32 bits: api.hip sends arrays of 4 GiB and more to the ORDER & 4 instantiations)
            const Slab4<float> sl = slab4_presel<RecOff>(L, wb_, rec_);
            const Wide4Tail w = *reinterpret_cast<const Wide4Tail *>(wb_ + (size_t)rec_ + 96);
            if constexpr ((ORDER & 1) == 1)
              NRT_STEP_NODE4_DIST(sl, w);
            else
              NRT_STEP_NODE4_SL(sl, w);
          } else
#endif
          {
            const Wide4Node<T> w = a.wide4[cur];
            NRT_STEP_NODE4(w);
          }
        } else {
#ifdef NRT_PROBE_EXTRA_LOADS // sensitivity probe (tools/variant_ab.sh): N more 16-byte reads of the record about to be fetched
          typedef uint32_t u4v_ __attribute__((ext_vector_type(4)));
          u4v_ d0_; // (issued before the record's own loads, which return after them; kept live past the step)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d0_) : "v"(a.wide + cur) : "memory");
          for (int x_ = 1; x_ < NRT_PROBE_EXTRA_LOADS; x_++) // (same destination: the returns are in order)
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "+v"(d0_) : "v"(a.wide + cur) : "memory");
#endif
#if NRT_W2_F64_PRESEL
          bool stepped_ = false;
          if constexpr (sizeof(T) == 8) {
            if (a.wide_below_4g) { // (wave-uniform)
              const char *wb_ = reinterpret_cast<const char *>(a.wide);
              const uint32_t rec_ = cur * (uint32_t)sizeof(WideNode<double>);
              const SlabPair<double> sl = slab_pair_presel(L, wb_, rec_);
              const WideTail w = *reinterpret_cast<const WideTail *>(wb_ + (size_t)rec_ + 96);
              NRT_STEP_NODE_SL(sl, w);
              stepped_ = true;
            }
          }
          if (!stepped_)
#endif
          {
          const WideNode<T> w = a.wide[cur];
#ifdef NRT_PROBE_EXTRA_VALU // ... N more dependent v_fma_f32 per step
          {
            float f_ = __uint_as_float(cur);
            for (int x_ = 0; x_ < NRT_PROBE_EXTRA_VALU; x_++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f_));
            asm volatile("" :: "v"(f_));
          }
#endif
          NRT_STEP_NODE(w);
          }
#ifdef NRT_PROBE_EXTRA_LOADS
          asm volatile("" :: "v"(d0_));
#endif
        }
      }
      }
      // Leave when only a few lanes still walk — unless nothing else could be done anyway: no lane waits at a leaf
      // and there are no rays left to hand out (the drain of a launch: the last long rays then stay in this
      // tight loop instead of paying the outer loop's bookkeeping on every step).
      n_wait_leaf += (unsigned)__builtin_popcountll(__ballot(state == W_LEAF));
      if ((unsigned)__builtin_popcountll(__ballot(state == W_TRAV || state == W_POP)) < a.trav_min &&
          (n_wait_leaf != 0u || !ck.exhausted))
        break;
    }

    // ---- phase 2: leaves ------------------------------------------------------------------
    // With only a few lanes at a leaf and a refill due, take the refill first: the new rays walk to
    // their first leaves while these lanes wait, and the triangle loop then runs with many more
    // lanes.  (Every skip is followed by a refill that hands out at least one ray or marks the
    // claim exhausted, so this always makes progress.)
    if (STATS) {
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();
      st_t_p1 += now_ - st_stamp;
      st_stamp = now_;
    }
    const unsigned n_leaf = (unsigned)__builtin_popcountll(__ballot(state == W_LEAF));
    const bool refill_due = !ck.exhausted && (unsigned)__builtin_popcountll(__ballot(state == W_IDLE)) >= a.refill_min;
    if (n_leaf != 0u && !(n_leaf < a.leaf_min && refill_due)) {
      uint32_t cnt = 0, first = 0;
      if (state == W_LEAF) {
        if (a.packed_leaves) {
          cnt = (cur >> kPackedFirstBits) + 1u;
          first = cur & kPackedFirstMask;
        } else {
          const Node *nd = a.nodes + cur;
          cnt = nd->data[0];
          first = nd->data[1];
        }
      }
      if (a.debug_flags & 1u) cnt = 0;
      if (STATS) {
        st_entries2++;
        st_idle2 += (unsigned)__builtin_popcountll(__ballot(state == W_IDLE));
      }
      bool items_done_ = false; // (wave-uniform)
      if constexpr (LEAFC) {
        // ---- leaf items: the records of every waiting lane, spread over the wave --------------------------------------------
        // In the steady state fewer than half of the lanes wait at a leaf when this phase runs (tools/loop_stats.py: 30 of 64
        // on C3's primaries, 20 on its bounce rays) and a lane's 1-4 records are tested two per trip by their OWNER, the other
        // lanes idling.  When all the waiting lanes' records together fit ONE trip of the wave (<= 64: the rule on incoherent
        // waves), record k of owner o becomes ITEM base(o) + k, item j is tested by lane j with the owner's ray constants (fetched
        // by ds_bpermute: org, Sx Sy Sz, the packed axes), and the owner then takes its items' results IN RECORD ORDER through
        // the reference's own accept rule (`tt > t` / `tt < min_t` reject, equality and NaN accepted, nanort.h:1133-1139): the
        // same tests on the same operands, accepted in the same sequence, so the lane state after the leaf is bit for bit what
        // the owner's own loop leaves (tests/test_gpu_leaf_items.py).  More records than lanes: the owners' loop below.  Trees
        // whose leaves hold more than four records do not take this variant at all (api.hip).
        items_done_ = leaf_items_one_trip<PLAIN, uint32_t>(L, cnt, a.tris, first, lane, s_item_owner[tid / kWave], s_item_rec[tid / kWave], a.range0, a.range1,
                                                 a.skip_prim, cull);
      }
      if (items_done_) {
        // (the waiting lanes' records were tested as items)
      } else if constexpr (KIND == kPrimTriangles && (WIDTH == 4 ? NRT_W4_TRI_UNROLL : NRT_W2_TRI_UNROLL) > 1) {
        // several records per trip, all fetched before any is tested (same tests in the same order; fewer dependent
        // round trips per leaf — this variant has the registers for it)
        constexpr uint32_t U_ = WIDTH == 4 ? NRT_W4_TRI_UNROLL : NRT_W2_TRI_UNROLL;
        for (uint32_t i = 0; __ballot(i < cnt) != 0ull; i += U_) {
          LeafTri<T> t_[U_];
          if (STATS) {
            st_it2++;
            st_act2 += (unsigned)__builtin_popcountll(__ballot(i < cnt));
            st_act2b += (unsigned)__builtin_popcountll(__ballot(i + 1u < cnt));
            if (!ck.exhausted) {
              st_it2_s++;
              st_act2_s += (unsigned)__builtin_popcountll(__ballot(i < cnt)) + (unsigned)__builtin_popcountll(__ballot(i + 1u < cnt));
            }
            if (i < cnt) st_tris += (i + 1u < cnt) ? 2u : 1u;
          }
#pragma unroll
          for (uint32_t j = 0; j < U_; j++) t_[j] = a.tris[first + (i + j < cnt ? i + j : 0u)];
#pragma unroll
          for (uint32_t j = 0; j < U_; j++) {
            if (PLAIN)
              tri_test<T, true>(L, t_[j], i + j < cnt, 0u, 0u, 0u, false);
            else
              tri_test<T>(L, t_[j], i + j < cnt, a.range0, a.range1, a.skip_prim, cull);
          }
        }
      } else if constexpr (!STATS && KIND == kPrimSpheres && NRT_SPHERE_UNROLL > 1) { // the same for the 20-byte sphere records
        for (uint32_t i = 0; __ballot(i < cnt) != 0ull; i += NRT_SPHERE_UNROLL) {
          LeafSphere<T> s_[NRT_SPHERE_UNROLL];
#pragma unroll
          for (uint32_t j = 0; j < NRT_SPHERE_UNROLL; j++) s_[j] = a.spheres[first + (i + j < cnt ? i + j : 0u)];
#pragma unroll
          for (uint32_t j = 0; j < NRT_SPHERE_UNROLL; j++) sphere_test<T>(L, s_[j], i + j < cnt, a.range0, a.range1);
        }
      } else
      for (uint32_t i = 0; __ballot(i < cnt) != 0ull; i++) {
        if (STATS) {
          st_it2++;
          st_act2 += (unsigned)__builtin_popcountll(__ballot(i < cnt));
          if (i < cnt) st_tris++;
        }
        // no divergent region here: lanes past their count re-test their first record with ok = false
        NRT_TEST_PRIM(first + (i < cnt ? i : 0u), i < cnt);
      }
      // occlusion query: any accepted primitive settles the ray — drop what is left of its stack
      if (a.any_hit | a.batch_anyhit) sp = (state == W_LEAF && L.hit_t < L.max_t && (a.any_hit != 0u || (L.pk & 512u) != 0u)) ? 0 : sp;
      state = (state == W_LEAF) ? W_POP : state;
    }
    if (STATS) st_t_p2 += __builtin_amdgcn_s_memtime() - st_stamp;
  }
  if (rid != kInvalid) NRT_WRITE_RESULT(); // results still held in registers
#undef NRT_WRITE_RESULT
#undef NRT_TEST_PRIM
  done_end<T>(a, lane);
  if (clocked && lane == 0u) { // one record per wave, reduced on the host (atomics on one line would serialise the exits)
    unsigned long long *rec = a.wave_clock + 3ull * (size_t)(gslot / kWave);
    const unsigned long long clk_end = __builtin_amdgcn_s_memrealtime();
    rec[0] = clk_begin;
    rec[1] = clk_dry == 0ull ? clk_end : clk_dry;
    rec[2] = clk_end;
  }
  if (STATS && lane == 0) { // counters[0..7]: it1, act1, idle lanes at phase-2 entry, it2, act2, refills, refilled, phase-2 entries
    atomicAdd(&a.counters[0], st_it1);
    atomicAdd(&a.counters[1], st_act1);
    atomicAdd(&a.counters[2], st_idle2);
    atomicAdd(&a.counters[3], st_it2);
    atomicAdd(&a.counters[4], st_act2);
    atomicAdd(&a.counters[5], st_refills);
    atomicAdd(&a.counters[6], st_refilled);
    atomicAdd(&a.counters[7], st_entries2);
    atomicAdd(&a.counters[8], st_t_refill); // [8..10]: shader-clock ticks of the wave spent refilling / in the inner-node phase / in the leaf phase
    atomicAdd(&a.counters[9], st_t_p1);
    atomicAdd(&a.counters[10], st_t_p2);
    atomicAdd(&a.counters[11], st_act2b);   // leaf loop, two records per trip: lanes with a second record
    atomicAdd(&a.counters[12], st_it1_s);   // [12..15]: inner iterations / their active lanes / leaf trips / records tested while rays were still being handed out
    atomicAdd(&a.counters[13], st_act1_s);
    atomicAdd(&a.counters[14], st_it2_s);
    atomicAdd(&a.counters[15], st_act2_s);
  }
}

// ---------------------------------------------------------------------------
// Two-level (instanced) traversal, nanosg::Scene::Traverse (examples/nanosg/nanosg.h:778-870) for a whole batch in ONE
// launch.  A ray arrives with the list of instances whose world boxes it enters, nearest entry first (at most 64:
// BVHAccel::ListNodeIntersections, nanort.h:2608-2692 — built by scene.hip's listing kernel over the top-level BVH).
// The lane then does what the reference's loop does, candidate after candidate: skip the instance if the nearest world
// distance so far is below its entry distance (nanosg.h:795), else carry the ray into the instance's space (MultV with
// inv_xform / inv_xform33, :806-808, the local interval left at Ray()'s defaults), walk the instance's OWN tree with the
// single-level machinery above (same WideNode step, same watertight test, same order: the local record is what
// nrtTraverseBatchDevice would return for that local ray, bit for bit), measure the world distance of the local hit
// (:823-836) and keep it if strictly nearer (:838).  Lanes of a wave are at different candidates of different instances
// at the same time — every lane carries its instance's array bases — so there is no per-instance launch, no compaction
// and no host round trip; the wave alternates between the inner-node phase and the leaf phase like k_traverse_wide.
// A lane whose ray is finished takes the next one from a per-partition work cursor (a wave refills once `refill_min` lanes are
// free).  Since round 4 this kernel traces small scenes and the rays the single-pass walk (k_scene_walk, below) hands over;
// scenes of thousands of instances go through that walk, which keeps no list at all.
// ---------------------------------------------------------------------------
// A record fetched through a per-lane pointer that came out of memory (an instance's array bases in the scene kernels) is a FLAT
// access as far as the compiler can tell — flat loads also take a slot of the LDS queue and wait on both counters.  These
// pointers are device-memory addresses: say so (address space 1) and the loads become global_load.
template <typename R>
__device__ __forceinline__ R load_global_record(const R *p) {
  R r;
  if constexpr (sizeof(R) % 16 == 0 && alignof(R) >= 16) {
    typedef uint32_t u4g __attribute__((ext_vector_type(4)));
    const u4g __attribute__((address_space(1))) *g = (const u4g __attribute__((address_space(1))) *)p;
    u4g *d = reinterpret_cast<u4g *>(&r);
#pragma unroll
    for (unsigned q = 0; q < sizeof(R) / 16; q++) d[q] = g[q];
  } else {
    static_assert(sizeof(R) % 4 == 0, "whole words");
    const uint32_t __attribute__((address_space(1))) *g = (const uint32_t __attribute__((address_space(1))) *)p;
    uint32_t *d = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (unsigned q = 0; q < sizeof(R) / 4; q++) d[q] = g[q]; // (neighbouring words: the compiler merges them into the widest loads the alignment allows)
  }
  return r;
}
enum : int { S_NEXT = 4, S_FIN = 5, S_DONE = 6 }; // besides W_TRAV / W_LEAF / W_POP: pick the next candidate / a local walk ended / ray finished

__device__ __forceinline__ void scene_mult_v(float dst[3], const float m[4][4], const float v[3]) { // Matrix::MultV, nanosg.h:232-240
  const float t0 = m[0][0] * v[0] + m[1][0] * v[1] + m[2][0] * v[2] + m[3][0];
  const float t1 = m[0][1] * v[0] + m[1][1] * v[1] + m[2][1] * v[2] + m[3][1];
  const float t2 = m[0][2] * v[0] + m[1][2] * v[1] + m[2][2] * v[2] + m[3][2];
  dst[0] = t0;
  dst[1] = t1;
  dst[2] = t2;
}

// SCAN: a scene of a handful of instances (a.scan_nodes of them): the lane lists the instances its ray enters ITSELF when it fetches
// the ray — every world box tested in id order, exactly what the listing kernel of such scenes did in a launch of its own
// (scene.hip k_scene_list) — into its own slots of the list arrays, which it alone reads back: one launch per batch, no second
// pass over the rays.
template <int STACK, bool SCAN>
__global__ __launch_bounds__(kTraverseBlock) void k_scene_trace(const SceneTraceArgs a) {
  typedef float T;
  typedef StackEntry<float> SE;
  __shared__ SE::type s_stack[STACK][kTraverseBlock];

  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned gslot = blockIdx.x * kTraverseBlock + tid;

  // Persistent threads: a lane whose ray is finished takes the next one from a work cursor (claimed for the whole wave by
  // one atomic once `refill_min` lanes are free), so a wave keeps its lanes busy to the end of the batch instead of
  // waiting for the slowest of 64 fixed rays.
  Lane<float> L;
  uint32_t i = 0, ri = 0; // this lane's slot in the launch (lists, counts) and its ray (== i unless the launch traces a subset)
  uint32_t cur = 0;
  int state = S_DONE, sp = 0;
  uint32_t cnt = 0, j = 0, inst = 0; // candidates of this ray; how many of them have been taken; the instance being walked
  float last_t = 0.0f;               // (entry distance, id) of the candidate taken last: the next one is the smallest above it
  uint32_t last_id = 0u;
  float best_t = 3.402823466e+38f; // t_nearest = numeric_limits<T>::max(), nanosg.h:787
  bool has_hit = false;
  float worg[3] = {0.f, 0.f, 0.f}, wdir[3] = {0.f, 0.f, 0.f};
  const WideNode<float> *wide = nullptr;
  const Wide4Node<float> *wide4 = nullptr; // non-null: this instance's tree is walked two levels per step
  const LeafTri<float> *tris = nullptr;
  const nrt_node_f32 *nodes = nullptr;
  uint32_t packed = 1u;
  bool exhausted = false;
  uint32_t part = blockIdx.x % a.num_parts, tried = 0u; // ray partition this wave claims from (its home first), partitions found empty

  for (;;) {
    // ---- refill free lanes -------------------------------------------------------------------------------------------
    {
      const unsigned long long free_lanes = __ballot(state == S_DONE);
      const unsigned want = (unsigned)__builtin_popcountll(free_lanes);
      if (!exhausted && want >= a.refill_min) {
        // one cursor per ray partition (== XCD, as in k_traverse_wide): a wave drains its home range first, so an XCD walks one
        // band of the batch and its L2 keeps one part of the scene; then it steals from the others
        const uint32_t per = a.n / a.num_parts;
        uint32_t mine = a.n;
        while (tried < a.num_parts) {
          const uint32_t lo = part * per, len = (part + 1u == a.num_parts) ? a.n - lo : per;
          uint32_t base = 0;
          if (lane == (unsigned)__builtin_ctzll(free_lanes)) base = atomicAdd(a.cursor + kCursorStrideWords * part, want);
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(free_lanes));
          if (base < len) {
            const uint32_t off = base + (uint32_t)__builtin_popcountll(free_lanes & ((1ull << lane) - 1ull));
            mine = off < len ? lo + off : a.n;
            break;
          }
          part = (part + 1u == a.num_parts) ? 0u : part + 1u;
          tried++;
        }
        exhausted = tried >= a.num_parts;
        if (state == S_DONE && mine < a.n) {
          i = mine;
          ri = a.subset ? a.subset[i] : i;
          const nrt_ray_f32 r = a.rays[ri];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            worg[k] = r.org[k];
            wdir[k] = r.dir[k];
          }
          if (SCAN) {
            cnt = 0u;
            for (uint32_t k = 0; k < a.scan_nodes; k++) { // (wave-uniform addresses: the boxes arrive through the scalar cache)
              float t;
              if (!scene_node_interval(r, a.insts[k].xbmin, a.insts[k].xbmax, t)) continue;
              a.list_t[(size_t)cnt * a.n + i] = t;
              a.list_node[(size_t)cnt * a.n + i] = k;
              cnt++;
            }
          } else {
            cnt = a.count[i];
          }
          j = 0;
          last_t = 0.0f;
          last_id = 0u;
          best_t = 3.402823466e+38f;
          has_hit = false;
          nrt_scene_hit_f32 h; // the miss record; overwritten by every strictly nearer hit
          h.t = r.max_t;
          h.u = 0.0f;
          h.v = 0.0f;
          h.prim_id = 0xFFFFFFFFu;
          h.node_id = 0xFFFFFFFFu;
          a.hits[ri] = h;
          state = S_NEXT;
        }
      }
    }
    // ---- candidates: finish a local walk, pick the next instance ---------------------------------------------------
    // (Both steps are long and run under divergence — matrix products, a division-heavy lane set-up, the selection scan over the
    // ray's list: the wave runs them for SEVERAL lanes at a time.  Lanes between two instances wait until `cand_min` of them have
    // gathered, unless fewer than `cand_busy_max` lanes would be walking meanwhile.  Walking lanes always finish, so the waiting
    // ones are served at the latest when nobody walks.)
    const unsigned n_cand = (unsigned)__builtin_popcountll(__ballot(state == S_FIN || state == S_NEXT));
    const unsigned n_busy = (unsigned)__builtin_popcountll(__ballot(state == W_TRAV || state == W_POP || state == W_LEAF));
    const bool do_cand = n_cand != 0u && (n_cand >= a.cand_min || n_busy < a.cand_busy_max);
    if (do_cand) {
    if (state == S_FIN) {
      if (L.hit_t < L.max_t) { // the local Traverse() hit (strict final predicate, nanort.h:2552)
        const SceneInst &nd = a.insts[inst];
        float lp[3], wp[3];
        lp[0] = L.org0 + L.hit_t * L.d0; // nanosg.h:823-825
        lp[1] = L.org1 + L.hit_t * L.d1;
        lp[2] = L.org2 + L.hit_t * L.d2;
        scene_mult_v(wp, nd.xform, lp);
        const float px = wp[0] - worg[0], py = wp[1] - worg[1], pz = wp[2] - worg[2];
        const float t_world = __builtin_sqrtf(px * px + py * py + pz * pz); // vlength, nanort.h:383-385
        if (t_world < best_t) {                                             // strict, nanosg.h:838
          best_t = t_world;
          has_hit = true;
          nrt_scene_hit_f32 h;
          h.t = t_world;
          h.u = L.u;
          h.v = L.v;
          h.prim_id = L.prim;
          h.node_id = inst;
          a.hits[ri] = h;
        }
      }
      state = S_NEXT;
    }
    if (state == S_NEXT) {
      state = S_DONE;
      // The ray's candidates — the (at most 64 nearest) instances whose world boxes it enters — lie UNSORTED in its list
      // (scene.hip appends them as the top-level walk finds them).  The reference visits them in the order of (entry distance,
      // id) and skips one whose entry distance lies beyond the nearest hit so far (nanosg.h:795) — and then every later one
      // too, their entry distances being no smaller.  So: pick the smallest (t_min, id) above the one taken last; if the hit
      // so far is nearer than that, the ray is done.  Rays take one or two candidates before that happens, so a selection scan
      // per candidate taken costs far less than sorting every list (round 2: an insertion sort in global memory while listing).
      if (j < cnt) {
        float bt = 0.0f;
        uint32_t bid = 0xFFFFFFFFu;
        bool found = false;
        for (uint32_t q = 0; q < cnt; q++) {
          const float t = a.list_t[(size_t)q * a.n + i];
          if (j != 0u && t < last_t) continue;
          if (found && t > bt) continue;
          const uint32_t id = a.list_node[(size_t)q * a.n + i];
          if (j != 0u && t == last_t && id <= last_id) continue; // taken already (ids are unique)
          if (!found || t < bt || id < bid) { // (here t <= bt: nearer, or as near with the lower id)
            bt = t;
            bid = id;
            found = true;
          }
        }
        if (found && !(best_t < bt)) { // (else: early cull, nanosg.h:795 — for this candidate and all that follow)
          last_t = bt;
          last_id = bid;
          j++;
          inst = bid;
          const SceneInst &nd = a.insts[inst];
          nrt_ray_f32 lr;
          scene_mult_v(lr.org, nd.inv_xform, worg);   // nanosg.h:807
          scene_mult_v(lr.dir, nd.inv_xform33, wdir); // nanosg.h:808
          lr.min_t = 0.0f;                            // Ray() defaults (nanort.h:477-487): the world interval is not propagated
          lr.max_t = 3.402823466e+38f;
          lr.type = 0;
          lane_init<float>(L, lr);
          wide = (const WideNode<float> *)nd.wide;
          wide4 = (const Wide4Node<float> *)nd.wide4;
          tris = (const LeafTri<float> *)nd.tris;
          nodes = nd.nodes;
          packed = nd.packed_leaves;
          sp = 0;
          cur = 0u;
          if (nd.root_is_branch && nd.tree_nested) {
            state = W_TRAV; // (a ray that misses node 0's box misses both children's: see k_traverse_wide)
          } else {
            const nrt_node_f32 root = nodes[0];
            const bool root_hit = slab_test<float>(L, root.bmin, root.bmax);
            if (nd.root_is_branch) {
              state = root_hit ? W_TRAV : W_POP;
            } else { // single-leaf tree
              cur = packed ? (((root.data[0] - 1u) << kPackedFirstBits) | root.data[1]) : 0u;
              state = root_hit ? W_LEAF : W_POP;
            }
          }
        }
      }
      if (state == S_DONE && a.mask) a.mask[ri] = has_hit ? 1 : 0; // this ray is finished
    }
    }
    if (__ballot(state != S_DONE) == 0ull) {
      if (exhausted) break;
      continue;
    }

    // ---- phase 1: inner nodes / stack pops ---------------------------------------------------------------------
    unsigned n_wait = (unsigned)__builtin_popcountll(__ballot(state == W_LEAF || state == S_FIN));
    while (state == W_TRAV || state == W_POP) {
#pragma unroll
      for (int u_ = 0; u_ < NRT_SCENE_P1_UNROLL; u_++) { // (pop + step rounds per trip, as in k_traverse_wide)
      if (state == W_POP) {
        NRT_POP_ENTRY();
        state = (state == W_IDLE) ? S_FIN : state; // an empty stack ends this instance's walk
      }
      if (state == W_TRAV) {
        if (wide4 != nullptr) { // (trees whose child boxes lie inside their parents': two levels per step, NRT_STEP_NODE4)
          const Wide4Node<float> w = load_global_record(wide4 + cur);
          NRT_STEP_NODE4(w);
        } else {
          const WideNode<float> w = load_global_record(wide + cur);
          NRT_STEP_NODE(w);
        }
      }
      }
      n_wait += (unsigned)__builtin_popcountll(__ballot(state == W_LEAF || state == S_FIN));
      if ((unsigned)__builtin_popcountll(__ballot(state == W_TRAV || state == W_POP)) < a.trav_min && n_wait != 0u) break;
    }

    // ---- phase 2: leaves -----------------------------------------------------------------------------------------
    if (__ballot(state == W_LEAF) != 0ull) {
      uint32_t lcnt = 0, first = 0;
      if (state == W_LEAF) {
        if (packed) {
          lcnt = (cur >> kPackedFirstBits) + 1u;
          first = cur & kPackedFirstMask;
        } else {
          lcnt = nodes[cur].data[0];
          first = nodes[cur].data[1];
        }
      }
      for (uint32_t k = 0; __ballot(k < lcnt) != 0ull; k += 2u) { // two records per trip, as in k_traverse_wide
        if (k < lcnt) { // (divergent on purpose: the lanes' record arrays differ)
          const bool two = k + 1u < lcnt;
          const LeafTri<float> t0 = load_global_record(tris + first + k);
          const LeafTri<float> t1 = load_global_record(tris + first + (two ? k + 1u : k));
          tri_test<float, true>(L, t0, true, 0u, 0u, 0u, false); // default trace options (nanosg.h:817)
          tri_test<float, true>(L, t1, two, 0u, 0u, 0u, false);
        }
      }
      state = (state == W_LEAF) ? W_POP : state;
    }
  }
}

hipError_t launch_scene_trace(const SceneTraceArgs &args, unsigned grid, hipStream_t s) {
  if (args.n == 0) return hipSuccess;
  if (args.scan_nodes)
    hipLaunchKernelGGL((k_scene_trace<kSceneLdsStack, true>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  else
    hipLaunchKernelGGL((k_scene_trace<kSceneLdsStack, false>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  return hipGetLastError();
}
int scene_trace_blocks_per_cu() {
  int n = 0;
  int n1 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scene_trace<kSceneLdsStack, false>, kTraverseBlock, 0) != hipSuccess || n < 1) n = 4;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_scene_trace<kSceneLdsStack, true>, kTraverseBlock, 0) == hipSuccess && n1 >= 1 && n1 < n) n = n1;
  return n > 8 ? 8 : n;
}

// ---------------------------------------------------------------------------
// The single-pass scene walk: nanosg::Scene::Traverse (examples/nanosg/nanosg.h:778-870) WITHOUT the list.  The reference first
// lists the (at most 64 nearest) instances whose world boxes the ray enters (BVHAccel::ListNodeIntersections,
// nanort.h:2608-2692), sorted by (entry distance, id) =: rank, then walks the list: an instance is skipped once the nearest world
// distance so far lies below its entry distance (:795), else traced with a fresh local ray, and a strictly nearer world
// distance wins (:838).  The listing must follow the ray through the WHOLE scene before the first triangle is tested — on
// 100 000 instances that was three quarters of the time.  This kernel walks the top-level tree (its Wide4Node records, near
// slots first) and, whenever it reaches an entered instance, walks that instance's tree at once, in the same lane, on the same
// stack (top-level entries below, the open instance's entries above `base`).  What makes that exact:
//   * winner     = the traced hit with the smallest (t_world, rank) — the list order with strict '<' picks exactly that;
//   * skipping   an instance or a top-level subtree entered at e is allowed when the current winner w has t_world_w < e and
//                t_min_w < e: w ranks before everything in there, so the reference's cull has fired at or before it (entry
//                distances only grow from a box to a box inside it — for rays whose direction components are ordinary non-zero
//                numbers; other rays skip per instance only, never a subtree);
//   * certificate at the end: at most 64 instances traced, and no OTHER traced hit lies in front of the winner's own box entry
//                (t2 >= t_min_w).  Then the winner is inside the prefix of the list the reference processes, everything in
//                that prefix was traced here, and nothing behind it can win.  A ray without the certificate (a hit that rounds
//                to the near side of its own box while another hit lies in between; more than 64 boxes; direction vectors
//                far shorter than 1, where the reference's cull compares a distance with a parameter and fires early) is
//                appended to `redo` and traced by the listing path (k_scene_list* + k_scene_trace over the subset).
// The argument is spelled out and tested on the CPU, with random visiting orders and random skipping, by the model
// sgo_traverse_unordered_model (tests/test_scene_walk_model.py); tests/test_gpu_scene.py compares this kernel with the listing
// path and the restatement record by record.
// ---------------------------------------------------------------------------
enum : int { T_ENTER = 7, S_END = 8 }; // a top-level leaf was reached: open its instance / the top-level stack ran empty: the ray is finished

// The top-level tree is walked by the SAME step as the instances' trees: while a lane is in the top-level tree its Lane holds the
// WORLD ray (hit_t pinned at ray.max_t: the box test is then ListNodeIntersections' own, nanort.h:2651), its record pointer is
// the top-level tree's, and `cull_t` — max(winner's distance, winner's box entry, ray.min_t), +inf without a hit — takes the
// place of the hit distance where the walk skips (a box entered beyond it ranks behind a nearer hit; the clipped entry
// distance the step computes is the unclipped one whenever it exceeds ray.min_t).  A leaf reference of the top-level tree names
// instances instead of triangles: the lane leaves the loop, opens the instance (its Lane becomes the local ray, `base` = the
// stack height) and rejoins the same loop; when the stack is back at `base` the world ray returns.  Lanes in the top-level
// tree and lanes inside instances therefore execute the same instructions.
#define NRT_POP_ENTRY_SCENE()                                                                          \
do {                                                                                                 \
  const bool fin_ = (sp <= base);                     /* this level's part of the stack is empty */  \
  const int s1_ = fin_ ? sp : sp - 1;                                                                \
  const int sr_ = s1_ > 0 ? s1_ : 0;                                                                 \
  typename SE::type e_ = s_stack[sr_ < STACK ? sr_ : STACK - 1][tid];                                \
  if (!fin_ && s1_ >= STACK) {                                                                       \
    const size_t o_ = (size_t)(s1_ - STACK) * a.spill_stride + gslot;                                \
    e_ = SE::make(a.spill[o_], a.spill_tmin[o_]);                                                    \
  }                                                                                                  \
  const bool enter_ = !fin_ & (SE::tmin(e_) <= (in_top ? cull_t : L.hit_t));                         \
  const uint32_t ref_ = SE::ref(e_);                                                                 \
  sp = s1_;                                                                                          \
  cur = enter_ ? (ref_ & ~kLeafBit) : cur;                                                           \
  state = fin_ ? (in_top ? S_END : S_FIN) : (enter_ ? ((ref_ & kLeafBit) ? W_LEAF : W_TRAV) : W_POP); \
} while (0)

template <int STACK, bool STATS>
__global__ __launch_bounds__(kTraverseBlock, STATS ? 1 : NRT_SCENE_WALK_WAVES) void k_scene_walk(const SceneWalkArgs a) {
  typedef float T;
  typedef StackEntry<float> SE;
  __shared__ SE::type s_stack[STACK][kTraverseBlock];
  __shared__ uint32_t s_item_owner[kTraverseBlock / kWave][kWave]; // leaf items (leaf_items_one_trip)
  __shared__ const LeafTri<float> *s_item_rec[kTraverseBlock / kWave][kWave];
  // (profiling build only) per wave: [0] outer trips, [1..2] level-change blocks run / lanes served, [3..4] inner-phase trips / lane
  // steps, [5] of those steps in the top-level tree, [6..7] leaf-phase trips / lanes with a first record, [8] instances opened,
  // [9] top-level leaves reached, [10..12] shader-clock ticks in level changes / the inner phase / the leaf phase
  unsigned long long st[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned gslot = blockIdx.x * kTraverseBlock + tid;

  Lane<float> L;
  uint32_t i = 0;   // this lane's ray
  uint32_t cur = 0; // record due (of the top-level tree or of the open instance's tree) / leaf reference reached
  int state = S_DONE, sp = 0, base = 0;
  bool in_top = true, tame = true;
  float cur_tmin = 0.f; // box entry distance of the open instance
  uint32_t inst = 0, traced = 0; // the open instance's id; instances opened so far
  float best_t = 3.402823466e+38f, best_tmin = 0.f, t2 = __builtin_huge_valf(); // winner's distance and box entry; runner-up distance
  float cull_t = __builtin_huge_valf();
  uint32_t best_id = 0;
  bool has_hit = false;
  float worg[3] = {0.f, 0.f, 0.f}, wdir[3] = {0.f, 0.f, 0.f}, winv[3] = {0.f, 0.f, 0.f};
  float wmin_t = 0.f, wmax_t = 0.f;
  const Wide4Node<float> *wide4 = nullptr; // the tree being walked: the top-level tree's records or the open instance's
  const LeafTri<float> *tris = nullptr;
  bool exhausted = false;
  uint32_t part = blockIdx.x % a.num_parts, tried = 0u;

  auto world_ray = [&]() { // the lane (re-)enters the top-level tree
    L.org0 = worg[0];
    L.org1 = worg[1];
    L.org2 = worg[2];
    L.inv0 = winv[0];
    L.inv1 = winv[1];
    L.inv2 = winv[2];
    L.min_t = wmin_t;
    L.hit_t = wmax_t;
    L.max_t = wmax_t;
    L.pk = (wdir[0] < 0.0f ? 1u : 0u) | (wdir[1] < 0.0f ? 2u : 0u) | (wdir[2] < 0.0f ? 4u : 0u);
    wide4 = a.top_wide4;
    in_top = true;
    base = 0;
  };

  for (;;) {
    if (STATS) st[0]++;
    // ---- refill free lanes (as k_scene_trace) -------------------------------------------------------------------------
    {
      const unsigned long long free_lanes = __ballot(state == S_DONE);
      const unsigned want = (unsigned)__builtin_popcountll(free_lanes);
      if (!exhausted && want >= a.refill_min) {
        const uint32_t per = a.n / a.num_parts;
        uint32_t mine = a.n;
        while (tried < a.num_parts) {
          const uint32_t lo = part * per, len = (part + 1u == a.num_parts) ? a.n - lo : per;
          uint32_t b0 = 0;
          if (lane == (unsigned)__builtin_ctzll(free_lanes)) b0 = atomicAdd(a.cursor + kCursorStrideWords * part, want);
          b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, __builtin_ctzll(free_lanes));
          if (b0 < len) {
            const uint32_t off = b0 + (uint32_t)__builtin_popcountll(free_lanes & ((1ull << lane) - 1ull));
            mine = off < len ? lo + off : a.n;
            break;
          }
          part = (part + 1u == a.num_parts) ? 0u : part + 1u;
          tried++;
        }
        exhausted = tried >= a.num_parts;
        if (state == S_DONE && mine < a.n) {
          i = mine;
          const nrt_ray_f32 r = a.rays[i];
          // tame: ordinary non-zero direction components and a finite origin — entry distances then only grow from a box to a
          // box inside it, which is what skipping top-level SUBTREES rests on.  Any other ray (axis-parallel, say) walks the whole
          // top-level tree it enters (cull_t stays +inf) and skips per instance only, by the instance's own entry distance.
          tame = true;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            worg[k] = r.org[k];
            wdir[k] = r.dir[k];
            winv[k] = safe_inverse<float>(r.dir[k]); // vsafe_inverse (nanort.h:2509 via :349-357): what the top-level boxes are tested with
            tame = tame && (__builtin_fabsf(r.dir[k]) >= 1.1920928955078125e-07f) && (__builtin_fabsf(winv[k]) < __builtin_huge_valf()) &&
                   (__builtin_fabsf(r.org[k]) < __builtin_huge_valf());
          }
          wmin_t = r.min_t;
          wmax_t = r.max_t;
          {
            best_t = 3.402823466e+38f; // t_nearest = numeric_limits<T>::max(), nanosg.h:787
            best_tmin = 0.f;
            best_id = 0u;
            t2 = __builtin_huge_valf();
            cull_t = __builtin_huge_valf();
            has_hit = false;
            traced = 0u;
            nrt_scene_hit_f32 h; // the miss record; overwritten by every better hit
            h.t = r.max_t;
            h.u = 0.0f;
            h.v = 0.0f;
            h.prim_id = 0xFFFFFFFFu;
            h.node_id = 0xFFFFFFFFu;
            a.hits[i] = h;
            sp = 0;
            world_ray();
            cur = 0u; // record 0 == the root branch (its own box test is implied by its children's)
            state = W_TRAV;
          }
        }
      }
    }
    // ---- level changes: a local walk ended / an instance is opened / the ray is finished (long, divergent: for several lanes at a time) ----
    {
      const unsigned n_cand = (unsigned)__builtin_popcountll(__ballot(state == S_FIN || state == T_ENTER || state == S_END));
      const unsigned n_busy = (unsigned)__builtin_popcountll(__ballot(state == W_TRAV || state == W_POP || state == W_LEAF));
      if (n_cand != 0u && (n_cand >= a.cand_min || n_busy < a.cand_busy_max)) {
        const unsigned long long c0_ = STATS ? clock64() : 0ull;
        if (STATS) {
          st[1]++;
          st[2] += n_cand;
          st[9] += (unsigned)__builtin_popcountll(__ballot(state == T_ENTER));
        }
        if (state == S_FIN) {
          if (L.hit_t < L.max_t) { // the local Traverse() hit (strict final predicate, nanort.h:2552)
            const SceneInst &nd = a.insts[inst]; // (the full record, by id: only an instance that was hit needs its xform)
            float lp[3], wp[3];
            lp[0] = L.org0 + L.hit_t * L.d0; // nanosg.h:823-825
            lp[1] = L.org1 + L.hit_t * L.d1;
            lp[2] = L.org2 + L.hit_t * L.d2;
            scene_mult_v(wp, nd.xform, lp);
            const float px = wp[0] - worg[0], py = wp[1] - worg[1], pz = wp[2] - worg[2];
            const float t_world = __builtin_sqrtf(px * px + py * py + pz * pz); // vlength, nanort.h:383-385
            // strict '<' in rank order (nanosg.h:838) == smallest (t_world, rank) in any order
            const bool wins = t_world < best_t ||
                              (has_hit && t_world == best_t && (cur_tmin < best_tmin || (cur_tmin == best_tmin && inst < best_id)));
            if (wins) {
              if (has_hit && best_t < t2) t2 = best_t; // the old winner becomes the runner-up
              best_t = t_world;
              best_tmin = cur_tmin;
              best_id = inst;
              has_hit = true;
              const float c = best_t > best_tmin ? best_t : best_tmin;
              cull_t = tame ? (c > wmin_t ? c : wmin_t) : __builtin_huge_valf();
              nrt_scene_hit_f32 h;
              h.t = t_world;
              h.u = L.u;
              h.v = L.v;
              h.prim_id = L.prim;
              h.node_id = inst;
              a.hits[i] = h;
            } else if (t_world < t2) {
              t2 = t_world;
            }
          }
          world_ray();
          state = W_POP;
        } else if (state == T_ENTER) {
          // cur: a leaf reference of the top-level tree, (count - 1, first) into its index array.  One instance is opened now;
          // the others of a leaf of several (boxes the builder could not separate) wait on the stack as a leaf of one fewer.
          const uint32_t lcount = (cur >> kPackedFirstBits) + 1u, lfirst = cur & kPackedFirstMask;
          if (lcount > 1u) NRT_PUSH_IF(true, kLeafBit | ((lcount - 2u) << kPackedFirstBits) | (lfirst + 1u), -__builtin_huge_valf());
          const SceneOpen &nd = a.open_top[lfirst]; // (everything the opening needs in one 128-byte line: id, world box, matrices, mesh)
          const uint32_t k = nd.id;
          const float bx[6] = {nd.xbmin[0], nd.xbmin[1], nd.xbmin[2], nd.xbmax[0], nd.xbmax[1], nd.xbmax[2]};
          const bool s0 = wdir[0] < 0.0f, s1 = wdir[1] < 0.0f, s2 = wdir[2] < 0.0f;
          const float n0 = ((s0 ? bx[3] : bx[0]) - worg[0]) * winv[0], n1 = ((s1 ? bx[4] : bx[1]) - worg[1]) * winv[1],
                      n2 = ((s2 ? bx[5] : bx[2]) - worg[2]) * winv[2];
          const float f0 = ((s0 ? bx[0] : bx[3]) - worg[0]) * winv[0], f1 = ((s1 ? bx[1] : bx[4]) - worg[1]) * winv[1],
                      f2 = ((s2 ? bx[2] : bx[5]) - worg[2]) * winv[2];
          // the leaf's own box test of ListNodeIntersections (IntersectRayAABB, nanort.h:2285-2325, hit_t == ray.max_t) ...
          float tmn = wmin_t, tmx = wmax_t;
          tmn = (n0 > tmn) ? n0 : tmn;
          tmn = (n1 > tmn) ? n1 : tmn;
          tmn = (n2 > tmn) ? n2 : tmn;
          const float g0 = f0 * 1.00000024f, g1 = f1 * 1.00000024f, g2 = f2 * 1.00000024f;
          tmx = (g0 < tmx) ? g0 : tmx;
          tmx = (g1 < tmx) ? g1 : tmx;
          tmx = (g2 < tmx) ? g2 : tmx;
          // ... then NodeBBoxIntersector::Intersect (nanosg.h:603-639): the unclipped interval by the PLAIN reciprocal (for a tame
          // ray the same numbers as above), whose near end the list is sorted by
          float p0 = winv[0], p1 = winv[1], p2 = winv[2]; // (a tame ray's safe reciprocal IS the plain one: the same division)
          if (!tame) {
            p0 = 1.0f / wdir[0];
            p1 = 1.0f / wdir[1];
            p2 = 1.0f / wdir[2];
          }
          const float a0 = ((s0 ? bx[3] : bx[0]) - worg[0]) * p0, a1 = ((s1 ? bx[4] : bx[1]) - worg[1]) * p1,
                      a2 = ((s2 ? bx[5] : bx[2]) - worg[2]) * p2;
          const float b0 = ((s0 ? bx[0] : bx[3]) - worg[0]) * p0, b1 = ((s1 ? bx[1] : bx[4]) - worg[1]) * p1,
                      b2 = ((s2 ? bx[2] : bx[5]) - worg[2]) * p2;
          float e = (a1 > a0) ? a1 : a0;
          e = (a2 > e) ? a2 : e;
          float f = (b1 < b0) ? b1 : b0;
          f = (b2 < f) ? b2 : f;
          const bool entered = tmn <= tmx && e <= f;
          const bool behind = has_hit && best_t < e && (best_tmin < e || (best_tmin == e && best_id < k)); // ranks behind a nearer hit
          if (!entered || behind) {
            state = W_POP; // (still in the top-level tree)
          } else {
            inst = k;
            cur_tmin = e;
            traced++;
            nrt_ray_f32 lr;
            // Matrix::MultV (nanosg.h:232-240) with inv_xform / inv_xform33 (:807-808), the twelve entries it reads
            lr.org[0] = nd.inv[0][0] * worg[0] + nd.inv[1][0] * worg[1] + nd.inv[2][0] * worg[2] + nd.inv[3][0];
            lr.org[1] = nd.inv[0][1] * worg[0] + nd.inv[1][1] * worg[1] + nd.inv[2][1] * worg[2] + nd.inv[3][1];
            lr.org[2] = nd.inv[0][2] * worg[0] + nd.inv[1][2] * worg[1] + nd.inv[2][2] * worg[2] + nd.inv[3][2];
            lr.dir[0] = nd.inv33[0][0] * wdir[0] + nd.inv33[1][0] * wdir[1] + nd.inv33[2][0] * wdir[2] + nd.inv33[3][0];
            lr.dir[1] = nd.inv33[0][1] * wdir[0] + nd.inv33[1][1] * wdir[1] + nd.inv33[2][1] * wdir[2] + nd.inv33[3][1];
            lr.dir[2] = nd.inv33[0][2] * wdir[0] + nd.inv33[1][2] * wdir[1] + nd.inv33[2][2] * wdir[2] + nd.inv33[3][2];
            lr.min_t = 0.0f; // Ray() defaults (nanort.h:477-487): the world interval is not propagated
            lr.max_t = 3.402823466e+38f;
            lr.type = 0;
            lane_init<float>(L, lr);
            const SceneMesh &mesh = a.meshes[nd.mesh];
            wide4 = (const Wide4Node<float> *)mesh.wide4;
            tris = (const LeafTri<float> *)mesh.tris;
            in_top = false;
            base = sp;
            cur = 0u;
            state = W_TRAV; // (a branch root whose children's boxes lie inside it: scene.hip checks ...)
            if (mesh.root_leaf) { // (... or a tree of one leaf: node 0's own box test, nanort.h:2526-2531, then its triangles)
              cur = mesh.leaf_ref;
              state = slab_test<float>(L, mesh.bmin, mesh.bmax) ? W_LEAF : S_FIN;
            }
          }
        } else if (state == S_END) {
          const bool certified = traced <= 64u && (!has_hit || t2 >= best_tmin);
          if (certified) {
            if (a.mask) a.mask[i] = has_hit ? 1 : 0;
          } else {
            a.redo[atomicAdd(a.redo_count, 1u)] = i;
          }
          state = S_DONE;
        }
        if (STATS) {
          st[8] += (unsigned)__builtin_popcountll(__ballot(!in_top && base == sp && (state == W_TRAV || state == W_LEAF || state == S_FIN) && cur == 0u));
          st[10] += clock64() - c0_;
        }
      }
    }
    if (__ballot(state != S_DONE) == 0ull) {
      if (exhausted) break;
      continue;
    }

    // ---- phase 1: inner nodes / stack pops, of the top-level tree and of the open instances alike --------------------
    const unsigned long long c1_ = STATS ? clock64() : 0ull;
    unsigned n_wait = (unsigned)__builtin_popcountll(__ballot(state == W_LEAF || state == S_FIN || state == T_ENTER || state == S_END));
    while (state == W_TRAV || state == W_POP) {
      if (STATS) st[3]++;
#pragma unroll
      for (int u_ = 0; u_ < NRT_SCENE_P1_UNROLL; u_++) {
        if (state == W_POP) {
          NRT_POP_ENTRY_SCENE();
          if (state == S_FIN && !(L.hit_t < L.max_t)) { // the instance was missed (most are): back into the top-level tree right here
            world_ray();
            state = W_POP;
          }
        }
        if (STATS) {
          st[4] += (unsigned)__builtin_popcountll(__ballot(state == W_TRAV));
          st[5] += (unsigned)__builtin_popcountll(__ballot(state == W_TRAV && in_top));
        }
        if (state == W_TRAV) {
          const Wide4Node<float> w = load_global_record(wide4 + cur);
          Slab4<float> sl4_ = slab4(L, w);
          const float ct_ = in_top ? cull_t : __builtin_huge_valf(); // (in the top-level tree: nothing entered beyond a nearer hit that ranks before it)
#pragma unroll
          for (int j_ = 0; j_ < 4; j_++) sl4_.h[j_] = sl4_.h[j_] && (sl4_.tm[j_] <= ct_);
          NRT_STEP_NODE4_SL(sl4_, w);
        }
        state = (in_top && state == W_LEAF) ? T_ENTER : state; // a top-level leaf: instances, not triangles
      }
      n_wait += (unsigned)__builtin_popcountll(__ballot(state == W_LEAF || state == S_FIN || state == T_ENTER || state == S_END));
      if ((unsigned)__builtin_popcountll(__ballot(state == W_TRAV || state == W_POP)) < a.trav_min && n_wait != 0u) break;
    }

    // ---- phase 2: leaves ---------------------------------------------------------------------------------------------
    const unsigned long long c2_ = STATS ? clock64() : 0ull;
    if (STATS) st[11] += c2_ - c1_;
    if (__ballot(state == W_LEAF) != 0ull) {
      uint32_t lcnt = 0, first = 0;
      if (state == W_LEAF) { // (packed leaf references: scene.hip checks)
        lcnt = (cur >> kPackedFirstBits) + 1u;
        first = cur & kPackedFirstMask;
      }
      // (few lanes wait at a leaf in this kernel — 10 of 64 on the instanced scenes: their records nearly always fit one trip)
      const bool items_done_ = !STATS && a.leaf_items != 0u &&
                               leaf_items_one_trip<true, const LeafTri<float> *>(L, lcnt, nullptr, tris + first, lane, s_item_owner[tid / kWave], s_item_rec[tid / kWave], 0u, 0u, 0u, false);
      if (!items_done_)
      for (uint32_t k = 0; __ballot(k < lcnt) != 0ull; k += 2u) {
        if (STATS) {
          st[6]++;
          st[7] += (unsigned)__builtin_popcountll(__ballot(k < lcnt));
        }
        if (k < lcnt) {
          const bool two = k + 1u < lcnt;
          const LeafTri<float> t0 = load_global_record(tris + first + k);
          const LeafTri<float> t1 = load_global_record(tris + first + (two ? k + 1u : k));
          tri_test<float, true>(L, t0, true, 0u, 0u, 0u, false); // default trace options (nanosg.h:817)
          tri_test<float, true>(L, t1, two, 0u, 0u, 0u, false);
        }
      }
      state = (state == W_LEAF) ? W_POP : state;
      if (STATS) st[12] += clock64() - c2_;
    }
  }
  if (STATS && lane == 0 && a.counters) {
#pragma unroll
    for (int q = 0; q < 13; q++) atomicAdd(&a.counters[q], st[q]);
  }
}

hipError_t launch_scene_walk(const SceneWalkArgs &args, unsigned grid, hipStream_t s) {
  if (args.n == 0) return hipSuccess;
#ifdef NRT_PROF
  if (args.counters) {
    hipLaunchKernelGGL((k_scene_walk<kSceneWalkLdsStack, true>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
    return hipGetLastError();
  }
#endif
  NRT_RANGE("scene walk launch (k_scene_walk)");
  hipLaunchKernelGGL((k_scene_walk<kSceneWalkLdsStack, false>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  return hipGetLastError();
}
int scene_walk_blocks_per_cu() {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scene_walk<kSceneWalkLdsStack, false>, kTraverseBlock, 0) != hipSuccess || n < 1) n = 3;
  return n > 8 ? 8 : n;
}

// BVHNode[] -> dense WideNode[]: (1) branches per 1024-node tile, (2) exclusive scan of the
// tile counts, (3) dense index of every branch node, (4) the records.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *s_wave, uint32_t &total) {
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (unsigned)off) inc += t;
  }
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  uint32_t pre = 0;
  total = 0;
  for (unsigned j = 0; j < 4; j++) {
    if (j < w) pre += s_wave[j];
    total += s_wave[j];
  }
  __syncthreads();
  return pre + inc - v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_wide_count(const typename Wire<T>::Node *__restrict__ nodes, uint32_t n,
                                                    uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t s_wave[4];
  uint32_t c = 0;
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
  for (uint32_t k = 0; k < 4; k++)
    if (base + k < n && nodes[base + k].flag == 0) c++;
  uint32_t total;
  (void)block_exclusive_scan_256(c, s_wave, total);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_wide_scan_tiles(uint32_t *tile_count, uint32_t num_tiles) {
  __shared__ uint32_t s_wave[4];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < num_tiles; base += 256u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < num_tiles ? tile_count[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_256(v, s_wave, total);
    if (i < num_tiles) tile_count[i] = carry + ex;
    carry += total;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_wide_index(const typename Wire<T>::Node *__restrict__ nodes, uint32_t n,
                                                    const uint32_t *__restrict__ tile_base,
                                                    uint32_t *__restrict__ dense_of, uint32_t scramble_mod) {
  __shared__ uint32_t s_wave[4];
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
  uint32_t flags[4], c = 0;
  for (uint32_t k = 0; k < 4; k++) {
    flags[k] = (base + k < n && nodes[base + k].flag == 0) ? 1u : 0u;
    c += flags[k];
  }
  uint32_t total;
  uint32_t ex = tile_base[blockIdx.x] + block_exclusive_scan_256(c, s_wave, total);
  for (uint32_t k = 0; k < 4; k++) {
    // (layout probe, tunable wide_scramble: record j of the pre-order goes to slot j * P mod #records — a bijection that
    // keeps the root at 0 and tears every subtree apart; how much the walk slows down bounds what any re-ordering can win)
    if (base + k < n) dense_of[base + k] = scramble_mod ? (uint32_t)(((uint64_t)ex * 2654435761ull) % scramble_mod) : ex;
    ex += flags[k];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_make_wide(const typename Wire<T>::Node *__restrict__ nodes, uint32_t n,
                                                   const uint32_t *__restrict__ dense_of, uint32_t packed,
                                                   WideNode<T> *__restrict__ wide, Wide4Node<T> *__restrict__ wide4) {
  typedef typename Wire<T>::Node Node;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const Node nd = nodes[i];
  if (nd.flag != 0) return;
  if (nd.data[0] >= n || nd.data[1] >= n) return; // an unreachable record of a loaded tree: never visited
  const Node a = nodes[nd.data[0]], b = nodes[nd.data[1]];
  WideNode<T> w;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if constexpr (sizeof(T) == 8) { // (component-major: see WideNode<double>)
      w.mn[k][0] = a.bmin[k];
      w.mx[k][0] = a.bmax[k];
      w.mn[k][1] = b.bmin[k];
      w.mx[k][1] = b.bmax[k];
    } else {
      w.box0[k] = a.bmin[k];
      w.box0[3 + k] = a.bmax[k];
      w.box1[k] = b.bmin[k];
      w.box1[3 + k] = b.bmax[k];
    }
  }
  const uint32_t la = packed ? (((a.data[0] - 1u) << kPackedFirstBits) | a.data[1]) : nd.data[0];
  const uint32_t lb = packed ? (((b.data[0] - 1u) << kPackedFirstBits) | b.data[1]) : nd.data[1];
  w.c0 = a.flag != 0 ? (kLeafBit | la) : dense_of[nd.data[0]];
  w.c1 = b.flag != 0 ? (kLeafBit | lb) : dense_of[nd.data[1]];
  w.axis = nd.axis;
  w.pad = 0;
  wide[dense_of[i]] = w;
  if (wide4 == nullptr) return;
  // the record over two levels: each child's children (or the child itself, when it is a leaf)
  Wide4Node<T> q;
  q.axis0 = nd.axis;
  q.pad = 0;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const Node &ch = h == 0 ? a : b;
    const uint32_t ch_ref = h == 0 ? w.c0 : w.c1;
    const bool two = ch.flag == 0 && ch.data[0] < n && ch.data[1] < n;
    int32_t axis = 0;
    if (two) {
      axis = ch.axis;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint32_t gi = ch.data[j];
        const Node g = nodes[gi];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          q.bmin[k][2 * h + j] = g.bmin[k];
          q.bmax[k][2 * h + j] = g.bmax[k];
        }
        const uint32_t lg = packed ? (((g.data[0] - 1u) << kPackedFirstBits) | g.data[1]) : gi;
        q.c[2 * h + j] = g.flag != 0 ? (kLeafBit | lg) : dense_of[gi];
      }
    } else { // a leaf child (or a branch whose children cannot be read: then it is entered through its own record)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        q.bmin[k][2 * h] = ch.bmin[k];
        q.bmax[k][2 * h] = ch.bmax[k];
        q.bmin[k][2 * h + 1] = ch.bmin[k];
        q.bmax[k][2 * h + 1] = ch.bmax[k];
      }
      q.c[2 * h] = ch_ref;
      q.c[2 * h + 1] = kWide4Empty;
    }
    if (h == 0) q.axis1 = axis; else q.axis2 = axis;
  }
  wide4[dense_of[i]] = q;
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

// Leaf-ordered sphere records from (indices, centers, radii).
template <typename T>
__global__ __launch_bounds__(256) void k_gather_leaf_spheres(const uint32_t *__restrict__ indices,
                                                             const T *__restrict__ centers, const T *__restrict__ radii,
                                                             LeafSphere<T> *__restrict__ out, uint32_t n) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = indices[s];
  LeafSphere<T> r;
#pragma unroll
  for (int k = 0; k < 3; k++) r.c[k] = centers[3 * (size_t)prim + k];
  r.r = radii[prim];
  r.prim_id = prim;
  out[s] = r;
}

// Leaf-ordered cylinder records from (indices, end points, radii).
template <typename T>
__global__ __launch_bounds__(256) void k_gather_leaf_cylinders(const uint32_t *__restrict__ indices,
                                                               const T *__restrict__ verts, const T *__restrict__ radii,
                                                               LeafCylinder<T> *__restrict__ out, uint32_t n) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = indices[s];
  LeafCylinder<T> r;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    r.p0[k] = verts[3 * (size_t)(2 * prim) + k];
    r.p1[k] = verts[3 * (size_t)(2 * prim + 1) + k];
  }
  r.r0 = radii[2 * (size_t)prim];
  r.r1 = radii[2 * (size_t)prim + 1];
  r.prim_id = prim;
  out[s] = r;
}

// ---- host-side launchers (called from api.hip) ------------------------------

template <typename T, int STACK>
static hipError_t launch_traverse_s(const TraverseArgs<T> &args, unsigned grid, bool count, hipStream_t s) {
  if (count) {
    hipLaunchKernelGGL((k_traverse<T, true, STACK>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  } else {
    hipLaunchKernelGGL((k_traverse<T, false, STACK>), dim3(grid), dim3(kTraverseBlock), 0, s, args);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_traverse(const TraverseArgs<T> &args, unsigned grid, bool count, int lds_stack, hipStream_t s) {
  switch (lds_stack) {
    case 16: return launch_traverse_s<T, 16>(args, grid, count, s);
    case 24: return launch_traverse_s<T, 24>(args, grid, count, s);
    default: return launch_traverse_s<T, 32>(args, grid, count, s);
  }
}

// Resident blocks per CU of the (non-counting) traversal kernel for a given LDS stack depth.
template <typename T>
int traverse_blocks_per_cu(int lds_stack) {
  int n = 0;
  hipError_t e;
  switch (lds_stack) {
    case 16: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse<T, false, 16>, kTraverseBlock, 0); break;
    case 24: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse<T, false, 24>, kTraverseBlock, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse<T, false, 32>, kTraverseBlock, 0); break;
  }
  if (e != hipSuccess || n < 1) n = 4;
  return n > 8 ? 8 : n;
}

// `name_out` (optional) receives the name of the variant launched, as rocprofv3 prints it without the argument list.
static const char *variant_name(bool f32, int stack, bool stats, int kind, bool plain, bool clock, int width, int order) {
  static std::mutex m;
  static std::map<std::string, std::string> *names = new std::map<std::string, std::string>(); // (never destroyed: the pointers are handed out)
  char buf[160];
  snprintf(buf, sizeof(buf), "nrt::k_traverse_wide<%s, %d, %s, %d, %s, %s, %d, %d>", f32 ? "float" : "double", stack,
           stats ? "true" : "false", kind, plain ? "true" : "false", clock ? "true" : "false", width, order);
  std::lock_guard<std::mutex> lock(m);
  return names->emplace(buf, buf).first->second.c_str();
}
#define NRT_LAUNCH_WIDE_O(STACK_, STATS_, KIND_, PLAIN_, CLOCK_, WIDTH_, ORDER_)                                          \
  do {                                                                                                                  \
    const char *vn_ = variant_name(sizeof(T) == 4, STACK_, STATS_, KIND_, PLAIN_, CLOCK_, WIDTH_, ORDER_);              \
    NRT_RANGE_PUSH(vn_); /* (profiling library: a marker range per traversal launch, named like the kernel) */          \
    hipLaunchKernelGGL((k_traverse_wide<T, STACK_, STATS_, KIND_, PLAIN_, CLOCK_, WIDTH_, ORDER_>), dim3(grid),         \
                       dim3(kTraverseBlock), 0, s, args);                                                               \
    NRT_RANGE_POP();                                                                                                    \
    if (name_out) *name_out = vn_;                                                                                      \
  } while (0)
#define NRT_LAUNCH_WIDE(STACK_, STATS_, KIND_, PLAIN_, CLOCK_, WIDTH_) NRT_LAUNCH_WIDE_O(STACK_, STATS_, KIND_, PLAIN_, CLOCK_, WIDTH_, 0)

template <typename T>
hipError_t launch_traverse_wide(const TraverseArgs<T> &args, unsigned grid, int lds_stack, int prim_kind, hipStream_t s,
                                const char **name_out) {
  if (prim_kind == kPrimSpheres) { // 10 LDS entries walking one level per step, kWide4LdsStack walking two (the caller sizes the overflow stack)
    if constexpr (sizeof(T) == 4) {
      if (args.wide4)
        NRT_LAUNCH_WIDE(kWide4LdsStack, false, kPrimSpheres, false, false, 4);
      else
        NRT_LAUNCH_WIDE(10, false, kPrimSpheres, false, false, 2);
    } else {
      NRT_LAUNCH_WIDE(10, false, kPrimSpheres, false, false, 2);
    }
    if (args.hits) // (args.done_publish == 0: this pass closes the launch's completion record)
      hipLaunchKernelGGL((k_sphere_uv<T>), dim3(std::min((args.num_rays + 255u) / 256u, 2048u)), dim3(256), 0, s, args.rays, args.hits,
                         args.centers, args.num_rays, args.done_publish ? nullptr : args.done_rec, args.done_count, args.done_seq);
    return hipGetLastError();
  }
  if (prim_kind == kPrimCylinders) {
    if constexpr (sizeof(T) == 4) {
      if (args.wide4)
        NRT_LAUNCH_WIDE(kWide4LdsStack, false, kPrimCylinders, false, false, 4);
      else
        NRT_LAUNCH_WIDE(10, false, kPrimCylinders, false, false, 2);
    } else {
      NRT_LAUNCH_WIDE(10, false, kPrimCylinders, false, false, 2);
    }
    return hipGetLastError();
  }
  if (args.wide4) { // two tree levels per step (the caller checked what that needs)
    if constexpr (sizeof(T) == 4) {
#ifdef NRT_PROF // (the profiling instantiations exist in libnanort_hip_prof.so only: nanort_hip_prof.h)
      if (args.debug_flags & 32u)
        NRT_LAUNCH_WIDE(kWide4LdsStack, true, kPrimTriangles, true, false, 4); // profiling instantiation (default trace options only)
      else if (args.wave_clock)
        NRT_LAUNCH_WIDE(kWide4LdsStack, false, kPrimTriangles, true, true, 4); // per-wave time stamps (default trace options only)
      else
#endif
      if (args.wide4_big) { // a record array of 4 GiB or more: the default walk with 64-bit record offsets (api.hip sends nothing else here)
        if (args.leaf_items && args.plain_options)
          NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, true, false, 4, 6);
        else if (args.leaf_items)
          NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, false, false, 4, 6);
        else if (args.plain_options)
          NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, true, false, 4, 4);
        else
          NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, false, false, 4, 4);
      } else if (args.leaf_items && !args.order4 && args.plain_options) // leaf phase over items (tunable leaf_compact): the reference's walk, records bit-identical
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, true, false, 4, 2);
      else if (args.leaf_items && !args.order4)
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, false, false, 4, 2);
      else if (args.leaf_items && args.plain_options)
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, true, false, 4, 3);
      else if (args.leaf_items)
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, false, false, 4, 3);
      else if (args.order4 && args.plain_options) // slots entered by entry distance (tunable order4; contract-level parity: see NRT_STEP_NODE4_DIST)
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, true, false, 4, 1);
      else if (args.order4)
        NRT_LAUNCH_WIDE_O(kWide4LdsStack, false, kPrimTriangles, false, false, 4, 1);
      else if (args.plain_options)
        NRT_LAUNCH_WIDE(kWide4LdsStack, false, kPrimTriangles, true, false, 4);
      else
        NRT_LAUNCH_WIDE(kWide4LdsStack, false, kPrimTriangles, false, false, 4);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  switch (lds_stack) {
    case 8: NRT_LAUNCH_WIDE(8, false, kPrimTriangles, false, false, 2); break;
    case 10:
#ifdef NRT_PROF
      if (args.debug_flags & 32u) {
        NRT_LAUNCH_WIDE(10, true, kPrimTriangles, false, false, 2);
      } else if (args.wave_clock) { // profiling: per-wave time stamps (default trace options only)
        NRT_LAUNCH_WIDE(10, false, kPrimTriangles, true, true, 2);
      } else
#endif
      if (args.plain_options) {
        NRT_LAUNCH_WIDE(10, false, kPrimTriangles, true, false, 2);
      } else {
        NRT_LAUNCH_WIDE(10, false, kPrimTriangles, false, false, 2);
      }
      break;
    case 12: NRT_LAUNCH_WIDE(12, false, kPrimTriangles, false, false, 2); break;
    default: NRT_LAUNCH_WIDE(16, false, kPrimTriangles, false, false, 2); break;
  }
  return hipGetLastError();
}
#undef NRT_LAUNCH_WIDE
#undef NRT_LAUNCH_WIDE_O

template <typename T>
int traverse_wide_blocks_per_cu(int lds_stack, int prim_kind, bool wide4) {
  int n = 0;
  hipError_t e = hipErrorInvalidValue;
  if (prim_kind == kPrimSpheres) {
    if constexpr (sizeof(T) == 4) {
      if (wide4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, kWide4LdsStack, false, kPrimSpheres, false, false, 4>, kTraverseBlock, 0);
    }
    if (!wide4 || sizeof(T) != 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 10, false, kPrimSpheres>, kTraverseBlock, 0);
  } else if (prim_kind == kPrimCylinders) {
    if constexpr (sizeof(T) == 4) {
      if (wide4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, kWide4LdsStack, false, kPrimCylinders, false, false, 4>, kTraverseBlock, 0);
    }
    if (!wide4 || sizeof(T) != 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 10, false, kPrimCylinders>, kTraverseBlock, 0);
  } else if (wide4) {
    if constexpr (sizeof(T) == 4)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, kWide4LdsStack, false, kPrimTriangles, true, false, 4>, kTraverseBlock, 0);
  } else {
    switch (lds_stack) {
      case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 8, false, kPrimTriangles>, kTraverseBlock, 0); break;
      case 10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 10, false, kPrimTriangles, true>, kTraverseBlock, 0); break;
      case 12: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 12, false, kPrimTriangles>, kTraverseBlock, 0); break;
      default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_wide<T, 16, false, kPrimTriangles>, kTraverseBlock, 0); break;
    }
  }
  if (e != hipSuccess || n < 1) n = 4;
  return n > 8 ? 8 : n;
}

// scratch: tile counts (ceil(n/1024) u32) followed by dense_of (n u32)
template <typename T>
hipError_t launch_make_wide(const typename Wire<T>::Node *nodes, uint32_t n, uint32_t packed, uint32_t *scratch,
                            WideNode<T> *wide, Wide4Node<T> *wide4, uint32_t scramble_mod, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const uint32_t tiles = (n + 1023u) / 1024u;
  uint32_t *tile_count = scratch, *dense_of = scratch + tiles;
  hipLaunchKernelGGL((k_wide_count<T>), dim3(tiles), dim3(256), 0, s, nodes, n, tile_count);
  hipLaunchKernelGGL(k_wide_scan_tiles, dim3(1), dim3(256), 0, s, tile_count, tiles);
  hipLaunchKernelGGL((k_wide_index<T>), dim3(tiles), dim3(256), 0, s, nodes, n, tile_count, dense_of, scramble_mod);
  hipLaunchKernelGGL((k_make_wide<T>), dim3((n + 255u) / 256u), dim3(256), 0, s, nodes, n, dense_of, packed, wide, wide4);
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

template <typename T>
hipError_t launch_gather_leaf_spheres(const uint32_t *indices, const T *centers, const T *radii, LeafSphere<T> *out,
                                      uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_gather_leaf_spheres<T>), dim3((n + 255u) / 256u), dim3(256), 0, s, indices, centers, radii, out, n);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_gather_leaf_cylinders(const uint32_t *indices, const T *verts, const T *radii, LeafCylinder<T> *out,
                                        uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_gather_leaf_cylinders<T>), dim3((n + 255u) / 256u), dim3(256), 0, s, indices, verts, radii, out, n);
  return hipGetLastError();
}
template hipError_t launch_gather_leaf_cylinders<float>(const uint32_t *, const float *, const float *, LeafCylinder<float> *,
                                                        uint32_t, hipStream_t);
template hipError_t launch_gather_leaf_cylinders<double>(const uint32_t *, const double *, const double *,
                                                         LeafCylinder<double> *, uint32_t, hipStream_t);

hipError_t launch_cylinder_post(const nrt_ray_f32 *rays, const nrt_hit_f32 *compact, const uint8_t *bits, const float *verts,
                                uint32_t n, void *out, uint8_t *mask, DoneRec *done_rec, DoneCount *done_count, uint32_t done_seq,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_cylinder_post, dim3(std::min((n + 255u) / 256u, 2048u)), dim3(256), 0, s, rays, compact, bits, verts, n,
                     (CylHit32 *)out, mask, done_rec, done_count, done_seq);
  return hipGetLastError();
}

template hipError_t launch_traverse<float>(const TraverseArgs<float> &, unsigned, bool, int, hipStream_t);
template hipError_t launch_traverse<double>(const TraverseArgs<double> &, unsigned, bool, int, hipStream_t);
template hipError_t launch_traverse_wide<float>(const TraverseArgs<float> &, unsigned, int, int, hipStream_t, const char **);
template hipError_t launch_traverse_wide<double>(const TraverseArgs<double> &, unsigned, int, int, hipStream_t, const char **);
template int traverse_wide_blocks_per_cu<float>(int, int, bool);
template int traverse_wide_blocks_per_cu<double>(int, int, bool);
template hipError_t launch_gather_leaf_spheres<float>(const uint32_t *, const float *, const float *, LeafSphere<float> *,
                                                      uint32_t, hipStream_t);
template hipError_t launch_gather_leaf_spheres<double>(const uint32_t *, const double *, const double *,
                                                       LeafSphere<double> *, uint32_t, hipStream_t);
template hipError_t launch_make_wide<float>(const nrt_node_f32 *, uint32_t, uint32_t, uint32_t *, WideNode<float> *,
                                            Wide4Node<float> *, uint32_t, hipStream_t);
template hipError_t launch_make_wide<double>(const nrt_node_f64 *, uint32_t, uint32_t, uint32_t *, WideNode<double> *,
                                             Wide4Node<double> *, uint32_t, hipStream_t);
template int traverse_blocks_per_cu<float>(int);
template int traverse_blocks_per_cu<double>(int);
template hipError_t launch_gather_leaf_tris<float>(const uint32_t *, const uint32_t *, const float *,
                                                   LeafTri<float> *, uint32_t, hipStream_t);
template hipError_t launch_gather_leaf_tris<double>(const uint32_t *, const uint32_t *,
                                                    const double *, LeafTri<double> *, uint32_t,
                                                    hipStream_t);

} // namespace nrt

// tests/cpp/host_check.cc — drives include/nanort.h the way the reference's examples do.
//
//   host_check regress30 [x]          the scenario of the reference's
//                                     test/regression/possible-accuracy-problem-30 (fp64, one triangle)
//   host_check spheres N W H OUT      the reference's particle example (examples/particle_primitive/main.cc) with the
//                                     header's built-in sphere primitive: Build + per-ray Traverse (+ TraverseBatch)
//   host_check cylinders SCENE W H OUT  the same for the cylinder example (examples/cylinder_primitive/main.cc)
//   host_check trace MESH RAYS OUT    Build + per-ray Traverse (and, with the HIP backend compiled in,
//                                     TraverseBatch) over a raw mesh / ray file; hit records to OUT
//
// Built twice by tests/test_host_header.py: plain (generic host path) and with
// -DNANORT_USE_HIP_BACKEND -lnanort_hip (GPU Build + TraverseBatch).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include <string>
#include <vector>

#include "nanort.h"
#ifdef NANORT_USE_HIP_BACKEND
#include <hip/hip_runtime_api.h>  // the device-resident variant is checked with plain runtime calls
#endif

static int regress30(bool tiny) {
  typedef double real;
  real vertices[9] = {1.0, 2.0, -3.0, -1.0, 2.0, -3.0, 1.0, 2.0, 3.0};
  unsigned int triangles[3] = {0, 1, 2};
  real d[3] = {tiny ? -5.30287619e-17 : 0.0, -8.66025404e-01, -0.5};
  nanort::BVHBuildOptions<real> build_options;
  build_options.cache_bbox = true;
  nanort::TriangleMesh<real> mesh(vertices, triangles, sizeof(real) * 3);
  nanort::TriangleSAHPred<real> pred(vertices, triangles, sizeof(real) * 3);
  nanort::BVHAccel<real> accel;
  if (!accel.Build(1, mesh, pred, build_options)) return 2;
  nanort::Ray<real> ray;
  ray.org[0] = -0.36;
  ray.org[1] = 7.93890843;
  ray.org[2] = 1.2160368;
  const real len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  for (int k = 0; k < 3; k++) ray.dir[k] = d[k] / len;
  ray.min_t = 0.0;
  ray.max_t = 1.0e+30;
  nanort::TriangleIntersector<real, nanort::TriangleIntersection<real> > isector(vertices, triangles, sizeof(real) * 3);
  nanort::TriangleIntersection<real> isect;
  const bool hit = accel.Traverse(ray, isector, &isect);
  if (!hit) {
    printf("No intersection detected\n");
    return 1;
  }
  printf("isect.u =%g v = %g t = %.17g\n", isect.u, isect.v, isect.t);
  return 0;
}

template <typename T>
static int trace(const char *mesh_path, const char *rays_path, const char *out_path) {
  // mesh: u32 nv, u32 nf, T xyz[nv], u32 ijk[nf]; rays: u64 n, Ray<T>[n]
  FILE *fp = fopen(mesh_path, "rb");
  if (!fp) return 2;
  uint32_t nv = 0, nf = 0;
  if (fread(&nv, 4, 1, fp) != 1 || fread(&nf, 4, 1, fp) != 1) return 2;
  std::vector<T> verts(3 * (size_t)nv);
  std::vector<unsigned int> faces(3 * (size_t)nf);
  if (fread(verts.data(), sizeof(T), verts.size(), fp) != verts.size()) return 2;
  if (fread(faces.data(), 4, faces.size(), fp) != faces.size()) return 2;
  fclose(fp);
  fp = fopen(rays_path, "rb");
  if (!fp) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, fp) != 1) return 2;
  std::vector<nanort::Ray<T> > rays(n);
  if (fread(rays.data(), sizeof(nanort::Ray<T>), n, fp) != n) return 2;
  fclose(fp);

  nanort::TriangleMesh<T> mesh(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::TriangleSAHPred<T> pred(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::BVHAccel<T> accel;
  if (!accel.Build(nf, mesh, pred)) {
    fprintf(stderr, "Build failed\n");
    return 3;
  }
  nanort::BVHBuildStatistics st = accel.GetStatistics();
  printf("nodes %zu leaf %u branch %u depth %u build_secs %g\n", accel.GetNodes().size(), st.num_leaf_nodes, st.num_branch_nodes,
         st.max_tree_depth, (double)st.build_secs);

  // per-ray path, exactly like the reference's render loops
  std::vector<nanort::TriangleIntersection<T> > hits(n);
  std::vector<unsigned char> mask(n, 0);
  for (uint64_t i = 0; i < n; i++) {
    nanort::TriangleIntersector<T> isector(verts.data(), faces.data(), sizeof(T) * 3);
    memset(&hits[i], 0, sizeof(hits[i]));
    hits[i].t = rays[i].max_t;
    hits[i].prim_id = 0xFFFFFFFFu;
    mask[i] = accel.Traverse(rays[i], isector, &hits[i]) ? 1 : 0;
  }
#ifdef NANORT_USE_HIP_BACKEND
  // batched GPU path over the same (GPU-built) node array: must agree bit for bit (with NANORT_HIP_DEVICES set the batch is
  // split over one replica of the tree per listed device: BVHAccel::TraverseBatch -> nrtTraverseBatchMulti)
  printf("hip_devices %zu\n", accel.NumHipDevices());
  accel.SetTraverseBatchRowLength(320);
  std::vector<nanort::TriangleIntersection<T> > bhits(n);
  std::vector<unsigned char> bmask(n, 0);
  for (uint64_t i = 0; i < n; i++) {
    memset(&bhits[i], 0, sizeof(bhits[i]));
    bhits[i].t = rays[i].max_t;
    bhits[i].prim_id = 0xFFFFFFFFu;
  }
  if (!accel.TraverseBatch(rays.data(), n, bhits.data(), bmask.data())) {
    fprintf(stderr, "TraverseBatch failed: %s\n", accel.LastBackendError().c_str());
    return 4;
  }
  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (bmask[i] != mask[i] || bhits[i].t != hits[i].t || bhits[i].u != hits[i].u || bhits[i].v != hits[i].v ||
        bhits[i].prim_id != hits[i].prim_id)
      bad++;
  }
  printf("batch_vs_per_ray_mismatches %llu\n", (unsigned long long)bad);
  if (bad) return 5;
  {  // device-resident variant: rays and records stay in HBM, asynchronous on a stream of the caller's
    void *d_rays = NULL, *d_hits = NULL, *d_mask = NULL;
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess || hipMalloc(&d_rays, n * sizeof(rays[0])) != hipSuccess ||
        hipMalloc(&d_hits, n * sizeof(hits[0])) != hipSuccess || hipMalloc(&d_mask, n) != hipSuccess)
      return 6;
    hipMemcpyAsync(d_rays, rays.data(), n * sizeof(rays[0]), hipMemcpyHostToDevice, stream);
    if (!accel.TraverseBatchDevice(static_cast<const nanort::Ray<T> *>(d_rays), n, static_cast<nanort::TriangleIntersection<T> *>(d_hits),
                                   static_cast<unsigned char *>(d_mask), stream)) {
      fprintf(stderr, "TraverseBatchDevice failed: %s\n", accel.LastBackendError().c_str());
      return 6;
    }
    std::vector<nanort::TriangleIntersection<T> > dh(n);
    std::vector<unsigned char> dm(n);
    hipMemcpyAsync(dh.data(), d_hits, n * sizeof(hits[0]), hipMemcpyDeviceToHost, stream);
    hipMemcpyAsync(dm.data(), d_mask, n, hipMemcpyDeviceToHost, stream);
    hipStreamSynchronize(stream);
    uint64_t dbad = 0;
    for (uint64_t i = 0; i < n; i++) {
      if (dm[i] != mask[i]) dbad++;
      if (mask[i] ? (dh[i].t != hits[i].t || dh[i].u != hits[i].u || dh[i].v != hits[i].v || dh[i].prim_id != hits[i].prim_id)
                  : (dh[i].t != rays[i].max_t || dh[i].prim_id != 0xFFFFFFFFu))
        dbad++;
    }
    printf("device_variant_mismatches %llu\n", (unsigned long long)dbad);
    hipFree(d_rays);
    hipFree(d_hits);
    hipFree(d_mask);
    hipStreamDestroy(stream);
    if (dbad) return 6;
  }
#endif
  fp = fopen(out_path, "wb");
  if (!fp) return 2;
  fwrite(hits.data(), sizeof(hits[0]), n, fp);
  fwrite(mask.data(), 1, n, fp);
  // the tree, so the caller can validate it
  uint64_t nn = accel.GetNodes().size();
  fwrite(&nn, 8, 1, fp);
  fwrite(accel.GetNodes().data(), sizeof(nanort::BVHNode<T>), nn, fp);
  fwrite(accel.GetIndices().data(), 4, nf, fp);
  fclose(fp);
  return 0;
}

// The particle example's main() (scene, build, camera loop: examples/particle_primitive/main.cc:329-400) over the
// header's built-in sphere classes; the spheres come from a raw file {u32 n; float xyz[n]; float r[n]}.
static int spheres(const char *scene_path, int W, int H, const char *out_path) {
  FILE *fp = fopen(scene_path, "rb");
  if (!fp) return 2;
  uint32_t n = 0;
  if (fread(&n, 4, 1, fp) != 1) return 2;
  std::vector<float> centers(3 * (size_t)n), radii(n);
  if (fread(centers.data(), 4, centers.size(), fp) != centers.size() || fread(radii.data(), 4, n, fp) != n) return 2;
  fclose(fp);
  nanort::BVHBuildOptions<float> options;
  options.cache_bbox = false;
  nanort::SphereGeometry geom(centers.data(), radii.data());
  nanort::SpherePred pred(centers.data());
  nanort::BVHAccel<float> accel;
  if (!accel.Build(n, geom, pred, options)) return 3;
  const uint64_t nr = (uint64_t)W * H;
  std::vector<nanort::Ray<float> > rays(nr);
  std::vector<nanort::SphereIntersection> hits(nr);
  std::vector<unsigned char> mask(nr, 0);
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++) {
      nanort::Ray<float> &ray = rays[(size_t)y * W + x];
      ray.org[0] = 0.0f;
      ray.org[1] = 0.0f;
      ray.org[2] = 4.0f;
      nanort::real3<float> dir((x / (float)W) - 0.5f, (y / (float)H) - 0.5f, -1.0f);
      dir = vnormalize(dir);
      ray.dir[0] = dir[0];
      ray.dir[1] = dir[1];
      ray.dir[2] = dir[2];
      ray.min_t = 0.0f;
      ray.max_t = 1.0e+30f;
      nanort::SphereIntersector<nanort::SphereIntersection> isecter(centers.data(), radii.data());
      nanort::SphereIntersection isect;
      isect.u = isect.v = 0.0f;
      isect.t = ray.max_t;
      isect.prim_id = 0xFFFFFFFFu;
      mask[(size_t)y * W + x] = accel.Traverse(ray, isecter, &isect) ? 1 : 0;
      hits[(size_t)y * W + x] = isect;
    }
  }
#ifdef NANORT_USE_HIP_BACKEND
  std::vector<nanort::SphereIntersection> bhits(nr);
  std::vector<unsigned char> bmask(nr, 0);
  for (uint64_t i = 0; i < nr; i++) {
    bhits[i].u = bhits[i].v = 0.0f;
    bhits[i].t = rays[i].max_t;
    bhits[i].prim_id = 0xFFFFFFFFu;
  }
  if (!accel.TraverseBatch(rays.data(), nr, bhits.data(), bmask.data())) {
    fprintf(stderr, "TraverseBatch failed: %s\n", accel.LastBackendError().c_str());
    return 4;
  }
  uint64_t bad = 0;
  double worst_uv = 0.0;
  for (uint64_t i = 0; i < nr; i++) {
    if (bmask[i] != mask[i] || bhits[i].t != hits[i].t || bhits[i].prim_id != hits[i].prim_id) bad++;
    worst_uv = std::max(worst_uv, (double)std::fabs(bhits[i].u - hits[i].u));
    worst_uv = std::max(worst_uv, (double)std::fabs(bhits[i].v - hits[i].v));
  }
  printf("batch_vs_per_ray_mismatches %llu worst_uv %.3g\n", (unsigned long long)bad, worst_uv);
  if (bad || worst_uv > 1e-6) return 5;
  {  // records of one primitive kind must not be handed out as another's: the triangle and cylinder overloads refuse
    std::vector<nanort::TriangleIntersection<float> > th(16);
    std::vector<nanort::CylinderIntersection> ch(16);
    const bool tri_ok = accel.TraverseBatch(rays.data(), 16, th.data());
    const bool cyl_ok = accel.TraverseBatch(rays.data(), 16, ch.data());
    printf("wrong_kind_refused %d\n", (!tri_ok && !cyl_ok && !accel.LastBackendError().empty()) ? 1 : 0);
    if (tri_ok || cyl_ok) return 6;
  }
#endif
  fp = fopen(out_path, "wb");
  if (!fp) return 2;
  fwrite(hits.data(), sizeof(hits[0]), nr, fp);
  fwrite(mask.data(), 1, nr, fp);
  uint64_t nn = accel.GetNodes().size();
  fwrite(&nn, 8, 1, fp);
  fwrite(accel.GetNodes().data(), sizeof(nanort::BVHNode<float>), nn, fp);
  fwrite(accel.GetIndices().data(), 4, n, fp);
  fclose(fp);
  return 0;
}

// The cylinder example's main() over the header's built-in cylinder classes; scene file {u32 n; float ends[n][2][3];
// float radii[n][2]}.  Records are compared field by field (all of them are exact: no libm beyond sqrt).
static int cylinders(const char *scene_path, int W, int H, const char *out_path) {
  FILE *fp = fopen(scene_path, "rb");
  if (!fp) return 2;
  uint32_t n = 0;
  if (fread(&n, 4, 1, fp) != 1) return 2;
  std::vector<float> ends(6 * (size_t)n), radii(2 * (size_t)n);
  if (fread(ends.data(), 4, ends.size(), fp) != ends.size() || fread(radii.data(), 4, radii.size(), fp) != radii.size()) return 2;
  fclose(fp);
  nanort::BVHBuildOptions<float> options;
  options.cache_bbox = false;
  nanort::CylinderGeometry geom(ends.data(), radii.data());
  nanort::CylinderPred pred(ends.data());
  nanort::BVHAccel<float> accel;
  if (!accel.Build(n, geom, pred, options)) return 3;
  const uint64_t nr = (uint64_t)W * H;
  std::vector<nanort::Ray<float> > rays(nr);
  std::vector<nanort::CylinderIntersection> hits(nr);
  std::vector<unsigned char> mask(nr, 0);
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++) {
      nanort::Ray<float> &ray = rays[(size_t)y * W + x];
      ray.org[0] = 0.0f;
      ray.org[1] = 0.0f;
      ray.org[2] = 4.0f;
      nanort::real3<float> dir((x / (float)W) - 0.5f, (y / (float)H) - 0.5f, -1.0f);
      dir = vnormalize(dir);
      ray.dir[0] = dir[0];
      ray.dir[1] = dir[1];
      ray.dir[2] = dir[2];
      ray.min_t = 0.0f;
      ray.max_t = 1.0e+30f;
      nanort::CylinderIntersector<nanort::CylinderIntersection> isecter(ends.data(), radii.data());
      nanort::CylinderIntersection isect;
      memset(static_cast<void *>(&isect), 0, sizeof(isect));
      isect.t = ray.max_t;
      isect.prim_id = 0xFFFFFFFFu;
      mask[(size_t)y * W + x] = accel.Traverse(ray, isecter, &isect) ? 1 : 0;
      hits[(size_t)y * W + x] = isect;
    }
  }
#ifdef NANORT_USE_HIP_BACKEND
  std::vector<nanort::CylinderIntersection> bhits(nr);
  std::vector<unsigned char> bmask(nr, 0);
  for (uint64_t i = 0; i < nr; i++) {
    memset(static_cast<void *>(&bhits[i]), 0, sizeof(bhits[i]));
    bhits[i].t = rays[i].max_t;
    bhits[i].prim_id = 0xFFFFFFFFu;
  }
  if (!accel.TraverseBatch(rays.data(), nr, bhits.data(), bmask.data())) {
    fprintf(stderr, "TraverseBatch failed: %s\n", accel.LastBackendError().c_str());
    return 4;
  }
  uint64_t bad = 0;
  for (uint64_t i = 0; i < nr; i++)
    if (bmask[i] != mask[i] || memcmp(&bhits[i], &hits[i], sizeof(hits[i])) != 0) bad++;
  printf("batch_vs_per_ray_mismatches %llu\n", (unsigned long long)bad);
  if (bad) return 5;
#ifdef NANORT_ENABLE_SERIALIZATION
  {
    // Dump() / Load() of a SEGMENTED cylinder tree (its index array names a cylinder once per segment, so it is longer than
    // the primitive count): the file round-trips both arrays at their own lengths and the re-sent tree gives the same records
    const std::string path = std::string(out_path) + ".bvh";
    const size_t nn0 = accel.GetNodes().size(), ni0 = accel.GetIndices().size();
    if (!accel.Dump(path.c_str()) || !accel.Load(path.c_str())) return 6;
    printf("dump_load nodes %zu/%zu indices %zu/%zu primitives %u\n", accel.GetNodes().size(), nn0, accel.GetIndices().size(), ni0, n);
    std::vector<nanort::CylinderIntersection> lhits(bhits.size());
    std::vector<unsigned char> lmask(nr, 0);
    for (uint64_t i = 0; i < nr; i++) {
      memset(static_cast<void *>(&lhits[i]), 0, sizeof(lhits[i]));
      lhits[i].t = rays[i].max_t;
      lhits[i].prim_id = 0xFFFFFFFFu;
    }
    if (!accel.TraverseBatch(rays.data(), nr, lhits.data(), lmask.data())) {
      fprintf(stderr, "TraverseBatch after Load failed: %s\n", accel.LastBackendError().c_str());
      return 7;
    }
    uint64_t lbad = 0;
    for (uint64_t i = 0; i < nr; i++)
      if (lmask[i] != mask[i] || memcmp(&lhits[i], &hits[i], sizeof(hits[i])) != 0) lbad++;
    printf("after_dump_load_mismatches %llu\n", (unsigned long long)lbad);
    if (lbad || accel.GetNodes().size() != nn0 || accel.GetIndices().size() != ni0) return 8;
  }
#endif
#endif
  fp = fopen(out_path, "wb");
  if (!fp) return 2;
  fwrite(hits.data(), sizeof(hits[0]), nr, fp);
  fwrite(mask.data(), 1, nr, fp);
  uint64_t nn = accel.GetNodes().size();
  fwrite(&nn, 8, 1, fp);
  fwrite(accel.GetNodes().data(), sizeof(nanort::BVHNode<float>), nn, fp);
  // (the index array names a cylinder once per SEGMENT when the GPU builder cut the cylinders: its length is written too)
  uint64_t ni = accel.GetIndices().size();
  fwrite(&ni, 8, 1, fp);
  fwrite(accel.GetIndices().data(), 4, ni, fp);
  fclose(fp);
  return 0;
}

#ifdef NANORT_USE_HIP_BACKEND
#include <thread>
// The host copy of a GPU-built tree is fetched on demand (include/nanort.h: EnsureHostTree): Build() + TraverseBatch() never
// read it back; the first GetNodes() / Traverse() does, also when eight threads make that first call at once; BoundingBox()
// and IsValid() answer without it; a copy of the accel carries its own arrays.
template <typename T>
static int lazy(const char *mesh_path, const char *rays_path) {
  FILE *fp = fopen(mesh_path, "rb");
  if (!fp) return 2;
  uint32_t nv = 0, nf = 0;
  if (fread(&nv, 4, 1, fp) != 1 || fread(&nf, 4, 1, fp) != 1) return 2;
  std::vector<T> verts(3 * (size_t)nv);
  std::vector<unsigned int> faces(3 * (size_t)nf);
  if (fread(verts.data(), sizeof(T), verts.size(), fp) != verts.size()) return 2;
  if (fread(faces.data(), 4, faces.size(), fp) != faces.size()) return 2;
  fclose(fp);
  fp = fopen(rays_path, "rb");
  if (!fp) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, fp) != 1) return 2;
  std::vector<nanort::Ray<T> > rays(n);
  if (fread(rays.data(), sizeof(nanort::Ray<T>), n, fp) != n) return 2;
  fclose(fp);

  nanort::TriangleMesh<T> mesh(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::TriangleSAHPred<T> pred(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::BVHAccel<T> accel;
  if (!accel.Build(nf, mesh, pred)) return 3;
  printf("pending_after_build %d valid %d\n", accel.HostTreePending() ? 1 : 0, accel.IsValid() ? 1 : 0);
  T lo[3], hi[3];
  accel.BoundingBox(lo, hi);
  printf("pending_after_bounds %d\n", accel.HostTreePending() ? 1 : 0);
  std::vector<nanort::TriangleIntersection<T> > bh(n);
  std::vector<unsigned char> bm(n, 0);
  for (uint64_t i = 0; i < n; i++) {
    memset(&bh[i], 0, sizeof(bh[i]));
    bh[i].t = rays[i].max_t;
    bh[i].prim_id = 0xFFFFFFFFu;
  }
  if (!accel.TraverseBatch(rays.data(), n, bh.data(), bm.data())) return 4;
  printf("pending_after_batch %d\n", accel.HostTreePending() ? 1 : 0);
  // eight threads make the first per-ray call at the same time
  std::vector<nanort::TriangleIntersection<T> > hits(n);
  std::vector<unsigned char> mask(n, 0);
  std::vector<std::thread> pool;
  for (int w = 0; w < 8; w++)
    pool.push_back(std::thread([&, w]() {
      nanort::TriangleIntersector<T> isector(verts.data(), faces.data(), sizeof(T) * 3);
      for (uint64_t i = w; i < n; i += 8) {
        memset(&hits[i], 0, sizeof(hits[i]));
        hits[i].t = rays[i].max_t;
        hits[i].prim_id = 0xFFFFFFFFu;
        mask[i] = accel.Traverse(rays[i], isector, &hits[i]) ? 1 : 0;
      }
    }));
  for (size_t w = 0; w < pool.size(); w++) pool[w].join();
  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; i++)
    if (bm[i] != mask[i] || bh[i].t != hits[i].t || bh[i].u != hits[i].u || bh[i].v != hits[i].v || bh[i].prim_id != hits[i].prim_id) bad++;  // (fields: the fp64 record ends in padding)
  printf("pending_after_traverse %d threads_vs_batch_mismatches %llu\n", accel.HostTreePending() ? 1 : 0, (unsigned long long)bad);
  const std::vector<nanort::BVHNode<T> > &nodes = accel.GetNodes();
  int box_ok = !nodes.empty();
  for (int k = 0; k < 3 && box_ok; k++) box_ok = nodes[0].bmin[k] == lo[k] && nodes[0].bmax[k] == hi[k];
  printf("bounds_match_root %d nodes %zu indices %zu\n", box_ok, nodes.size(), accel.GetIndices().size());
  // a rebuild leaves the tree on the device again; a copy taken now carries the arrays, and so does the original afterwards
  if (!accel.Build(nf, mesh, pred)) return 3;
  const int pend = accel.HostTreePending() ? 1 : 0;
  nanort::BVHAccel<T> copy(accel);
  nanort::BVHAccel<T> assigned;
  assigned = accel;
  printf("pending_after_rebuild %d copy_pending %d copy_nodes %zu assigned_nodes %zu same_bytes %d\n", pend, copy.HostTreePending() ? 1 : 0,
         copy.GetNodes().size(), assigned.GetNodes().size(),
         copy.GetNodes().size() == nodes.size() && memcmp(copy.GetNodes().data(), accel.GetNodes().data(), nodes.size() * sizeof(nanort::BVHNode<T>)) == 0 ? 1 : 0);
  // the copy traces through the shared device context as before
  std::vector<nanort::TriangleIntersection<T> > ch(bh);
  std::vector<unsigned char> cm(n, 0);
  if (!copy.TraverseBatch(rays.data(), n, ch.data(), cm.data())) return 4;
  bad = 0;
  for (uint64_t i = 0; i < n; i++)
    if (bm[i] != cm[i] || (cm[i] && (bh[i].t != ch[i].t || bh[i].u != ch[i].u || bh[i].v != ch[i].v || bh[i].prim_id != ch[i].prim_id))) bad++;
  printf("copy_batch_mismatches %llu\n", (unsigned long long)bad);
  return 0;
}
#endif

int main(int argc, char **argv) {
#ifdef NANORT_USE_HIP_BACKEND
  if (argc == 5 && !strcmp(argv[1], "lazy")) return !strcmp(argv[2], "f64") ? lazy<double>(argv[3], argv[4]) : lazy<float>(argv[3], argv[4]);
#endif
  if (argc >= 2 && !strcmp(argv[1], "regress30")) return regress30(argc > 2);
  if (argc == 6 && !strcmp(argv[1], "cylinders")) return cylinders(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5]);
  if (argc == 6 && !strcmp(argv[1], "spheres")) return spheres(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5]);
  if (argc == 6 && !strcmp(argv[1], "trace")) {
    return !strcmp(argv[2], "f64") ? trace<double>(argv[3], argv[4], argv[5]) : trace<float>(argv[3], argv[4], argv[5]);
  }
  fprintf(stderr, "usage: host_check regress30 [x] | trace f32|f64 MESH RAYS OUT | spheres|cylinders SCENE W H OUT\n");
  return 64;
}

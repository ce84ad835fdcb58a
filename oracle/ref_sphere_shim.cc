// oracle/ref_sphere_shim.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the UNMODIFIED custom-primitive example of the reference,
// examples/particle_primitive/main.cc (SpherePred :82-108, SphereGeometry :113-147, SphereIntersector
// :161-291, GenerateRandomSpheres :295-325), on top of the unmodified nanort.h.  The example's translation
// unit is included where it lies (its main() renamed by the preprocessor); compiled into
// oracle/_ref/libsphere_ref.so by oracle/Makefile.  Used to generate tests/golden/spheres_ref.npz and to pin
// the C restatement (oracle/sphere_oracle.c).
#include <stdint.h>
#include <string.h>

#define main nrt_particle_example_main
#include "main.cc"  // -I$(REFERENCE)/examples/particle_primitive -I$(REFERENCE)/examples/common -I$(REFERENCE)
#undef main

extern "C" {

struct RefSphereAccel {
  std::vector<float> centers, radii;
  nanort::BVHAccel<float> accel;
};

void refsp_generate(float *centers, float *radii, uint64_t n, const float bmin[3], const float bmax[3]) {
  GenerateRandomSpheres(centers, radii, (size_t)n, bmin, bmax);
}

void *refsp_build(const float *centers, const float *radii, uint32_t n, uint32_t *num_nodes, uint32_t stats[3]) {
  RefSphereAccel *a = new RefSphereAccel();
  a->centers.assign(centers, centers + 3 * (size_t)n);
  a->radii.assign(radii, radii + n);
  nanort::BVHBuildOptions<float> options;  // the example's options (main.cc:340-341)
  options.cache_bbox = false;
  SphereGeometry geom(a->centers.data(), a->radii.data());
  SpherePred pred(a->centers.data());
  if (!a->accel.Build(n, geom, pred, options)) {
    delete a;
    return NULL;
  }
  nanort::BVHBuildStatistics st = a->accel.GetStatistics();
  *num_nodes = (uint32_t)a->accel.GetNodes().size();
  stats[0] = st.max_tree_depth;
  stats[1] = st.num_leaf_nodes;
  stats[2] = st.num_branch_nodes;
  return a;
}

void refsp_get_tree(void *h, void *nodes_out, uint32_t *indices_out) {
  RefSphereAccel *a = static_cast<RefSphereAccel *>(h);
  memcpy(nodes_out, a->accel.GetNodes().data(), a->accel.GetNodes().size() * sizeof(nanort::BVHNode<float>));
  memcpy(indices_out, a->accel.GetIndices().data(), a->accel.GetIndices().size() * sizeof(unsigned int));
}

void refsp_destroy(void *h) { delete static_cast<RefSphereAccel *>(h); }

// rays: nanort::Ray<float>[n] (36 B); hits: {u, v, t, prim_id}[n]; a miss leaves {0, 0, max_t, 0xFFFFFFFF}.
void refsp_traverse(void *h, const void *rays, uint64_t n, uint32_t range0, uint32_t range1, void *hits,
                    uint8_t *mask) {
  RefSphereAccel *a = static_cast<RefSphereAccel *>(h);
  const nanort::Ray<float> *r = static_cast<const nanort::Ray<float> *>(rays);
  struct Out {
    float u, v, t;
    uint32_t prim_id;
  } *o = static_cast<Out *>(hits);
  nanort::BVHTraceOptions opt;
  opt.prim_ids_range[0] = range0;
  opt.prim_ids_range[1] = range1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (int64_t i = 0; i < (int64_t)n; i++) {
    SphereIntersector<SphereIntersection> isecter(a->centers.data(), a->radii.data());
    SphereIntersection isect;
    isect.u = isect.v = 0.f;
    isect.t = r[i].max_t;
    isect.prim_id = 0xFFFFFFFFu;
    const bool hit = a->accel.Traverse(r[i], isecter, &isect, opt);
    o[i].u = isect.u;
    o[i].v = isect.v;
    o[i].t = isect.t;
    o[i].prim_id = isect.prim_id;
    if (mask) mask[i] = hit ? 1 : 0;
  }
}

}  // extern "C"

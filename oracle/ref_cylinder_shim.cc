// oracle/ref_cylinder_shim.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the UNMODIFIED custom-primitive example examples/cylinder_primitive/main.cc (solve2e
// :61-90, CylinderPred :94-120, CylinderGeometry :124-210, CylinderIntersector :226-424, GenerateRandomCylinders
// :428-462) on top of the unmodified nanort.h; the example's translation unit is included where it lies with its
// main() renamed.  Compiled into oracle/_ref/libcylinder_ref.so by oracle/Makefile; generates
// tests/golden/cylinders_ref.npz and pins oracle/cylinder_oracle.c.
#include <stdint.h>
#include <string.h>

#define main nrt_cylinder_example_main
#include "main.cc"  // -I$(REFERENCE)/examples/cylinder_primitive -I$(REFERENCE)
#undef main

extern "C" {

struct RefCylAccel {
  std::vector<float> verts, radii;
  nanort::BVHAccel<float> accel;
};

void refcy_generate(float *verts, float *radii, uint64_t n, const float bmin[3], const float bmax[3]) {
  GenerateRandomCylinders(verts, radii, (size_t)n, bmin, bmax);
}

void *refcy_build(const float *verts, const float *radii, uint32_t n, uint32_t *num_nodes, uint32_t stats[3]) {
  RefCylAccel *a = new RefCylAccel();
  a->verts.assign(verts, verts + 6 * (size_t)n);
  a->radii.assign(radii, radii + 2 * (size_t)n);
  nanort::BVHBuildOptions<float> options;  // the example's options (main.cc:476-477)
  options.cache_bbox = false;
  CylinderGeometry geom(a->verts.data(), a->radii.data());
  CylinderPred pred(a->verts.data());
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

void refcy_get_tree(void *h, void *nodes_out, uint32_t *indices_out) {
  RefCylAccel *a = static_cast<RefCylAccel *>(h);
  memcpy(nodes_out, a->accel.GetNodes().data(), a->accel.GetNodes().size() * sizeof(nanort::BVHNode<float>));
  memcpy(indices_out, a->accel.GetIndices().data(), a->accel.GetIndices().size() * sizeof(unsigned int));
}

void refcy_destroy(void *h) { delete static_cast<RefCylAccel *>(h); }

// hits: {u, v, normal[3], t, prim_id}[n] (28 B).  The example's PostTraversal never writes isect->t
// (main.cc:367-418); the shim reports the intersector's t through GetT().  A miss leaves
// {0, 0, (0,0,0), max_t, 0xFFFFFFFF}.
void refcy_traverse(void *h, const void *rays, uint64_t n, uint32_t range0, uint32_t range1, int test_cap, void *hits,
                    uint8_t *mask) {
  RefCylAccel *a = static_cast<RefCylAccel *>(h);
  const nanort::Ray<float> *r = static_cast<const nanort::Ray<float> *>(rays);
  struct Out {
    float u, v, normal[3], t;
    uint32_t prim_id;
  } *o = static_cast<Out *>(hits);
  nanort::BVHTraceOptions opt;
  opt.prim_ids_range[0] = range0;
  opt.prim_ids_range[1] = range1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (int64_t i = 0; i < (int64_t)n; i++) {
    CylinderIntersector<CylinderIntersection> isecter(a->verts.data(), a->radii.data(), test_cap != 0);
    CylinderIntersection isect;
    isect.u = isect.v = 0.f;
    isect.normal[0] = isect.normal[1] = isect.normal[2] = 0.f;
    isect.t = r[i].max_t;
    isect.prim_id = 0xFFFFFFFFu;
    const bool hit = a->accel.Traverse(r[i], isecter, &isect, opt);
    o[i].u = isect.u;
    o[i].v = isect.v;
    o[i].normal[0] = isect.normal[0];
    o[i].normal[1] = isect.normal[1];
    o[i].normal[2] = isect.normal[2];
    o[i].t = hit ? isecter.GetT() : r[i].max_t;
    o[i].prim_id = isect.prim_id;
    if (mask) mask[i] = hit ? 1 : 0;
  }
}

}  // extern "C"

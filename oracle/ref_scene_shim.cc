// oracle/ref_scene_shim.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the UNMODIFIED two-level scene graph of the reference,
// examples/nanosg/nanosg.h (Scene::Commit :700-760, Scene::Traverse :780-880) on top of the unmodified
// nanort.h (ListNodeIntersections :2608-2692).  Compiled from where the sources lie into
// oracle/_ref/libnanosg_ref.so by oracle/Makefile; used to generate tests/golden/scene_*.npz and to pin
// the C restatement (oracle/nanosg_oracle.c).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "nanosg.h"  // -I$(REFERENCE)/examples/nanosg -I$(REFERENCE)

namespace {

// The mesh concept nanosg needs (nanosg.h:400-411, 828-846): flat arrays + a normal query.
struct ShimMesh {
  std::vector<float> vertices;
  std::vector<unsigned int> faces;
  size_t stride;
  const float *GetVertices() const { return vertices.data(); }
  const unsigned int *GetFaces() const { return faces.data(); }
  size_t GetVertexStrideBytes() const { return stride; }
  void GetNormal(float Ng[3], float Ns[3], unsigned int prim, float u, float v) const {
    (void)prim;
    (void)u;
    (void)v;
    Ng[0] = Ns[0] = 0.f;
    Ng[1] = Ns[1] = 0.f;
    Ng[2] = Ns[2] = 1.f;
  }
};

typedef nanosg::Node<float, ShimMesh> Node;
typedef nanosg::Scene<float, ShimMesh> Scene;
typedef nanosg::Intersection<float> Isect;

struct RefScene {
  std::vector<ShimMesh *> meshes;
  Scene scene;
  ~RefScene() {
    for (size_t i = 0; i < meshes.size(); i++) delete meshes[i];
  }
};

}  // namespace

extern "C" {

void *refsg_create(void) { return new RefScene(); }
void refsg_destroy(void *h) { delete static_cast<RefScene *>(h); }

// xform: nanosg's T[4][4] (row-major storage; MultV treats rows 0..2 as the basis and row 3 as the
// translation, nanosg.h:232-240).
int refsg_add_node(void *h, const float *verts, uint32_t num_verts, const uint32_t *faces, uint32_t num_faces,
                   const float xform[16]) {
  RefScene *s = static_cast<RefScene *>(h);
  ShimMesh *m = new ShimMesh();
  m->vertices.assign(verts, verts + 3 * (size_t)num_verts);
  m->faces.assign(faces, faces + 3 * (size_t)num_faces);
  m->stride = sizeof(float) * 3;
  s->meshes.push_back(m);
  Node node(m);
  float x[4][4];
  memcpy(x, xform, sizeof(x));
  node.SetLocalXform(x);
  s->scene.AddNode(node);
  return (int)s->meshes.size() - 1;
}

int refsg_commit(void *h) { return static_cast<RefScene *>(h)->scene.Commit() ? 1 : 0; }

// Per-node state after Commit(), for pinning the restatement: world AABB (6), inv_xform (16), inv_xform33 (16), xform (16).
void refsg_node_state(void *h, uint32_t node, float *out54) {
  const Node &n = static_cast<RefScene *>(h)->scene.GetNodes()[node];
  float bmin[3], bmax[3];
  n.GetWorldBoundingBox(bmin, bmax);
  memcpy(out54, bmin, 12);
  memcpy(out54 + 3, bmax, 12);
  memcpy(out54 + 6, n.inv_xform_, 64);
  memcpy(out54 + 22, n.inv_xform33_, 64);
  memcpy(out54 + 38, n.xform_, 64);
}

// Scene::GetBoundingBox (nanosg.h:761-769) after Commit().
void refsg_bounds(void *h, float bmin[3], float bmax[3]) { static_cast<RefScene *>(h)->scene.GetBoundingBox(bmin, bmax); }

// hits_out: {t, u, v, prim_id(u32), node_id(u32)} = 20 bytes per ray
void refsg_traverse(void *h, const void *rays, uint64_t n, int cull_back_face, void *hits_out, uint8_t *mask) {
  RefScene *s = static_cast<RefScene *>(h);
  const nanort::Ray<float> *r = static_cast<const nanort::Ray<float> *>(rays);
  unsigned char *out = static_cast<unsigned char *>(hits_out);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (int64_t i = 0; i < (int64_t)n; i++) {
    nanort::Ray<float> ray = r[i];
    Isect isect;
    memset(&isect, 0, sizeof(isect));
    const bool hit = s->scene.Traverse<Isect, nanort::TriangleIntersector<float, Isect> >(ray, &isect, cull_back_face != 0);
    float rec[3] = {hit ? isect.t : ray.max_t, hit ? isect.u : 0.f, hit ? isect.v : 0.f};
    uint32_t ids[2] = {hit ? isect.prim_id : 0xFFFFFFFFu, hit ? isect.node_id : 0xFFFFFFFFu};
    memcpy(out + 20 * i, rec, 12);
    memcpy(out + 20 * i + 12, ids, 8);
    if (mask) mask[i] = hit ? 1 : 0;
  }
}

}  // extern "C"

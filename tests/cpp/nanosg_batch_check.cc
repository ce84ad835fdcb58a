// tests/cpp/nanosg_batch_check.cc — drives include/nanosg_hip.h (BatchTracer) the way a NanoSG application would.
//
//   nanosg_batch_check SCENE RAYS OUT
//
// SCENE: u32 count, then per instance {u32 nv, u32 nf, float xyz[nv], u32 ijk[nf], float xform[16]};
// RAYS: u64 n, Ray<float>[n]; OUT: per ray {t, u, v, prim_id, node_id, P[3], Ns[3], Ng[3]} (56 B) then the mask.
//
// The scene classes below are a stand-in for the reference's nanosg::Node / nanosg::Scene with exactly the public
// surface BatchTracer uses (the reference header is not available on the GPU box; tests/test_host_header.py checks
// in the build container that nanosg_hip.h also compiles against the unmodified nanosg.h).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "nanort.h"
#include "nanosg_hip.h"

struct Mesh {
  std::vector<float> vertices;
  std::vector<unsigned int> faces;
  size_t stride;
  // flat shading: the triangle's unit normal from its vertices
  void GetNormal(float Ng[3], float Ns[3], unsigned int prim, float, float) const {
    const float *a = &vertices[3 * faces[3 * prim]], *b = &vertices[3 * faces[3 * prim + 1]], *c = &vertices[3 * faces[3 * prim + 2]];
    const nanort::real3<float> n = vnormalize(vcross(nanort::real3<float>(b) - nanort::real3<float>(a), nanort::real3<float>(c) - nanort::real3<float>(a)));
    for (int k = 0; k < 3; k++) Ng[k] = Ns[k] = n[k];
  }
};

struct Isect {  // the fields of nanosg::Intersection<float>
  float t;
  unsigned int prim_id;
  float u, v;
  unsigned int node_id;
  nanort::real3<float> P, Ns, Ng;
};

class MiniNode {
 public:
  explicit MiniNode(const Mesh *m) : mesh_(m) { memset(local_, 0, sizeof(local_)); }
  bool Update() {  // the part of nanosg::Node::Update that builds the local BVH (nanosg.h:400-415)
    nanort::TriangleMesh<float> tm(mesh_->vertices.data(), mesh_->faces.data(), mesh_->stride);
    nanort::TriangleSAHPred<float> tp(mesh_->vertices.data(), mesh_->faces.data(), mesh_->stride);
    return accel_.Build(static_cast<unsigned int>(mesh_->faces.size()) / 3, tm, tp);
  }
  void SetLocalXform(const float x[16]) { memcpy(local_, x, sizeof(local_)); }
  const float *GetLocalXformPtr() const { return local_; }
  const Mesh *GetMesh() const { return mesh_; }
  const nanort::BVHAccel<float> &GetAccel() const { return accel_; }

 private:
  const Mesh *mesh_;
  nanort::BVHAccel<float> accel_;
  float local_[16];
};

struct MiniScene {
  std::vector<MiniNode> nodes;
  const std::vector<MiniNode> &GetNodes() const { return nodes; }
};

int main(int argc, char **argv) {
  if (argc != 4) return 64;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  uint32_t count = 0;
  if (fread(&count, 4, 1, fp) != 1) return 2;
  std::vector<Mesh> meshes(count);
  std::vector<std::vector<float> > xforms(count, std::vector<float>(16));
  for (uint32_t i = 0; i < count; i++) {
    uint32_t nv = 0, nf = 0;
    if (fread(&nv, 4, 1, fp) != 1 || fread(&nf, 4, 1, fp) != 1) return 2;
    meshes[i].vertices.resize(3 * (size_t)nv);
    meshes[i].faces.resize(3 * (size_t)nf);
    meshes[i].stride = 12;
    if (fread(meshes[i].vertices.data(), 4, meshes[i].vertices.size(), fp) != meshes[i].vertices.size()) return 2;
    if (fread(meshes[i].faces.data(), 4, meshes[i].faces.size(), fp) != meshes[i].faces.size()) return 2;
    if (fread(xforms[i].data(), 4, 16, fp) != 16) return 2;
  }
  fclose(fp);
  fp = fopen(argv[2], "rb");
  if (!fp) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, fp) != 1) return 2;
  std::vector<nanort::Ray<float> > rays(n);
  if (fread(rays.data(), sizeof(nanort::Ray<float>), n, fp) != n) return 2;
  fclose(fp);

  MiniScene scene;
  for (uint32_t i = 0; i < count; i++) {
    MiniNode node(&meshes[i]);
    if (!node.Update()) return 3;
    node.SetLocalXform(xforms[i].data());
    scene.nodes.push_back(node);  // copies share the GPU context, like nanosg::Scene::AddNode's copy
  }
  nanosg::BatchTracer<MiniScene> tracer(scene);
  if (!tracer.IsValid()) {
    fprintf(stderr, "BatchTracer: %s\n", tracer.LastError().c_str());
    return 4;
  }
  std::vector<Isect> isects(n);
  std::vector<unsigned char> mask(n, 0);
  memset(static_cast<void *>(isects.data()), 0, n * sizeof(Isect));
  if (!tracer.Traverse(rays.data(), n, isects.data(), mask.data())) {
    fprintf(stderr, "Traverse: %s\n", tracer.LastError().c_str());
    return 5;
  }
  fp = fopen(argv[3], "wb");
  if (!fp) return 2;
  for (uint64_t i = 0; i < n; i++) {
    const Isect &s = isects[i];
    float rec[14] = {s.t, s.u, s.v, 0, 0, s.P[0], s.P[1], s.P[2], s.Ns[0], s.Ns[1], s.Ns[2], s.Ng[0], s.Ng[1], s.Ng[2]};
    memcpy(&rec[3], &s.prim_id, 4);
    memcpy(&rec[4], &s.node_id, 4);
    fwrite(rec, 4, 14, fp);
  }
  fwrite(mask.data(), 1, n, fp);
  fclose(fp);
  printf("rays %llu hits %llu\n", (unsigned long long)n, (unsigned long long)std::count(mask.begin(), mask.end(), 1));
  return 0;
}

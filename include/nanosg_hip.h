// include/nanosg_hip.h — batched, GPU-backed traversal for NanoSG scenes.
//
// NanoSG (reference examples/nanosg/nanosg.h) is user-level code on top of nanort.h: it compiles unchanged against
// this repository's include/nanort.h, and with -DNANORT_USE_HIP_BACKEND every Node::Update() already builds its
// local BVH on the GPU (nanosg.h:400-415 calls BVHAccel::Build with the built-in triangle types).  What NanoSG lacks
// is a way to trace more than one ray per call.  This add-on supplies it without touching nanosg.h:
//
//     #include "nanort.h"        // this repository's, -DNANORT_USE_HIP_BACKEND
//     #include "nanosg.h"        // the reference's, unmodified (or anything with the same public surface)
//     #include "nanosg_hip.h"
//     nanosg::Scene<float, Mesh> scene;  ... AddNode ... ; scene.Commit();
//     nanosg::BatchTracer<nanosg::Scene<float, Mesh> > tracer(scene);
//     tracer.Traverse(rays, n, isects, hit);     // == n x scene.Traverse<...>(rays[i], &isects[i]), one GPU pass
//
// It replaces Scene::Traverse (nanosg.h:773-870): ListNodeIntersections over the node AABBs, per-node Traverse of the
// ray transformed into the node's space, world-distance comparison, and the Intersection record {t, prim_id, u, v,
// node_id, P, Ns, Ng} — through the nrtScene* entry points of include/nanort_hip.h.  Required of the scene type:
// GetNodes() -> container of nodes; of a node: GetMesh() (-> vertices, faces, stride, GetNormal(Ng, Ns, prim, u, v)),
// GetAccel() (a nanort::BVHAccel<float> built by Build()), GetLocalXformPtr().  Children of a node are not traced,
// exactly as in the reference (its Traverse walks the top-level nodes only).  fp32 (the reference's Node::Update is).
#ifndef NANOSG_HIP_H_
#define NANOSG_HIP_H_

#ifndef NANORT_USE_HIP_BACKEND
#error "nanosg_hip.h needs include/nanort.h compiled with -DNANORT_USE_HIP_BACKEND"
#endif

#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "nanort.h"
#include "nanort_hip.h"

namespace nanosg {

template <class SceneT>
class BatchTracer {
 public:
  // `scene` must be committed and must outlive the tracer; so must its nodes' meshes.
  explicit BatchTracer(const SceneT &scene) : scene_(&scene), handle_(NULL) {
    int device = 0;
    if (const char *env = std::getenv("NANORT_HIP_DEVICE")) device = std::atoi(env);
    if (nrtSceneCreate(device, &handle_) != NRT_OK) {
      error_ = nrtSceneLastError(NULL);
      handle_ = NULL;
      return;
    }
    for (size_t i = 0; i < scene.GetNodes().size(); i++) {
      nrt_ctx *ctx = scene.GetNodes()[i].GetAccel().HipContext();
      uint32_t id = 0;
      if (!ctx || nrtSceneAddNode_f32(handle_, ctx, scene.GetNodes()[i].GetLocalXformPtr(), &id) != NRT_OK) {
        error_ = ctx ? nrtSceneLastError(handle_) : "a node's BVHAccel was not built on the GPU (Node::Update() before AddNode/Commit?)";
        Release();
        return;
      }
    }
    if (nrtSceneCommit(handle_) != NRT_OK) {
      error_ = nrtSceneLastError(handle_);
      Release();
      return;
    }
    state_.resize(scene.GetNodes().size());
    for (size_t i = 0; i < state_.size(); i++) {
      if (nrtSceneNodeState_f32(handle_, static_cast<uint32_t>(i), state_[i].m) != NRT_OK) {
        error_ = nrtSceneLastError(handle_);
        Release();
        return;
      }
    }
  }
  ~BatchTracer() { Release(); }

  bool IsValid() const { return handle_ != NULL; }
  const std::string &LastError() const { return error_; }

  // isects[i] is written only when ray i hits, like Scene::Traverse; hit_out[i] (optional) receives 1 / 0.
  template <class IsectT>
  bool Traverse(const nanort::Ray<float> *rays, size_t num_rays, IsectT *isects, unsigned char *hit_out = NULL) {
    if (!handle_) return false;
    if (num_rays == 0) return true;
    std::vector<nrt_scene_hit_f32> hits(num_rays);
    std::vector<unsigned char> mask(num_rays);
    if (nrtSceneTraverseBatch_f32(handle_, reinterpret_cast<const nrt_ray_f32 *>(rays), num_rays, &hits[0], &mask[0]) != NRT_OK) {
      error_ = nrtSceneLastError(handle_);
      return false;
    }
    for (size_t i = 0; i < num_rays; i++) {
      if (hit_out) hit_out[i] = mask[i];
      if (!mask[i]) continue;
      Finish(rays[i], hits[i], &isects[i]);
    }
    return true;
  }

  // For callers that keep their ray waves in HBM: device pointers in and out, compact records {t, u, v, prim_id, node_id}
  // (every record is written: a miss carries t = ray.max_t and ids 0xFFFFFFFF), optional hit flags.  The rest of the
  // reference's Intersection record (P, Ns, Ng) is the shader's business on the device.  Synchronous.
  bool TraverseDevice(const nanort::Ray<float> *d_rays, size_t num_rays, nrt_scene_hit_f32 *d_hits, unsigned char *d_hit_flags = NULL) {
    if (!handle_) return false;
    if (nrtSceneTraverseBatchDevice_f32(handle_, reinterpret_cast<const nrt_ray_f32 *>(d_rays), num_rays, d_hits, d_hit_flags) != NRT_OK) {
      error_ = nrtSceneLastError(handle_);
      return false;
    }
    return true;
  }

 private:
  struct NodeState {
    float m[64];  // xform, inv_xform, inv_xform33, inv_transpose_xform33 (nrtSceneNodeState_f32)
    const float *Xform() const { return m; }
    const float *InvXform() const { return m + 16; }
    const float *InvXform33() const { return m + 32; }
    const float *InvTransposeXform33() const { return m + 48; }
  };

  // dst = v * M with M's row 3 as the translation: Matrix::MultV, nanosg.h:232-240
  static void MultV(float dst[3], const float *M, const float v[3]) {
    const float x = M[0] * v[0] + M[4] * v[1] + M[8] * v[2] + M[12];
    const float y = M[1] * v[0] + M[5] * v[1] + M[9] * v[2] + M[13];
    const float z = M[2] * v[0] + M[6] * v[1] + M[10] * v[2] + M[14];
    dst[0] = x;
    dst[1] = y;
    dst[2] = z;
  }

  // The tail of Scene::Traverse for the winning node (nanosg.h:834-862).  The local hit distance is not part of the
  // compact record; it is recovered exactly by intersecting the local ray with the one winning triangle (t depends on
  // the ray and the triangle only).
  template <class IsectT>
  void Finish(const nanort::Ray<float> &ray, const nrt_scene_hit_f32 &h, IsectT *isect) const {
    const NodeState &st = state_[h.node_id];
    const typename std::remove_reference<decltype(scene_->GetNodes()[0])>::type &node = scene_->GetNodes()[h.node_id];
    nanort::Ray<float> local_ray;  // default min_t / max_t, as in the reference (nanosg.h:803-808)
    MultV(local_ray.org, st.InvXform(), ray.org);
    MultV(local_ray.dir, st.InvXform33(), ray.dir);
    nanort::TriangleIntersector<float> one(node.GetMesh()->vertices.data(), node.GetMesh()->faces.data(), node.GetMesh()->stride);
    one.PrepareTraversal(local_ray, nanort::BVHTraceOptions());
    float t_local = std::numeric_limits<float>::max();
    (void)one.Intersect(&t_local, h.prim_id);
    float local_P[3];
    for (int k = 0; k < 3; k++) local_P[k] = local_ray.org[k] + t_local * local_ray.dir[k];
    isect->node_id = h.node_id;
    isect->prim_id = h.prim_id;
    isect->u = h.u;
    isect->v = h.v;
    float Ng[3], Ns[3];
    node.GetMesh()->GetNormal(Ng, Ns, h.prim_id, h.u, h.v);
    isect->t = h.t;
    float P[3], wNg[3], wNs[3];
    MultV(P, st.Xform(), local_P);
    MultV(wNg, st.InvTransposeXform33(), Ng);
    MultV(wNs, st.InvTransposeXform33(), Ns);
    for (int k = 0; k < 3; k++) {
      isect->P[k] = P[k];
      isect->Ng[k] = wNg[k];
      isect->Ns[k] = wNs[k];
    }
  }

  void Release() {
    if (handle_) nrtSceneDestroy(handle_);
    handle_ = NULL;
  }

  BatchTracer(const BatchTracer &);             // not copyable
  BatchTracer &operator=(const BatchTracer &);

  const SceneT *scene_;
  nrt_scene *handle_;
  std::vector<NodeState> state_;
  std::string error_;
};

}  // namespace nanosg

#endif  // NANOSG_HIP_H_

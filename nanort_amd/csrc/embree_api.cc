// nanort_amd/csrc/embree_api.cc — libnanort_embree.so: the Embree 2.x C API over the GPU two-level traversal.
//
// Replaces examples/embree-api/nanort-embree.cc of the reference (an Embree2 shim over nanosg + nanort, single rays
// on the CPU).  Same entry points, same observable behaviour (include/embree2/rtcore.h lists where that differs from
// real Embree), plus ray streams.  Host-only C++ over the C ABI of libnanort_hip.so: one nrt_ctx per triangle mesh
// (built by nrtBuild_f32 with default options, as nanosg::Node::Update does, nanosg.h:400-415), one nrt_scene per
// RTCScene with identity node transforms (nanort-embree.cc:321-332 adds every mesh as a root node).
//
// Nothing here computes an intersection: every query goes through nrtSceneTraverseBatch_f32.  There is no CPU
// fallback; when the GPU library fails the error is recorded on the device (rtcDeviceGetError / the error callback)
// and the rays come back as misses.
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/embree2/rtcore.h"
#include "../../include/embree2/rtcore_ray.h"
#include "../../include/nanort_hip.h"

static_assert(sizeof(RTCRay) == 96 && alignof(RTCRay) == 16, "Embree 2 RTCRay is 96 bytes, 16-byte aligned");
static_assert(offsetof(RTCRay, dir) == 16 && offsetof(RTCRay, tnear) == 32 && offsetof(RTCRay, tfar) == 36, "RTCRay layout");
static_assert(offsetof(RTCRay, Ng) == 48 && offsetof(RTCRay, u) == 64 && offsetof(RTCRay, geomID) == 72, "RTCRay layout");
static_assert(offsetof(RTCRay, primID) == 76 && offsetof(RTCRay, instID) == 80, "RTCRay layout");
static_assert(sizeof(RTCBounds) == 32, "RTCBounds layout");

namespace {

struct Device;

struct Mesh {
  std::vector<float> vertices;  // xyz + pad per vertex: Embree's 16-byte stride (nanort-embree.cc:150-155)
  std::vector<uint32_t> faces;  // 3 per triangle
  nrt_ctx *ctx = nullptr;
  bool dirty = true;  // host buffers changed since the last upload + build
  ~Mesh() {
    if (ctx) nrtDestroy(ctx);
  }
};

// Grow-only host staging for a query's rays / records: page-locked when the backend can provide it (the copies of
// nrtSceneTraverseBatch_f32 then run at PCIe speed), plain malloc otherwise.
struct HostBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void *ensure(size_t bytes) {
    if (bytes <= cap && p) return p;
    release();
    const size_t want = bytes + bytes / 4;
    void *q = nullptr;
    if (nrtHostAlloc(want, &q) == NRT_OK && q) {
      p = q;
      pinned = true;
    } else {
      p = malloc(want);
      pinned = false;
    }
    cap = p ? want : 0;
    return p;
  }
  void release() {
    if (p) {
      if (pinned)
        nrtHostFree(p);
      else
        free(p);
    }
    p = nullptr;
    cap = 0;
  }
  ~HostBuf() { release(); }
};

struct Scene {
  Device *device = nullptr;
  std::map<uint32_t, Mesh *> meshes;  // by geometry id; iteration order = node order, as in the reference (:321-329)
  uint32_t next_id = 1;               // id 0 is reserved (:226)
  nrt_scene *scene = nullptr;
  bool committed = false;
  std::mutex mu;  // the staging buffers below and the nrt_scene are used by one query at a time
  HostBuf rays, hits, mask;
  ~Scene() {
    if (scene) nrtSceneDestroy(scene);
    for (auto &kv : meshes) delete kv.second;
  }
};

struct Device {
  int hip_device = 0;
  std::mutex mu;
  std::map<Scene *, Scene *> scenes;
  RTCErrorFunc2 error_func = nullptr;
  void *user_ptr = nullptr;
  RTCError error = RTC_NO_ERROR;
  std::string message;
};

std::mutex g_mu;
std::map<Device *, Device *> &devices() {
  static std::map<Device *, Device *> d;
  return d;
}
std::string g_error;  // errors that have no device to land on (the reference keeps one global string, :430-445)

void report(Device *d, RTCError code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
void report(Device *d, RTCError code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  RTCErrorFunc2 f = nullptr;
  void *up = nullptr;
  if (d) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (d->error == RTC_NO_ERROR) {
      d->error = code;
      d->message = buf;
    }
    f = d->error_func;
    up = d->user_ptr;
  } else {
    std::lock_guard<std::mutex> lk(g_mu);
    g_error = buf;
  }
  if (f) f(up, code, buf);
}

Scene *as_scene(RTCScene s) { return reinterpret_cast<Scene *>(s); }

Mesh *find_mesh(Scene *s, unsigned id) {
  auto it = s->meshes.find(id);
  return it == s->meshes.end() ? nullptr : it->second;
}

const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// (Re)build what changed and assemble the two-level scene.  Caller holds s->mu.
bool commit_locked(Scene *s) {
  s->committed = false;
  if (s->scene) {
    nrtSceneDestroy(s->scene);
    s->scene = nullptr;
  }
  if (s->meshes.empty()) {
    report(s->device, RTC_INVALID_OPERATION, "rtcCommit: the scene has no geometry (nanosg's Commit() fails the same way)");
    return false;
  }
  const int dev = s->device ? s->device->hip_device : 0;
  if (nrtSceneCreate(dev, &s->scene) != NRT_OK) {
    report(s->device, RTC_UNKNOWN_ERROR, "rtcCommit: nrtSceneCreate failed on HIP device %d", dev);
    s->scene = nullptr;
    return false;
  }
  for (auto &kv : s->meshes) {
    Mesh *m = kv.second;
    if (m->dirty || !m->ctx) {
      if (!m->ctx && nrtCreate(dev, &m->ctx) != NRT_OK) {
        report(s->device, RTC_UNKNOWN_ERROR, "rtcCommit: nrtCreate failed: %s", nrtLastError(nullptr));
        m->ctx = nullptr;
        return false;
      }
      if (nrtSetMesh_f32(m->ctx, m->vertices.data(), 4 * sizeof(float), m->faces.data(), (uint32_t)(m->faces.size() / 3)) != NRT_OK ||
          nrtBuild_f32(m->ctx, nullptr, nullptr, nullptr) != NRT_OK) {
        report(s->device, RTC_INVALID_ARGUMENT, "rtcCommit: geometry %u: %s", kv.first, nrtLastError(m->ctx));
        return false;
      }
      m->dirty = false;
    }
    if (nrtSceneAddNode_f32(s->scene, m->ctx, kIdentity, nullptr) != NRT_OK) {
      report(s->device, RTC_UNKNOWN_ERROR, "rtcCommit: geometry %u: %s", kv.first, nrtSceneLastError(s->scene));
      return false;
    }
  }
  if (nrtSceneCommit(s->scene) != NRT_OK) {
    report(s->device, RTC_UNKNOWN_ERROR, "rtcCommit: %s", nrtSceneLastError(s->scene));
    return false;
  }
  s->committed = true;
  return true;
}

// Host threads for the record conversion: what the process may use (cgroup CPU quota), at most 16 — never the OpenMP
// default, which on a many-core host inside a quota'd container starves the HIP runtime's own threads.
static int host_threads_uncached() {
  long n = 8;
#ifdef _OPENMP
  n = omp_get_num_procs();
#endif
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32];
    long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0 && atol(q) / period >= 1) n = std::min(n, atol(q) / period);
    fclose(f);
  }
  return (int)std::max(1l, std::min(16l, n));
}
static int host_threads() {
  static const int cached = host_threads_uncached();  // (function-local static: initialised once, thread-safely)
  return cached;
}
constexpr size_t kParallelMin = 1u << 16;
constexpr size_t kPiece = 1u << 19; // rays per pipeline stage of a stream query

inline RTCRay *ray_at(RTCRay *base, size_t i, size_t stride) {
  return reinterpret_cast<RTCRay *>(reinterpret_cast<char *>(base) + i * stride);
}

// The one query path.  `get(i)` yields the i-th ray of the caller's stream.  A stream is 96 bytes of RTCRay per ray, of
// which the traversal needs 32 and returns 20: the records are converted on the host into the (page-locked) staging the
// scene call copies from.  Long streams go piece by piece with the three stages overlapped — while the GPU traces piece k
// (a helper thread sits in nrtSceneTraverseBatch_f32) this thread's team converts piece k+1 and writes piece k-1 back.
template <class Get>
void trace(Scene *s, size_t n, bool occluded, Get get) {
  if (!s || n == 0) return;
  std::lock_guard<std::mutex> lk(s->mu);
  bool ok = s->committed;
  if (!ok) report(s->device, RTC_INVALID_OPERATION, "rtcIntersect/rtcOccluded: the scene is not committed");
  nrt_ray_f32 *rays = nullptr;
  nrt_scene_hit_f32 *hits = nullptr;
  uint8_t *mask = nullptr;
  if (ok) {
    rays = static_cast<nrt_ray_f32 *>(s->rays.ensure(n * sizeof(nrt_ray_f32)));
    hits = static_cast<nrt_scene_hit_f32 *>(s->hits.ensure(n * sizeof(nrt_scene_hit_f32)));
    mask = static_cast<uint8_t *>(s->mask.ensure(n));
    if (!rays || !hits || !mask) {
      report(s->device, RTC_OUT_OF_MEMORY, "rtcIntersect/rtcOccluded: no host memory for %zu rays", n);
      ok = false;
    }
  }
  const int threads = host_threads();
  (void)threads;
  auto convert = [&](size_t lo, size_t hi) {
#pragma omp parallel for schedule(static) num_threads(threads) if (hi - lo > kParallelMin)
    for (size_t i = lo; i < hi; i++) {
      const RTCRay *r = get(i);
      nrt_ray_f32 &o = rays[i];
      for (int k = 0; k < 3; k++) {
        o.org[k] = r->org[k];
        o.dir[k] = r->dir[k];
      }
      o.min_t = r->tnear;
      o.max_t = r->tfar;
      o.type = 0u;  // RAY_TYPE_NONE: the reference's Ray default, not read by traversal
    }
  };
  auto write_back = [&](size_t lo, size_t hi, bool good) {
#pragma omp parallel for schedule(static) num_threads(threads) if (hi - lo > kParallelMin)
    for (size_t i = lo; i < hi; i++) {
      RTCRay *r = get(i);
      const bool hit = good && mask[i] != 0;
      if (occluded) {
        if (hit) r->geomID = 0;
        continue;
      }
      if (hit) {  // nanort-embree.cc:541-548
        const nrt_scene_hit_f32 &h = hits[i];
        r->tfar = h.t;
        r->u = h.u;
        r->v = h.v;
        r->geomID = h.node_id;
        r->primID = h.prim_id;
        r->instID = RTC_INVALID_GEOMETRY_ID;
      } else {  // :549-553
        r->geomID = RTC_INVALID_GEOMETRY_ID;
        r->primID = RTC_INVALID_GEOMETRY_ID;
        r->instID = RTC_INVALID_GEOMETRY_ID;
      }
    }
  };
  auto gpu = [&](size_t lo, size_t hi) {
    return nrtSceneTraverseBatch_f32(s->scene, rays + lo, hi - lo, hits + lo, mask + lo) == NRT_OK;
  };
  if (!ok) {
    write_back(0, n, false);
    return;
  }
  if (n < 2 * kPiece) {
    convert(0, n);
    ok = gpu(0, n);
    if (!ok) report(s->device, RTC_UNKNOWN_ERROR, "rtcIntersect/rtcOccluded: %s", nrtSceneLastError(s->scene));
    write_back(0, n, ok);
    return;
  }
  const size_t pieces = (n + kPiece - 1) / kPiece;
  std::vector<char> good(pieces, 0);
  std::vector<std::string> why(pieces);  // the scene's error text of a failed piece, captured by the thread that traced it
  std::thread worker;
  nrt_scene *scene = s->scene;
  for (size_t k = 0; k <= pieces; k++) {
    if (k < pieces) convert(k * kPiece, std::min(n, (k + 1) * kPiece));
    if (worker.joinable()) worker.join();  // piece k-1 is traced
    if (k < pieces) {
      const size_t lo = k * kPiece, hi = std::min(n, (k + 1) * kPiece);
      worker = std::thread([&good, &why, &gpu, scene, k, lo, hi]() {
        good[k] = gpu(lo, hi) ? 1 : 0;
        if (!good[k]) why[k] = nrtSceneLastError(scene);  // (read here: the next piece's trace may overwrite it)
      });
    }
    if (k >= 1) {  // overlaps the trace of piece k
      if (!good[k - 1] && ok) {
        ok = false;
        report(s->device, RTC_UNKNOWN_ERROR, "rtcIntersect/rtcOccluded: %s", why[k - 1].c_str());
      }
      write_back((k - 1) * kPiece, std::min(n, k * kPiece), good[k - 1] != 0);
    }
  }
}

}  // namespace

// ---- devices ----------------------------------------------------------------------------------------------------

RTCORE_API RTCDevice rtcNewDevice(const char *cfg) {
  Device *d = new Device;
  if (cfg) {
    const char *p = strstr(cfg, "device=");
    if (p) d->hip_device = atoi(p + 7);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  devices()[d] = d;
  return reinterpret_cast<RTCDevice>(d);
}

RTCORE_API void rtcDeleteDevice(RTCDevice device) {
  Device *d = reinterpret_cast<Device *>(device);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = devices().find(d);
    if (it == devices().end()) {
      char buf[64];
      snprintf(buf, sizeof(buf), "Invalid device : %p", (void *)device);
      g_error = buf;  // :479-483
      return;
    }
    devices().erase(it);
  }
  for (auto &kv : d->scenes) delete kv.second;  // scenes die with their device, as in the reference (:366-378)
  delete d;
}

RTCORE_API void rtcDeviceSetErrorFunction2(RTCDevice device, RTCErrorFunc2 func, void *userPtr) {
  Device *d = reinterpret_cast<Device *>(device);
  if (!d) return;
  std::lock_guard<std::mutex> lk(d->mu);
  d->error_func = func;
  d->user_ptr = userPtr;
}

RTCORE_API RTCError rtcDeviceGetError(RTCDevice device) {
  Device *d = reinterpret_cast<Device *>(device);
  if (!d) return RTC_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(d->mu);
  const RTCError e = d->error;
  d->error = RTC_NO_ERROR;
  d->message.clear();
  return e;
}

// ---- scenes -----------------------------------------------------------------------------------------------------

RTCORE_API RTCScene rtcDeviceNewScene(RTCDevice device, RTCSceneFlags, RTCAlgorithmFlags) {
  Device *d = reinterpret_cast<Device *>(device);
  if (!d) return nullptr;
  Scene *s = new Scene;  // the flags are hints to Embree's builders; the reference ignores them too (:236-241)
  s->device = d;
  std::lock_guard<std::mutex> lk(d->mu);
  d->scenes[s] = s;
  return reinterpret_cast<RTCScene>(s);
}

RTCORE_API void rtcDeleteScene(RTCScene scene) {
  Scene *s = as_scene(scene);
  Device *owner = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : devices()) {  // the reference searches every device for the scene (:404-417)
      std::lock_guard<std::mutex> lk2(kv.second->mu);
      auto it = kv.second->scenes.find(s);
      if (it != kv.second->scenes.end()) {
        kv.second->scenes.erase(it);
        owner = kv.second;
        break;
      }
    }
    if (!owner) {
      char buf[64];
      snprintf(buf, sizeof(buf), "Invalid scene : %p", (void *)scene);
      g_error = buf;  // :465-469
      return;
    }
  }
  delete s;
}

RTCORE_API void rtcCommit(RTCScene scene) {
  Scene *s = as_scene(scene);
  if (!s) return;
  std::lock_guard<std::mutex> lk(s->mu);
  commit_locked(s);
}

RTCORE_API void rtcGetBounds(RTCScene scene, RTCBounds &b) {
  Scene *s = as_scene(scene);
  // an uncommitted / failed scene reports nanosg's "invalid" box (nanosg.h:745-753)
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  if (s) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->committed && nrtSceneBounds_f32(s->scene, lo, hi) != NRT_OK)
      report(s->device, RTC_UNKNOWN_ERROR, "rtcGetBounds: %s", nrtSceneLastError(s->scene));
  }
  b.lower_x = lo[0];
  b.lower_y = lo[1];
  b.lower_z = lo[2];
  b.upper_x = hi[0];
  b.upper_y = hi[1];
  b.upper_z = hi[2];
}

// ---- triangle meshes --------------------------------------------------------------------------------------------

RTCORE_API unsigned rtcNewTriangleMesh(RTCScene scene, RTCGeometryFlags, size_t numTriangles, size_t numVertices, size_t numTimeSteps) {
  Scene *s = as_scene(scene);
  if (!s) return 0;
  // the reference's three argument checks (:567-588); 0 is the "no geometry" id
  if (numTimeSteps != 1) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcNewTriMesh] Motion blur is not supported. numTimeSteps : %zu", numTimeSteps);
    return 0;
  }
  if (numTriangles < 1) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcNewTriMesh] Invalid numTriangles : %zu", numTriangles);
    return 0;
  }
  if (numVertices < 1) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcNewTriMesh] Invalid numVertices : %zu", numVertices);
    return 0;
  }
  if (numTriangles > 0xFFFFFFFFull / 3) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcNewTriMesh] numTriangles : %zu exceeds the 32-bit primitive ids of nanort", numTriangles);
    return 0;
  }
  Mesh *m = new Mesh;
  m->vertices.assign(numVertices * 4, 0.0f);
  m->faces.assign(numTriangles * 3, 0u);
  std::lock_guard<std::mutex> lk(s->mu);
  const uint32_t id = s->next_id++;
  s->meshes[id] = m;
  return id;
}

RTCORE_API void *rtcMapBuffer(RTCScene scene, unsigned geomID, RTCBufferType type) {
  Scene *s = as_scene(scene);
  if (!s) return nullptr;
  if (type != RTC_VERTEX_BUFFER && type != RTC_INDEX_BUFFER) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcMapBuffer] Unsupported type : %d", (int)type);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(s->mu);
  Mesh *m = find_mesh(s, geomID);
  if (!m) {
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcMapBuffer] geomID : %u not found in the scene.", geomID);
    return nullptr;
  }
  m->dirty = true;  // the caller may write through the pointer: upload + rebuild at the next rtcCommit
  return type == RTC_VERTEX_BUFFER ? (void *)m->vertices.data() : (void *)m->faces.data();
}

RTCORE_API void rtcUnmapBuffer(RTCScene scene, unsigned, RTCBufferType type) {
  Scene *s = as_scene(scene);
  if (s && type != RTC_VERTEX_BUFFER && type != RTC_INDEX_BUFFER)
    report(s->device, RTC_INVALID_ARGUMENT, "[rtcUnmapBuffer] Unsupported type : %d", (int)type);
}

RTCORE_API void rtcUpdate(RTCScene scene, unsigned geomID) {
  Scene *s = as_scene(scene);
  if (!s) return;
  std::lock_guard<std::mutex> lk(s->mu);
  Mesh *m = find_mesh(s, geomID);
  if (m) m->dirty = true;
}

RTCORE_API unsigned rtcNewInstance2(RTCScene target, RTCScene, size_t numTimeSteps) {
  Scene *s = as_scene(target);
  if (numTimeSteps != 1)
    report(s ? s->device : nullptr, RTC_INVALID_ARGUMENT, "[rtcNewInstance2] numTimeSteps must be 1");
  else
    report(s ? s->device : nullptr, RTC_INVALID_OPERATION, "[rtcNewInstance2] instancing is not implemented (nor in the reference shim)");
  return 0;
}

RTCORE_API void rtcSetTransform2(RTCScene, unsigned, RTCMatrixType, const float *, size_t) {}

// ---- queries ----------------------------------------------------------------------------------------------------

RTCORE_API void rtcIntersect(RTCScene scene, RTCRay &ray) {
  RTCRay *p = &ray;
  trace(as_scene(scene), 1, false, [p](size_t) { return p; });
}

RTCORE_API void rtcIntersect1M(RTCScene scene, const RTCIntersectContext *, RTCRay *rays, const size_t M, const size_t stride) {
  trace(as_scene(scene), M, false, [rays, stride](size_t i) { return ray_at(rays, i, stride); });
}

RTCORE_API void rtcIntersect1Mp(RTCScene scene, const RTCIntersectContext *, RTCRay **rays, const size_t M) {
  trace(as_scene(scene), M, false, [rays](size_t i) { return rays[i]; });
}

RTCORE_API void rtcOccluded(RTCScene scene, RTCRay &ray) {
  RTCRay *p = &ray;
  trace(as_scene(scene), 1, true, [p](size_t) { return p; });
}

RTCORE_API void rtcOccluded1M(RTCScene scene, const RTCIntersectContext *, RTCRay *rays, const size_t M, const size_t stride) {
  trace(as_scene(scene), M, true, [rays, stride](size_t i) { return ray_at(rays, i, stride); });
}

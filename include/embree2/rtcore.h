/* include/embree2/rtcore.h — the part of the Embree 2.x C API that libnanort_embree.so implements on the
 * MI355X: what the reference's own shim provides (examples/embree-api/nanort-embree.cc:454-693 — devices,
 * scenes, triangle meshes with mapped buffers, rtcCommit, rtcGetBounds, rtcIntersect) plus the ray-STREAM
 * entry points that shim lists as TODO (README "Ray stream API") and a GPU needs: rtcIntersect1M,
 * rtcIntersect1Mp, rtcOccluded, rtcOccluded1M.
 *
 * One self-contained header (Embree splits it into rtcore_scene.h / rtcore_geometry.h; an application only
 * ever includes <embree2/rtcore.h> and <embree2/rtcore_ray.h>).  Enumerator values and struct layouts are
 * Embree 2.17's ABI; the declarations are written from that ABI so that code compiled against the real
 * Embree 2 headers links against this library unchanged.  Anything not declared here is not implemented —
 * a program using it fails at compile or link time, never silently.
 *
 * Semantics are the REFERENCE SHIM's, not Embree's, wherever the two differ:
 *  - rtcNewTriangleMesh returns ids 1, 2, 3, …, but a hit reports geomID = the mesh's 0-based position in id
 *    order (nanort-embree.cc:545 stores nanosg's node index; the reference demo indexes meshes[ray.geomID]);
 *  - a miss sets geomID = primID = instID = RTC_INVALID_GEOMETRY_ID and leaves tfar alone; Ng is never written;
 *  - no back-face culling, ray masks and time are ignored; the ray's [tnear, tfar] selects candidate meshes by
 *    their boxes, the per-mesh search itself is unbounded (nanosg.h:817, see nanort_hip.h "two-level scenes"). */
#ifndef NANORT_EMBREE2_RTCORE_H_
#define NANORT_EMBREE2_RTCORE_H_

#include <stddef.h>
#include <sys/types.h>

#ifndef RTCORE_API
#define RTCORE_API extern "C" __attribute__((visibility("default")))
#endif

#define RTC_INVALID_GEOMETRY_ID ((unsigned)-1)

struct __attribute__((aligned(16))) RTCBounds {
  float lower_x, lower_y, lower_z, align0;
  float upper_x, upper_y, upper_z, align1;
};

typedef struct __RTCDevice {} *RTCDevice;
typedef struct __RTCScene {} *RTCScene;

enum RTCError {
  RTC_NO_ERROR = 0,
  RTC_UNKNOWN_ERROR = 1,
  RTC_INVALID_ARGUMENT = 2,
  RTC_INVALID_OPERATION = 3,
  RTC_OUT_OF_MEMORY = 4,
  RTC_UNSUPPORTED_CPU = 5,
  RTC_CANCELLED = 6
};
typedef void (*RTCErrorFunc2)(void *userPtr, const RTCError code, const char *str);

enum RTCSceneFlags {
  RTC_SCENE_STATIC = 0,
  RTC_SCENE_DYNAMIC = 1 << 0,
  RTC_SCENE_COMPACT = 1 << 8,
  RTC_SCENE_COHERENT = 1 << 9,
  RTC_SCENE_INCOHERENT = 1 << 10,
  RTC_SCENE_HIGH_QUALITY = 1 << 11,
  RTC_SCENE_ROBUST = 1 << 16
};
enum RTCAlgorithmFlags {
  RTC_INTERSECT1 = 1 << 0,
  RTC_INTERSECT4 = 1 << 1,
  RTC_INTERSECT8 = 1 << 2,
  RTC_INTERSECT16 = 1 << 3,
  RTC_INTERPOLATE = 1 << 4,
  RTC_INTERSECT_STREAM = 1 << 5
};
/* flag sets are combined with | in application code (the reference demo: RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT) */
inline RTCSceneFlags operator|(RTCSceneFlags a, RTCSceneFlags b) { return (RTCSceneFlags)((unsigned)a | (unsigned)b); }
inline RTCAlgorithmFlags operator|(RTCAlgorithmFlags a, RTCAlgorithmFlags b) { return (RTCAlgorithmFlags)((unsigned)a | (unsigned)b); }

enum RTCIntersectFlags { RTC_INTERSECT_COHERENT = 0, RTC_INTERSECT_INCOHERENT = 1 };
struct RTCIntersectContext {
  RTCIntersectFlags flags; /* a hint; ignored */
  void *userRayExt;        /* ignored (no callbacks) */
};

enum RTCBufferType { RTC_INDEX_BUFFER = 0x01000000, RTC_VERTEX_BUFFER = 0x02000000 };
enum RTCGeometryFlags { RTC_GEOMETRY_STATIC = 0, RTC_GEOMETRY_DEFORMABLE = 1, RTC_GEOMETRY_DYNAMIC = 2 };
enum RTCMatrixType { RTC_MATRIX_ROW_MAJOR = 0, RTC_MATRIX_COLUMN_MAJOR = 1, RTC_MATRIX_COLUMN_MAJOR_ALIGNED16 = 2 };

struct RTCRay;

/* ---- devices (nanort-embree.cc:454-493) ---- */
RTCORE_API RTCDevice rtcNewDevice(const char *cfg = NULL); /* cfg: "device=N" selects the HIP device (default 0) */
RTCORE_API void rtcDeleteDevice(RTCDevice device);
RTCORE_API void rtcDeviceSetErrorFunction2(RTCDevice device, RTCErrorFunc2 func, void *userPtr);
RTCORE_API RTCError rtcDeviceGetError(RTCDevice device); /* returns and clears the first error recorded */

/* ---- scenes (:495-513, :688-693) ---- */
RTCORE_API RTCScene rtcDeviceNewScene(RTCDevice device, RTCSceneFlags flags, RTCAlgorithmFlags aflags);
RTCORE_API void rtcDeleteScene(RTCScene scene);
RTCORE_API void rtcCommit(RTCScene scene); /* builds one BVH per mesh on the GPU + the two-level scene */
RTCORE_API void rtcGetBounds(RTCScene scene, RTCBounds &bounds_o);

/* ---- triangle meshes (:560-646): 16-byte vertex stride, 3 x 32-bit indices per triangle ---- */
RTCORE_API unsigned rtcNewTriangleMesh(RTCScene scene, RTCGeometryFlags flags, size_t numTriangles, size_t numVertices,
                                       size_t numTimeSteps = 1);
RTCORE_API void *rtcMapBuffer(RTCScene scene, unsigned geomID, RTCBufferType type);
RTCORE_API void rtcUnmapBuffer(RTCScene scene, unsigned geomID, RTCBufferType type);
RTCORE_API void rtcUpdate(RTCScene scene, unsigned geomID); /* marks the mesh for re-upload + rebuild at the next rtcCommit */
/* Declared because the reference shim exports them; like there (:648-680) instancing is not implemented:
 * rtcNewInstance2 returns 0 and records an error, rtcSetTransform2 does nothing. */
RTCORE_API unsigned rtcNewInstance2(RTCScene target, RTCScene source, size_t numTimeSteps = 1);
RTCORE_API void rtcSetTransform2(RTCScene scene, unsigned geomID, RTCMatrixType layout, const float *xfm, size_t timeStep = 0);

/* ---- queries ---- */
/* One ray = one GPU round trip (tens of microseconds): kept for compatibility (:515-558), use the streams. */
RTCORE_API void rtcIntersect(RTCScene scene, RTCRay &ray);
/* M rays, `stride` bytes apart, in ONE batched two-level traversal; same per-ray result as rtcIntersect. */
RTCORE_API void rtcIntersect1M(RTCScene scene, const RTCIntersectContext *context, RTCRay *rays, const size_t M, const size_t stride);
RTCORE_API void rtcIntersect1Mp(RTCScene scene, const RTCIntersectContext *context, RTCRay **rays, const size_t M);
/* Occlusion: geomID = 0 if anything is hit, untouched otherwise (Embree's convention); nothing else is written. */
RTCORE_API void rtcOccluded(RTCScene scene, RTCRay &ray);
RTCORE_API void rtcOccluded1M(RTCScene scene, const RTCIntersectContext *context, RTCRay *rays, const size_t M, const size_t stride);

#endif /* NANORT_EMBREE2_RTCORE_H_ */

// tests/cpp/embree_check.cc — drives an Embree-2 API implementation the way the reference's demo does
// (examples/embree-api/main.cc:92-124, 167-196: rtcNewDevice, rtcDeviceNewScene, rtcNewTriangleMesh, rtcMapBuffer with
// the 16-byte vertex stride, rtcCommit, rtcGetBounds, rtcIntersect) and dumps what comes back, byte for byte.
//
//   embree_check SCENE RAYS OUT MODE      MODE: single | stream | streamp | occluded | recommit
//
// SCENE: u32 count, then per mesh {u32 nv, u32 nf, float xyz[3 nv], u32 ijk[3 nf]};
// RAYS:  u64 n, then per ray 8 floats {org[3], dir[3], tnear, tfar};
// OUT:   RTCBounds (32 B), u32 ids[count] (what rtcNewTriangleMesh returned), RTCRay[n] (96 B each).
//
// Every RTCRay starts out filled with 0xA5 bytes, so fields the implementation must leave alone show up if it writes them.
// Built against include/embree2 + libnanort_embree.so in the GPU tests and compared with tests/golden/embree_ref.npz
// (oracle/gen_golden_embree.py: the unmodified nanosg + the reference shim's field mapping; the reference's shim itself
// does not compile at this revision, see oracle/Makefile).  tests/test_embree_api.py also type-checks this file against
// the Embree 2.17 headers the reference vendors, so the source is valid for a real Embree application too.
#include <embree2/rtcore.h>
#include <embree2/rtcore_ray.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

static bool read_all(const char *path, std::vector<unsigned char> *out) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize((size_t)n);
  const bool ok = fread(out->data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

static void on_error(void *, const RTCError code, const char *str) { fprintf(stderr, "embree_check: RTC error %d: %s\n", (int)code, str); }

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: embree_check SCENE RAYS OUT MODE\n");
    return 2;
  }
  const std::string mode = argv[4];
  std::vector<unsigned char> sb, rb;
  if (!read_all(argv[1], &sb) || !read_all(argv[2], &rb)) {
    fprintf(stderr, "embree_check: cannot read the inputs\n");
    return 2;
  }

  RTCDevice device = rtcNewDevice(NULL);
  rtcDeviceSetErrorFunction2(device, on_error, NULL);
  RTCScene scene = rtcDeviceNewScene(device, RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT, RTC_INTERSECT1);

  const unsigned char *p = sb.data();
  uint32_t count;
  memcpy(&count, p, 4);
  p += 4;
  std::vector<uint32_t> ids;
  for (uint32_t m = 0; m < count; m++) {
    uint32_t nv, nf;
    memcpy(&nv, p, 4);
    memcpy(&nf, p + 4, 4);
    p += 8;
    const float *xyz = reinterpret_cast<const float *>(p);
    p += 12ull * nv;
    const uint32_t *ijk = reinterpret_cast<const uint32_t *>(p);
    p += 12ull * nf;
    const unsigned id = rtcNewTriangleMesh(scene, RTC_GEOMETRY_STATIC, nf, nv, 1);
    ids.push_back(id);
    float *v = reinterpret_cast<float *>(rtcMapBuffer(scene, id, RTC_VERTEX_BUFFER));
    int *f = reinterpret_cast<int *>(rtcMapBuffer(scene, id, RTC_INDEX_BUFFER));
    if (!v || !f) {
      fprintf(stderr, "embree_check: rtcMapBuffer failed for mesh %u\n", m);
      return 1;
    }
    for (uint32_t i = 0; i < nv; i++) {
      v[4 * i + 0] = xyz[3 * i + 0];
      v[4 * i + 1] = xyz[3 * i + 1];
      v[4 * i + 2] = xyz[3 * i + 2];
      v[4 * i + 3] = 0.0f;
    }
    for (uint32_t i = 0; i < 3 * nf; i++) f[i] = (int)ijk[i];
    rtcUnmapBuffer(scene, id, RTC_VERTEX_BUFFER);
    rtcUnmapBuffer(scene, id, RTC_INDEX_BUFFER);
  }
  rtcCommit(scene);
  if (mode == "recommit") {
    // move mesh 0 away and back through the mapped buffer, committing in between: the final state is the original scene
    float *v = reinterpret_cast<float *>(rtcMapBuffer(scene, ids[0], RTC_VERTEX_BUFFER));
    uint32_t nv0;
    memcpy(&nv0, sb.data() + 4, 4);
    for (uint32_t i = 0; i < nv0; i++) v[4 * i + 1] += 1000.0f;
    rtcUnmapBuffer(scene, ids[0], RTC_VERTEX_BUFFER);
    rtcUpdate(scene, ids[0]);
    rtcCommit(scene);
    const float *xyz = reinterpret_cast<const float *>(sb.data() + 12);
    v = reinterpret_cast<float *>(rtcMapBuffer(scene, ids[0], RTC_VERTEX_BUFFER));
    for (uint32_t i = 0; i < nv0; i++) v[4 * i + 1] = xyz[3 * i + 1];
    rtcUnmapBuffer(scene, ids[0], RTC_VERTEX_BUFFER);
    rtcUpdate(scene, ids[0]);
    rtcCommit(scene);
  }
  RTCBounds bounds;
  memset(&bounds, 0, sizeof(bounds));
  rtcGetBounds(scene, bounds);
  bounds.align0 = bounds.align1 = 0.0f;

  uint64_t n;
  memcpy(&n, rb.data(), 8);
  const float *rf = reinterpret_cast<const float *>(rb.data() + 8);
  RTCRay *rays = NULL;
  if (posix_memalign(reinterpret_cast<void **>(&rays), 64, sizeof(RTCRay) * (n ? n : 1)) != 0) return 1;
  memset(rays, 0xA5, sizeof(RTCRay) * n);
  for (uint64_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      rays[i].org[k] = rf[8 * i + k];
      rays[i].dir[k] = rf[8 * i + 3 + k];
    }
    rays[i].tnear = rf[8 * i + 6];
    rays[i].tfar = rf[8 * i + 7];
  }

  // EMBREE_CHECK_TIMING=k: run the query k times on fresh copies of the rays and print the wall time of each pass (the
  // first includes the library's lazy device allocations).
  const int reps = getenv("EMBREE_CHECK_TIMING") ? (atoi(getenv("EMBREE_CHECK_TIMING")) > 0 ? atoi(getenv("EMBREE_CHECK_TIMING")) : 1) : 1;
  std::vector<unsigned char> pristine(reinterpret_cast<unsigned char *>(rays), reinterpret_cast<unsigned char *>(rays) + sizeof(RTCRay) * n);
  for (int rep = 0; rep < reps; rep++) {
    memcpy(rays, pristine.data(), pristine.size());
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    if (mode == "single") {
      for (uint64_t i = 0; i < n; i++) rtcIntersect(scene, rays[i]);
    }
    else if (mode == "stream" || mode == "recommit") {
      RTCIntersectContext ctx;
      ctx.flags = RTC_INTERSECT_INCOHERENT;
      ctx.userRayExt = NULL;
      rtcIntersect1M(scene, &ctx, rays, n, sizeof(RTCRay));
    } else if (mode == "streamp") {
      std::vector<RTCRay *> ptrs(n);
      for (uint64_t i = 0; i < n; i++) ptrs[i] = &rays[n - 1 - i];  // any order
      RTCIntersectContext ctx;
      ctx.flags = RTC_INTERSECT_COHERENT;
      ctx.userRayExt = NULL;
      rtcIntersect1Mp(scene, &ctx, ptrs.data(), n);
    } else if (mode == "occluded") {
      RTCIntersectContext ctx;
      ctx.flags = RTC_INTERSECT_INCOHERENT;
      ctx.userRayExt = NULL;
      if (n > 0) rtcOccluded(scene, rays[0]);
      if (n > 1) rtcOccluded1M(scene, &ctx, rays + 1, n - 1, sizeof(RTCRay));
    }
    else {
      fprintf(stderr, "embree_check: unknown mode %s\n", mode.c_str());
      return 2;
    }

    if (getenv("EMBREE_CHECK_TIMING")) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("embree_check: mode %s, pass %d, %llu rays, %.3f ms, %.3f Mrays/s\n", mode.c_str(), rep, (unsigned long long)n, ms, n / ms * 1e-3);
    }
  }

  FILE *out = fopen(argv[3], "wb");
  if (!out) return 1;
  fwrite(&bounds, sizeof(bounds), 1, out);
  fwrite(ids.data(), 4, ids.size(), out);
  fwrite(rays, sizeof(RTCRay), n, out);
  fclose(out);
  free(rays);
  rtcDeleteScene(scene);
  rtcDeleteDevice(device);
  return 0;
}

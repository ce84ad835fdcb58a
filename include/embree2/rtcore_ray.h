/* include/embree2/rtcore_ray.h — the single-ray record of the Embree 2.x C API, as far as
 * libnanort_embree.so (nanort_amd/csrc/embree_api.cc) uses it.
 *
 * Written from the ABI, not from Embree's header text: field order, sizes and the 16-byte alignment are
 * what an application compiled against Embree 2.17 (the version the reference vendors under
 * examples/embree-api/include/embree2/) passes in, so such an application links against this library
 * without recompiling.  tests/test_embree_api.py checks every offset below against that header in the
 * build container.  Packet rays (RTCRay4/8/16) and RTCRayN are not declared: the library takes single
 * rays and STREAMS of single rays (rtcIntersect1M), which is the shape a GPU wants. */
#ifndef NANORT_EMBREE2_RTCORE_RAY_H_
#define NANORT_EMBREE2_RTCORE_RAY_H_

#ifndef __RTCRay__
#define __RTCRay__
struct __attribute__((aligned(16))) RTCRay {
  /* in */
  float org[3];
  float align0;
  float dir[3];
  float align1;
  float tnear;
  float tfar; /* in: end of the segment; out: hit distance when geomID != RTC_INVALID_GEOMETRY_ID */
  float time; /* ignored (no motion blur, as in the reference shim) */
  unsigned mask; /* ignored (as in the reference shim) */
  /* out */
  float Ng[3]; /* left untouched: the reference shim does not fill it (nanort-embree.cc:541-548) */
  float align2;
  float u;
  float v;
  unsigned geomID;
  unsigned primID;
  unsigned instID;
};
#endif

#endif /* NANORT_EMBREE2_RTCORE_RAY_H_ */

// examples/wavefront_path_tracer_gpu/main.hip — the wavefront path tracer of examples/wavefront_path_tracer with the
// SHADING on the GPU as well: ray generation, next-event estimation, cosine bounces and image accumulation are HIP
// kernels, every wave goes through BVHAccel::TraverseBatchDevice(), and nothing but the final image crosses PCIe.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -DNANORT_USE_HIP_BACKEND -I../../include main.hip \
//         -L../../nanort_amd/lib -lnanort_hip -Wl,-rpath,$PWD/../../nanort_amd/lib -o wavefront_gpu
//   ./wavefront_gpu [--size W H] [--spp N] [--depth D] [--grid NX NY] [--out image.f32] [--streams 0|1|2]
//
// Two streams (the default): the shadow query of depth d and the path wave of depth d + 1 do not depend on each other — both come
// out of k_shade(d) — so the shadow query and its resolve run on a second stream while the first goes on with the next wave.
// One BVHAccel serves both (every launch owns a launch slot of the context); the end of one launch — a handful of long rays —
// is filled by the other.  The shadow contributions then accumulate in an image of their own that is added at the end, so a
// pixel's sum is associated differently than with --streams 1 (last-bit differences).
//
// Same scene, camera, light, sampler and per-pixel accumulation order as the host-shaded example (a pixel's path is a
// pure function of (pixel, sample)); waves keep one slot per pixel — a dead path's slot holds a ray that cannot hit
// (min_t > max_t), so no compaction is needed and the image does not depend on scheduling.  The images agree with the
// host-shaded ones up to the difference between the device's and glibc's sinf/cosf (tests allow 2e-3 absolute).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "nanort.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                                   \
    }                                                                                            \
  } while (0)

// Device code uses the C ABI's PODs (layout-identical to nanort::Ray<float> / TriangleIntersection<float>, whose
// constructors are host functions).
typedef nrt_ray_f32 Ray;
typedef nrt_hit_f32 Hit;
static_assert(sizeof(Ray) == sizeof(nanort::Ray<float>) && sizeof(Hit) == sizeof(nanort::TriangleIntersection<float>), "wire layouts");
static bool Trace(const nanort::BVHAccel<float> &accel, const Ray *d_rays, size_t n, Hit *d_hits, unsigned char *d_mask, hipStream_t s) {
  return accel.TraverseBatchDevice(reinterpret_cast<const nanort::Ray<float> *>(d_rays), n,
                                   reinterpret_cast<nanort::TriangleIntersection<float> *>(d_hits), d_mask, s);
}

struct V3 {
  float x, y, z;
};
__host__ __device__ inline V3 v3(float x, float y, float z) {
  V3 r = {x, y, z};
  return r;
}
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__host__ __device__ inline V3 normalize(V3 a) {  // nanort::vnormalize
  const float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  if (fabsf(len) > 1.1920928955078125e-07f) {
    const float inv = 1.0f / len;
    return v3(a.x * inv, a.y * inv, a.z * inv);
  }
  return a;
}

__host__ __device__ inline uint32_t pcg_hash(uint32_t v) {
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
__device__ inline float rnd01(uint32_t &s) {
  s = pcg_hash(s);
  return (float)(s >> 8) / 16777216.0f;
}

struct PathState {  // one slot per pixel
  uint32_t rng;
  float tx, ty, tz;  // throughput
  uint32_t alive;
};

__device__ inline void dead_ray(Ray *r) {  // cannot pass the root's slab test
  r->org[0] = r->org[1] = r->org[2] = 0.0f;
  r->dir[0] = r->dir[1] = 0.0f;
  r->dir[2] = 1.0f;
  r->min_t = 1.0f;
  r->max_t = 0.0f;
}

__global__ void k_camera(int W, int H, int sample, Ray *rays, PathState *paths) {
  const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (uint32_t)(W * H)) return;
  const int x = pix % W, y = pix / W;
  PathState p;
  p.rng = pcg_hash(pix * 9781u + (uint32_t)sample * 6271u + 17u);
  p.tx = p.ty = p.tz = 1.0f;
  p.alive = 1;
  const float jx = rnd01(p.rng), jy = rnd01(p.rng);
  const V3 d = normalize(v3(((float)x + jx) / (float)W - 0.5f, ((float)y + jy) / (float)H - 0.5f, -1.0f));
  Ray r;
  r.org[0] = 0.0f;
  r.org[1] = 5.0f;
  r.org[2] = 20.0f;
  r.dir[0] = d.x;
  r.dir[1] = d.y;
  r.dir[2] = d.z;
  r.min_t = 0.001f;
  r.max_t = 1.0e30f;
  r.type = 0;
  rays[pix] = r;
  paths[pix] = p;
}

// Shade one wave in place: sky on a miss, a shadow ray + its pending contribution on a hit, and (below the depth limit)
// the bounce ray that replaces this slot's ray.
__global__ void k_shade(int n, int spp, int depth, int max_depth, const float *verts, const unsigned int *faces, Ray *rays,
                        const Hit *hits, const unsigned char *mask, PathState *paths, Ray *shadow_rays, float *shadow_contrib,
                        float *image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)n) return;
  PathState p = paths[i];
  Ray sr;
  dead_ray(&sr);
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  if (p.alive) {
    const V3 light = v3(8.0f, 12.0f, 15.0f), albedo = v3(0.75f, 0.7f, 0.6f), sky = v3(0.4f, 0.5f, 0.7f);
    const float light_power = 600.0f;
    float *px = image + 3 * (size_t)i;
    if (!mask[i]) {
      px[0] += p.tx * sky.x / (float)spp;
      px[1] += p.ty * sky.y / (float)spp;
      px[2] += p.tz * sky.z / (float)spp;
      p.alive = 0;
      dead_ray(&rays[i]);
    } else {
      const Ray r = rays[i];
      const Hit h = hits[i];
      const V3 org = v3(r.org[0], r.org[1], r.org[2]), dir = v3(r.dir[0], r.dir[1], r.dir[2]);
      const V3 P = org + dir * h.t;
      const unsigned int f0 = faces[3 * h.prim_id], f1 = faces[3 * h.prim_id + 1], f2 = faces[3 * h.prim_id + 2];
      const V3 p0 = v3(verts[3 * f0], verts[3 * f0 + 1], verts[3 * f0 + 2]), p1 = v3(verts[3 * f1], verts[3 * f1 + 1], verts[3 * f1 + 2]),
               p2 = v3(verts[3 * f2], verts[3 * f2 + 1], verts[3 * f2 + 2]);
      V3 N = normalize(cross(p1 - p0, p2 - p0));
      if (dot(N, dir) > 0.0f) N = v3(-N.x, -N.y, -N.z);
      const V3 toL = light - P;
      const float dist = sqrtf(dot(toL, toL));
      const V3 wl = toL * (1.0f / dist);
      const float cosl = dot(N, wl);
      if (cosl > 0.0f) {
        sr.org[0] = P.x;
        sr.org[1] = P.y;
        sr.org[2] = P.z;
        sr.dir[0] = wl.x;
        sr.dir[1] = wl.y;
        sr.dir[2] = wl.z;
        sr.min_t = 1.0e-3f;
        sr.max_t = dist - 1.0e-3f;
        const float g = cosl / (dist * dist) * (1.0f / 3.14159265f);
        c0 = p.tx * albedo.x * light_power * g;
        c1 = p.ty * albedo.y * light_power * g;
        c2 = p.tz * albedo.z * light_power * g;
      }
      if (depth < max_depth) {
        const float u1 = rnd01(p.rng), phi = 6.28318530718f * rnd01(p.rng);
        const float rr = sqrtf(u1);
        V3 b1, b2;  // Building an Orthonormal Basis, Revisited
        if (N.z < 0.0f) {
          const float a = 1.0f / (1.0f - N.z), b = N.x * N.y * a;
          b1 = v3(1.0f - N.x * N.x * a, -b, N.x);
          b2 = v3(b, N.y * N.y * a - 1.0f, -N.y);
        } else {
          const float a = 1.0f / (1.0f + N.z), b = -N.x * N.y * a;
          b1 = v3(1.0f - N.x * N.x * a, b, -N.x);
          b2 = v3(b, 1.0f - N.y * N.y * a, -N.y);
        }
        const V3 wi = normalize(b1 * (rr * cosf(phi)) + b2 * (rr * sinf(phi)) + N * sqrtf(1.0f - u1));
        Ray br;
        br.org[0] = P.x;
        br.org[1] = P.y;
        br.org[2] = P.z;
        br.dir[0] = wi.x;
        br.dir[1] = wi.y;
        br.dir[2] = wi.z;
        br.min_t = 1.0e-3f;
        br.max_t = 1.0e30f;
        br.type = 0;
        rays[i] = br;
        p.tx *= albedo.x;
        p.ty *= albedo.y;
        p.tz *= albedo.z;
      } else {
        p.alive = 0;
        dead_ray(&rays[i]);
      }
    }
    paths[i] = p;
  }
  sr.type = 0;
  shadow_rays[i] = sr;
  shadow_contrib[3 * (size_t)i + 0] = c0;
  shadow_contrib[3 * (size_t)i + 1] = c1;
  shadow_contrib[3 * (size_t)i + 2] = c2;
}

__global__ void k_add_images(int n3, float *image, const float *other) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (uint32_t)n3) image[i] += other[i];
}

__global__ void k_resolve_shadows(int n, int spp, const unsigned char *shadow_mask, const float *shadow_contrib, float *image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)n || shadow_mask[i]) return;  // occluded (a dead shadow ray never hits and carries 0)
  for (int k = 0; k < 3; k++) image[3 * (size_t)i + k] += shadow_contrib[3 * (size_t)i + k] / (float)spp;
}

static void MakeGrid(std::vector<float> *vertices, std::vector<unsigned int> *faces, int nx, int ny) {
  vertices->resize(3 * (size_t)(nx + 1) * (ny + 1));
  faces->resize(3 * 2 * (size_t)nx * ny);
  for (int j = 0; j <= ny; j++)
    for (int i = 0; i <= nx; i++) {
      const float x = -10.0f + 20.0f * (float)i / (float)nx, y = -5.0f + 20.0f * (float)j / (float)ny;
      const size_t v = (size_t)j * (nx + 1) + i;
      (*vertices)[3 * v + 0] = x;
      (*vertices)[3 * v + 1] = y;
      (*vertices)[3 * v + 2] = 0.8f * sinf(0.9f * x) * cosf(1.1f * y);
    }
  size_t f = 0;
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const unsigned int a = (unsigned int)(j * (nx + 1) + i), b = a + 1, c = a + (unsigned int)(nx + 1), d = c + 1;
      const unsigned int t[6] = {a, b, d, a, d, c};
      for (int k = 0; k < 6; k++) (*faces)[f++] = t[k];
    }
}

int main(int argc, char **argv) {
  int W = 480, H = 270, spp = 2, depth = 3, nx = 400, ny = 200, streams = 0;
  std::string out;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--size") && i + 2 < argc) {
      W = atoi(argv[++i]);
      H = atoi(argv[++i]);
    } else if (!strcmp(argv[i], "--spp") && i + 1 < argc) {
      spp = atoi(argv[++i]);
    } else if (!strcmp(argv[i], "--depth") && i + 1 < argc) {
      depth = atoi(argv[++i]);
    } else if (!strcmp(argv[i], "--grid") && i + 2 < argc) {
      nx = atoi(argv[++i]);
      ny = atoi(argv[++i]);
    } else if (!strcmp(argv[i], "--out") && i + 1 < argc) {
      out = argv[++i];
    } else if (!strcmp(argv[i], "--streams") && i + 1 < argc) {
      streams = atoi(argv[++i]);
      if (streams < 0 || streams > 2) streams = 0;
    }
  }
  std::vector<float> vertices;
  std::vector<unsigned int> faces;
  MakeGrid(&vertices, &faces, nx, ny);
  nanort::TriangleMesh<float> mesh(vertices.data(), faces.data(), sizeof(float) * 3);
  nanort::TriangleSAHPred<float> pred(vertices.data(), faces.data(), sizeof(float) * 3);
  nanort::BVHAccel<float> accel;
  if (!accel.Build((unsigned int)(faces.size() / 3), mesh, pred)) {
    fprintf(stderr, "Build failed: %s\n", accel.LastBackendError().c_str());
    return 1;
  }
  const int n = W * H;
  float *d_verts, *d_contrib[2], *d_image, *d_image2;
  unsigned int *d_faces;
  Ray *d_rays, *d_shadow[2];
  Hit *d_hits;
  unsigned char *d_mask, *d_smask[2];
  PathState *d_paths;
  hipStream_t stream, stream2;
  hipEvent_t ev_shaded[2], ev_resolved[2]; // per shadow buffer: k_shade has filled it / its query has been resolved
  CHECK(hipStreamCreate(&stream));
  CHECK(hipStreamCreate(&stream2));
  for (int b = 0; b < 2; b++) {
    CHECK(hipEventCreateWithFlags(&ev_shaded[b], hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&ev_resolved[b], hipEventDisableTiming));
  }
  CHECK(hipMalloc(&d_verts, vertices.size() * 4));
  CHECK(hipMalloc(&d_faces, faces.size() * 4));
  CHECK(hipMalloc(&d_rays, (size_t)n * sizeof(Ray)));
  for (int b = 0; b < 2; b++) { // shadow rays, their flags and pending contributions: double-buffered across depths
    CHECK(hipMalloc(&d_shadow[b], (size_t)n * sizeof(Ray)));
    CHECK(hipMalloc(&d_smask[b], n));
    CHECK(hipMalloc(&d_contrib[b], (size_t)n * 12));
  }
  CHECK(hipMalloc(&d_hits, (size_t)n * sizeof(Hit)));
  CHECK(hipMalloc(&d_mask, n));
  CHECK(hipMalloc(&d_paths, (size_t)n * sizeof(PathState)));
  CHECK(hipMalloc(&d_image, (size_t)n * 12));
  CHECK(hipMalloc(&d_image2, (size_t)n * 12));
  CHECK(hipMemcpy(d_verts, vertices.data(), vertices.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_faces, faces.data(), faces.size() * 4, hipMemcpyHostToDevice));
  const dim3 grid((n + 255) / 256), block(256);
  uint64_t rays_traced = 0;
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {  // first repetition warms up; the image is the same every time
    CHECK(hipMemsetAsync(d_image, 0, (size_t)n * 12, stream));
    CHECK(hipMemsetAsync(d_image2, 0, (size_t)n * 12, stream));
    CHECK(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    rays_traced = 0;
    hipStream_t shadow_stream = streams == 2 ? stream2 : stream;
    float *shadow_image = streams == 2 ? d_image2 : d_image; // (one stream: the shadow terms join the pixel's sum in path order)
    int wave = 0; // shadow buffers alternate from wave to wave
    if (streams == 0) {
      // ONE stream, ONE traversal launch per depth (the default): the shadow query of depth d and the path wave of depth d + 1 both
      // come out of k_shade(d) and do not depend on each other — they go into one persistent launch as two waves
      // (BVHAccel::TraverseBatchesDevice), which has one tail where two launches have two.  Per pixel the terms are added in
      // the order of --streams 1 (shade(d), shadows(d), shade(d + 1), ...): the image is the same in every bit.
      // (the camera wave of sample s + 1 does not depend on sample s either: it rides with the last shadow query of sample s)
      hipLaunchKernelGGL(k_camera, grid, block, 0, stream, W, H, 0, d_rays, d_paths);
      if (!Trace(accel, d_rays, n, d_hits, d_mask, stream)) {
        fprintf(stderr, "TraverseBatchDevice: %s\n", accel.LastBackendError().c_str());
        return 1;
      }
      rays_traced += n;
      for (int s = 0; s < spp; s++) {
        for (int d = 0; d <= depth; d++, wave++) {
          const int b = wave & 1;
          hipLaunchKernelGGL(k_shade, grid, block, 0, stream, n, spp, d, depth, d_verts, d_faces, d_rays, d_hits, d_mask, d_paths, d_shadow[b],
                             d_contrib[b], d_image);
          const bool next_wave = d < depth || s + 1 < spp; // the next path wave of this sample, or the camera wave of the next one
          if (d == depth && s + 1 < spp) hipLaunchKernelGGL(k_camera, grid, block, 0, stream, W, H, s + 1, d_rays, d_paths);
          const nanort::Ray<float> *waves[2] = {reinterpret_cast<const nanort::Ray<float> *>(d_shadow[b]),
                                                reinterpret_cast<const nanort::Ray<float> *>(d_rays)};
          nanort::TriangleIntersection<float> *recs[2] = {NULL, reinterpret_cast<nanort::TriangleIntersection<float> *>(d_hits)};
          unsigned char *flags[2] = {d_smask[b], d_mask};
          const size_t counts[2] = {(size_t)n, (size_t)n};
          const unsigned char occlusion[2] = {1, 0};
          if (!accel.TraverseBatchesDevice(next_wave ? 2 : 1, waves, counts, recs, flags, occlusion, stream)) {
            fprintf(stderr, "TraverseBatchesDevice: %s\n", accel.LastBackendError().c_str());
            return 1;
          }
          hipLaunchKernelGGL(k_resolve_shadows, grid, block, 0, stream, n, spp, d_smask[b], d_contrib[b], d_image);
          rays_traced += next_wave ? 2ull * n : (uint64_t)n;
        }
      }
    } else
    for (int s = 0; s < spp; s++) {
      hipLaunchKernelGGL(k_camera, grid, block, 0, stream, W, H, s, d_rays, d_paths);
      for (int d = 0; d <= depth; d++, wave++) {
        const int b = wave & 1;
        if (!Trace(accel, d_rays, n, d_hits, d_mask, stream)) {
          fprintf(stderr, "TraverseBatchDevice: %s\n", accel.LastBackendError().c_str());
          return 1;
        }
        // (k_shade overwrites shadow buffer b: the query that read it two waves ago must have been resolved)
        if (streams == 2 && wave >= 2) CHECK(hipStreamWaitEvent(stream, ev_resolved[b], 0));
        hipLaunchKernelGGL(k_shade, grid, block, 0, stream, n, spp, d, depth, d_verts, d_faces, d_rays, d_hits, d_mask, d_paths, d_shadow[b],
                           d_contrib[b], d_image);
        if (streams == 2) {
          CHECK(hipEventRecord(ev_shaded[b], stream));
          CHECK(hipStreamWaitEvent(shadow_stream, ev_shaded[b], 0));
        }
        // shadow rays only ask "is anything in the way?": the opt-in occlusion query (same flags, early exit) — on the second
        // stream it overlaps the next path wave, which the first stream goes on with at once
        if (!accel.OccludedBatchDevice(reinterpret_cast<const nanort::Ray<float> *>(d_shadow[b]), n, d_smask[b], shadow_stream)) return 1;
        hipLaunchKernelGGL(k_resolve_shadows, grid, block, 0, shadow_stream, n, spp, d_smask[b], d_contrib[b], shadow_image);
        if (streams == 2) CHECK(hipEventRecord(ev_resolved[b], shadow_stream));
        rays_traced += 2ull * n;
      }
    }
    if (streams == 2) {
      CHECK(hipStreamSynchronize(stream2));
      hipLaunchKernelGGL(k_add_images, dim3((3 * n + 255) / 256), block, 0, stream, 3 * n, d_image, d_image2);
    }
    CHECK(hipStreamSynchronize(stream));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rep > 0 && secs < best) best = secs;
  }
  std::vector<float> image(3 * (size_t)n);
  CHECK(hipMemcpy(image.data(), d_image, (size_t)n * 12, hipMemcpyDeviceToHost));
  double sum = 0.0;
  for (size_t i = 0; i < image.size(); i++) sum += image[i];
  printf("triangles %zu image %dx%d spp %d depth %d ray_slots %llu frame_ms %.3f Mray_slots_per_s %.1f image_sum %.6f\n", faces.size() / 3, W, H,
         spp, depth, (unsigned long long)rays_traced, best * 1e3, (double)rays_traced / best / 1e6, sum);
  if (!out.empty()) {
    FILE *fp = fopen(out.c_str(), "wb");
    if (!fp) return 2;
    fwrite(image.data(), 4, image.size(), fp);  // raw float RGB, row-major
    fclose(fp);
  }
  return 0;
}

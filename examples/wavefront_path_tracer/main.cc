// examples/wavefront_path_tracer/main.cc — SURVEY.md §8(f) row 1.
//
// The reference's examples/path_tracer traces one synchronous ray per Traverse() call inside a
// per-pixel loop (examples/path_tracer/main.cc:804-992), which cannot feed a GPU.  This is the same
// kind of renderer (uni-directional path tracing of a diffuse triangle scene, next-event estimation
// with a closest-hit shadow query as in CheckForOccluder, main.cc:675-701, cosine-weighted bounces
// on the revisedONB basis, main.cc:214-250) restructured into RAY WAVES: every bounce builds one
// buffer of rays for all live paths, hands it to BVHAccel::TraverseBatch(), and shades the returned
// hit records on the host.  With -DNANORT_USE_HIP_BACKEND both Build() and TraverseBatch() run on the
// MI355X through the C ABI; without it the same waves are traced with the per-ray host Traverse()
// (OpenMP), which is also what --verify uses to check the GPU image bit for bit.
//
//   g++ -O2 -std=c++11 -fopenmp -DNANORT_USE_HIP_BACKEND -I../../include main.cc
//       -L../../nanort_amd/lib -lnanort_hip -Wl,-rpath,$PWD/../../nanort_amd/lib -L/opt/rocm/lib -lamdhip64
//   ./a.out [--size W H] [--spp N] [--depth D] [--grid NX NY] [--out image.ppm] [--raw image.f32] [--verify]
//
// Scene: a displaced grid (procedural, 2*NX*NY triangles) under a point light; per-pixel RNG is a
// counter-based hash, so the image is a pure function of the arguments (unlike the reference's
// rand()-driven example) and CPU/GPU runs can be compared exactly.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "nanort.h"

typedef nanort::real3<float> float3;

static uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
static float rnd01(uint32_t &s) {
  s = pcg_hash(s);
  return (float)(s >> 8) / 16777216.0f;
}

struct Scene {
  std::vector<float> vertices;
  std::vector<unsigned int> faces;
};

static void MakeGrid(Scene *sc, int nx, int ny) {
  sc->vertices.resize(3 * (size_t)(nx + 1) * (ny + 1));
  sc->faces.resize(3 * 2 * (size_t)nx * ny);
  for (int j = 0; j <= ny; j++)
    for (int i = 0; i <= nx; i++) {
      const float x = -10.0f + 20.0f * (float)i / (float)nx, y = -5.0f + 20.0f * (float)j / (float)ny;
      const size_t v = (size_t)j * (nx + 1) + i;
      sc->vertices[3 * v + 0] = x;
      sc->vertices[3 * v + 1] = y;
      sc->vertices[3 * v + 2] = 0.8f * sinf(0.9f * x) * cosf(1.1f * y);
    }
  size_t f = 0;
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const unsigned a = (unsigned)(j * (nx + 1) + i), b = a + 1, c = a + (unsigned)(nx + 1), d = c + 1;
      const unsigned tri[6] = {a, b, d, a, d, c};
      for (int k = 0; k < 6; k++) sc->faces[f++] = tri[k];
    }
}

static float3 FaceNormal(const Scene &sc, unsigned int prim) {
  const float3 p0(&sc.vertices[3 * sc.faces[3 * prim + 0]]), p1(&sc.vertices[3 * sc.faces[3 * prim + 1]]),
      p2(&sc.vertices[3 * sc.faces[3 * prim + 2]]);
  return nanort::vnormalize(nanort::vcross(p1 - p0, p2 - p0));
}

// Building an Orthonormal Basis, Revisited (as the reference's path tracer does).
static void Onb(const float3 &n, float3 *b1, float3 *b2) {
  if (n[2] < 0.0f) {
    const float a = 1.0f / (1.0f - n[2]), b = n[0] * n[1] * a;
    *b1 = float3(1.0f - n[0] * n[0] * a, -b, n[0]);
    *b2 = float3(b, n[1] * n[1] * a - 1.0f, -n[1]);
  } else {
    const float a = 1.0f / (1.0f + n[2]), b = -n[0] * n[1] * a;
    *b1 = float3(1.0f - n[0] * n[0] * a, b, -n[0]);
    *b2 = float3(b, 1.0f - n[1] * n[1] * a, -n[1]);
  }
}

struct Path {  // one live path of the current wave
  uint32_t pixel;
  uint32_t rng;
  float3 throughput;
};

typedef nanort::Ray<float> Ray;
typedef nanort::TriangleIntersection<float> Hit;

// One wave through the accelerator: GPU batch, or the per-ray host loop of the reference API.
static void TraceWave(const nanort::BVHAccel<float> &accel, const Scene &sc, const std::vector<Ray> &rays, bool use_batch,
                      std::vector<Hit> *hits, std::vector<unsigned char> *mask, double *secs, uint64_t *count) {
  const size_t n = rays.size();
  hits->resize(n);
  mask->assign(n, 0);
  if (n == 0) return;
  auto t0 = std::chrono::steady_clock::now();
#ifdef NANORT_USE_HIP_BACKEND
  if (use_batch) {
    if (!accel.TraverseBatch(rays.data(), n, hits->data(), mask->data())) {
      fprintf(stderr, "TraverseBatch failed: %s\n", accel.LastBackendError().c_str());
      exit(1);
    }
  } else
#endif
  {
    (void)use_batch;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (long long i = 0; i < (long long)n; i++) {
      nanort::TriangleIntersector<float> isector(sc.vertices.data(), sc.faces.data(), sizeof(float) * 3);
      (*mask)[i] = accel.Traverse(rays[i], isector, &(*hits)[i]) ? 1 : 0;
    }
  }
  *secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *count += n;
}

static bool g_separate_waves = false;  // --separate-waves: one TraverseBatch() call per wave (the round-4 behaviour), for comparison

// Two independent waves that are ready at the same time — the shadow query of one depth and the path wave of the next:
// with the GPU backend ONE TraverseBatches() call (one upload, one persistent launch over both waves — one launch tail
// instead of two —, one download; the shadow wave as an occlusion query, whose flags equal the closest-hit flags);
// otherwise the two per-ray host loops.  Records, flags and therefore the image are the same either way.
static void TraceShadowAndNext(const nanort::BVHAccel<float> &accel, const Scene &sc, const std::vector<Ray> &shadow_rays,
                               const std::vector<Ray> &next_rays, bool use_batch, std::vector<Hit> *shadow_hits,
                               std::vector<unsigned char> *shadow_mask, std::vector<Hit> *next_hits, std::vector<unsigned char> *next_mask,
                               double *secs, uint64_t *count) {
#ifdef NANORT_USE_HIP_BACKEND
  if (use_batch && !g_separate_waves) {
    shadow_mask->assign(shadow_rays.size(), 0);
    next_hits->resize(next_rays.size());
    next_mask->assign(next_rays.size(), 0);
    if (shadow_rays.empty() && next_rays.empty()) return;
    auto t0 = std::chrono::steady_clock::now();
    const Ray *r[2] = {shadow_rays.data(), next_rays.data()};
    const size_t n[2] = {shadow_rays.size(), next_rays.size()};
    Hit *h[2] = {NULL, next_hits->data()};
    unsigned char *m[2] = {shadow_mask->data(), next_mask->data()};
    const unsigned char occ[2] = {1, 0};
    if (!accel.TraverseBatches(2, r, n, h, m, occ)) {
      fprintf(stderr, "TraverseBatches failed: %s\n", accel.LastBackendError().c_str());
      exit(1);
    }
    *secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *count += shadow_rays.size() + next_rays.size();
    return;
  }
#endif
  TraceWave(accel, sc, shadow_rays, use_batch, shadow_hits, shadow_mask, secs, count);
  TraceWave(accel, sc, next_rays, use_batch, next_hits, next_mask, secs, count);
}

static void Render(const nanort::BVHAccel<float> &accel, const Scene &sc, int W, int H, int spp, int max_depth, bool use_batch,
                   std::vector<float> *image, double *trace_secs, uint64_t *rays_traced) {
  const float3 light(8.0f, 12.0f, 15.0f), light_power(600.0f), albedo(0.75f, 0.7f, 0.6f), sky(0.4f, 0.5f, 0.7f);
  image->assign(3 * (size_t)W * H, 0.0f);
  std::vector<Path> paths, next_paths;
  std::vector<Ray> rays, shadow_rays, next_rays;
  std::vector<Hit> hits, shadow_hits, next_hits;
  std::vector<unsigned char> mask, shadow_mask, next_mask;
  std::vector<float3> shadow_contrib;
  std::vector<uint32_t> shadow_pixel;

  for (int s = 0; s < spp; s++) {
    // wave 0: one camera ray per pixel (the objrender camera, jittered per sample)
    paths.resize((size_t)W * H);
    rays.resize((size_t)W * H);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const uint32_t pix = (uint32_t)(y * W + x);
        Path &p = paths[pix];
        p.pixel = pix;
        p.rng = pcg_hash(pix * 9781u + (uint32_t)s * 6271u + 17u);
        p.throughput = float3(1.0f);
        const float jx = rnd01(p.rng), jy = rnd01(p.rng);
        Ray &r = rays[pix];
        r.org[0] = 0.0f;
        r.org[1] = 5.0f;
        r.org[2] = 20.0f;
        const float3 d = nanort::vnormalize(float3(((float)x + jx) / (float)W - 0.5f, ((float)y + jy) / (float)H - 0.5f, -1.0f));
        r.dir[0] = d[0];
        r.dir[1] = d[1];
        r.dir[2] = d[2];
        r.min_t = 0.001f;
        r.max_t = 1.0e30f;
      }
    TraceWave(accel, sc, rays, use_batch, &hits, &mask, trace_secs, rays_traced);  // the camera wave; deeper waves ride with the shadow queries below
    for (int depth = 0; depth <= max_depth && !paths.empty(); depth++) {
      // shade: misses pick up the sky; hits queue a shadow ray (NEE) and, below max depth, a bounce
      shadow_rays.clear();
      shadow_contrib.clear();
      shadow_pixel.clear();
      next_paths.clear();
      next_rays.clear();
      for (size_t i = 0; i < paths.size(); i++) {
        Path p = paths[i];
        float *px = &(*image)[3 * (size_t)p.pixel];
        if (!mask[i]) {
          for (int k = 0; k < 3; k++) px[k] += p.throughput[k] * sky[k] / (float)spp;
          continue;
        }
        const Ray &r = rays[i];
        const float3 org(r.org[0], r.org[1], r.org[2]), dir(r.dir[0], r.dir[1], r.dir[2]);
        const float3 P = org + dir * hits[i].t;
        float3 N = FaceNormal(sc, hits[i].prim_id);
        if (nanort::vdot(N, dir) > 0.0f) N = -N;
        // direct light: closest-hit query used as an occlusion test, like CheckForOccluder
        const float3 toL = light - P;
        const float dist = nanort::vlength(toL);
        const float3 wl = toL * (1.0f / dist);
        const float cosl = nanort::vdot(N, wl);
        if (cosl > 0.0f) {
          Ray sr;
          for (int k = 0; k < 3; k++) {
            sr.org[k] = P[k];
            sr.dir[k] = wl[k];
          }
          sr.min_t = 1.0e-3f;
          sr.max_t = dist - 1.0e-3f;
          shadow_rays.push_back(sr);
          const float g = cosl / (dist * dist) * (1.0f / 3.14159265f);
          shadow_contrib.push_back(p.throughput * albedo * light_power * g);
          shadow_pixel.push_back(p.pixel);
        }
        if (depth < max_depth) {
          const float u1 = rnd01(p.rng), phi = 6.28318530718f * rnd01(p.rng);
          const float rr = sqrtf(u1);
          float3 b1, b2;
          Onb(N, &b1, &b2);
          const float3 wi = nanort::vnormalize(b1 * (rr * cosf(phi)) + b2 * (rr * sinf(phi)) + N * sqrtf(1.0f - u1));
          Ray br;
          for (int k = 0; k < 3; k++) {
            br.org[k] = P[k];
            br.dir[k] = wi[k];
          }
          br.min_t = 1.0e-3f;
          br.max_t = 1.0e30f;
          p.throughput = p.throughput * albedo;  // cosine-weighted sampling: pdf cancels cos/pi
          next_paths.push_back(p);
          next_rays.push_back(br);
        }
      }
      TraceShadowAndNext(accel, sc, shadow_rays, next_rays, use_batch, &shadow_hits, &shadow_mask, &next_hits, &next_mask, trace_secs, rays_traced);
      for (size_t i = 0; i < shadow_rays.size(); i++) {
        if (shadow_mask[i]) continue;  // occluded
        float *px = &(*image)[3 * (size_t)shadow_pixel[i]];
        for (int k = 0; k < 3; k++) px[k] += shadow_contrib[i][k] / (float)spp;
      }
      paths.swap(next_paths);
      rays.swap(next_rays);
      hits.swap(next_hits);
      mask.swap(next_mask);
    }
  }
}

static void SavePPM(const char *path, const std::vector<float> &img, int W, int H) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return;
  fprintf(fp, "P6\n%d %d\n255\n", W, H);
  for (int y = H - 1; y >= 0; y--)
    for (int x = 0; x < W; x++)
      for (int k = 0; k < 3; k++) {
        float v = powf(std::min(1.0f, std::max(0.0f, img[3 * ((size_t)y * W + x) + k])), 1.0f / 2.2f);
        fputc((int)(v * 255.0f + 0.5f), fp);
      }
  fclose(fp);
}

int main(int argc, char **argv) {
  int W = 512, H = 288, spp = 4, depth = 3, nx = 300, ny = 150;
  bool verify = false;
  std::string out = "wavefront.ppm", raw;
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
    } else if (!strcmp(argv[i], "--raw") && i + 1 < argc) {
      raw = argv[++i];  // the accumulated float RGB image, row-major, for comparisons
    } else if (!strcmp(argv[i], "--verify")) {
      verify = true;
    } else if (!strcmp(argv[i], "--separate-waves")) {
      g_separate_waves = true;
    }
  }
  Scene sc;
  MakeGrid(&sc, nx, ny);
  const unsigned int num_faces = (unsigned int)(sc.faces.size() / 3);

  nanort::TriangleMesh<float> mesh(sc.vertices.data(), sc.faces.data(), sizeof(float) * 3);
  nanort::TriangleSAHPred<float> pred(sc.vertices.data(), sc.faces.data(), sizeof(float) * 3);
  nanort::BVHAccel<float> accel;
  auto t0 = std::chrono::steady_clock::now();
  if (!accel.Build(num_faces, mesh, pred)) {
    fprintf(stderr, "Build failed\n");
    return 1;
  }
  const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  nanort::BVHBuildStatistics st = accel.GetStatistics();
  printf("triangles %u  nodes %zu  depth %u  build %.3f ms (device %.3f ms)\n", num_faces, accel.GetNodes().size(), st.max_tree_depth,
         build_s * 1e3, (double)st.build_secs * 1e3);

#ifdef NANORT_USE_HIP_BACKEND
  const bool batch = true;
#else
  const bool batch = false;
#endif
  std::vector<float> image;
  double secs = 0.0;
  uint64_t rays = 0;
  Render(accel, sc, W, H, spp, depth, batch, &image, &secs, &rays);
  printf("%s: %llu rays in %.3f s of tracing = %.2f Mrays/s (host-visible, PCIe included)\n",
         batch ? (g_separate_waves ? "TraverseBatch (one call per wave)" : "TraverseBatch + TraverseBatches (shadow and next wave in one launch)") : "per-ray Traverse",
         (unsigned long long)rays, secs, (double)rays / secs / 1e6);
  SavePPM(out.c_str(), image, W, H);
  if (!raw.empty()) {
    if (FILE *fp = fopen(raw.c_str(), "wb")) {
      fwrite(image.data(), sizeof(float), image.size(), fp);
      fclose(fp);
    }
  }

  if (verify && batch) {
    std::vector<float> ref;
    double rsecs = 0.0;
    uint64_t rrays = 0;
    Render(accel, sc, W, H, spp, depth, false, &ref, &rsecs, &rrays);
    size_t bad = 0;
    for (size_t i = 0; i < image.size(); i++)
      if (image[i] != ref[i]) bad++;
    printf("per-ray host Traverse over the same tree: %llu rays in %.3f s = %.2f Mrays/s\n", (unsigned long long)rrays, rsecs,
           (double)rrays / rsecs / 1e6);
    printf("verify: %zu differing float components of %zu\n", bad, image.size());
    return bad == 0 ? 0 : 2;
  }
  return 0;
}

/* nanort_amd/csrc/scenes.c — deterministic synthetic workloads (host, plain C).
 *
 * The meshes and ray waves BASELINE.json's configs are quoted on, as specified
 * in SURVEY.md §8(d).  Everything is a pure function of its integer arguments
 * (no RNG state), generated in fp32 and widened by the caller for the fp64
 * config, so the CPU reference and the GPU consume byte-identical buffers.
 *
 * Built with gcc into nanort_amd/lib/libnrt_scenes.so (glibc sinf/cosf, no
 * fast-math, no contraction) so the bytes are the same in the build container
 * and on the GPU box.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
  float org[3];
  float dir[3];
  float min_t;
  float max_t;
  uint32_t type;
} ray_f32; /* wire format of nanort::Ray<float>, reference nanort.h:474-496 */

typedef struct {
  float u, v, t;
  uint32_t prim_id;
} hit_f32; /* nanort::TriangleIntersection<float>, reference nanort.h:996-1005 */

static uint32_t wang_hash(uint32_t s) {
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  return s;
}

static uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

/* Plane(nx, ny): (nx+1)*(ny+1) vertices, 2*nx*ny triangles.
 * C3/C5: nx=1000, ny=500 -> 1 000 000 triangles; C4: 2500 x 2000 -> 10 M. */
void nrt_scene_plane(uint32_t nx, uint32_t ny, float *verts, uint32_t *faces) {
  uint32_t i, j;
  for (j = 0; j <= ny; j++) {
    for (i = 0; i <= nx; i++) {
      float x = -10.0f + 20.0f * (float)i / (float)nx;
      float y = -5.0f + 20.0f * (float)j / (float)ny;
      uint32_t h = wang_hash((j * 73856093u) ^ (i * 19349663u)) >> 8;
      float noise = (float)h / 16777216.0f;
      float z = 0.5f * sinf(0.9f * x) * cosf(1.1f * y) + 0.05f * noise - 0.025f;
      size_t v = (size_t)j * (nx + 1) + i;
      verts[3 * v + 0] = x;
      verts[3 * v + 1] = y;
      verts[3 * v + 2] = z;
    }
  }
  for (j = 0; j < ny; j++) {
    for (i = 0; i < nx; i++) {
      uint32_t a = j * (nx + 1) + i, b = a + 1, c = a + (nx + 1), d = c + 1;
      size_t f = 2 * ((size_t)j * nx + i);
      faces[3 * f + 0] = a;
      faces[3 * f + 1] = b;
      faces[3 * f + 2] = d;
      faces[3 * f + 3] = a;
      faces[3 * f + 4] = d;
      faces[3 * f + 5] = c;
    }
  }
}

/* Closed lat-long "lumpy sphere": the C2 stand-in for the Stanford Bunny
 * (not in the reference tree; no network).  nu=264, nv=132 -> 69 696 tris.
 * Vertices: nu*(nv-1) ring vertices + 2 poles.  Triangles: 2*nu*(nv-1). */
void nrt_scene_sphere(uint32_t nu, uint32_t nv, float *verts, uint32_t *faces) {
  const float PI = 3.14159265358979323846f;
  uint32_t i, j;
  size_t f = 0;
  uint32_t nring = nv - 1;
  uint32_t south = nu * nring, north = south + 1;
  for (j = 1; j < nv; j++) {
    for (i = 0; i < nu; i++) {
      float theta = 2.0f * PI * (float)i / (float)nu;
      float phi = PI * (float)j / (float)nv;
      float noise = (float)(wang_hash((j * 73856093u) ^ (i * 19349663u)) >> 8) / 16777216.0f;
      float r = 7.5f * (1.0f + 0.15f * sinf(5.0f * theta) * cosf(3.0f * phi) + 0.01f * noise);
      size_t v = (size_t)(j - 1) * nu + i;
      verts[3 * v + 0] = r * sinf(phi) * cosf(theta);
      verts[3 * v + 1] = 5.0f + r * cosf(phi);
      verts[3 * v + 2] = r * sinf(phi) * sinf(theta);
    }
  }
  verts[3 * north + 0] = 0.0f;
  verts[3 * north + 1] = 5.0f + 7.5f;
  verts[3 * north + 2] = 0.0f;
  verts[3 * south + 0] = 0.0f;
  verts[3 * south + 1] = 5.0f - 7.5f;
  verts[3 * south + 2] = 0.0f;
  for (i = 0; i < nu; i++) { /* caps */
    uint32_t i1 = (i + 1) % nu;
    faces[3 * f + 0] = north;
    faces[3 * f + 1] = i1;
    faces[3 * f + 2] = i;
    f++;
    faces[3 * f + 0] = south;
    faces[3 * f + 1] = (nring - 1) * nu + i;
    faces[3 * f + 2] = (nring - 1) * nu + i1;
    f++;
  }
  for (j = 0; j + 1 < nring; j++) {
    for (i = 0; i < nu; i++) {
      uint32_t i1 = (i + 1) % nu;
      uint32_t a = j * nu + i, b = j * nu + i1, c = (j + 1) * nu + i, d = (j + 1) * nu + i1;
      faces[3 * f + 0] = a;
      faces[3 * f + 1] = b;
      faces[3 * f + 2] = d;
      f++;
      faces[3 * f + 0] = a;
      faces[3 * f + 1] = d;
      faces[3 * f + 2] = c;
      f++;
    }
  }
}

/* Wave 1: the camera of the reference's examples/objrender (main.cc:654-670):
 * org=(0,5,20), dir = normalize(x/W-0.5, y/H-0.5, -1) with the example's
 * float3::normalize (multiply by 1/len when len > 1e-6, main.cc:190-198),
 * min_t=0, max_t=1e30; row-major.  Rows [y0, y1) of a W x H image are written
 * (a tile for the multi-GPU split); out must hold (y1-y0)*W rays. */
static void camera_rows(uint32_t W, uint32_t H, uint32_t y0, uint32_t y_step, uint32_t rows, ray_f32 *out);

void nrt_rays_camera(uint32_t W, uint32_t H, uint32_t y0, uint32_t y1, ray_f32 *out) {
  camera_rows(W, H, y0, 1, y1 - y0, out);
}

/* Interleaved row split for the multi-GPU tile partition: rows y0, y0+y_step,
 * ... (`rows` of them) of the same W x H image. */
void nrt_rays_camera_rows(uint32_t W, uint32_t H, uint32_t y0, uint32_t y_step, uint32_t rows,
                          ray_f32 *out) {
  camera_rows(W, H, y0, y_step, rows, out);
}

static void camera_rows(uint32_t W, uint32_t H, uint32_t y0, uint32_t y_step, uint32_t rows, ray_f32 *out) {
  uint32_t x, j;
  for (j = 0; j < rows; j++) {
    const uint32_t y = y0 + j * y_step;
    for (x = 0; x < W; x++) {
      ray_f32 *r = &out[(size_t)j * W + x];
      float dx = ((float)x / (float)W) - 0.5f;
      float dy = ((float)y / (float)H) - 0.5f;
      float dz = -1.0f;
      float len = sqrtf(dx * dx + dy * dy + dz * dz);
      if (fabsf(len) > 1.0e-6f) {
        float inv = 1.0f / len;
        dx *= inv;
        dy *= inv;
        dz *= inv;
      }
      r->org[0] = 0.0f;
      r->org[1] = 5.0f;
      r->org[2] = 20.0f;
      r->dir[0] = dx;
      r->dir[1] = dy;
      r->dir[2] = dz;
      r->min_t = 0.0f;
      r->max_t = 1.0e30f;
      r->type = 0x1u; /* RAY_TYPE_PRIMARY */
    }
  }
}

static void face_normal_toward(const float *verts, const uint32_t *faces, uint32_t prim,
                               const float *view_dir, float n[3]) {
  const float *p0 = verts + 3 * (size_t)faces[3 * prim + 0];
  const float *p1 = verts + 3 * (size_t)faces[3 * prim + 1];
  const float *p2 = verts + 3 * (size_t)faces[3 * prim + 2];
  float e1[3], e2[3], len;
  int k;
  for (k = 0; k < 3; k++) {
    e1[k] = p1[k] - p0[k];
    e2[k] = p2[k] - p0[k];
  }
  n[0] = e1[1] * e2[2] - e1[2] * e2[1];
  n[1] = e1[2] * e2[0] - e1[0] * e2[2];
  n[2] = e1[0] * e2[1] - e1[1] * e2[0];
  len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (len > 0.0f) {
    n[0] /= len;
    n[1] /= len;
    n[2] /= len;
  } else {
    n[0] = 0.0f;
    n[1] = 0.0f;
    n[2] = 1.0f;
  }
  if (n[0] * view_dir[0] + n[1] * view_dir[1] + n[2] * view_dir[2] > 0.0f) {
    n[0] = -n[0];
    n[1] = -n[1];
    n[2] = -n[2];
  }
}

/* Wave 2 generators.  Built on the host from wave-1 hit records so that every
 * consumer sees the same bytes.  `mask[i]` != 0 marks a wave-1 hit; output is
 * compacted (one ray per wave-1 hit, in ray order); returns the count.
 * `pixel_base` is the global index of rays[0] (for tiles).
 *
 * kind 0 — shadow ray to the point light L=(8,12,15): dir = normalize(L-P),
 *          min_t=1e-3, max_t=|L-P|-1e-3  (closest-hit query used as an
 *          occlusion test, as the reference's CheckForOccluder does,
 *          examples/path_tracer/main.cc:675-701).
 * kind 1 — diffuse bounce: cosine-weighted direction about the geometric
 *          normal flipped toward the viewer, basis from the reference path
 *          tracer's revisedONB (examples/path_tracer/main.cc:214-228),
 *          (u1, u2) from pcg_hash(pixel), min_t=1e-3, max_t=1e30. */
uint64_t nrt_rays_secondary(int kind, const float *verts, const uint32_t *faces,
                            const ray_f32 *rays, const hit_f32 *hits, const uint8_t *mask,
                            uint64_t n, uint64_t pixel_base, ray_f32 *out) {
  const float L[3] = {8.0f, 12.0f, 15.0f};
  const float TWO_PI = 6.28318530717958647692f;
  uint64_t i, m = 0;
  for (i = 0; i < n; i++) {
    const ray_f32 *r = &rays[i];
    ray_f32 *o;
    float P[3], nrm[3];
    int k;
    if (!mask[i]) continue;
    o = &out[m++];
    for (k = 0; k < 3; k++) P[k] = r->org[k] + hits[i].t * r->dir[k];
    face_normal_toward(verts, faces, hits[i].prim_id, r->dir, nrm);
    for (k = 0; k < 3; k++) o->org[k] = P[k];
    if (kind == 0) {
      float d[3], dist, inv;
      for (k = 0; k < 3; k++) d[k] = L[k] - P[k];
      dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      inv = 1.0f / dist;
      for (k = 0; k < 3; k++) o->dir[k] = d[k] * inv;
      o->min_t = 1.0e-3f;
      o->max_t = dist - 1.0e-3f;
      o->type = 0x2u; /* RAY_TYPE_SECONDARY */
    } else {
      uint32_t h1 = pcg_hash((uint32_t)(pixel_base + i));
      uint32_t h2 = pcg_hash(h1);
      float u1 = (float)(h1 >> 8) / 16777216.0f;
      float phi = TWO_PI * ((float)(h2 >> 8) / 16777216.0f);
      float rr = sqrtf(u1);
      float x = rr * cosf(phi), y = rr * sinf(phi), z = sqrtf(1.0f - u1);
      float b1[3], b2[3], len, inv;
      if (nrm[2] < 0.0f) {
        const float a = 1.0f / (1.0f - nrm[2]);
        const float b = nrm[0] * nrm[1] * a;
        b1[0] = 1.0f - nrm[0] * nrm[0] * a;
        b1[1] = -b;
        b1[2] = nrm[0];
        b2[0] = b;
        b2[1] = nrm[1] * nrm[1] * a - 1.0f;
        b2[2] = -nrm[1];
      } else {
        const float a = 1.0f / (1.0f + nrm[2]);
        const float b = -nrm[0] * nrm[1] * a;
        b1[0] = 1.0f - nrm[0] * nrm[0] * a;
        b1[1] = b;
        b1[2] = -nrm[0];
        b2[0] = b;
        b2[1] = 1.0f - nrm[1] * nrm[1] * a;
        b2[2] = -nrm[1];
      }
      for (k = 0; k < 3; k++) o->dir[k] = b1[k] * x + b2[k] * y + nrm[k] * z;
      len = sqrtf(o->dir[0] * o->dir[0] + o->dir[1] * o->dir[1] + o->dir[2] * o->dir[2]);
      inv = 1.0f / len;
      for (k = 0; k < 3; k++) o->dir[k] *= inv;
      o->min_t = 1.0e-3f;
      o->max_t = 1.0e30f;
      o->type = 0x4u; /* RAY_TYPE_DIFFUSE */
    }
  }
  return m;
}

/* ---- the particle example's workload (examples/particle_primitive/main.cc) ------------------------------ */

/* PCG32 as that example uses it (main.cc:23-52): one float in [0, 1) per call. */
typedef struct {
  unsigned long long state, inc;
} particle_rng;
static float particle_random(particle_rng *rng) {
  const unsigned long long old = rng->state;
  unsigned int xorshifted, rot, ret;
  rng->state = old * 6364136223846793005ULL + rng->inc;
  xorshifted = (unsigned int)(((old >> 18u) ^ old) >> 27u);
  rot = (unsigned int)(old >> 59u);
  ret = (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
  return (float)((double)ret / 4294967296.0);
}

/* GenerateRandomSpheres (main.cc:295-325): n spheres with centres uniform in [bmin, bmax), seed (0, 1), all
 * radii = largest box extent / sqrt(n). */
void nrt_scene_random_spheres(uint64_t n, const float bmin[3], const float bmax[3], float *centers, float *radii) {
  particle_rng rng;
  float bsize = bmax[0] - bmin[0];
  uint64_t i;
  rng.state = 0u;
  rng.inc = (1ULL << 1u) | 1u;
  (void)particle_random(&rng);
  rng.state += 0ULL;
  (void)particle_random(&rng);
  if (bsize < bmax[1] - bmin[1]) bsize = bmax[1] - bmin[1];
  if (bsize < bmax[2] - bmin[2]) bsize = bmax[2] - bmin[2];
  for (i = 0; i < n; i++) {
    const float x = particle_random(&rng), y = particle_random(&rng), z = particle_random(&rng);
    centers[3 * i + 0] = x * (bmax[0] - bmin[0]) + bmin[0];
    centers[3 * i + 1] = y * (bmax[1] - bmin[1]) + bmin[1];
    centers[3 * i + 2] = z * (bmax[2] - bmin[2]) + bmin[2];
    radii[i] = bsize / (float)sqrt((double)n);
  }
}

/* That example's camera (main.cc:367-389): org = (0, 0, 4), dir = vnormalize(x/W - 0.5, y/H - 0.5, -1)
 * (nanort.h:383-398: scaled by 1/len when len > FLT_EPSILON), min_t = 0, max_t = 1e30; row-major. */
void nrt_rays_particle_camera(uint32_t W, uint32_t H, ray_f32 *out) {
  uint32_t x, y;
  for (y = 0; y < H; y++) {
    for (x = 0; x < W; x++) {
      ray_f32 *r = &out[(size_t)y * W + x];
      float d[3], len;
      d[0] = (x / (float)W) - 0.5f;
      d[1] = (y / (float)H) - 0.5f;
      d[2] = -1.0f;
      len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      if (fabsf(len) > 1.1920928955078125e-07f) {
        const float inv_len = 1.0f / len;
        d[0] *= inv_len;
        d[1] *= inv_len;
        d[2] *= inv_len;
      }
      r->org[0] = 0.0f;
      r->org[1] = 0.0f;
      r->org[2] = 4.0f;
      r->dir[0] = d[0];
      r->dir[1] = d[1];
      r->dir[2] = d[2];
      r->min_t = 0.0f;
      r->max_t = 1.0e30f;
      r->type = 0x1u;
    }
  }
}

/* GenerateRandomCylinders (examples/cylinder_primitive/main.cc:428-462): n cylinders with both end points uniform in
 * [bmin, bmax), seed (0, 1), radii = 0.25 * largest box extent / sqrt(n) (double arithmetic, then float).
 * verts: 2 x xyz per cylinder; radii: 2 per cylinder. */
void nrt_scene_random_cylinders(uint64_t n, const float bmin[3], const float bmax[3], float *verts, float *radii) {
  particle_rng rng;
  float bsize = bmax[0] - bmin[0];
  uint64_t i;
  int k;
  rng.state = 0u;
  rng.inc = (1ULL << 1u) | 1u;
  (void)particle_random(&rng);
  rng.state += 0ULL;
  (void)particle_random(&rng);
  if (bsize < bmax[1] - bmin[1]) bsize = bmax[1] - bmin[1];
  if (bsize < bmax[2] - bmin[2]) bsize = bmax[2] - bmin[2];
  for (i = 0; i < n; i++) {
    float u[6];
    for (k = 0; k < 6; k++) u[k] = particle_random(&rng);
    for (k = 0; k < 3; k++) {
      verts[3 * (2 * i + 0) + k] = u[k] * (bmax[k] - bmin[k]) + bmin[k];
      verts[3 * (2 * i + 1) + k] = u[3 + k] * (bmax[k] - bmin[k]) + bmin[k];
    }
    radii[2 * i + 0] = (float)((0.25 * bsize) / sqrt((double)n));
    radii[2 * i + 1] = (float)((0.25 * bsize) / sqrt((double)n));
  }
}

// tools/build_host.cc — the application-visible build time: wall clock of nanort::BVHAccel<T>::Build() through include/nanort.h
// (NANORT_USE_HIP_BACKEND), i.e. what /root/reference/examples/path_tracer/main.cc:742-766 prints as "BVH build time": mesh
// compaction + upload, the device build, and whatever the header copies back.  Beside it the same three steps through the C ABI
// (nrtSetMesh / nrtBuild / nrtGetTree), timed one by one.  Mesh: SURVEY 8(d)'s Plane(nx, ny).
//
//   build_host NX NY f32|f64 [reps]      -> one JSON object on stdout
//
// Built by __graft_entry__.build() into tools/bin/build_host; bench.py reports its output as `build_host_ms`.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define NANORT_USE_HIP_BACKEND
#include "nanort.h"

extern "C" void nrt_scene_plane(uint32_t nx, uint32_t ny, float *verts, uint32_t *faces);  // nanort_amd/csrc/scenes.c

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v.empty() ? 0.0 : v[v.size() / 2];
}

template <typename T>
struct Api;
template <>
struct Api<float> {
  static nrt_status SetMesh(nrt_ctx *c, const float *v, const unsigned *f, unsigned n) { return nrtSetMesh_f32(c, v, 12, f, n); }
  static nrt_status Build(nrt_ctx *c, uint64_t *nn) { return nrtBuild_f32(c, NULL, NULL, nn); }
  static nrt_status GetTree(nrt_ctx *c, void *n, uint32_t *i) { return nrtGetTree_f32(c, (nrt_node_f32 *)n, i); }
  static size_t NodeBytes() { return sizeof(nrt_node_f32); }
};
template <>
struct Api<double> {
  static nrt_status SetMesh(nrt_ctx *c, const double *v, const unsigned *f, unsigned n) { return nrtSetMesh_f64(c, v, 24, f, n); }
  static nrt_status Build(nrt_ctx *c, uint64_t *nn) { return nrtBuild_f64(c, NULL, NULL, nn); }
  static nrt_status GetTree(nrt_ctx *c, void *n, uint32_t *i) { return nrtGetTree_f64(c, (nrt_node_f64 *)n, i); }
  static size_t NodeBytes() { return sizeof(nrt_node_f64); }
};

template <typename T>
static int run(unsigned nx, unsigned ny, int reps, const char *real) {
  const size_t nv = (size_t)(nx + 1) * (ny + 1), nf = (size_t)2 * nx * ny;
  std::vector<float> v32(3 * nv);
  std::vector<unsigned> faces(3 * nf);
  nrt_scene_plane(nx, ny, v32.data(), faces.data());
  std::vector<T> verts(v32.begin(), v32.end());

  // (a) the header: what an application times
  nanort::TriangleMesh<T> mesh(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::TriangleSAHPred<T> pred(verts.data(), faces.data(), sizeof(T) * 3);
  nanort::BVHAccel<T> accel;
  Clock::time_point t0 = Clock::now();
  if (!accel.Build((unsigned)nf, mesh, pred)) {
    fprintf(stderr, "Build failed: %s\n", accel.LastBackendError().c_str());
    return 1;
  }
  const double first = ms_since(t0);
  std::vector<double> steady;
  for (int r = 0; r < reps; r++) {
    t0 = Clock::now();
    if (!accel.Build((unsigned)nf, mesh, pred)) return 1;
    steady.push_back(ms_since(t0));
  }
  const nanort::BVHBuildStatistics st = accel.GetStatistics();
  // the tree on the host, when the application asks for it (GetNodes / Traverse / Dump): the lazy part
  t0 = Clock::now();
  const size_t nn = accel.GetNodes().size();
  const size_t ni = accel.GetIndices().size();
  const double fetch = ms_since(t0);
  // a rebuild after the host tree was materialised once (vectors keep their capacity), Build() + GetNodes()
  std::vector<double> eager;
  for (int r = 0; r < reps; r++) {
    t0 = Clock::now();
    if (!accel.Build((unsigned)nf, mesh, pred)) return 1;
    (void)accel.GetNodes().size();
    eager.push_back(ms_since(t0));
  }

  // (b) the same steps through the C ABI, one by one
  nrt_ctx *c = NULL;
  if (nrtCreate(0, &c) != NRT_OK) return 1;
  std::vector<double> up, dev, dev_wall, down;
  std::vector<unsigned char> nodes;
  std::vector<uint32_t> idx(nf);
  for (int r = 0; r < reps + 1; r++) {
    t0 = Clock::now();
    if (Api<T>::SetMesh(c, verts.data(), faces.data(), (unsigned)nf) != NRT_OK) return 1;
    const double a = ms_since(t0);
    uint64_t n_nodes = 0;
    t0 = Clock::now();
    if (Api<T>::Build(c, &n_nodes) != NRT_OK) return 1;
    const double b = ms_since(t0);
    nodes.resize((size_t)n_nodes * Api<T>::NodeBytes());
    t0 = Clock::now();
    if (Api<T>::GetTree(c, nodes.data(), idx.data()) != NRT_OK) return 1;
    const double d = ms_since(t0);
    if (r) {  // (first round: allocations)
      up.push_back(a);
      dev_wall.push_back(b);
      dev.push_back(nrtLastBuildMs(c));
      down.push_back(d);
    }
  }
  nrtDestroy(c);
  printf("{\"unit\": \"ms\", \"what\": \"wall clock of BVHAccel<%s>::Build() through include/nanort.h, Plane(%u,%u) = %zu triangles\", "
         "\"first\": %.3f, \"steady\": %.3f, \"device\": %.3f, \"steady_over_device\": %.2f, \"host_tree_on_demand\": %.3f, "
         "\"steady_with_host_tree\": %.3f, \"c_abi\": {\"set_mesh\": %.3f, \"build_wall\": %.3f, \"get_tree\": %.3f}, "
         "\"nodes\": %zu, \"indices\": %zu, \"reps\": %d}\n",
         real, nx, ny, nf, first, median(steady), (double)st.build_secs * 1e3, median(steady) / std::max(1e-9, (double)st.build_secs * 1e3), fetch,
         median(eager), median(up), median(dev_wall), median(down), nn, ni, reps);
  return 0;
}

int main(int argc, char **argv) {
  const unsigned nx = argc > 1 ? (unsigned)atoi(argv[1]) : 1000, ny = argc > 2 ? (unsigned)atoi(argv[2]) : 500;
  const char *real = argc > 3 ? argv[3] : "f32";
  const int reps = argc > 4 ? atoi(argv[4]) : 7;
  return strcmp(real, "f64") == 0 ? run<double>(nx, ny, reps, real) : run<float>(nx, ny, reps, real);
}

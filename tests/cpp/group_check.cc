// tests/cpp/group_check.cc — a C++ host drives the multi-GPU C ABI with device-resident rays (include/nanort_hip.h: nrtGroup*):
// N contexts (replicas of one tree; device k % nrtDeviceCount()), the frame cut into row-interleaved tiles whose rays live in
// HBM, one nrtGroupTraverseGather_f32/_f64 per frame, and the gathered frame compared bit for bit (every field) with
// nrtTraverseBatchDevice over the whole ray array on one context.
//
//   group_check f32|f64 mesh.bin rays.bin NUM_TILES ROW_LEN [transport=0|1] [self_send=0|1] [ranked=0|1] [root=K]
//
// mesh.bin: u32 nv, u32 nf, T xyz[nv], u32 ijk[nf]; rays.bin: u64 n, Ray<T>[n]   (tests/test_host_header.py write_inputs)
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nanort_hip.h"

#define HIP(call)                                                                   \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                    \
      return 10;                                                                    \
    }                                                                               \
  } while (0)

template <typename T>
struct Api;
template <>
struct Api<float> {
  typedef nrt_ray_f32 Ray;
  typedef nrt_hit_f32 Hit;
  static nrt_status SetMesh(nrt_ctx *c, const float *v, const uint32_t *f, uint32_t n) { return nrtSetMesh_f32(c, v, 12, f, n); }
  static nrt_status Build(nrt_ctx *c) { return nrtBuild_f32(c, NULL, NULL, NULL); }
  static nrt_status Trace(nrt_ctx *c, const Ray *r, uint64_t n, Hit *h, uint8_t *m) { return nrtTraverseBatchDevice_f32(c, r, n, NULL, h, m, NULL); }
  static nrt_status Gather(nrt_group *g, const Ray *const *r, const uint64_t *cnt, uint64_t total, uint64_t row, uint32_t root, Hit *h, uint8_t *m) {
    return nrtGroupTraverseGather_f32(g, r, cnt, total, row, NULL, root, h, m);
  }
  static nrt_status GatherTiles(nrt_group *g, const Ray *const *r, const uint64_t *cnt, uint64_t slot, uint32_t root, Hit *h, uint8_t *m) {
    return nrtGroupTraverseGatherTiles_f32(g, r, cnt, slot, NULL, root, h, m);
  }
};
template <>
struct Api<double> {
  typedef nrt_ray_f64 Ray;
  typedef nrt_hit_f64 Hit;
  static nrt_status SetMesh(nrt_ctx *c, const double *v, const uint32_t *f, uint32_t n) { return nrtSetMesh_f64(c, v, 24, f, n); }
  static nrt_status Build(nrt_ctx *c) { return nrtBuild_f64(c, NULL, NULL, NULL); }
  static nrt_status Trace(nrt_ctx *c, const Ray *r, uint64_t n, Hit *h, uint8_t *m) { return nrtTraverseBatchDevice_f64(c, r, n, NULL, h, m, NULL); }
  static nrt_status Gather(nrt_group *g, const Ray *const *r, const uint64_t *cnt, uint64_t total, uint64_t row, uint32_t root, Hit *h, uint8_t *m) {
    return nrtGroupTraverseGather_f64(g, r, cnt, total, row, NULL, root, h, m);
  }
  static nrt_status GatherTiles(nrt_group *g, const Ray *const *r, const uint64_t *cnt, uint64_t slot, uint32_t root, Hit *h, uint8_t *m) {
    return nrtGroupTraverseGatherTiles_f64(g, r, cnt, slot, NULL, root, h, m);
  }
};

static long long opt(int argc, char **argv, const char *name, long long dflt) {
  const size_t len = strlen(name);
  for (int i = 6; i < argc; i++)
    if (!strncmp(argv[i], name, len) && argv[i][len] == '=') return atoll(argv[i] + len + 1);
  return dflt;
}

template <typename T>
static int run(int argc, char **argv) {
  typedef typename Api<T>::Ray Ray;
  typedef typename Api<T>::Hit Hit;
  FILE *fp = fopen(argv[2], "rb");
  if (!fp) return 2;
  uint32_t nv = 0, nf = 0;
  if (fread(&nv, 4, 1, fp) != 1 || fread(&nf, 4, 1, fp) != 1) return 2;
  std::vector<T> verts(3 * (size_t)nv);
  std::vector<uint32_t> faces(3 * (size_t)nf);
  if (fread(verts.data(), sizeof(T), verts.size(), fp) != verts.size() || fread(faces.data(), 4, faces.size(), fp) != faces.size()) return 2;
  fclose(fp);
  fp = fopen(argv[3], "rb");
  if (!fp) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, fp) != 1) return 2;
  std::vector<Ray> rays(n);
  if (fread(rays.data(), sizeof(Ray), n, fp) != n) return 2;
  fclose(fp);
  const uint32_t N = (uint32_t)atoi(argv[4]);
  const uint64_t row_len = (uint64_t)atoll(argv[5]);
  const long long transport = opt(argc, argv, "transport", 0), self_send = opt(argc, argv, "self_send", 0), ranked = opt(argc, argv, "ranked", 0);
  const uint32_t root = (uint32_t)opt(argc, argv, "root", 0);
  const int ndev = nrtDeviceCount();
  if (ndev < 1) return 3;

  // replicas: the deterministic build of the same mesh on every context
  std::vector<nrt_ctx *> ctx(N, (nrt_ctx *)NULL);
  for (uint32_t k = 0; k < N; k++) {
    if (nrtCreate((int)(k % (uint32_t)ndev), &ctx[k]) != NRT_OK || Api<T>::SetMesh(ctx[k], verts.data(), faces.data(), nf) != NRT_OK ||
        Api<T>::Build(ctx[k]) != NRT_OK) {
      fprintf(stderr, "context %u: %s\n", k, nrtLastError(ctx[k]));
      return 4;
    }
  }
  // the reference frame: one context, the whole ray array
  HIP(hipSetDevice(0));
  Ray *d_all = NULL;
  Hit *d_ref = NULL;
  uint8_t *d_refm = NULL;
  HIP(hipMalloc((void **)&d_all, n * sizeof(Ray)));
  HIP(hipMalloc((void **)&d_ref, n * sizeof(Hit)));
  HIP(hipMalloc((void **)&d_refm, n));
  HIP(hipMemcpy(d_all, rays.data(), n * sizeof(Ray), hipMemcpyHostToDevice));
  if (Api<T>::Trace(ctx[0], d_all, n, d_ref, d_refm) != NRT_OK) return 5;
  HIP(hipDeviceSynchronize());
  std::vector<Hit> ref(n);
  std::vector<uint8_t> refm(n);
  HIP(hipMemcpy(ref.data(), d_ref, n * sizeof(Hit), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(refm.data(), d_refm, n, hipMemcpyDeviceToHost));

  // the tiles' rays, resident on their devices
  const uint64_t rows = (n + row_len - 1) / row_len;
  std::vector<Ray *> d_tile(N, (Ray *)NULL);
  std::vector<uint64_t> cnt(N, 0);
  for (uint32_t t = 0; t < N; t++) {
    std::vector<Ray> mine;
    for (uint64_t r = t; r < rows; r += N)
      for (uint64_t x = r * row_len; x < (r + 1) * row_len && x < n; x++) mine.push_back(rays[x]);
    cnt[t] = mine.size();
    if (cnt[t] != nrtGroupTileRays(n, row_len, t, N)) {
      fprintf(stderr, "nrtGroupTileRays(%u) = %llu, expected %llu\n", t, (unsigned long long)nrtGroupTileRays(n, row_len, t, N), (unsigned long long)cnt[t]);
      return 6;
    }
    HIP(hipSetDevice((int)(t % (uint32_t)ndev)));
    HIP(hipMalloc((void **)&d_tile[t], (cnt[t] ? cnt[t] : 1) * sizeof(Ray)));
    if (cnt[t]) HIP(hipMemcpy(d_tile[t], mine.data(), cnt[t] * sizeof(Ray), hipMemcpyHostToDevice));
  }

  nrt_group *g = NULL;
  if (ranked) {  // the one-process-per-GPU form with a world of one: this process owns tile 0 of 1
    if (N != 1) return 7;
    char id[128];
    if (nrtGroupUniqueId(id, sizeof(id)) != NRT_OK || nrtGroupCreateRanked(ctx[0], id, 0, 1, &g) != NRT_OK) {
      fprintf(stderr, "ranked group: %s\n", nrtGroupLastError(NULL));
      return 8;
    }
  } else if (nrtGroupCreate(ctx.data(), N, &g) != NRT_OK) {
    fprintf(stderr, "group: %s\n", nrtGroupLastError(NULL));
    return 8;
  }
  uint32_t tiles = 0, local = 0;
  int nranks = 0, bound = 0;
  nrtGroupInfo(g, &tiles, &local, &nranks, &bound);
  printf("tiles %u local %u ranks %d rccl_bound %d note '%s'\n", tiles, local, nranks, bound, nrtGroupLastError(g));
  if (nrtGroupSetTunable(g, "transport", transport) != NRT_OK || nrtGroupSetTunable(g, "self_send", self_send) != NRT_OK) {
    fprintf(stderr, "tunable: %s\n", nrtGroupLastError(g));
    return 9;
  }
  const int root_dev = (int)(root % (uint32_t)ndev);
  HIP(hipSetDevice(root_dev));
  Hit *d_frame = NULL;
  uint8_t *d_fmask = NULL;
  HIP(hipMalloc((void **)&d_frame, n * sizeof(Hit)));
  HIP(hipMalloc((void **)&d_fmask, n));
  std::vector<Hit> frame(n);
  std::vector<uint8_t> fmask(n);
  unsigned long long bad_total = 0;
  // u, v, t, prim_id: every bit (the fp64 record ends in four bytes of padding that no launch defines)
  const size_t kFieldBytes = 3 * sizeof(T) + sizeof(uint32_t);
  for (int round = 0; round < 3; round++) {  // (several frames through the same group: buffers and events are reused)
    HIP(hipMemset(d_frame, 0xCD, n * sizeof(Hit)));
    HIP(hipMemset(d_fmask, 0xCD, n));
    HIP(hipDeviceSynchronize());
    if (Api<T>::Gather(g, (const Ray *const *)d_tile.data(), cnt.data(), n, row_len, root, d_frame, d_fmask) != NRT_OK || nrtGroupSynchronize(g) != NRT_OK) {
      fprintf(stderr, "gather: %s\n", nrtGroupLastError(g));
      return 11;
    }
    HIP(hipMemcpy(frame.data(), d_frame, n * sizeof(Hit), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(fmask.data(), d_fmask, n, hipMemcpyDeviceToHost));
    unsigned long long bad = 0;
    for (uint64_t i = 0; i < n; i++)
      if (memcmp(&frame[i], &ref[i], kFieldBytes) != 0 || fmask[i] != refm[i]) bad++;
    bad_total += bad;
  }
  uint64_t b_rccl = 0, b_peer = 0, b_place = 0;
  nrtGroupLastTraffic(g, &b_rccl, &b_peer, &b_place);
  unsigned long long hits = 0;
  for (uint64_t i = 0; i < n; i++) hits += refm[i];
  printf("rays %llu hits %llu frame_mismatches %llu bytes_rccl %llu bytes_peer %llu bytes_in_place %llu\n", (unsigned long long)n, hits, bad_total,
         (unsigned long long)b_rccl, (unsigned long long)b_peer, (unsigned long long)b_place);
  // ragged waves (secondary rays): tile t traces only its first cnt[t] - 13 * t rays; the root receives every tile's slot, tile-major
  {
    uint64_t slot = 0;
    for (uint32_t t = 0; t < N; t++) slot = cnt[t] > slot ? cnt[t] : slot;
    std::vector<uint64_t> rag(N);
    for (uint32_t t = 0; t < N; t++) rag[t] = cnt[t] > 13ull * t ? cnt[t] - 13ull * t : 0;
    Hit *d_slots = NULL;
    uint8_t *d_smask = NULL;
    HIP(hipSetDevice(root_dev));
    HIP(hipMalloc((void **)&d_slots, (size_t)N * slot * sizeof(Hit)));
    HIP(hipMalloc((void **)&d_smask, (size_t)N * slot));
    unsigned long long tbad = 0;
    for (int round = 0; round < 2; round++) {
      HIP(hipMemset(d_slots, 0xCD, (size_t)N * slot * sizeof(Hit)));
      HIP(hipDeviceSynchronize());
      if (Api<T>::GatherTiles(g, (const Ray *const *)d_tile.data(), rag.data(), slot, root, d_slots, d_smask) != NRT_OK || nrtGroupSynchronize(g) != NRT_OK) {
        fprintf(stderr, "gather tiles: %s\n", nrtGroupLastError(g));
        return 12;
      }
      std::vector<Hit> slots((size_t)N * slot);
      std::vector<uint8_t> smask((size_t)N * slot);
      HIP(hipMemcpy(slots.data(), d_slots, slots.size() * sizeof(Hit), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(smask.data(), d_smask, smask.size(), hipMemcpyDeviceToHost));
      for (uint32_t t = 0; t < N; t++) {
        uint64_t i = 0;  // the tile's i-th ray is the frame's ray of row t + (i / row_len) * N, column i % row_len
        for (uint64_t r = t; r < rows && i < rag[t]; r += N)
          for (uint64_t x = r * row_len; x < (r + 1) * row_len && x < n && i < rag[t]; x++, i++)
            if (memcmp(&slots[(size_t)t * slot + i], &ref[x], kFieldBytes) != 0 || smask[(size_t)t * slot + i] != refm[x]) tbad++;
      }
    }
    printf("tile_slot_mismatches %llu slot %llu\n", tbad, (unsigned long long)slot);
    bad_total += tbad;
    std::vector<uint64_t> too_many(rag);
    too_many[0] = slot + 1;
    printf("slot_overflow_status %d\n", (int)Api<T>::GatherTiles(g, (const Ray *const *)d_tile.data(), too_many.data(), slot, root, d_slots, d_smask));
  }
  // misuse is reported, not executed
  std::vector<uint64_t> wrong(cnt);
  wrong[0] += 1;
  const nrt_status st = Api<T>::Gather(g, (const Ray *const *)d_tile.data(), wrong.data(), n, row_len, root, d_frame, d_fmask);
  printf("wrong_count_status %d '%s'\n", (int)st, nrtGroupLastError(g));
  nrtGroupDestroy(g);
  for (uint32_t k = 0; k < N; k++) nrtDestroy(ctx[k]);
  return bad_total == 0 ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: group_check f32|f64 mesh.bin rays.bin NUM_TILES ROW_LEN [transport=] [self_send=] [ranked=] [root=]\n");
    return 64;
  }
  return !strcmp(argv[1], "f64") ? run<double>(argc, argv) : run<float>(argc, argv);
}

// nanort_amd/csrc/group.hip — multi-GPU traversal of DEVICE-RESIDENT ray tiles with an RCCL gather of the hit records.
//
// SURVEY.md §8(e) / BASELINE.json north_star: "rays partition embarrassingly across the 8 GPUs of one node (replicated BVH,
// image-tile split, RCCL gather of hit records over xGMI)".  The reference has no counterpart (its only parallel loop is the
// example's row loop, /root/reference/examples/path_tracer/main.cc:785-806); this is the C-ABI form of that split for a C++
// host that keeps its ray waves in HBM: one context (== one replica of the tree) per tile, every tile traced on a stream of
// its own, the 16 / 32-byte records sent to the root GPU and put into frame order there by a small kernel.
//
//   tile g of N owns the interleaved rows  g, g + N, g + 2N, ...  of a frame of `total_rays` rays cut into rows of `row_len`
//   (the split of nrtTraverseBatchMulti and of bench.py's ranks).
//
// Two ways to form a group:
//   nrtGroupCreate(ctxs, n)                 one process drives n contexts (one per GPU; several on one GPU work too);
//                                           one RCCL rank per distinct device (ncclCommInitAll)
//   nrtGroupCreateRanked(ctx, id, r, n)     one process per GPU (the bench's launch model): rank r of n, the 128-byte id of
//                                           nrtGroupUniqueId() handed round by the host program (MPI, a file, a torch store)
// RCCL is bound at run time (dlopen of librccl.so.1: the process's own copy when a framework already loaded one), so the
// library has no link-time dependency on it; records of tiles that live on the root's own device are read in place, tiles on
// other devices travel by ncclSend / ncclRecv in one ncclGroupStart / ncclGroupEnd (or, tunable "transport" = 1 in a
// single-process group, by hipMemcpyPeerAsync).
#include <dlfcn.h>
#include <string.h>

#include <algorithm>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

int nrt_internal_device(const nrt_ctx *c); // api.hip
using namespace nrt;

namespace {

// ---- the nine RCCL entry points, bound at run time ---------------------------------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId; // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 }; // rccl.h:459-460

struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      r.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "?");
      return;
    }
#define NRT_BIND(field, sym)                                            \
  do {                                                                  \
    *(void **)(&r.field) = dlsym(r.handle, sym);                        \
    if (!r.field) {                                                     \
      r.why = std::string("librccl.so.1 lacks ") + sym;                 \
      return;                                                           \
    }                                                                   \
  } while (0)
    NRT_BIND(GetUniqueId, "ncclGetUniqueId");
    NRT_BIND(CommInitRank, "ncclCommInitRank");
    NRT_BIND(CommInitAll, "ncclCommInitAll");
    NRT_BIND(CommDestroy, "ncclCommDestroy");
    NRT_BIND(GroupStart, "ncclGroupStart");
    NRT_BIND(GroupEnd, "ncclGroupEnd");
    NRT_BIND(Send, "ncclSend");
    NRT_BIND(Recv, "ncclRecv");
    NRT_BIND(GetErrorString, "ncclGetErrorString");
#undef NRT_BIND
    r.ok = true;
  });
  return r;
}

// rays of tile g: its rows are g, g + N, ...; only the frame's final row can be short
uint64_t tile_share(uint64_t total, uint64_t row_len, uint32_t g, uint32_t N) {
  const uint64_t rows = (total + row_len - 1) / row_len;
  if (g >= rows) return 0;
  const uint64_t mine = (rows - 1 - g) / N + 1;
  const uint64_t last_row = g + (mine - 1) * N;
  const uint64_t last_len = (last_row == rows - 1) ? total - last_row * row_len : row_len;
  return (mine - 1) * row_len + last_len;
}

// Frame order from tile order: element i of tile g (row i / row_len of the tile, column i % row_len) goes to frame row
// (i / row_len) * N + g.  UNITS 16-byte pieces per element (1: fp32 record, 2: fp64 record); the flag variant moves bytes.
template <typename V>
__global__ __launch_bounds__(256) void k_tile_to_frame(const V *__restrict__ src, V *__restrict__ dst, uint64_t count, uint32_t units,
                                                       uint64_t row_len, uint32_t g, uint32_t N) {
  const uint64_t total = count * units;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t e = i / units, part = i - e * units;
    const uint64_t row = e / row_len, x = e - row * row_len;
    dst[((row * N + g) * row_len + x) * units + part] = src[i];
  }
}

} // namespace

struct nrt_group {
  uint32_t num_tiles = 0; // N: tiles of the frame over all processes
  int nranks = 1, my_rank = 0;
  bool ranked = false;
  struct Local {
    nrt_ctx *ctx = nullptr;
    int device = 0;
    uint32_t tile = 0;       // global tile index
    int comm = -1;           // index into comms
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    DevBuf hits, mask;       // the tile's own records / flags (on its device)
  };
  std::vector<Local> local;
  std::vector<int> tile_rank;     // RCCL rank holding tile g
  std::vector<ncclComm_t> comms;  // single process: one per distinct device; ranked: one
  std::vector<int> comm_device;
  // root side (grow-only, on the root tile's device): where the other devices' records arrive
  std::vector<DevBuf> stage_hits, stage_mask;
  hipEvent_t frame_done = nullptr; // recorded on the root's stream after a frame's kernels; the next frame's tiles wait for it
  int frame_done_device = -1;
  int transport = 0; // 0: RCCL send / recv, 1: hipMemcpyPeerAsync (single-process groups only)
  int self_send = 0; // 1: the root tile's own records go through ncclSend / ncclRecv too (exercises the exchange on a one-GPU box)
  uint64_t last_bytes_rccl = 0, last_bytes_peer = 0, last_bytes_in_place = 0;
  std::string err;
  std::mutex mu;
};

static thread_local std::string g_group_error;

static nrt_status gfail(nrt_group *g, nrt_status st, const std::string &msg) {
  if (g)
    g->err = msg;
  else
    g_group_error = msg;
  return st;
}

#define GHIP(g, call)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) return gfail(g, NRT_ERR_DEVICE, std::string(#call ": ") + hipGetErrorString(e_));    \
  } while (0)
// ... inside an open ncclGroupStart: the group is closed before the error is reported, so that the communicator stays usable
#define GNCCL_IN_GROUP(g, call)                                                                                           \
  do {                                                                                                                     \
    int r_ = (call);                                                                                                       \
    if (r_ != ncclSuccess) {                                                                                               \
      (void)rccl().GroupEnd();                                                                                             \
      return gfail(g, NRT_ERR_DEVICE, std::string(#call ": ") + rccl().GetErrorString(r_));                                \
    }                                                                                                                      \
  } while (0)
#define GNCCL(g, call)                                                                                                     \
  do {                                                                                                                     \
    int r_ = (call);                                                                                                       \
    if (r_ != ncclSuccess) return gfail(g, NRT_ERR_DEVICE, std::string(#call ": ") + rccl().GetErrorString(r_));           \
  } while (0)

static nrt_status group_init_locals(nrt_group *g) {
  for (nrt_group::Local &l : g->local) {
    GHIP(g, hipSetDevice(l.device));
    GHIP(g, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    GHIP(g, hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
  }
  g->stage_hits.resize(g->num_tiles);
  g->stage_mask.resize(g->num_tiles);
  return NRT_OK;
}

extern "C" {

const char *nrtGroupLastError(const nrt_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }

nrt_status nrtGroupUniqueId(void *id_out, size_t bytes) {
  if (!id_out || bytes < sizeof(ncclUniqueId)) return gfail(nullptr, NRT_ERR_INVALID, "nrtGroupUniqueId: 128 bytes are needed");
  Rccl &r = rccl();
  if (!r.ok) return gfail(nullptr, NRT_ERR_DEVICE, "nrtGroupUniqueId: " + r.why);
  ncclUniqueId id;
  GNCCL(nullptr, r.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return NRT_OK;
}

nrt_status nrtGroupCreate(nrt_ctx *const *ctxs, uint32_t n, nrt_group **out) {
  if (!ctxs || !out || n == 0) return gfail(nullptr, NRT_ERR_INVALID, "nrtGroupCreate: no contexts");
  for (uint32_t k = 0; k < n; k++) {
    if (!ctxs[k]) return gfail(nullptr, NRT_ERR_INVALID, "nrtGroupCreate: context " + std::to_string(k) + " is NULL");
    for (uint32_t j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return gfail(nullptr, NRT_ERR_INVALID, "nrtGroupCreate: context " + std::to_string(k) + " is listed twice");
  }
  nrt_group *g = new nrt_group();
  g->num_tiles = n;
  g->tile_rank.resize(n);
  for (uint32_t k = 0; k < n; k++) {
    nrt_group::Local l;
    l.ctx = ctxs[k];
    l.device = nrt_internal_device(ctxs[k]);
    l.tile = k;
    int ci = -1;
    for (size_t j = 0; j < g->comm_device.size(); j++)
      if (g->comm_device[j] == l.device) ci = (int)j;
    if (ci < 0) {
      ci = (int)g->comm_device.size();
      g->comm_device.push_back(l.device);
    }
    l.comm = ci;
    g->tile_rank[k] = ci;
    g->local.push_back(l);
  }
  g->nranks = (int)g->comm_device.size();
  nrt_status st = group_init_locals(g);
  if (st) {
    g_group_error = g->err;
    delete g;
    return st;
  }
  // one RCCL rank per distinct device.  Without RCCL a group on one device still works (nothing travels), and a group over
  // several devices falls back to peer copies.
  Rccl &r = rccl();
  if (r.ok) {
    g->comms.resize(g->comm_device.size());
    int rc = r.CommInitAll(g->comms.data(), (int)g->comm_device.size(), g->comm_device.data());
    if (rc != ncclSuccess) {
      g->comms.clear();
      g->err = std::string("ncclCommInitAll: ") + r.GetErrorString(rc) + " (peer copies are used instead)";
      g->transport = 1;
    }
  } else {
    g->transport = 1;
    g->err = "RCCL unavailable (" + r.why + "): peer copies are used instead";
  }
  *out = g;
  return NRT_OK;
}

nrt_status nrtGroupCreateRanked(nrt_ctx *ctx, const void *unique_id, int rank, int nranks, nrt_group **out) {
  if (!ctx || !unique_id || !out || nranks < 1 || rank < 0 || rank >= nranks)
    return gfail(nullptr, NRT_ERR_INVALID, "nrtGroupCreateRanked: bad arguments");
  Rccl &r = rccl();
  if (!r.ok) return gfail(nullptr, NRT_ERR_DEVICE, "nrtGroupCreateRanked: " + r.why);
  nrt_group *g = new nrt_group();
  g->ranked = true;
  g->num_tiles = (uint32_t)nranks;
  g->nranks = nranks;
  g->my_rank = rank;
  g->tile_rank.resize(nranks);
  for (int k = 0; k < nranks; k++) g->tile_rank[k] = k;
  nrt_group::Local l;
  l.ctx = ctx;
  l.device = nrt_internal_device(ctx);
  l.tile = (uint32_t)rank;
  l.comm = 0;
  g->local.push_back(l);
  g->comm_device.push_back(l.device);
  nrt_status st = group_init_locals(g);
  if (st == NRT_OK) {
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    g->comms.resize(1);
    hipError_t he = hipSetDevice(l.device);
    int rc = he == hipSuccess ? r.CommInitRank(&g->comms[0], nranks, id, rank) : -1;
    if (rc != ncclSuccess) {
      g->comms.clear();
      st = gfail(g, NRT_ERR_DEVICE, std::string("ncclCommInitRank: ") + (rc < 0 ? hipGetErrorString(he) : r.GetErrorString(rc)));
    }
  }
  if (st) {
    g_group_error = g->err;
    delete g;
    return st;
  }
  *out = g;
  return NRT_OK;
}

void nrtGroupDestroy(nrt_group *g) {
  if (!g) return;
  for (nrt_group::Local &l : g->local) {
    (void)hipSetDevice(l.device);
    if (l.stream) {
      (void)hipStreamSynchronize(l.stream);
      (void)hipStreamDestroy(l.stream);
    }
    if (l.done) (void)hipEventDestroy(l.done);
    if (l.hits.p) (void)hipFree(l.hits.p);
    if (l.mask.p) (void)hipFree(l.mask.p);
  }
  if (g->frame_done) (void)hipEventDestroy(g->frame_done);
  for (DevBuf &b : g->stage_hits)
    if (b.p) (void)hipFree(b.p);
  for (DevBuf &b : g->stage_mask)
    if (b.p) (void)hipFree(b.p);
  for (ncclComm_t c : g->comms)
    if (c) (void)rccl().CommDestroy(c);
  delete g;
}

nrt_status nrtGroupSetTunable(nrt_group *g, const char *name, long long value) {
  if (!g || !name) return NRT_ERR_INVALID;
  if (!strcmp(name, "transport")) {
    if (value == 1 && g->ranked) return gfail(g, NRT_ERR_INVALID, "transport = 1 (peer copies) needs a single-process group");
    if (value == 0 && g->comms.empty()) return gfail(g, NRT_ERR_INVALID, "transport = 0 (RCCL) is unavailable in this group: " + g->err);
    g->transport = value ? 1 : 0;
    return NRT_OK;
  }
  if (!strcmp(name, "self_send")) {
    if (value && g->comms.empty()) return gfail(g, NRT_ERR_INVALID, "self_send needs RCCL: " + g->err);
    g->self_send = value ? 1 : 0;
    return NRT_OK;
  }
  return gfail(g, NRT_ERR_INVALID, std::string("unknown group tunable ") + name);
}

nrt_status nrtGroupInfo(const nrt_group *g, uint32_t *num_tiles, uint32_t *num_local, int *nranks, int *rccl_bound) {
  if (!g) return NRT_ERR_INVALID;
  if (num_tiles) *num_tiles = g->num_tiles;
  if (num_local) *num_local = (uint32_t)g->local.size();
  if (nranks) *nranks = g->nranks;
  if (rccl_bound) *rccl_bound = g->comms.empty() ? 0 : 1;
  return NRT_OK;
}

nrt_status nrtGroupLastTraffic(const nrt_group *g, uint64_t *bytes_rccl, uint64_t *bytes_peer, uint64_t *bytes_in_place) {
  if (!g) return NRT_ERR_INVALID;
  if (bytes_rccl) *bytes_rccl = g->last_bytes_rccl;
  if (bytes_peer) *bytes_peer = g->last_bytes_peer;
  if (bytes_in_place) *bytes_in_place = g->last_bytes_in_place;
  return NRT_OK;
}

uint64_t nrtGroupTileRays(uint64_t total_rays, uint64_t row_len, uint32_t tile, uint32_t num_tiles) {
  if (row_len == 0) row_len = 4096;
  if (num_tiles == 0) return 0;
  return tile_share(total_rays, row_len, tile, num_tiles);
}

nrt_status nrtGroupSynchronize(nrt_group *g) {
  if (!g) return NRT_ERR_INVALID;
  for (nrt_group::Local &l : g->local) {
    GHIP(g, hipSetDevice(l.device));
    GHIP(g, hipStreamSynchronize(l.stream));
  }
  return NRT_OK;
}

} // extern "C"

// One frame: trace every local tile, bring the records (and flags) to the root tile's device, frame order there.
template <int HIT_BYTES>
static nrt_status group_traverse_gather(nrt_group *g, const void *const *d_rays, const uint64_t *counts, uint64_t total_rays, uint64_t row_len,
                                        const nrt_trace_options *opt, uint32_t root_tile, void *d_frame_hits, uint8_t *d_frame_mask) {
  if (!g) return NRT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g->mu);
  if (row_len == 0) row_len = 4096;
  const uint32_t N = g->num_tiles;
  if (root_tile >= N) return gfail(g, NRT_ERR_INVALID, "root tile out of range");
  if (!d_rays || !counts) return gfail(g, NRT_ERR_INVALID, "NULL ray table");
  int root_local = -1;
  for (size_t k = 0; k < g->local.size(); k++) {
    const nrt_group::Local &l = g->local[k];
    if (counts[k] != tile_share(total_rays, row_len, l.tile, N))
      return gfail(g, NRT_ERR_INVALID, "tile " + std::to_string(l.tile) + ": " + std::to_string(counts[k]) + " rays given, its interleaved rows hold " +
                                           std::to_string(tile_share(total_rays, row_len, l.tile, N)) + " (nrtGroupTileRays)");
    if (counts[k] && !d_rays[k]) return gfail(g, NRT_ERR_INVALID, "tile " + std::to_string(l.tile) + ": NULL rays");
    if (l.tile == root_tile) root_local = (int)k;
  }
  const bool i_am_root = root_local >= 0;
  if (i_am_root && total_rays && !d_frame_hits) return gfail(g, NRT_ERR_INVALID, "the root needs a frame buffer");
  const bool send_mask = g->ranked ? true : d_frame_mask != nullptr; // (a non-root process cannot know: in ranked groups the flags always travel)
  g->last_bytes_rccl = g->last_bytes_peer = g->last_bytes_in_place = 0;

  // ---- A. every local tile on its own stream -----------------------------------------------------------------------------
  for (size_t k = 0; k < g->local.size(); k++) {
    nrt_group::Local &l = g->local[k];
    if (!counts[k]) continue;
    GHIP(g, hipSetDevice(l.device));
    // (the previous frame's kernels on the root's stream may still be reading this tile's records in place)
    if (g->frame_done_device >= 0) GHIP(g, hipStreamWaitEvent(l.stream, g->frame_done, 0));
    GHIP(g, devbuf_ensure(&l.hits, counts[k] * HIT_BYTES));
    GHIP(g, devbuf_ensure(&l.mask, counts[k]));
    nrt_status st = HIT_BYTES == 16
                        ? nrtTraverseBatchDevice_f32(l.ctx, (const nrt_ray_f32 *)d_rays[k], counts[k], opt, (nrt_hit_f32 *)l.hits.p, (uint8_t *)l.mask.p, l.stream)
                        : nrtTraverseBatchDevice_f64(l.ctx, (const nrt_ray_f64 *)d_rays[k], counts[k], opt, (nrt_hit_f64 *)l.hits.p, (uint8_t *)l.mask.p, l.stream);
    if (st) return gfail(g, st, "tile " + std::to_string(l.tile) + ": " + nrtLastError(l.ctx));
  }

  // ---- B. the exchange ----------------------------------------------------------------------------------------------------
  const int root_dev = i_am_root ? g->local[root_local].device : -1;
  hipStream_t root_stream = i_am_root ? g->local[root_local].stream : nullptr;
  // where tile t's records are on the root device once the exchange is done (root process only)
  std::vector<const void *> src_hits(N, nullptr);
  std::vector<const uint8_t *> src_mask(N, nullptr);
  if (i_am_root) { // staging for what arrives from other devices / processes
    GHIP(g, hipSetDevice(root_dev));
    for (uint32_t t = 0; t < N; t++) {
      const uint64_t cnt = tile_share(total_rays, row_len, t, N);
      if (!cnt) continue;
      int lk = -1;
      for (size_t k = 0; k < g->local.size(); k++)
        if (g->local[k].tile == t) lk = (int)k;
      const bool in_place = lk >= 0 && g->local[lk].device == root_dev && !(g->self_send && (int)t == (int)root_tile);
      if (in_place) {
        src_hits[t] = g->local[lk].hits.p;
        src_mask[t] = (const uint8_t *)g->local[lk].mask.p;
        g->last_bytes_in_place += cnt * HIT_BYTES;
      } else {
        GHIP(g, devbuf_ensure(&g->stage_hits[t], cnt * HIT_BYTES));
        src_hits[t] = g->stage_hits[t].p;
        if (send_mask) {
          GHIP(g, devbuf_ensure(&g->stage_mask[t], cnt));
          src_mask[t] = (const uint8_t *)g->stage_mask[t].p;
        }
      }
    }
  }
  auto travels = [&](const nrt_group::Local &l) { // does this local tile's data leave its buffers?
    if (g->ranked) return (int)l.tile != (int)root_tile || g->self_send != 0;
    if (!i_am_root) return false;
    return l.device != root_dev || (g->self_send && l.tile == root_tile);
  };
  if (g->transport == 0 && !g->comms.empty()) {
    Rccl &r = rccl();
    bool any = false;
    for (size_t k = 0; k < g->local.size(); k++) any = any || (counts[k] && travels(g->local[k]));
    if (i_am_root)
      for (uint32_t t = 0; t < N; t++) any = any || (src_hits[t] && src_hits[t] == g->stage_hits[t].p);
    if (any) {
      GNCCL(g, r.GroupStart());
      // sends: tile order (two tiles of one rank reach the root in the order the receives are posted)
      for (size_t k = 0; k < g->local.size(); k++) {
        nrt_group::Local &l = g->local[k];
        if (!counts[k] || !travels(l)) continue;
        GHIP(g, hipSetDevice(l.device));
        GNCCL_IN_GROUP(g, r.Send(l.hits.p, counts[k] * HIT_BYTES, ncclUint8, g->tile_rank[root_tile], g->comms[l.comm], l.stream));
        if (send_mask) GNCCL_IN_GROUP(g, r.Send(l.mask.p, counts[k], ncclUint8, g->tile_rank[root_tile], g->comms[l.comm], l.stream));
        g->last_bytes_rccl += counts[k] * HIT_BYTES + (send_mask ? counts[k] : 0);
      }
      if (i_am_root) {
        GHIP(g, hipSetDevice(root_dev));
        for (uint32_t t = 0; t < N; t++) {
          if (!src_hits[t] || src_hits[t] != g->stage_hits[t].p) continue;
          const uint64_t cnt = tile_share(total_rays, row_len, t, N);
          GNCCL_IN_GROUP(g, r.Recv(g->stage_hits[t].p, cnt * HIT_BYTES, ncclUint8, g->tile_rank[t], g->comms[g->local[root_local].comm], root_stream));
          if (send_mask) GNCCL_IN_GROUP(g, r.Recv(g->stage_mask[t].p, cnt, ncclUint8, g->tile_rank[t], g->comms[g->local[root_local].comm], root_stream));
        }
      }
      GNCCL(g, r.GroupEnd());
    }
  } else if (i_am_root) { // peer copies (single-process groups): issued on the sender's stream, after its traversal
    for (size_t k = 0; k < g->local.size(); k++) {
      nrt_group::Local &l = g->local[k];
      if (!counts[k] || !travels(l)) continue;
      GHIP(g, hipSetDevice(l.device));
      GHIP(g, hipMemcpyPeerAsync(g->stage_hits[l.tile].p, root_dev, l.hits.p, l.device, counts[k] * HIT_BYTES, l.stream));
      if (send_mask) GHIP(g, hipMemcpyPeerAsync(g->stage_mask[l.tile].p, root_dev, l.mask.p, l.device, counts[k], l.stream));
      g->last_bytes_peer += counts[k] * HIT_BYTES + (send_mask ? counts[k] : 0);
    }
  } else if (g->ranked) {
    return gfail(g, NRT_ERR_INVALID, "a ranked group needs RCCL");
  }

  // ---- C. frame order on the root's stream --------------------------------------------------------------------------------
  if (i_am_root) {
    // the root stream waits for every other local stream (their traversals and copies)
    for (size_t k = 0; k < g->local.size(); k++) {
      nrt_group::Local &l = g->local[k];
      if ((int)k == root_local || !counts[k]) continue;
      GHIP(g, hipSetDevice(l.device));
      GHIP(g, hipEventRecord(l.done, l.stream));
      GHIP(g, hipSetDevice(root_dev));
      GHIP(g, hipStreamWaitEvent(root_stream, l.done, 0));
    }
    GHIP(g, hipSetDevice(root_dev));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    for (uint32_t t = 0; t < N; t++) {
      const uint64_t cnt = tile_share(total_rays, row_len, t, N);
      if (!cnt) continue;
      const uint32_t units = HIT_BYTES / 16;
      const unsigned grid = (unsigned)std::min<uint64_t>((cnt * units + 255) / 256, 4096);
      hipLaunchKernelGGL((k_tile_to_frame<u4>), dim3(grid), dim3(256), 0, root_stream, (const u4 *)src_hits[t], (u4 *)d_frame_hits, cnt, units, row_len, t, N);
      if (d_frame_mask && src_mask[t])
        hipLaunchKernelGGL((k_tile_to_frame<uint8_t>), dim3((unsigned)std::min<uint64_t>((cnt + 255) / 256, 4096)), dim3(256), 0, root_stream, src_mask[t],
                           d_frame_mask, cnt, 1u, row_len, t, N);
    }
    GHIP(g, hipGetLastError());
    if (!g->frame_done) GHIP(g, hipEventCreateWithFlags(&g->frame_done, hipEventDisableTiming));
    GHIP(g, hipEventRecord(g->frame_done, root_stream));
    g->frame_done_device = root_dev;
  }
  return NRT_OK;
}

// Ragged waves (secondary rays: every tile has its own number of them): tile-major gather.  Tile t traces counts[k] <= slot_rays
// rays into a buffer of slot_rays records; the root receives every tile's WHOLE slot at d_out + t * slot_rays (records past a
// tile's count are undefined) — no frame order to restore, so records arrive where they stay: RCCL receives straight into the
// caller's array, tiles on the root's own device are copied device to device.
template <int HIT_BYTES>
static nrt_status group_traverse_gather_tiles(nrt_group *g, const void *const *d_rays, const uint64_t *counts, uint64_t slot_rays,
                                              const nrt_trace_options *opt, uint32_t root_tile, void *d_tiles_hits, uint8_t *d_tiles_mask) {
  if (!g) return NRT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g->mu);
  const uint32_t N = g->num_tiles;
  if (root_tile >= N) return gfail(g, NRT_ERR_INVALID, "root tile out of range");
  if (!d_rays || !counts || slot_rays == 0) return gfail(g, NRT_ERR_INVALID, "NULL ray table / empty slot");
  int root_local = -1;
  for (size_t k = 0; k < g->local.size(); k++) {
    const nrt_group::Local &l = g->local[k];
    if (counts[k] > slot_rays) return gfail(g, NRT_ERR_INVALID, "tile " + std::to_string(l.tile) + ": " + std::to_string(counts[k]) + " rays do not fit its slot of " + std::to_string(slot_rays));
    if (counts[k] && !d_rays[k]) return gfail(g, NRT_ERR_INVALID, "tile " + std::to_string(l.tile) + ": NULL rays");
    if (l.tile == root_tile) root_local = (int)k;
  }
  const bool i_am_root = root_local >= 0;
  if (i_am_root && !d_tiles_hits) return gfail(g, NRT_ERR_INVALID, "the root needs an output buffer");
  const bool send_mask = g->ranked ? true : d_tiles_mask != nullptr;
  g->last_bytes_rccl = g->last_bytes_peer = g->last_bytes_in_place = 0;
  const size_t slot_b = (size_t)slot_rays * HIT_BYTES;
  for (size_t k = 0; k < g->local.size(); k++) {
    nrt_group::Local &l = g->local[k];
    GHIP(g, hipSetDevice(l.device));
    if (g->frame_done_device >= 0) GHIP(g, hipStreamWaitEvent(l.stream, g->frame_done, 0));
    GHIP(g, devbuf_ensure(&l.hits, slot_b));
    GHIP(g, devbuf_ensure(&l.mask, slot_rays));
    if (!counts[k]) continue;
    nrt_status st = HIT_BYTES == 16
                        ? nrtTraverseBatchDevice_f32(l.ctx, (const nrt_ray_f32 *)d_rays[k], counts[k], opt, (nrt_hit_f32 *)l.hits.p, (uint8_t *)l.mask.p, l.stream)
                        : nrtTraverseBatchDevice_f64(l.ctx, (const nrt_ray_f64 *)d_rays[k], counts[k], opt, (nrt_hit_f64 *)l.hits.p, (uint8_t *)l.mask.p, l.stream);
    if (st) return gfail(g, st, "tile " + std::to_string(l.tile) + ": " + nrtLastError(l.ctx));
  }
  const int root_dev = i_am_root ? g->local[root_local].device : -1;
  hipStream_t root_stream = i_am_root ? g->local[root_local].stream : nullptr;
  auto local_of = [&](uint32_t t) {
    for (size_t k = 0; k < g->local.size(); k++)
      if (g->local[k].tile == t) return (int)k;
    return -1;
  };
  auto in_place = [&](uint32_t t) { // the tile's records are on the root's device already: a device-to-device copy
    const int lk = local_of(t);
    return i_am_root && lk >= 0 && g->local[lk].device == root_dev && !(g->self_send && t == root_tile);
  };
  auto travels = [&](const nrt_group::Local &l) {
    if (g->ranked) return l.tile != root_tile || g->self_send != 0;
    return i_am_root && (l.device != root_dev || (g->self_send && l.tile == root_tile));
  };
  if (g->transport == 0 && !g->comms.empty()) {
    Rccl &r = rccl();
    bool any = false;
    for (size_t k = 0; k < g->local.size(); k++) any = any || travels(g->local[k]);
    if (i_am_root)
      for (uint32_t t = 0; t < N; t++) any = any || !in_place(t);
    if (any) {
      GNCCL(g, r.GroupStart());
      for (size_t k = 0; k < g->local.size(); k++) {
        nrt_group::Local &l = g->local[k];
        if (!travels(l)) continue;
        GHIP(g, hipSetDevice(l.device));
        GNCCL_IN_GROUP(g, r.Send(l.hits.p, slot_b, ncclUint8, g->tile_rank[root_tile], g->comms[l.comm], l.stream));
        if (send_mask) GNCCL_IN_GROUP(g, r.Send(l.mask.p, slot_rays, ncclUint8, g->tile_rank[root_tile], g->comms[l.comm], l.stream));
        g->last_bytes_rccl += slot_b + (send_mask ? slot_rays : 0);
      }
      if (i_am_root) {
        GHIP(g, hipSetDevice(root_dev));
        for (uint32_t t = 0; t < N; t++) {
          if (in_place(t)) continue;
          GNCCL_IN_GROUP(g, r.Recv((char *)d_tiles_hits + (size_t)t * slot_b, slot_b, ncclUint8, g->tile_rank[t], g->comms[g->local[root_local].comm], root_stream));
          if (send_mask) {
            // (the flags travel even when the root does not want them: a sender cannot know; they land in the tile's staging then)
            uint8_t *dst = d_tiles_mask ? d_tiles_mask + (size_t)t * slot_rays : nullptr;
            if (!dst) {
              GHIP(g, devbuf_ensure(&g->stage_mask[t], slot_rays));
              dst = (uint8_t *)g->stage_mask[t].p;
            }
            GNCCL_IN_GROUP(g, r.Recv(dst, slot_rays, ncclUint8, g->tile_rank[t], g->comms[g->local[root_local].comm], root_stream));
          }
        }
      }
      GNCCL(g, r.GroupEnd());
    }
  } else if (i_am_root) {
    for (size_t k = 0; k < g->local.size(); k++) {
      nrt_group::Local &l = g->local[k];
      if (!travels(l)) continue;
      GHIP(g, hipSetDevice(l.device));
      GHIP(g, hipMemcpyPeerAsync((char *)d_tiles_hits + (size_t)l.tile * slot_b, root_dev, l.hits.p, l.device, slot_b, l.stream));
      if (d_tiles_mask) GHIP(g, hipMemcpyPeerAsync(d_tiles_mask + (size_t)l.tile * slot_rays, root_dev, l.mask.p, l.device, slot_rays, l.stream));
      g->last_bytes_peer += slot_b + (d_tiles_mask ? slot_rays : 0);
    }
  } else if (g->ranked) {
    return gfail(g, NRT_ERR_INVALID, "a ranked group needs RCCL");
  }
  if (i_am_root) {
    for (size_t k = 0; k < g->local.size(); k++) { // tiles on the root's own device: copied on their own streams, which the root's then waits for
      nrt_group::Local &l = g->local[k];
      if (in_place(l.tile)) {
        GHIP(g, hipSetDevice(l.device));
        GHIP(g, hipMemcpyAsync((char *)d_tiles_hits + (size_t)l.tile * slot_b, l.hits.p, slot_b, hipMemcpyDeviceToDevice, l.stream));
        if (d_tiles_mask) GHIP(g, hipMemcpyAsync(d_tiles_mask + (size_t)l.tile * slot_rays, l.mask.p, slot_rays, hipMemcpyDeviceToDevice, l.stream));
        g->last_bytes_in_place += slot_b;
      }
      if ((int)k == root_local) continue;
      GHIP(g, hipSetDevice(l.device));
      GHIP(g, hipEventRecord(l.done, l.stream));
      GHIP(g, hipSetDevice(root_dev));
      GHIP(g, hipStreamWaitEvent(root_stream, l.done, 0));
    }
    GHIP(g, hipSetDevice(root_dev));
    if (!g->frame_done) GHIP(g, hipEventCreateWithFlags(&g->frame_done, hipEventDisableTiming));
    GHIP(g, hipEventRecord(g->frame_done, root_stream));
    g->frame_done_device = root_dev;
  }
  return NRT_OK;
}

extern "C" {

nrt_status nrtGroupTraverseGather_f32(nrt_group *g, const nrt_ray_f32 *const *d_rays, const uint64_t *counts, uint64_t total_rays, uint64_t row_len,
                                      const nrt_trace_options *opt, uint32_t root_tile, nrt_hit_f32 *d_frame_hits, uint8_t *d_frame_mask) {
  return group_traverse_gather<16>(g, (const void *const *)d_rays, counts, total_rays, row_len, opt, root_tile, d_frame_hits, d_frame_mask);
}
nrt_status nrtGroupTraverseGather_f64(nrt_group *g, const nrt_ray_f64 *const *d_rays, const uint64_t *counts, uint64_t total_rays, uint64_t row_len,
                                      const nrt_trace_options *opt, uint32_t root_tile, nrt_hit_f64 *d_frame_hits, uint8_t *d_frame_mask) {
  return group_traverse_gather<32>(g, (const void *const *)d_rays, counts, total_rays, row_len, opt, root_tile, d_frame_hits, d_frame_mask);
}
nrt_status nrtGroupTraverseGatherTiles_f32(nrt_group *g, const nrt_ray_f32 *const *d_rays, const uint64_t *counts, uint64_t slot_rays,
                                           const nrt_trace_options *opt, uint32_t root_tile, nrt_hit_f32 *d_tiles_hits, uint8_t *d_tiles_mask) {
  return group_traverse_gather_tiles<16>(g, (const void *const *)d_rays, counts, slot_rays, opt, root_tile, d_tiles_hits, d_tiles_mask);
}
nrt_status nrtGroupTraverseGatherTiles_f64(nrt_group *g, const nrt_ray_f64 *const *d_rays, const uint64_t *counts, uint64_t slot_rays,
                                           const nrt_trace_options *opt, uint32_t root_tile, nrt_hit_f64 *d_tiles_hits, uint8_t *d_tiles_mask) {
  return group_traverse_gather_tiles<32>(g, (const void *const *)d_rays, counts, slot_rays, opt, root_tile, d_tiles_hits, d_tiles_mask);
}

} // extern "C"

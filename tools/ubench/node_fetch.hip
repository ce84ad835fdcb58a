// tools/ubench/node_fetch.hip — what does a wave pay at the vector L1 (TCP) for fetching one 64-byte record per lane?
//   per-lane  : every lane issues 4 x global_load_dwordx4 over its own record (what k_traverse_wide's step does)
//   quad      : 4 instructions; in instruction j the four lanes of a quad read the four 16-byte pieces of the record of
//               the quad's lane j (one contiguous 64-byte access per quad), pieces not exchanged afterwards
// with all lanes or a random half of the lanes active.  Records are picked at random from an 18 MB table (the size of the
// C3 tree's WideNode array).  Build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE, int HALF>
__global__ __launch_bounds__(256) void k_fetch(const uint4 *__restrict__ table, uint32_t n_rec, int steps, uint32_t *out) {
  const unsigned lane = threadIdx.x & 63u;
  uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 1u;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; s++) {
    seed = mix(seed + s);
    const uint32_t idx = seed % n_rec;
    const bool on = HALF ? ((seed >> 20) & 1u) != 0u : true;
    if (MODE == 0) {
      if (on) {
        const uint4 *p = table + (size_t)idx * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y ^= a.y ^ b.z ^ c.w ^ d.x;
      }
    } else if (MODE == 2 || MODE == 3) {
      // quad loads, then every lane collects the four pieces of ITS record through LDS (MODE 3: the loads write LDS directly)
      __shared__ uint4 s_x[4][4][65];
      const unsigned w = threadIdx.x >> 6;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int src = (int)((lane & ~3u) + j);
        const uint32_t idx_j = __shfl(idx, src);
        const bool on_j = __shfl((int)on, src) != 0;
        if (on_j) {
          const uint4 *p = table + (size_t)idx_j * 4 + (lane & 3u);
          if (MODE == 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)&s_x[w][j][0], 16, 0, 0);
          else
            s_x[w][j][lane] = *p;
        }
      }
      if (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0f70 & 0); // vmcnt(0)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on) {
        const uint4 *q = &s_x[w][lane & 3u][lane & ~3u];
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y ^= a.y ^ b.z ^ c.w ^ d.x;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int src = (int)((lane & ~3u) + j);
        const uint32_t idx_j = __shfl(idx, src);
        const bool on_j = __shfl((int)on, src) != 0;
        if (on_j) {
          const uint4 a = table[(size_t)idx_j * 4 + (lane & 3u)];
          acc.x ^= a.x ^ a.w;
          acc.y ^= a.y ^ a.z;
        }
      }
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y;
}

int main(int argc, char **argv) {
  const uint32_t n_rec = argc > 1 ? (uint32_t)atoi(argv[1]) : 284000u; // records in the table
  uint4 *table; uint32_t *out;
  (void)hipMalloc(&table, (size_t)n_rec * 64);
  {
    std::vector<uint32_t> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
    (void)hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  const int grid = 256 * 6, steps = 2000;
  (void)hipMalloc(&out, grid * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<uint32_t> ref, got((size_t)grid * 256);
  auto run = [&](const char *name, void (*k)(const uint4 *, uint32_t, int, uint32_t *), double lanes_on, int check = 0) {
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, table, n_rec, steps, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    (void)hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
    if (check == 1) ref = got;
    if (check == 2 && got != ref) printf("   !! records differ from the per-lane fetch\n");
    const double recs = (double)grid * 256 * steps * lanes_on;
    printf("%-28s %8.3f ms   %7.2f G records/s   %6.2f TB/s of records\n", name, best, recs / best * 1e-6, recs * 64 / best * 1e-9);
  };
  printf("table: %u records, %.2f MB\n", n_rec, n_rec * 64e-6);
  run("per-lane, all lanes", k_fetch<0, 0>, 1.0, 1);
  run("per-lane, half the lanes", k_fetch<0, 1>, 0.5);
  run("quad, all lanes", k_fetch<1, 0>, 1.0);
  run("quad, half the lanes", k_fetch<1, 1>, 0.5);
  run("quad + LDS exchange, all", k_fetch<2, 0>, 1.0, 2);
  run("quad + LDS exchange, half", k_fetch<2, 1>, 0.5);
  run("quad -> LDS direct, all", k_fetch<3, 0>, 1.0, 2);
  run("quad -> LDS direct, half", k_fetch<3, 1>, 0.5);
  return 0;
}

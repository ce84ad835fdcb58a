// tools/ubench/launch_floor.hip — what does a chain of DEPENDENT kernel launches on one stream cost on this part, whatever the
// kernels do?  (The level-synchronous top phase of the builder is ~70 launches; profiles/r04j_launch_floor.txt)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor && tools/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(unsigned *p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void k_touch(unsigned *p, unsigned n) { // a little real work: one store per thread
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  unsigned *d;
  hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grids[] = {1, 64, 1024, 16384, 262144};
  for (int g : grids) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, d);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("200 dependent launches of an empty kernel, grid %7d x 256: %.2f us per launch\n", g, ms * 1000 / 200);
    }
  }
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, d, 1u << 20);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("200 dependent launches storing 4 MB each (grid 4096 x 256): %.2f us per launch\n", ms * 1000 / 200);
  }
  return 0;
}

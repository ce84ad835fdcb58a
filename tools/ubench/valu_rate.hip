// tools/ubench/valu_rate.hip — what does one wave64 VALU instruction cost a gfx950 SIMD?
// Settles the VALU peak used by bench.py's roofline.valu: N independent chains of one opcode, all 4 SIMDs of
// every CU loaded with W waves each; reports wave-instructions per SIMD per cycle (cycles from s_memtime-free
// wall time x the clock rocm-smi/hipDeviceProp reports, and from clock64()).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define ITER 4096

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float b = 1.0000001f, c = 1e-9f;
  const f2 pb = {b, b}, pc = {c, c};
  const unsigned long long t0 = clock64();
  for (int i = 0; i < ITER; i++) {
    if (OP == 0) { // v_fma_f32, 8 independent chains
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == 1) { // v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (OP == 2) { // v_cndmask_b32 with an SGPR-pair mask (vcc)
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    } else if (OP == 3) { // v_pk_mul_f32, 4 independent chains (8 multiplies)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
    } else if (OP == 4) { // v_max3_f32
      asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                   "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == 5) { // v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
    } else { // one dependent chain of v_fma_f32 (latency)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(a0) : "v"(b), "v"(c));
    }
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int waves_per_simd, float *d_out, unsigned long long *d_cyc, int cus) {
  // 256-thread blocks = one wave per SIMD; `waves_per_simd` blocks per CU
  const int grid = cus * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long *h = (unsigned long long *)malloc(sizeof(unsigned long long) * grid);
  hipMemcpy(h, d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < grid; i++) mean += (double)h[i];
  mean /= grid;
  free(h);
  const double insts_per_wave = 8.0 * ITER;
  // clock64() ticks at a fixed 100 MHz on this part; the wall figure is the one to read
  const double wave_insts_per_simd = insts_per_wave * waves_per_simd;
  printf("%-14s waves/SIMD %d  %8.3f ms  %6.2f ns per wave-instruction per SIMD  (= %.2f cycles @2.4 GHz)  clock64 ticks/inst %.3f\n", name,
         waves_per_simd, ms, ms * 1e6 / wave_insts_per_simd, ms * 1e6 / wave_insts_per_simd * 2.4, mean / insts_per_wave);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
  float *d_out;
  unsigned long long *d_cyc;
  hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
  hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 8);
  for (int w = 1; w <= 8; w *= 2) {
    run<0>("v_fma_f32", w, d_out, d_cyc, cus);
    run<1>("v_mul_f32", w, d_out, d_cyc, cus);
    run<2>("v_cndmask_b32", w, d_out, d_cyc, cus);
    run<3>("v_pk_mul_f32", w, d_out, d_cyc, cus);
    run<4>("v_max3_f32", w, d_out, d_cyc, cus);
    run<5>("v_pk_fma_f32", w, d_out, d_cyc, cus);
    run<6>("fma dep chain", w, d_out, d_cyc, cus);
  }
  return 0;
}

// Issue rate of integer multiplies on gfx950: v_mul_lo_u32 vs v_mul_i32_i24 vs v_mad_u64_u32 vs v_mad_i32_i24.
// One wave per SIMD (grid = 1024 single-wave workgroups), 8 independent chains per lane, so the
// pipe is issue-bound, not latency-bound.  Prints cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void k(int *out, int a0, int b0, int iters) {
  int x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x + i;
  int b = b0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 1) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 4) { long long r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x[i]), "v"(b) : "vcc"); x[i] = (int)r; }
      }
  }
  long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[65536] = (int)(t1 - t0);
}
int main() {
  int *d;
  hipMalloc(&d, (65536 + 16) * 4);
  const int iters = 2000;
  const char *names[5] = {"v_mul_lo_u32", "v_mul_i32_i24", "v_mad_i32_i24", "v_add_u32", "v_mad_u64_u32"};
  for (int kind = 0; kind < 5; kind++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      switch (kind) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, 3, 5, iters); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, d, 3, 5, iters); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, d, 3, 5, iters); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, d, 3, 5, iters); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), 0, 0, d, 3, 5, iters); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int cyc;
    hipMemcpy(&cyc, d + 65536, 4, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32;
    printf("%-14s  %.3f ms  %.2f ns/instr/wave  (s_memtime ticks/instr %.2f)\n", names[kind], ms, ms * 1e6 / n, cyc / n);
  }
  return 0;
}

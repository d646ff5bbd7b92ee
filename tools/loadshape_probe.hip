// What a "window row" costs in the texture-address / L1 path on gfx950: every lane fetches 24 bytes (12 samples) and
// neighbouring lanes are 8 bytes apart (adjacent 4x4 cells of a picture row), as in k_mc_cells.  Variants differ in
// the load shape and in the alignment of the lane address.  Also the issue rate of v_dot2_i32_i16.
// Build: hipcc --offload-arch=gfx950 -O3 -o loadshape_probe tools/loadshape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const char gchar;
// explicit instructions (the compiler would merge adjacent loads back into x4 + x2)
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define LD1(dst, q, o) asm volatile("global_load_dword %0, %1, off offset:" #o : "=v"(dst) : "v"(q) : "memory")
#define LD2(dst, q, o) asm volatile("global_load_dwordx2 %0, %1, off offset:" #o : "=v"(dst) : "v"(q) : "memory")
#define LD3(dst, q, o) asm volatile("global_load_dwordx3 %0, %1, off offset:" #o : "=v"(dst) : "v"(q) : "memory")
#define LD4(dst, q, o) asm volatile("global_load_dwordx4 %0, %1, off offset:" #o : "=v"(dst) : "v"(q) : "memory")
#define LDH(dst, q, o) asm volatile("global_load_sshort %0, %1, off offset:" #o : "=v"(dst) : "v"(q) : "memory")
template <int KIND>
__global__ __launch_bounds__(256) void k(const char *buf, int *out, int off, int rows, int row_bytes, size_t mask) {
  gchar *p = (gchar *)buf;
  const size_t wave = blockIdx.x * 4 + threadIdx.x / 64;
  size_t a = (wave * 64 * 8 * 11 + (threadIdx.x & 63) * (KIND == 3 ? 16 : 8) + off) & mask;
  int s = 0;
  for (int r = 0; r < rows; r++) {
    gchar *q = p + ((a + (size_t)r * row_bytes) & mask);
    int d[6] = {0, 0, 0, 0, 0, 0};
    if (KIND == 0) { v4i x; v2i y; LD4(x, q, 0); LD2(y, q, 16); asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y)); d[0] = x.x ^ x.y, d[1] = x.z ^ x.w, d[2] = y.x, d[3] = y.y; }
    if (KIND == 1) { v2i x, y, z; LD2(x, q, 0); LD2(y, q, 8); LD2(z, q, 16); asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y), "+v"(z)); d[0] = x.x ^ x.y, d[1] = y.x ^ y.y, d[2] = z.x, d[3] = z.y; }
    if (KIND == 2) { LD1(d[0], q, 0); LD1(d[1], q, 4); LD1(d[2], q, 8); LD1(d[3], q, 12); LD1(d[4], q, 16); LD1(d[5], q, 20);
                     asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])); }
    if (KIND == 3) { v4i x; LD4(x, q, 0); asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)); d[0] = x.x ^ x.y, d[1] = x.z ^ x.w; }
    if (KIND == 4) { v3i x, y; LD3(x, q, 0); LD3(y, q, 12); asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y)); d[0] = x.x ^ x.y, d[1] = x.z ^ y.x, d[2] = y.y, d[3] = y.z; }
    if (KIND == 5) { v2i x; LD2(x, q, 0); asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)); d[0] = x.x, d[1] = x.y; }
    if (KIND == 6) { int h[12]; LDH(h[0], q, 0); LDH(h[1], q, 2); LDH(h[2], q, 4); LDH(h[3], q, 6); LDH(h[4], q, 8); LDH(h[5], q, 10);
                     LDH(h[6], q, 12); LDH(h[7], q, 14); LDH(h[8], q, 16); LDH(h[9], q, 18); LDH(h[10], q, 20); LDH(h[11], q, 22);
                     asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]), "+v"(h[9]), "+v"(h[10]), "+v"(h[11]));
                     for (int i = 0; i < 12; i++) d[i / 2] += h[i]; }
    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
    s += d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
__global__ __launch_bounds__(64) void kv(int *out, int a0, int b0, int iters) {
  int x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x + i;
  int b = b0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(x[(i + 1) & 7]));
        if (KIND == 1) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(b), "v"(x[(i + 1) & 7]));
        if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(x[(i + 1) & 7]));
      }
  int s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND> float run(const char *buf, int *out, int off, int grid, int rows, int row_bytes, size_t mask) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, out, off, rows, row_bytes, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  const size_t bytes = 256u << 20;
  char *buf; int *out;
  hipMalloc(&buf, bytes + 4096); hipMemset(buf, 1, bytes + 4096);
  hipMalloc(&out, 8192 * 256 * 4);
  const int grid = 8192, rows = 44, row_bytes = 4096; // 32768 waves x 44 window rows
  const char *names[7] = {"x4+x2", "3 x x2", "6 x dword", "coalesced x4 (16 B/lane)", "2 x x3", "coalesced x2 (8 B/lane)", "12 x short"};
  for (int span = 0; span < 2; span++) {
    const size_t mask = (span ? bytes : (8u << 20)) - 1;
    printf("footprint %zu MiB\n", (mask + 1) >> 20);
    for (int kind = 0; kind < 7; kind++)
      for (int off : {0, 2, 4}) {
        if ((kind == 3 || kind == 5) && off) continue;
        float ms = 0;
        switch (kind) {
        case 0: ms = run<0>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 1: ms = run<1>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 2: ms = run<2>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 3: ms = run<3>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 4: ms = run<4>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 5: ms = run<5>(buf, out, off, grid, rows, row_bytes, mask); break;
        case 6: ms = run<6>(buf, out, off, grid, rows, row_bytes, mask); break;
        }
        const double wave_rows = (double)grid * 4 * rows;
        printf("  %-26s offset %d: %7.3f ms  %6.1f cycles/CU per wave-row (2.4 GHz)\n", names[kind], off, ms,
               ms * 1e-3 * 2.4e9 * 256 / wave_rows);
      }
  }
  const char *vn[4] = {"v_dot2c_i32_i16", "v_mad_i32_i24", "v_dot2_i32_i16 (vop3p)", "v_perm_b32"};
  for (int kind = 0; kind < 4; kind++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(kv<0>, dim3(1024), dim3(64), 0, 0, out, 3, 5, iters);
      if (kind == 1) hipLaunchKernelGGL(kv<1>, dim3(1024), dim3(64), 0, 0, out, 3, 5, iters);
      if (kind == 2) hipLaunchKernelGGL(kv<2>, dim3(1024), dim3(64), 0, 0, out, 3, 5, iters);
      if (kind == 3) hipLaunchKernelGGL(kv<3>, dim3(1024), dim3(64), 0, 0, out, 3, 5, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-24s %.2f cycles per wave instruction (one wave per SIMD)\n", vn[kind], ms * 1e-3 * 2.4e9 / (iters * 32.0));
  }
  return 0;
}

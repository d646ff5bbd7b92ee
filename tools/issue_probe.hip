// Two calibrations the round-2 verdict asked for (Weak 2 and 3), one binary:
//
//  (a) VALU issue rate against the number of waves resident on a SIMD.  tools/mulrate_probe.hip ran ONE wave per SIMD
//      (4 cycles per wave64 instruction); MI355X_MICROARCH.md says the SIMD-32 issues a wave64 instruction over 2 cycles,
//      so 2+ waves per SIMD should reach one instruction per 2 cycles per SIMD.  k_issue<KIND> runs W waves per SIMD
//      (grid = 256 CUs x 4 SIMDs x W single-wave workgroups, all resident) and reports SIMD cycles per instruction.
//  (b) What FETCH_SIZE counts for the access shapes k_intra_packed uses: 8-byte-per-lane loads (plain, non-temporal and
//      sc1 = relaxed agent-scope atomic), contiguous over the wave (512 B per instruction) or one 8-byte piece per
//      128-byte line, next to the 16-byte-per-lane streaming load the guide calibrated.  Each k_fetch_* kernel reads a
//      KNOWN number of bytes from a 2 GiB buffer (8x the Infinity Cache); run under
//        rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/bin/issue_probe fetch
//      and compare the counter (KB) per kernel with the "bytes" the program prints.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(64) void k_issue(int *out, int a0, int b0, int iters) {
  int x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x + i;
  int b = b0;
  unsigned long long msk = 0x5555555555555555ull + (unsigned)b0;
  int sc = b0 * 3;
  asm volatile("" : "+s"(msk), "+s"(sc));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 1) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 4) asm volatile("v_bfi_b32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 5) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 7) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
        if (KIND == 8) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x[i]));
        if (KIND == 9) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
        if (KIND == 11) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 12) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 13) asm volatile("v_med3_i32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 14) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 15) asm volatile("v_pk_mad_i16 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 16) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 17) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 18) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 19) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(x[i]));
        if (KIND == 20) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 21) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc");
        if (KIND == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 23) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 24) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 25) asm volatile("v_sad_u32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 26) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(x[i]) : "v"(b));
        if (KIND == 27) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 28) asm volatile("v_min3_i32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 29) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x[i]) : "v"(b));
        if (KIND == 30) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 31) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(long long*)&x[i & 6]));
        if (KIND == 33) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 34) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(b) : "vcc");
        if (KIND == 35) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 36) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(x[i]));
        if (KIND == 37) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(x[i]));
        if (KIND == 38) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 39) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 40) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc"); // 2 instructions
        if (KIND == 41) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "s"(msk));
        if (KIND == 42) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
        if (KIND == 43) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(b));
        if (KIND == 44) asm volatile("v_cmp_gt_i32_e64 %1, %0, %2" : : "v"(x[i]), "s"(msk), "v"(b));
        if (KIND == 45) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 46) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 47) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]));
        if (KIND == 48) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 49) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 50) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 51) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 52) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[i]));
        if (KIND == 53) asm volatile("v_readfirstlane_b32 %1, %0" : : "v"(x[i]), "s"(sc));
        if (KIND == 54) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "s"(sc));
        if (KIND == 55) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x[i]));
        if (KIND == 56) asm volatile("v_mul_i32_i24 %0, 83, %0" : "+v"(x[i]));
        if (KIND == 57) asm volatile("v_mad_i32_i24 %0, %0, 36, %1" : "+v"(x[i]) : "v"(b));
        if (KIND == 58) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
        if (KIND == 59) asm volatile("v_not_b32 %0, %0" : "+v"(x[i]));
      }
  }
  int s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 0x7fffffff) out[0] = s; // keeps the chain alive without a store per wave
}

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef int i4 __attribute__((ext_vector_type(4)));

// MODE 0: 16 B per lane, contiguous (the guide's calibration shape)
// MODE 1: 8 B per lane, contiguous, plain        MODE 2: 8 B per lane, contiguous, non-temporal
// MODE 3: 8 B per lane, contiguous, sc1 (relaxed agent-scope atomic load)
// MODE 4: 8 B per lane, ONE piece per 128-byte line, sc1  (the reference gather of a 4x4 / 8x8 block column)
// MODE 5: 8 B per lane, one piece per 128-byte line, plain
template <int MODE>
__global__ __launch_bounds__(256) void k_fetch(const char *buf, size_t bytes_per_thread_step, size_t n_steps, u64 *sink) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
  u64 acc = 0;
  for (size_t s = 0; s < n_steps; s++) {
    const size_t idx = s * nthreads + tid;
    if (MODE == 0) {
      const i4 v = *reinterpret_cast<const i4 *>(buf + idx * 16);
      acc += (u64)(unsigned)v[0] + (unsigned)v[1] + (unsigned)v[2] + (unsigned)v[3];
    } else if (MODE == 1) {
      acc += *reinterpret_cast<const u64 *>(buf + idx * 8);
    } else if (MODE == 2) {
      acc += __builtin_nontemporal_load(reinterpret_cast<const u64 *>(buf + idx * 8));
    } else if (MODE == 3) {
      acc += __hip_atomic_load((gu64 *)(buf + idx * 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 4) {
      acc += __hip_atomic_load((gu64 *)(buf + idx * 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 5) {
      acc += *reinterpret_cast<const u64 *>(buf + idx * 128);
    } else if (MODE == 6) { // 8 B of every 64-byte half line
      acc += *reinterpret_cast<const u64 *>(buf + idx * 64);
    } else if (MODE == 7) { // 8 B of every 32-byte sector
      acc += *reinterpret_cast<const u64 *>(buf + idx * 32);
    } else { // 8 B of every second 128-byte line
      acc += *reinterpret_cast<const u64 *>(buf + idx * 256);
    }
  }
  (void)bytes_per_thread_step;
  if (acc == 0x123456789abcdefull) sink[0] = acc;
}

// WRITE_SIZE: 8 B per lane, one piece per 128-byte line (a partially written line) vs whole lines
template <int MODE>
__global__ __launch_bounds__(256) void k_write(char *buf, size_t n_steps) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t s = 0; s < n_steps; s++) {
    const size_t idx = s * nthreads + tid;
    if (MODE == 0) *reinterpret_cast<u64 *>(buf + idx * 8) = idx;           // contiguous 8 B per lane
    else if (MODE == 1) *reinterpret_cast<u64 *>(buf + idx * 128) = idx;    // 8 B of every 128-byte line
    else if (MODE == 2) *reinterpret_cast<u64 *>(buf + idx * 128 + (idx & 1) * 64) = idx; // 8 B of every line, alternating halves
    else if (MODE == 3) __builtin_nontemporal_store((i4){(int)idx, 1, 2, 3}, reinterpret_cast<i4 *>(buf + idx * 16)); // 16 B per lane, nt
    else if (MODE == 4) { // a lane writes 64 contiguous bytes as four 16-byte NON-TEMPORAL stores (the levels of a 4x4 block, one lane per block)
#pragma unroll
      for (int q = 0; q < 4; q++) __builtin_nontemporal_store((i4){(int)idx, q, 2, 3}, reinterpret_cast<i4 *>(buf + idx * 64 + q * 16));
    } else if (MODE == 5) { // the same with plain stores
#pragma unroll
      for (int q = 0; q < 4; q++) *reinterpret_cast<i4 *>(buf + idx * 64 + q * 16) = (i4){(int)idx, q, 2, 3};
    } else if (MODE == 6) { // a lane writes 32 contiguous bytes as two plain 16-byte stores (the reconstruction of a 4x4 tile)
#pragma unroll
      for (int q = 0; q < 2; q++) *reinterpret_cast<i4 *>(buf + idx * 32 + q * 16) = (i4){(int)idx, q, 2, 3};
    } else { // 8 lanes write the 8 x 8-byte first-tile-rows of a 128-byte line pair: lane r -> line (idx / 8), piece (r & 3) * 8 + (r >> 2) * 32 ... then the second half
      const size_t line = idx >> 3, r = idx & 7;
      *reinterpret_cast<u64 *>(buf + line * 128 + (r >> 2) * 64 + (r & 3) * 8) = idx;
      *reinterpret_cast<u64 *>(buf + line * 128 + (r >> 2) * 64 + 32 + (r & 3) * 8) = idx;
    }
  }
}

static void run_issue() {
  int *d;
  hipMalloc(&d, 64);
  const int iters = 4000;
  const char *names[60] = {"v_add_u32", "v_mul_i32_i24", "v_mad_i32_i24", "v_mul_lo_u32", "v_bfi_b32", "v_max_i32", "v_and_b32", "v_lshlrev_b32",
                           "v_ashrrev_i32", "v_sub_u32", "v_cndmask_b32", "v_add3_u32", "v_lshl_add_u32", "v_med3_i32", "v_pk_add_i16", "v_pk_mad_i16",
                           "v_dot2_i32_i16", "v_perm_b32", "v_mov_b32", "v_bfe_i32", "v_xor_b32", "v_cmp_gt_i32", "v_mad_u32_u24", "v_pk_mul_lo_u16",
                           "v_pk_max_i16", "v_sad_u32", "v_add_lshl_u32", "v_and_or_b32", "v_min3_i32", "v_alignbit_b32", "v_add_f32", "v_fma_f32",
                           "v_pk_fma_f32", "v_mul_u32_u24", "v_add_co_u32", "v_pk_add_u16", "v_pk_lshlrev_b16", "v_pk_ashrrev_i16", "v_dot4_i32_i8",
                           "v_mul_hi_u32", "cmp+cndmask(2)", "v_cndmask_e64_sgpr", "v_mov_b32_dpp", "v_add_u32_dpp", "v_cmp_e64_sgpr", "v_max_i16", "v_or_b32",
                           "v_lshrrev_b32", "v_lshlrev_b32_vv", "v_min_i32", "v_subrev_u32", "v_mul_f32", "v_cvt_f32_i32", "v_readfirstlane", "v_add_u32_sgpr",
                           "v_add_u32_lit", "v_mul_i24_const", "v_mad_i24_const", "v_sub_u32_rev", "v_not_b32"};
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double ghz = prop.clockRate / 1e6;
  printf("device clock %.3f GHz, %d CUs\n", ghz, prop.multiProcessorCount);
  const int simds = prop.multiProcessorCount * 4;
  typedef void (*kern_t)(int *, int, int, int);
  kern_t kerns[60] = {k_issue<0>,  k_issue<1>,  k_issue<2>,  k_issue<3>,  k_issue<4>,  k_issue<5>,  k_issue<6>,  k_issue<7>,  k_issue<8>,  k_issue<9>,
                      k_issue<10>, k_issue<11>, k_issue<12>, k_issue<13>, k_issue<14>, k_issue<15>, k_issue<16>, k_issue<17>, k_issue<18>, k_issue<19>,
                      k_issue<20>, k_issue<21>, k_issue<22>, k_issue<23>, k_issue<24>, k_issue<25>, k_issue<26>, k_issue<27>, k_issue<28>, k_issue<29>,
                      k_issue<30>, k_issue<31>, k_issue<32>, k_issue<33>, k_issue<34>, k_issue<35>, k_issue<36>, k_issue<37>, k_issue<38>, k_issue<39>,
                      k_issue<40>, k_issue<41>, k_issue<42>, k_issue<43>, k_issue<44>, k_issue<45>, k_issue<46>, k_issue<47>, k_issue<48>, k_issue<49>,
                      k_issue<50>, k_issue<51>, k_issue<52>, k_issue<53>, k_issue<54>, k_issue<55>, k_issue<56>, k_issue<57>, k_issue<58>, k_issue<59>};
  for (int kind = (getenv("ISSUE_FROM") ? atoi(getenv("ISSUE_FROM")) : 0); kind < 60; kind++)
    for (int W : {1, 2, 4, 8}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kerns[kind], dim3(simds * W), dim3(64), 0, 0, d, 3, 5, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double n = (double)iters * 32;            // instructions per wave
      const double simd_cycles = best * 1e-3 * ghz * 1e9; // cycles every SIMD lived
      printf("%-16s W=%d waves/SIMD  %.3f ms  SIMD cycles per instruction %.2f  (per wave: %.2f)\n", names[kind], W, best,
             simd_cycles / (n * W), simd_cycles / n);
    }
  hipFree(d);
}

static void run_fetch() {
  const size_t total = (size_t)2 << 30; // 2 GiB
  char *buf;
  u64 *sink;
  if (hipMalloc(&buf, total) != hipSuccess) {
    printf("hipMalloc failed\n");
    return;
  }
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, total);
  hipDeviceSynchronize();
  const int wg = 4096, th = 256;
  const size_t nthreads = (size_t)wg * th;
  struct {
    int mode;
    size_t stride, useful;
    const char *what;
  } M[9] = {{6, 64, 8, "8B per 64B half line plain"}, {7, 32, 8, "8B per 32B sector plain"}, {8, 256, 8, "8B per 256B plain"},
            {0, 16, 16, "16B/lane contiguous plain"},   {1, 8, 8, "8B/lane contiguous plain"},   {2, 8, 8, "8B/lane contiguous nontemporal"},
            {3, 8, 8, "8B/lane contiguous sc1"},         {4, 128, 8, "8B per 128B line sc1"},     {5, 128, 8, "8B per 128B line plain"}};
  for (auto &m : M) {
    const size_t steps = total / (nthreads * m.stride);
    switch (m.mode) {
    case 0: hipLaunchKernelGGL(k_fetch<0>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 1: hipLaunchKernelGGL(k_fetch<1>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 2: hipLaunchKernelGGL(k_fetch<2>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 3: hipLaunchKernelGGL(k_fetch<3>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 4: hipLaunchKernelGGL(k_fetch<4>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 5: hipLaunchKernelGGL(k_fetch<5>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 6: hipLaunchKernelGGL(k_fetch<6>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 7: hipLaunchKernelGGL(k_fetch<7>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    case 8: hipLaunchKernelGGL(k_fetch<8>, dim3(wg), dim3(th), 0, 0, buf, m.stride, steps, sink); break;
    }
    hipDeviceSynchronize();
    printf("k_fetch<%d> %-32s useful bytes %zu  lines touched bytes %zu\n", m.mode, m.what, steps * nthreads * m.useful,
           steps * nthreads * (m.stride < 128 ? m.stride : 128));
  }
  for (int mode = 0; mode < 8; mode++) {
    const size_t stride = mode == 0 ? 8 : mode == 3 ? 16 : mode == 4 || mode == 5 ? 64 : mode == 6 ? 32 : mode == 7 ? 16 : 128, steps = total / (nthreads * stride);
    const size_t useful = mode == 1 || mode == 2 ? 8 : stride;
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_write<0>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 1: hipLaunchKernelGGL(k_write<1>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 2: hipLaunchKernelGGL(k_write<2>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 3: hipLaunchKernelGGL(k_write<3>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 4: hipLaunchKernelGGL(k_write<4>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 5: hipLaunchKernelGGL(k_write<5>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 6: hipLaunchKernelGGL(k_write<6>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    case 7: hipLaunchKernelGGL(k_write<7>, dim3(wg), dim3(th), 0, 0, buf, steps); break;
    }
    hipDeviceSynchronize();
    printf("k_write<%d> useful bytes %zu  lines touched bytes %zu\n", mode, steps * nthreads * useful, steps * nthreads * (stride < 128 ? stride : 128));
  }
  hipFree(buf);
  hipFree(sink);
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "fetch")) run_fetch();
  else run_issue();
  return 0;
}

// tools/rdoq_walk_probe/probe.hip -- how long does ONE walk of a coefficient group take (hmx_rdoq_core.h rdoq_walk_cg_in with
// the variants' sink, tables and coefficients in LDS as in rdoq_wave_tiles)?  One wave alone on the chip, then a full chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I thevc_amd/csrc -o gpurun_out/probe tools/rdoq_walk_probe/probe.hip && gpurun_out/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "hmx_kernels.h"
#include "hmx_rdoq.h"
using namespace hmx;

__global__ __launch_bounds__(64) void k_walk(const int *coefs, RdoqConst C0, const EstBitsDev *est, RdoqSpec *out, int last_pos, int reps, unsigned long long *ticks) {
  __shared__ TuLds<8> L;
  __shared__ RdoqWaveLdsT<1> W;
  const int lane = threadIdx.x;
  for (int i = lane; i < 64; i += 64) L.tile[i >> 3][i & 7] = coefs[i];
  for (int i = lane; i < 254; i += 64) ((int *)&W.est[0])[i] = ((const int *)est)[i];
  rdoq_stage_scan(W, 1, lane);
  __syncthreads();
  RdoqConst C = C0;
  auto bp_of = [&](int sp) { return (unsigned)W.scan[C.scan_idx * 64 + sp]; };
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
    RdoqSpecSink sink{&W.u.spec[lane]};
    const RdoqCgSums S = rdoq_walk_cg_in(C, W.est[0], (lane >> 3) & 3, bp_of, RdoqTileIn<8>{&L, &C}, lane & 3, (lane >> 2) & 1, last_pos, sink);
    W.u.spec[lane].S = S;
    wave_sync();
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  out[blockIdx.x * 64 + lane] = W.u.spec[lane];
}

int main() {
  EstBitsDev e;
  srand(7);
  int *ei = (int *)&e;
  for (int i = 0; i < 254; i++) ei[i] = 5000 + rand() % 60000;
  for (int style = 0; style < 3; style++) {
    std::vector<int> coef(64);
    for (int i = 0; i < 64; i++) coef[i] = style == 0 ? (rand() % 9) - 4 : style == 1 ? (rand() % 2001) - 1000 : ((rand() % 4) ? (rand() % 9) - 4 : (rand() % 401) - 200);
    RdoqConst C{};
    C.lg = 3, C.scan_idx = 0, C.is_luma = 1, C.q = 20560, C.qbits = 23, C.root_cbf = 0, C.cbf_ctx = 1, C.sign_hide = 1;
    C.lambda = 57.9, C.err_scale = 32768.0 * ldexp(1.0, -8) / 20560.0 / 20560.0, C.rd_factor = 1000;
    int *d_coef;
    EstBitsDev *d_est;
    RdoqSpec *d_out;
    unsigned long long *d_t, t;
    hipMalloc(&d_coef, 256), hipMalloc(&d_est, sizeof(e)), hipMalloc(&d_out, sizeof(RdoqSpec) * 64 * 4096), hipMalloc(&d_t, 8);
    hipMemcpy(d_coef, coef.data(), 256, hipMemcpyHostToDevice), hipMemcpy(d_est, &e, sizeof(e), hipMemcpyHostToDevice);
    for (int grid : {1, 256 * 4, 256 * 8}) {
      const int reps = 200;
      hipLaunchKernelGGL(k_walk, dim3(grid), dim3(64), 0, 0, d_coef, C, d_est, d_out, 63, reps, d_t);
      hipDeviceSynchronize();
      hipEvent_t a, b;
      hipEventCreate(&a), hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_walk, dim3(grid), dim3(64), 0, 0, d_coef, C, d_est, d_out, 63, reps, d_t);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, a, b);
      hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
      printf("style %d (%s), %d waves: %.2f us per walk in wave 0 (clock), %.2f us per walk by events\n", style,
             style == 0 ? "all levels zero" : style == 1 ? "dense levels" : "mixed", grid, t / 100.0 / reps, ms * 1e3 / reps);
    }
  }
  return 0;
}

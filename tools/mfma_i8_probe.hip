// Probe the operand lane/byte map of v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const signed char *A, const signed char *B, int *C, int variant) {
  // A: 32x32 row-major [row][k]; B: 32x32 [k][col]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  signed char a[16], b[16];
  for (int j = 0; j < 16; j++) {
    int k = variant == 0 ? 16 * h + j : (variant == 1 ? 8 * h + (j & 7) + 16 * (j >> 3) : 2 * j + h);
    a[j] = A[r * 32 + k];
    b[j] = B[k * 32 + r];
  }
  v4i av, bv;
  for (int q = 0; q < 4; q++) {
    av[q] = (a[4 * q] & 255) | ((a[4 * q + 1] & 255) << 8) | ((a[4 * q + 2] & 255) << 16) | ((a[4 * q + 3] & 255) << 24);
    bv[q] = (b[4 * q] & 255) | ((b[4 * q + 1] & 255) << 8) | ((b[4 * q + 2] & 255) << 16) | ((b[4 * q + 3] & 255) << 24);
  }
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
  for (int g = 0; g < 16; g++) {
    int row = (g & 3) + 8 * (g >> 2) + 4 * h;
    C[row * 32 + r] = c[g];
  }
}

int main() {
  std::vector<signed char> A(1024), B(1024);
  srand(1);
  for (auto &v : A) v = (signed char)(rand() % 256 - 128);
  for (auto &v : B) v = (signed char)(rand() % 256 - 128);
  std::vector<int> ref(1024, 0), got(1024);
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) {
      int s = 0;
      for (int k = 0; k < 32; k++) s += A[i * 32 + k] * B[k * 32 + j];
      ref[i * 32 + j] = s;
    }
  signed char *dA, *dB;
  int *dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  for (int v = 0; v < 3; v++) {
    hipMemset(dC, 0, 4096);
    probe<<<1, 64>>>(dA, dB, dC, v);
    hipMemcpy(got.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += got[i] != ref[i];
    printf("variant %d: %d mismatches\n", v, bad);
  }
  return 0;
}

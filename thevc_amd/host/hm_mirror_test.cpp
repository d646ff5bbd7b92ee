// hm_mirror_test.cpp -- drives the C++ host mirror (hmx_hm.hpp) the way TEncSearch::xIntraCodingLumaBlk
// does (ENC/TEncSearch.cpp:1006-1165): setQPforQuant, transformNxN, invtransformNxN on one block, and
// prints the results as text so that tests/test_host_mirror.py can compare them with the oracle.
// Usage: hm_mirror_test <bitDepth> <N> <qp> <mode> <seed>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "hmx_hm.hpp"

// Usage: hm_mirror_test recur <bitDepth> <qp> <seed>: a 32x32 inter CU whose transform tree splits down to 8x8 in two
// quadrants; prints the coefficient buffer (z-order) and the residual of invRecurTransformNxN.
static int recur_main(int argc, char **argv) {
  if (argc < 5) return 2;
  const int B = atoi(argv[2]), qp = atoi(argv[3]);
  unsigned seed = (unsigned)atoi(argv[4]);
  hmx_hm::Context ctx(B);
  hmx_hm::TComTrQuant tq(ctx);
  tq.setQPforQuant(qp, hmx_hm::TEXT_LUMA, 6 * (B - 8), 0);
  const int W = 32, parts = 64; // 64 partitions of 4x4 in a 32x32 CU (depth 1 of a 64x64 LCU)
  std::vector<unsigned char> trIdx(parts), cbf(parts);
  // quadrant 0: one 16x16 leaf (trIdx 1); quadrant 1: four 8x8 leaves (trIdx 2); quadrant 2: 16x16 not coded; quadrant 3: 8x8 leaves, two coded
  for (int p = 0; p < parts; p++) {
    const int quad = p / 16, sub = (p % 16) / 4;
    trIdx[p] = (quad == 1 || quad == 3) ? 2 : 1;
    unsigned char f = 1; // depth 0: something is coded in the CU
    if (quad != 2) f |= 2;
    if (quad == 1 || (quad == 3 && (sub == 0 || sub == 3))) f |= 4;
    cbf[p] = f;
  }
  std::vector<int> coef(W * W);
  for (auto &v : coef) {
    seed = seed * 1664525u + 1013904223u;
    v = ((seed >> 20) % 7 == 0) ? (int)((seed >> 8) % 41) - 20 : 0;
  }
  std::vector<short> resi(W * W, 0);
  hmx_hm::TComTrQuant::CuTransformTree cu{trIdx.data(), cbf.data(), nullptr, 1, 64, 256, (hmx_hm::UInt)parts, false};
  tq.invRecurTransformNxN(cu, 0, hmx_hm::TEXT_LUMA, resi.data(), 0, W, W, W, 2, 0, coef.data());
  for (int v : coef) printf("%d ", v);
  printf("\n");
  for (int v : resi) printf("%d ", v);
  printf("\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && std::string(argv[1]) == "recur") {
    try {
      return recur_main(argc, argv);
    } catch (const std::exception &e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
  }
  if (argc < 6) return 2;
  const int B = atoi(argv[1]), N = atoi(argv[2]), qp = atoi(argv[3]), mode = atoi(argv[4]);
  unsigned seed = (unsigned)atoi(argv[5]);
  try {
    hmx_hm::Context ctx(B);
    hmx_hm::TComTrQuant tq(ctx);
    std::vector<short> resi(N * N), rec(N * N);
    std::vector<int> lev(N * N);
    for (auto &v : resi) {
      seed = seed * 1664525u + 1013904223u;
      v = (short)((int)((seed >> 16) % 61) - 30);
    }
    tq.setQPforQuant(qp, hmx_hm::TEXT_LUMA, 6 * (B - 8), 0);
    tq.setBlockState(true, mode, HMX_I_SLICE, true);
    hmx_hm::UInt absSum = 0;
    tq.transformNxN(resi.data(), N, lev.data(), N, N, absSum, hmx_hm::TEXT_LUMA);
    tq.invtransformNxN(false, hmx_hm::TEXT_LUMA, mode, rec.data(), N, lev.data(), N, N, 0);
    printf("%u\n", absSum);
    for (int v : resi) printf("%d ", v);
    printf("\n");
    for (int v : lev) printf("%d ", v);
    printf("\n");
    for (int v : rec) printf("%d ", v);
    printf("\n");
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}

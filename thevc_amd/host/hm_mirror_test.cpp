// hm_mirror_test.cpp -- drives the C++ host mirror (hmx_hm.hpp) the way TEncSearch::xIntraCodingLumaBlk
// does (ENC/TEncSearch.cpp:1006-1165): setQPforQuant, transformNxN, invtransformNxN on one block, and
// prints the results as text so that tests/test_host_mirror.py can compare them with the oracle.
// Usage: hm_mirror_test <bitDepth> <N> <qp> <mode> <seed>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hmx_hm.hpp"

int main(int argc, char **argv) {
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

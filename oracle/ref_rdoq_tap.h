/* oracle/ref_rdoq_tap.h -- TEST INFRASTRUCTURE (oracle/build_ref_enc_shim.sh, the TAppEncoder_rdoqtap build).  Force-included in
 * front of the reference's TComTrQuant.cpp, whose xRateDistOptQuant gets ONE statement at its entry (oracle/ref_shim_edit.py
 * TComTrQuant rdoqtap) that constructs this recorder; everything else of the function is the reference's.  At scope exit -- the
 * early return included -- the recorder appends the call to the file named by the environment variable HMX_RDOQ_TAP:
 *   int32 header[16] = {magic, w, text type, qp, per, rem, bits, sign hide, intra, scan index, root cbf, cbf ctx, CU QP, POC, abs sum, table bytes}
 *   double lambda; the bit-estimate table as it stood (estBitsSbacStruct, TComTrQuant.h:59-72); int32 coef[w*w]; int32 level[w*w]
 * What the reference ENCODER fed its RDOQ and what came out, block by block: tests/golden/make_rdoq_enc_tap.py samples it into
 * the fixture tests/golden/rdoq_enc_tap.npz. */
#ifndef HMX_REF_RDOQ_TAP_H
#define HMX_REF_RDOQ_TAP_H
#include <cstdio>
#include <cstdlib>
#include <cstring>
struct HmxRdoqTap {
  int h[16];
  double lambda;
  unsigned char table[2048];
  const int *coef;
  const int *level;
  const unsigned *abs_sum;
  int *coef_copy;
  HmxRdoqTap(const void *est, unsigned long est_bytes, double lam, int qp, int per, int rem, int bits, int sign_hide, int intra, int scan, int root_cbf,
             int cbf_ctx, int ttype, int cu_qp, int poc, const int *src, const int *dst, int w, const unsigned *sum)
      : lambda(lam), coef(src), level(dst), abs_sum(sum), coef_copy(0) {
    const int v[16] = {0x52444f51, w, ttype, qp, per, rem, bits, sign_hide, intra, scan, root_cbf, cbf_ctx, cu_qp, poc, 0, (int)est_bytes};
    memcpy(h, v, sizeof(h));
    memcpy(table, est, est_bytes < sizeof(table) ? est_bytes : sizeof(table));
    if (getenv("HMX_RDOQ_TAP")) { /* the function may reuse its source buffer: keep the coefficients as they came in */
      coef_copy = (int *)malloc(sizeof(int) * w * w);
      memcpy(coef_copy, src, sizeof(int) * w * w);
    }
  }
  ~HmxRdoqTap() {
    const char *path = getenv("HMX_RDOQ_TAP");
    if (!path || !coef_copy) return;
    FILE *f = fopen(path, "ab");
    if (f) {
      h[14] = (int)*abs_sum;
      const int n = h[1] * h[1];
      fwrite(h, sizeof(h), 1, f);
      fwrite(&lambda, sizeof(lambda), 1, f);
      fwrite(table, h[15], 1, f);
      fwrite(coef_copy, sizeof(int), n, f);
      fwrite(level, sizeof(int), n, f);
      fclose(f);
    }
    free(coef_copy);
  }
};
#endif

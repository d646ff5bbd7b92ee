// ref_tap.cpp -- tap harness around the REFERENCE's own hot-path functions.  TEST INFRASTRUCTURE.
//
// Compiled by oracle/build_ref.sh together with /root/reference's TLibCommon objects into
// oracle/_ref/libhmref.so (never committed, never shipped as product).  It contains no
// reference code: it includes the reference headers where they lie, opens their private /
// protected sections, and forwards plain-C calls to the reference's member functions so that
// tests/ can (a) check oracle/hmx_oracle.c against the real thing and (b) generate the golden
// vectors of tests/golden/.
#define private public
#define protected public
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComInterpolationFilter.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibCommon/TComPattern.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComPicYuv.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComSampleAdaptiveOffset.h"
#include "TLibCommon/TComSlice.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComYuv.h"
#include "TLibVideoIO/TVideoIOYuv.h"
#undef private
#undef protected

#include <cstring>

// free functions with external linkage in TComTrQuant.cpp (MATRIX_MULT 0)
void partialButterfly4(short *src, short *dst, int shift, int line);
void partialButterfly8(short *src, short *dst, int shift, int line);
void partialButterfly16(short *src, short *dst, int shift, int line);
void partialButterfly32(short *src, short *dst, int shift, int line);
void partialButterflyInverse4(short *src, short *dst, int shift, int line);
void partialButterflyInverse8(short *src, short *dst, int shift, int line);
void partialButterflyInverse16(short *src, short *dst, int shift, int line);
void partialButterflyInverse32(short *src, short *dst, int shift, int line);
void fastForwardDst(short *block, short *coeff, int shift);
void fastInverseDst(short *tmp, short *block, int shift);
void xTrMxN(short *block, short *coeff, int iWidth, int iHeight, UInt uiMode);
void xITrMxN(short *coeff, short *block, int iWidth, int iHeight, UInt uiMode);

namespace {
struct State {
  TComTrQuant tq;
  TComPrediction pred;
  TComPattern pattern;
  TComInterpolationFilter filt;
  TComRdCost rd;
  TComSPS sps;
  TComPPS pps;
  TComPic *pic = nullptr;
  TComDataCU sub; // scratch "temp CU" like the encoder's m_ppcTempCU
  TComYuv yuv[3];
  int pic_w = 0, pic_h = 0;
  bool rom = false;
};
State *S = nullptr;
} // namespace

extern "C" {

// bit depth / globals as TAppEncCfg::xSetGlobal does for the shipped cfgs (CTU 64, depth 4)
int ref_init(int bit_depth, int pic_w, int pic_h, int sign_hide) {
  if (!S) S = new State;
  g_uiMaxCUWidth = g_uiMaxCUHeight = 64;
  g_uiMaxCUDepth = 4;
  g_uiAddCUDepth = 1;
  g_uiBitDepth = 8;
  g_uiBitIncrement = bit_depth - 8;
  g_uiBASE_MAX = 255;
  g_uiIBDI_MAX = (1u << bit_depth) - 1;
  if (!S->rom) {
    initROM();
    UInt *p = &g_auiZscanToRaster[0];
    initZscanToRaster(5, 1, 0, p); // total depth + 1, as TEncCu::create (ENC/TEncCu.cpp:63,101)
    initRasterToZscan(64, 64, 5);
    initRasterToPelXY(64, 64, 5);
    initMotionReferIdx(64, 64, 5); // total depth + 1, as TEncCu::create (ENC/TEncCu.cpp:63,106)
    S->tq.init(64, 64, 32, 0, NULL, NULL, NULL, false, true, true, false);
    S->tq.setFlatScalingList();
    S->tq.setUseScalingList(false);
    S->pred.initTempBuff();
    S->rd.init();
    S->sub.create(256, 64, 64, false, 4, true);
    for (int i = 0; i < 3; i++) S->yuv[i].create(64, 64);
    S->rom = true;
  }
  S->sps.setMaxCUWidth(64);
  S->sps.setMaxCUHeight(64);
  S->sps.setMaxCUDepth(4);
  S->sps.setMaxTrSize(32);
  S->sps.setQpBDOffsetY(6 * (bit_depth - 8));
  S->sps.setQpBDOffsetC(6 * (bit_depth - 8));
  S->tq.setFlatScalingList(); // the error-scale tables of RDOQ depend on the bit depth set above
  S->pps.setSignHideFlag(sign_hide);
  S->pps.setConstrainedIntraPred(false);
  S->pps.setSPS(&S->sps);
  if (pic_w > 0 && (pic_w != S->pic_w || pic_h != S->pic_h)) {
    // the previous picture is leaked on purpose: TComPicSym::destroy() walks the tile array that
    // only the full encoder allocates
    S->pic = new TComPic;
    S->pic->create(pic_w, pic_h, 64, 64, 4);
    S->pic_w = pic_w;
    S->pic_h = pic_h;
  }
  if (S->pic) {
    S->sps.setPicWidthInLumaSamples(S->pic_w);
    S->sps.setPicHeightInLumaSamples(S->pic_h);
    TComSlice *sl = S->pic->getSlice(0);
    sl->setSPS(&S->sps);
    sl->setPPS(&S->pps);
    sl->setSliceType(I_SLICE);
    sl->setSliceCurStartCUAddr(0);
    sl->setDependentSliceCurStartCUAddr(0);
    // one tile: the encoder's tile set-up (ENC/TEncGOP.cpp, TComPicSym::xInitTiles) leaves index 0
    for (UInt a = 0; a < S->pic->getNumCUsInFrame(); a++) S->pic->getPicSym()->m_puiTileIdxMap[a] = 0;
    for (UInt a = 0; a < S->pic->getNumCUsInFrame(); a++) S->pic->getCU(a)->initCU(S->pic, a);
  }
  return 0;
}

void ref_tables(short *t4, short *t8, short *t16, short *t32, short *dst4, int *q, int *iq,
                unsigned char *chroma58) {
  memcpy(t4, g_aiT4, sizeof(g_aiT4));
  memcpy(t8, g_aiT8, sizeof(g_aiT8));
  memcpy(t16, g_aiT16, sizeof(g_aiT16));
  memcpy(t32, g_aiT32, sizeof(g_aiT32));
  memcpy(dst4, g_as_DST_MAT_4, sizeof(g_as_DST_MAT_4));
  memcpy(q, g_quantScales, 6 * sizeof(int));
  memcpy(iq, g_invQuantScales, 6 * sizeof(int));
  memcpy(chroma58, g_aucChromaScale, 58);
}
void ref_scan(int scan_idx, int log2n, unsigned *out) {
  memcpy(out, g_auiSigLastScan[scan_idx][log2n - 1], sizeof(unsigned) << (2 * log2n));
}

void ref_partialButterfly(int N, short *src, short *dst, int shift, int line) {
  if (N == 4) partialButterfly4(src, dst, shift, line);
  if (N == 8) partialButterfly8(src, dst, shift, line);
  if (N == 16) partialButterfly16(src, dst, shift, line);
  if (N == 32) partialButterfly32(src, dst, shift, line);
}
void ref_partialButterflyInverse(int N, short *src, short *dst, int shift, int line) {
  if (N == 4) partialButterflyInverse4(src, dst, shift, line);
  if (N == 8) partialButterflyInverse8(src, dst, shift, line);
  if (N == 16) partialButterflyInverse16(src, dst, shift, line);
  if (N == 32) partialButterflyInverse32(src, dst, shift, line);
}
void ref_fastForwardDst(short *b, short *c, int shift) { fastForwardDst(b, c, shift); }
void ref_fastInverseDst(short *t, short *b, int shift) { fastInverseDst(t, b, shift); }
void ref_xTrMxN(short *b, short *c, int N, unsigned mode) { xTrMxN(b, c, N, N, mode); }
void ref_xITrMxN(short *c, short *b, int N, unsigned mode) { xITrMxN(c, b, N, N, mode); }

void ref_xT(unsigned mode, short *resi, unsigned stride, int *coef, int N) {
  S->tq.xT(mode, resi, stride, coef, N, N);
}
void ref_xIT(unsigned mode, int *coef, short *resi, unsigned stride, int N) {
  S->tq.xIT(mode, coef, resi, stride, N, N);
}
void ref_xTransformSkip(short *resi, unsigned stride, int *coef, int N) {
  S->tq.xTransformSkip(resi, stride, coef, N, N);
}
void ref_xITransformSkip(int *coef, short *resi, unsigned stride, int N) {
  S->tq.xITransformSkip(coef, resi, stride, N, N);
}
void ref_setQPforQuant(int qpy, int ttype, int bd_off, int c_off, int *out3) {
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd_off, c_off);
  out3[0] = S->tq.m_cQP.m_iQP;
  out3[1] = S->tq.m_cQP.m_iPer;
  out3[2] = S->tq.m_cQP.m_iRem;
}
void ref_xDeQuant(int qpy, int ttype, int bd_off, int c_off, const int *src, int *dst, int N) {
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd_off, c_off);
  S->tq.xDeQuant(src, dst, N, N, 0);
}

// transformNxN / invtransformNxN through CTU 0 of the picture with the per-TU CU state the
// encoder would have set (ENC/TEncSearch.cpp:1006-1165): intra CU, given dir mode / TS flag.
void ref_transformNxN(int qpy, int slice_type /*0 B,1 P,2 I*/, int ttype, int is_intra, int dir_mode,
                      int ts, int bypass, short *resi, unsigned stride, int *level, int N,
                      unsigned *abs_sum) {
  TComDataCU *cu = S->pic->getCU(0);
  TComSlice *sl = S->pic->getSlice(0);
  sl->setSliceType((SliceType)slice_type);
  sl->setSliceQp(qpy);
  sl->setSliceQpBase(qpy);
  cu->m_pcSlice = sl;
  cu->m_pePredMode[0] = is_intra ? MODE_INTRA : MODE_INTER;
  cu->m_puhLumaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhChromaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhDepth[0] = 0;
  cu->m_CUTransquantBypass[0] = bypass;
  for (int t = 0; t < 3; t++) cu->m_puhTransformSkip[t][0] = (UChar)ts;
  Int bd = (ttype == 0) ? S->sps.getQpBDOffsetY() : S->sps.getQpBDOffsetC();
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd, 0);
  Int *arl = NULL;
  UInt sum = 0;
  S->tq.transformNxN(cu, resi, stride, level, arl, N, N, sum, (TextType)ttype, 0, ts != 0);
  *abs_sum = sum;
}
// xRateDistOptQuant with a caller-supplied bit-estimate table (what TEncSbac::estBit would have left in
// m_pcEstBitsSbac for this block size / texture type) and Lagrange multiplier.
int ref_sizeof_estbits() { return (int)sizeof(estBitsSbacStruct); }
int ref_cbf_ctx(int ttype, int tr_idx) {
  TComDataCU *cu = S->pic->getCU(0);
  int ctx = (int)cu->getCtxQtCbf(0, (TextType)ttype, (UInt)tr_idx);
  return (ttype ? TEXT_CHROMA : ttype) * NUM_QT_CBF_CTX + ctx;
}
void ref_xRateDistOptQuant(int qpy, int slice_type, int ttype, int is_intra, int dir_mode, int tr_idx, double lambda,
                           const int *est_blob, int *coef, int *level, int N, unsigned *abs_sum) {
  TComDataCU *cu = S->pic->getCU(0);
  TComSlice *sl = S->pic->getSlice(0);
  sl->setSliceType((SliceType)slice_type);
  sl->setSliceQp(qpy);
  sl->setSliceQpBase(qpy);
  cu->m_pcSlice = sl;
  cu->m_pePredMode[0] = is_intra ? MODE_INTRA : MODE_INTER;
  cu->m_puhLumaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhChromaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhDepth[0] = 0;
  cu->m_puhTrIdx[0] = (UChar)tr_idx;
  cu->m_CUTransquantBypass[0] = false;
  Int bd = (ttype == 0) ? S->sps.getQpBDOffsetY() : S->sps.getQpBDOffsetC();
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd, 0);
  memcpy(S->tq.m_pcEstBitsSbac, est_blob, sizeof(estBitsSbacStruct));
  S->tq.m_dLambda = lambda;
  static Int arl_buf[32 * 32];
  Int *arl = arl_buf;
  UInt sum = 0;
  S->tq.xRateDistOptQuant(cu, coef, level, arl, N, N, sum, (TextType)ttype, 0);
  *abs_sum = sum;
}
// xDeQuant's scaling-list branch with a caller-supplied table: the table of (list type, QP remainder, size) is overwritten with
// dq_coef for the call and restored to the flat one afterwards.
void ref_xDeQuant_scaled(int qpy, int ttype, int bd_off, int list_type, const int *dq_coef, const int *src, int *dst, int N) {
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd_off, 0);
  const UInt size_id = g_aucConvertToBit[N];
  Int *tab = S->tq.getDequantCoeff(list_type, S->tq.m_cQP.m_iRem, size_id, SCALING_LIST_SQT);
  memcpy(tab, dq_coef, sizeof(Int) * N * N);
  S->tq.setUseScalingList(true);
  S->tq.xDeQuant(src, dst, N, N, list_type);
  S->tq.setUseScalingList(false);
  S->tq.xsetFlatScalingList(list_type, size_id, S->tq.m_cQP.m_iRem);
}
// xQuant as the encoder runs it under AdaptiveQpSelection: the slice's base QP differs from the block's QP (cQpBase,
// TComTrQuant.cpp:1162-1193), m_bUseAdaptQpSelect is on and pArlDes receives the ARL coefficients -- from the flat branch or,
// with rdoq != 0, from xRateDistOptQuant (est_blob / lambda as in ref_xRateDistOptQuant).
// q_coef / err_scale != NULL: the quantiser's per-position tables of a scaling list (getQuantCoeff / getErrScaleCoeff of the block's
// list type, QP remainder and size) are overwritten for the call and restored to the flat ones afterwards.
void ref_xQuant_arl(int qpy, int qp_base, int slice_type, int ttype, int is_intra, int dir_mode, int tr_idx, int rdoq, double lambda,
                    const int *est_blob, int *coef, int *level, int *arl, int N, unsigned *abs_sum, const int *q_coef, const double *err_scale) {
  TComDataCU *cu = S->pic->getCU(0);
  TComSlice *sl = S->pic->getSlice(0);
  sl->setSliceType((SliceType)slice_type);
  sl->setSliceQp(qpy);
  sl->setSliceQpBase(qp_base);
  cu->m_pcSlice = sl;
  cu->m_pePredMode[0] = is_intra ? MODE_INTRA : MODE_INTER;
  cu->m_puhLumaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhChromaIntraDir[0] = (UChar)dir_mode;
  cu->m_puhDepth[0] = 0;
  cu->m_puhTrIdx[0] = (UChar)tr_idx;
  cu->m_CUTransquantBypass[0] = false;
  for (int t = 0; t < 3; t++) cu->m_puhTransformSkip[t][0] = 0;
  Int bd = (ttype == 0) ? S->sps.getQpBDOffsetY() : S->sps.getQpBDOffsetC();
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd, 0);
  const Bool was_rdoq = S->tq.m_bUseRDOQ, was_arl = S->tq.m_bUseAdaptQpSelect;
  S->tq.m_bUseRDOQ = rdoq != 0;
  S->tq.m_bUseAdaptQpSelect = true;
  if (rdoq) {
    memcpy(S->tq.m_pcEstBitsSbac, est_blob, sizeof(estBitsSbacStruct));
    S->tq.m_dLambda = lambda;
  }
  const UInt size_id = g_aucConvertToBit[N], list_type = (is_intra ? 0 : 3) + g_eTTable[ttype];
  if (q_coef) memcpy(S->tq.getQuantCoeff(list_type, S->tq.m_cQP.m_iRem, size_id, SCALING_LIST_SQT), q_coef, sizeof(Int) * N * N);
  if (err_scale) memcpy(S->tq.getErrScaleCoeff(list_type, size_id, S->tq.m_cQP.m_iRem, SCALING_LIST_SQT), err_scale, sizeof(double) * N * N);
  Int *a = arl;
  UInt sum = 0;
  S->tq.xQuant(cu, coef, level, a, N, N, sum, (TextType)ttype, 0);
  *abs_sum = sum;
  if (q_coef || err_scale) {
    S->tq.xsetFlatScalingList(list_type, size_id, S->tq.m_cQP.m_iRem);
    S->tq.setErrScaleCoeff(list_type, size_id, S->tq.m_cQP.m_iRem, SCALING_LIST_SQT);
  }
  S->tq.m_bUseRDOQ = was_rdoq;
  S->tq.m_bUseAdaptQpSelect = was_arl;
  sl->setSliceQpBase(qpy);
}
void ref_invtransformNxN(int qpy, int ttype, int bypass, unsigned mode, short *resi, unsigned stride,
                         int *level, int N, int ts) {
  Int bd = (ttype == 0) ? S->sps.getQpBDOffsetY() : S->sps.getQpBDOffsetC();
  S->tq.setQPforQuant(qpy, (TextType)ttype, bd, 0);
  S->tq.invtransformNxN(bypass != 0, (TextType)ttype, mode, resi, stride, level, N, N, 0, ts != 0);
}

// ---- intra ----
void ref_fillReferenceSamples(short *rec_origin, int stride, const unsigned char *flags, int n_avail,
                              int unit, int N, int *adi) {
  Bool f[4 * 32 + 1];
  int n = N / unit, total = 4 * n + 1;
  for (int i = 0; i < total; i++) f[i] = flags[i] != 0;
  S->pattern.fillReferenceSamples(NULL, rec_origin, adi, f, n_avail, unit, n, total, N, N, 2 * N + 1,
                                  2 * N + 1, stride, false);
}

// set the picture's reconstruction planes (test input for initAdiPattern)
void ref_set_recon(const short *y, const short *cb, const short *cr) {
  TComPicYuv *r = S->pic->getPicYuvRec();
  for (int j = 0; j < S->pic_h; j++)
    memcpy(r->getLumaAddr() + j * r->getStride(), y + j * S->pic_w, 2 * S->pic_w);
  for (int j = 0; j < S->pic_h / 2; j++) {
    memcpy(r->getCbAddr() + j * r->getCStride(), cb + j * (S->pic_w / 2), S->pic_w);
    memcpy(r->getCrAddr() + j * r->getCStride(), cr + j * (S->pic_w / 2), S->pic_w);
  }
}

static TComDataCU *setup_sub(int x, int y, int cu_size) {
  UInt wcu = S->pic->getFrameWidthInCU();
  UInt addr = (y / 64) * wcu + (x / 64);
  TComDataCU *ctu = S->pic->getCU(addr);
  TComDataCU *c = &S->sub;
  c->m_pcPic = S->pic;
  c->m_pcSlice = S->pic->getSlice(0);
  c->m_uiCUAddr = addr;
  c->m_uiCUPelX = x;
  c->m_uiCUPelY = y;
  UInt raster = ((y % 64) / 4) * 16 + (x % 64) / 4;
  c->m_uiAbsIdxInLCU = g_auiRasterToZscan[raster];
  c->m_uiNumPartition = (cu_size / 4) * (cu_size / 4);
  for (UInt i = 0; i < c->m_uiNumPartition; i++) {
    c->m_puhWidth[i] = c->m_puhHeight[i] = (UChar)cu_size;
    c->m_pePredMode[i] = MODE_INTRA;
  }
  c->m_pcCULeft = ctu->m_pcCULeft;
  c->m_pcCUAbove = ctu->m_pcCUAbove;
  c->m_pcCUAboveLeft = ctu->m_pcCUAboveLeft;
  c->m_pcCUAboveRight = ctu->m_pcCUAboveRight;
  for (UInt i = 0; i < 256; i++) {
    c->m_uiSliceStartCU[i] = 0;
    c->m_uiDependentSliceStartCU[i] = 0;
  }
  return c;
}

// initAdiPattern for the block at luma (x,y): cu_size = CU size, part_depth / part_z select the
// TU inside it (the 4x4 case is CU 8, depth 1).  Copies the (2N+1)^2 x2 buffer out.
void ref_initAdiPattern(int x_cu, int y_cu, int cu_size, int part_depth, int part_z, int *adi_out) {
  TComDataCU *c = setup_sub(x_cu, y_cu, cu_size);
  Bool a, l;
  S->pattern.initAdiPattern(c, part_z, part_depth, S->pred.m_piYuvExt, S->pred.m_iYuvExtStride,
                            S->pred.m_iYuvExtHeight, a, l, false);
  int N = cu_size >> part_depth, W = 2 * N + 1;
  memcpy(adi_out, S->pred.m_piYuvExt, sizeof(int) * 2 * W * W);
}
void ref_initAdiPatternChroma(int x_cu, int y_cu, int cu_size, int part_depth, int part_z,
                              int *adi_out) {
  TComDataCU *c = setup_sub(x_cu, y_cu, cu_size);
  Bool a, l;
  S->pattern.initAdiPatternChroma(c, part_z, part_depth, S->pred.m_piYuvExt, S->pred.m_iYuvExtStride,
                                  S->pred.m_iYuvExtHeight, a, l);
  int N = (cu_size >> part_depth) >> 1, W = 2 * N + 1;
  memcpy(adi_out, S->pred.m_piYuvExt, sizeof(int) * 2 * W * W); // Cb buffer then Cr buffer
}

void ref_predIntraLumaAng(const int *adi, int mode, short *dst, unsigned stride, int N) {
  int W = 2 * N + 1;
  memcpy(S->pred.m_piYuvExt, adi, sizeof(int) * 2 * W * W);
  S->pred.predIntraLumaAng(&S->pattern, mode, dst, stride, N, N, NULL, true, true);
}
void ref_predIntraChromaAng(const int *adi, int mode, short *dst, unsigned stride, int N) {
  int W = 2 * N + 1;
  memcpy(S->pred.m_piYuvExt, adi, sizeof(int) * W * W);
  S->pred.predIntraChromaAng(&S->pattern, S->pred.m_piYuvExt, mode, dst, stride, N, N, NULL, true,
                             true);
}

// the protected building blocks named by the north star, on a caller-supplied border buffer
int ref_predIntraGetPredValDC(const int *adi, int N, int above, int left) {
  int W = 2 * N + 1;
  return S->pred.predIntraGetPredValDC(const_cast<int *>(adi) + W + 1, W, N, N, above != 0, left != 0);
}
void ref_xPredIntraAng(const int *adi, int N, int mode, int above, int left, int filter, short *dst) {
  int W = 2 * N + 1;
  Pel *d = dst;
  S->pred.xPredIntraAng(const_cast<int *>(adi) + W + 1, W, d, N, N, N, mode, above != 0, left != 0, filter != 0);
}
void ref_xPredIntraPlanar(const int *adi, int N, short *dst) {
  int W = 2 * N + 1;
  S->pred.xPredIntraPlanar(const_cast<int *>(adi) + W + 1, W, dst, N, N, N);
}

// ---- deblocking, application part: the reference's own edge filters driven with caller-supplied boundary strengths.
// The picture set by ref_init / ref_set_recon must be a whole number of 64x64 LCUs.  Maps are per 4x4 luma unit,
// raster order: bs_ver[u] = strength of the edge on the unit's left side, bs_hor[u] = on its top side.  Every LCU is
// driven as ONE depth-0 CU, edge by edge, as xDeblockCU does (all vertical edges of the picture, then all horizontal).
void ref_deblock_picture(const unsigned char *bs_ver, const unsigned char *bs_hor, const signed char *qp, const unsigned char *no_filter,
                         int beta_off_div2, int tc_off_div2, short *y, short *cb, short *cr) {
  TComPic *pic = S->pic;
  const int w = S->pic_w, h = S->pic_h, uw = w / 4, lw = w / 64, lh = h / 64;
  static TComLoopFilter *lf = nullptr;
  if (!lf) {
    lf = new TComLoopFilter;
    lf->create(4);
  }
  lf->setCfg(true, 0, beta_off_div2, tc_off_div2, true);
  TComSlice *sl = pic->getSlice(0);
  sl->setLFCrossSliceBoundaryFlag(true);
  // no_filter is driven through the IPCM + pcm_loop_filter_disable path (:609-614), where the P/Q flags are re-read
  // for every unit; the lossless path ORs them into flags that stay set for the rest of the CU's edge (:616-617)
  S->sps.setUsePCM(no_filter != NULL);
  S->sps.setPCMFilterDisableFlag(no_filter != NULL);
  S->pps.setTransquantBypassEnableFlag(false);
  for (int a = 0; a < lw * lh; a++) {
    TComDataCU *cu = pic->getCU(a);
    cu->initCU(pic, a);
    const int lx = (a % lw) * 16, ly = (a / lw) * 16;
    for (int r = 0; r < 256; r++) {
      const int z = g_auiRasterToZscan[r], u = (ly + r / 16) * uw + lx + r % 16;
      cu->m_phQP[z] = qp[u];
      cu->m_pbIPCMFlag[z] = no_filter ? no_filter[u] != 0 : false;
      cu->m_CUTransquantBypass[z] = false;
    }
  }
  for (int dir = 0; dir < 2; dir++)
    for (int a = 0; a < lw * lh; a++) {
      TComDataCU *cu = pic->getCU(a);
      const int lx = (a % lw) * 16, ly = (a / lw) * 16;
      const unsigned char *bs = dir ? bs_hor : bs_ver;
      for (int r = 0; r < 256; r++) lf->m_aapucBS[dir][g_auiRasterToZscan[r]] = bs[(ly + r / 16) * uw + lx + r % 16];
      for (int e = 0; e < 16; e += 2) {
        lf->xEdgeFilterLuma(cu, 0, 0, dir, e);
        if (e % 4 == 0) lf->xEdgeFilterChroma(cu, 0, 0, dir, e);
      }
    }
  TComPicYuv *r = pic->getPicYuvRec();
  for (int j = 0; j < h; j++) memcpy(y + j * w, r->getLumaAddr() + j * r->getStride(), 2 * w);
  for (int j = 0; j < h / 2; j++) {
    memcpy(cb + j * (w / 2), r->getCbAddr() + j * r->getCStride(), w);
    memcpy(cr + j * (w / 2), r->getCrAddr() + j * r->getCStride(), w);
  }
}

// xGetBoundaryStrengthSingle for every edge of the 8x8 grid flagged in edge_ver / edge_hor (bit0: filtered edge,
// bit1: the value left in m_aapucBS by xSetEdgefilterTU / xSetEdgefilterMultiple).  units: 12 bytes per 4x4 unit =
// {intra, cbf, ref[2] (picture ids, <0 unused), mv[2][2]}; the picture must be whole 64x64 LCUs.
struct TapDbkUnit {
  unsigned char intra, cbf;
  signed char ref[2];
  short mv[2][2];
};
void ref_deblock_strengths(const TapDbkUnit *units, const unsigned char *edge_ver, const unsigned char *edge_hor, int is_b,
                           unsigned char *bs_ver, unsigned char *bs_hor) {
  TComPic *pic = S->pic;
  const int w = S->pic_w, h = S->pic_h, uw = w / 4, lw = w / 64, lh = h / 64;
  static TComLoopFilter *lf = nullptr;
  if (!lf) {
    lf = new TComLoopFilter;
    lf->create(4);
  }
  lf->setCfg(true, 0, 0, 0, true);
  TComSlice *sl = pic->getSlice(0);
  sl->setLFCrossSliceBoundaryFlag(true);
  sl->setSliceType(is_b ? B_SLICE : P_SLICE);
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < 16; i++) sl->m_apcRefPicList[l][i] = reinterpret_cast<TComPic *>(static_cast<uintptr_t>(0x10000 + 64 * i));
  for (int a = 0; a < lw * lh; a++) {
    TComDataCU *cu = pic->getCU(a);
    cu->initCU(pic, a);
    const int lx = (a % lw) * 16, ly = (a / lw) * 16;
    for (int r = 0; r < 256; r++) {
      const int z = g_auiRasterToZscan[r];
      const TapDbkUnit &q = units[(ly + r / 16) * uw + lx + r % 16];
      cu->m_pePredMode[z] = q.intra ? MODE_INTRA : MODE_INTER;
      cu->m_puhTrIdx[z] = 0;
      cu->m_puhCbf[0][z] = q.cbf ? 1 : 0;
      for (int l = 0; l < 2; l++) {
        cu->m_acCUMvField[l].m_pcMv[z].set(q.mv[l][0], q.mv[l][1]);
        cu->m_acCUMvField[l].m_piRefIdx[z] = q.ref[l] < 0 ? -1 : q.ref[l];
      }
    }
  }
  for (int dir = 0; dir < 2; dir++)
    for (int a = 0; a < lw * lh; a++) {
      TComDataCU *cu = pic->getCU(a);
      const int lx = (a % lw) * 16, ly = (a / lw) * 16;
      const unsigned char *edge = dir ? edge_hor : edge_ver;
      unsigned char *bs = dir ? bs_hor : bs_ver;
      for (int r = 0; r < 256; r++) {
        const int ux = lx + r % 16, uy = ly + r / 16, u = uy * uw + ux, z = g_auiRasterToZscan[r];
        bs[u] = 0;
        if (!(edge[u] & 1) || ((dir ? uy : ux) & 1) || (dir ? uy : ux) == 0) continue;
        lf->m_aapucBS[dir][z] = (edge[u] >> 1) & 1;
        lf->xGetBoundaryStrengthSingle(cu, 0, dir, z);
        bs[u] = lf->m_aapucBS[dir][z];
      }
    }
}

// ---- SAO: SAOProcess on the picture set by ref_set_recon, with per-LCU parameters (6 bytes each: type, band, 4 offsets),
// params = [Y LCUs][Cb LCUs][Cr LCUs] ----
void ref_sao_picture(const signed char *params, int n_lcu, short *y, short *cb, short *cr) {
  TComPic *pic = S->pic;
  const int w = S->pic_w, h = S->pic_h;
  TComSampleAdaptiveOffset sao;
  sao.create(w, h, 64, 64, 4);
  sao.createPicSaoInfo(pic, 1);
  sao.setSaoLcuBasedOptimization(true);
  SAOParam prm;
  sao.allocSaoParam(&prm);
  sao.resetSAOParam(&prm);
  prm.bSaoFlag[0] = prm.bSaoFlag[1] = true;
  for (int c = 0; c < 3; c++) {
    prm.oneUnitFlag[c] = false;
    for (int a = 0; a < n_lcu; a++) {
      const signed char *q = params + (c * n_lcu + a) * 6;
      SaoLcuParam &L = prm.saoLcuParam[c][a];
      L.mergeUpFlag = L.mergeLeftFlag = false;
      L.typeIdx = q[0];
      L.subTypeIdx = (unsigned char)q[1];
      L.length = 4;
      for (int i = 0; i < 4; i++) L.offset[i] = q[2 + i];
    }
  }
  sao.SAOProcess(pic, &prm);
  sao.freeSaoParam(&prm);
  sao.destroy();
  TComPicYuv *r = pic->getPicYuvRec();
  for (int j = 0; j < h; j++) memcpy(y + j * w, r->getLumaAddr() + j * r->getStride(), 2 * w);
  for (int j = 0; j < h / 2; j++) {
    memcpy(cb + j * (w / 2), r->getCbAddr() + j * r->getCStride(), w);
    memcpy(cr + j * (w / 2), r->getCrAddr() + j * r->getCStride(), w);
  }
}

// ---- planar YUV files (TLibVideoIO/TVideoIOYuv.cpp): one frame in, one frame out ----
// read: the file holds (w_full - pad_x) x (h_full - pad_y) samples; planes come back dense w_full x h_full
int ref_yuv_read(const char *path, int file_bits, int internal_bits, int w_full, int h_full, int pad_x, int pad_y, short *y,
                 short *cb, short *cr) {
  TVideoIOYuv io;
  io.open(const_cast<char *>(path), false, file_bits, internal_bits);
  TComPicYuv pic;
  pic.create(w_full, h_full, 64, 64, 4);
  Int pad[2] = {pad_x, pad_y};
  bool ok = io.read(&pic, pad);
  io.close();
  if (ok) {
    for (int j = 0; j < h_full; j++) memcpy(y + j * w_full, pic.getLumaAddr() + j * pic.getStride(), 2 * w_full);
    for (int j = 0; j < h_full / 2; j++) {
      memcpy(cb + j * (w_full / 2), pic.getCbAddr() + j * pic.getCStride(), w_full);
      memcpy(cr + j * (w_full / 2), pic.getCrAddr() + j * pic.getCStride(), w_full);
    }
  }
  pic.destroy();
  return ok ? 1 : 0;
}
int ref_yuv_write(const char *path, int file_bits, int internal_bits, int w, int h, int crop_right, int crop_bottom, const short *y,
                  const short *cb, const short *cr) {
  TVideoIOYuv io;
  io.open(const_cast<char *>(path), true, file_bits, internal_bits);
  TComPicYuv pic;
  pic.create(w, h, 64, 64, 4);
  for (int j = 0; j < h; j++) memcpy(pic.getLumaAddr() + j * pic.getStride(), y + j * w, 2 * w);
  for (int j = 0; j < h / 2; j++) {
    memcpy(pic.getCbAddr() + j * pic.getCStride(), cb + j * (w / 2), w);
    memcpy(pic.getCrAddr() + j * pic.getCStride(), cr + j * (w / 2), w);
  }
  bool ok = io.write(&pic, 0, crop_right, 0, crop_bottom);
  io.close();
  pic.destroy();
  return ok ? 1 : 0;
}

// ---- distortion ----
unsigned ref_calcHAD(short *org, int so, short *cur, int sc, int w, int h) { return S->rd.calcHAD(org, so, cur, sc, w, h); }
unsigned ref_getDistPart(short *cur, int sc, short *org, int so, int w, int h, int hads) {
  return S->rd.getDistPart(cur, sc, org, so, w, h, false, hads ? DF_HADS : DF_SSE);
}

// ---- inter ----
void ref_filterHorLuma(short *s, int ss, short *d, int ds, int w, int h, int frac, int last) {
  S->filt.filterHorLuma(s, ss, d, ds, w, h, frac, last != 0);
}
void ref_filterVerLuma(short *s, int ss, short *d, int ds, int w, int h, int frac, int first,
                       int last) {
  S->filt.filterVerLuma(s, ss, d, ds, w, h, frac, first != 0, last != 0);
}
void ref_filterHorChroma(short *s, int ss, short *d, int ds, int w, int h, int frac, int last) {
  S->filt.filterHorChroma(s, ss, d, ds, w, h, frac, last != 0);
}
void ref_filterVerChroma(short *s, int ss, short *d, int ds, int w, int h, int frac, int first,
                         int last) {
  S->filt.filterVerChroma(s, ss, d, ds, w, h, frac, first != 0, last != 0);
}

// xPredInterLumaBlk + xPredInterChromaBlk for one PU at luma (x,y) against the picture's recon
// planes used as the reference picture (borders extended first).  Outputs dense w*h / (w/2)*(h/2).
void ref_predInterBlk(int x, int y, int w, int h, int mvx, int mvy, int bi, short *out_y,
                      short *out_cb, short *out_cr, int do_clip_mv) {
  TComPicYuv *ref = S->pic->getPicYuvRec();
  ref->m_bIsBorderExtended = false;
  ref->extendPicBorder();
  TComDataCU *c = setup_sub(x & ~63, y & ~63, 64);
  // the "CU" starts at the PU origin so that partAddr 0 addresses both the reference block
  // (cu->getZorderIdxInCU() + partAddr) and the origin of the 64x64 destination buffer
  c->m_uiAbsIdxInLCU = g_auiRasterToZscan[((y % 64) / 4) * 16 + (x % 64) / 4];
  c->m_uiCUPelX = x & ~63;
  c->m_uiCUPelY = y & ~63;
  UInt part = 0;
  TComMv mv((Short)mvx, (Short)mvy);
  if (do_clip_mv) {
    // clipMv reads the CU origin; for parity with the oracle use the PU's own position
    c->m_uiCUPelX = x;
    c->m_uiCUPelY = y;
    c->clipMv(mv);
    c->m_uiCUPelX = x & ~63;
    c->m_uiCUPelY = y & ~63;
  }
  TComYuv *dst = &S->yuv[0];
  S->pred.xPredInterLumaBlk(c, ref, part, &mv, w, h, dst, bi != 0);
  S->pred.xPredInterChromaBlk(c, ref, part, &mv, w, h, dst, bi != 0);
  Pel *py = dst->getLumaAddr(part), *pu = dst->getCbAddr(part), *pv = dst->getCrAddr(part);
  for (int j = 0; j < h; j++) memcpy(out_y + j * w, py + j * dst->getStride(), 2 * w);
  for (int j = 0; j < h / 2; j++) {
    memcpy(out_cb + j * (w / 2), pu + j * dst->getCStride(), w);
    memcpy(out_cr + j * (w / 2), pv + j * dst->getCStride(), w);
  }
}
void ref_clipMv(int cu_x, int cu_y, int *mvx, int *mvy) {
  TComDataCU *c = setup_sub(cu_x & ~63, cu_y & ~63, 64);
  c->m_uiCUPelX = cu_x;
  c->m_uiCUPelY = cu_y;
  TComMv mv((Short)*mvx, (Short)*mvy);
  c->clipMv(mv);
  *mvx = mv.getHor();
  *mvy = mv.getVer();
}

// TComYuv::addAvg on dense w*h luma + (w/2)*(h/2) chroma inputs
void ref_addAvg(const short *a[3], const short *b[3], short *o[3], int w, int h) {
  TComYuv *A = &S->yuv[0], *B = &S->yuv[1], *O = &S->yuv[2];
  for (int j = 0; j < h; j++) {
    memcpy(A->getLumaAddr() + j * A->getStride(), a[0] + j * w, 2 * w);
    memcpy(B->getLumaAddr() + j * B->getStride(), b[0] + j * w, 2 * w);
  }
  for (int j = 0; j < h / 2; j++) {
    memcpy(A->getCbAddr() + j * A->getCStride(), a[1] + j * (w / 2), w);
    memcpy(A->getCrAddr() + j * A->getCStride(), a[2] + j * (w / 2), w);
    memcpy(B->getCbAddr() + j * B->getCStride(), b[1] + j * (w / 2), w);
    memcpy(B->getCrAddr() + j * B->getCStride(), b[2] + j * (w / 2), w);
  }
  O->addAvg(A, B, 0, w, h);
  for (int j = 0; j < h; j++) memcpy(o[0] + j * w, O->getLumaAddr() + j * O->getStride(), 2 * w);
  for (int j = 0; j < h / 2; j++) {
    memcpy(o[1] + j * (w / 2), O->getCbAddr() + j * O->getCStride(), w);
    memcpy(o[2] + j * (w / 2), O->getCrAddr() + j * O->getCStride(), w);
  }
}

// recon plane with margins after extendPicBorder: copies (w+2mx)*(h+2my) luma samples out
void ref_extended_luma(short *out) {
  TComPicYuv *r = S->pic->getPicYuvRec();
  r->m_bIsBorderExtended = false;
  r->extendPicBorder();
  int mx = r->m_iLumaMarginX, my = r->m_iLumaMarginY, st = r->getStride();
  for (int j = 0; j < S->pic_h + 2 * my; j++)
    memcpy(out + j * (S->pic_w + 2 * mx), r->getLumaAddr() + (j - my) * st - mx, 2 * (S->pic_w + 2 * mx));
}
int ref_luma_margin() { return S->pic->getPicYuvRec()->m_iLumaMarginX; }

// ---- whole-picture all-intra reconstruction with the reference's OWN functions ----
// The hot loop of ENC/TEncSearch.cpp:1006-1165 (luma) / 1167-1390 (chroma) with RDOQ off, driven
// from a decision list: initAdiPattern[Chroma] (neighbour availability from the reference's
// getPU* functions on the real TComPic) -> predIntraLumaAng/ChromaAng -> residual ->
// transformNxN -> invtransformNxN -> Clip(pred + resi) into the picture's reconstruction.
// Used (a) to pin the oracle's frame driver and (b) as bench.py's "reference" CPU baseline.
struct RefTu {
  unsigned short x, y;
  unsigned char log2n, plane, mode, flags;
};
void ref_intra_frame_encode(const RefTu *tus, int n_tu, int qp, const short *org_y, const short *org_cb,
                            const short *org_cr, short *rec_y, short *rec_cb, short *rec_cr, int *lev_y, int *lev_cb,
                            int *lev_cr) {
  TComPicYuv *r = S->pic->getPicYuvRec();
  const short *org[3] = {org_y, org_cb, org_cr};
  int *lev[3] = {lev_y, lev_cb, lev_cr};
  Pel *rp[3] = {r->getLumaAddr(), r->getCbAddr(), r->getCrAddr()};
  int rs[3] = {r->getStride(), r->getCStride(), r->getCStride()};
  int pw[3] = {S->pic_w, S->pic_w / 2, S->pic_w / 2}, ph[3] = {S->pic_h, S->pic_h / 2, S->pic_h / 2};
  for (int p = 0; p < 3; p++)
    for (int j = 0; j < ph[p]; j++) memset(rp[p] + j * rs[p], 0, 2 * pw[p]);
  static short pred[32 * 32], resi[32 * 32];
  static int lvl[32 * 32];
  TComSlice *sl = S->pic->getSlice(0);
  sl->setSliceType(I_SLICE);
  sl->setSliceQp(qp);
  sl->setSliceQpBase(qp);
  const int maxv = (int)g_uiIBDI_MAX;
  for (int i = 0; i < n_tu; i++) {
    const RefTu &t = tus[i];
    const int N = 1 << t.log2n, p = t.plane, ts = t.flags & 1;
    Bool a, l;
    if (p == 0) {
      TComDataCU *c;
      int depth = 0, part = 0;
      if (N == 4) {
        c = setup_sub(t.x & ~7, t.y & ~7, 8);
        depth = 1;
        part = ((t.y >> 2) & 1) * 2 + ((t.x >> 2) & 1);
      } else
        c = setup_sub(t.x, t.y, N);
      S->pattern.initAdiPattern(c, part, depth, S->pred.m_piYuvExt, S->pred.m_iYuvExtStride, S->pred.m_iYuvExtHeight,
                                a, l, false);
      S->pred.predIntraLumaAng(&S->pattern, t.mode, pred, N, N, N, c, a, l);
    } else {
      TComDataCU *c = setup_sub(t.x * 2, t.y * 2, 2 * N);
      S->pattern.initAdiPatternChroma(c, 0, 0, S->pred.m_piYuvExt, S->pred.m_iYuvExtStride, S->pred.m_iYuvExtHeight, a, l);
      Int *src = p == 1 ? S->pattern.getAdiCbBuf(N, N, S->pred.m_piYuvExt) : S->pattern.getAdiCrBuf(N, N, S->pred.m_piYuvExt);
      S->pred.predIntraChromaAng(&S->pattern, src, t.mode, pred, N, N, N, c, a, l);
    }
    const short *o = org[p] + t.y * pw[p] + t.x;
    for (int j = 0; j < N; j++)
      for (int k = 0; k < N; k++) resi[j * N + k] = o[j * pw[p] + k] - pred[j * N + k];
    unsigned abs_sum = 0;
    const int ttype = p == 0 ? 0 : p + 1; // TEXT_LUMA, TEXT_CHROMA_U, TEXT_CHROMA_V
    ref_transformNxN(qp, I_SLICE, ttype, 1, t.mode, ts, 0, resi, N, lvl, N, &abs_sum);
    if (abs_sum)
      ref_invtransformNxN(qp, ttype, 0, p == 0 ? t.mode : REG_DCT, resi, N, lvl, N, ts);
    else
      memset(resi, 0, sizeof(short) * N * N);
    Pel *d = rp[p] + t.y * rs[p] + t.x;
    int *lo = lev[p] + t.y * pw[p] + t.x;
    for (int j = 0; j < N; j++)
      for (int k = 0; k < N; k++) {
        int v = pred[j * N + k] + resi[j * N + k];
        d[j * rs[p] + k] = (Pel)(v < 0 ? 0 : (v > maxv ? maxv : v));
        lo[j * pw[p] + k] = lvl[j * N + k];
      }
  }
  short *out[3] = {rec_y, rec_cb, rec_cr};
  for (int p = 0; p < 3; p++)
    for (int j = 0; j < ph[p]; j++) memcpy(out[p] + j * pw[p], rp[p] + j * rs[p], 2 * pw[p]);
}

} // extern "C"

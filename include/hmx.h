/*
 * hmx.h -- C-ABI of libhmx: the MI355X (gfx950) implementation of the HM block hot path.
 *
 * Drop-in boundary for the reference's block kernels.  The reference (fr34k8/thevc = HM 7.2/8-dev)
 * has no plugin/FFI layer: the boundary is the member-function surface of TComTrQuant,
 * TComPrediction/TComPattern and TComInterpolationFilter, called only from TEncSearch, TEncCu and
 * TDecCu.  Every entry point below names the reference member it replaces (file:line, paths
 * relative to /root/reference/source/Lib/).  Two families:
 *
 *   hmx_<name>        scalar drop-ins: the reference's own argument order, HOST pointers, one block
 *                     per call.  Hidden reference state (bit-depth globals, m_cQP, TComDataCU
 *                     getters) becomes the context handle and explicit scalars.  They run the
 *                     same HIP kernels as the batched path with a batch of one (H2D, launch, D2H);
 *                     there is no CPU arithmetic in this library.
 *   hmx_batch_<name>  the throughput path: DEVICE pointers, descriptor arrays, asynchronous on
 *   hmx_frame_<name>  the context's HIP stream.
 *
 * Types: Pel = int16_t, TCoeff = int32_t (TLibCommon/TypeDef.h:297-298); strides in elements.
 * All functions return 0 on success, a negative hmx_status otherwise (the reference returns Void
 * and asserts; hmx_last_error() gives the text).
 */
#ifndef HMX_H
#define HMX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t hmx_pel;
typedef int32_t hmx_coeff;
typedef struct hmx_ctx hmx_ctx;

enum hmx_status {
  HMX_OK = 0,
  HMX_ERR_ARG = -1,     /* unsupported size / null pointer (the reference asserts or falls through) */
  HMX_ERR_DEVICE = -2,  /* HIP runtime error */
  HMX_ERR_NOMEM = -3
};

#define HMX_REG_DCT 65535u /* TLibCommon/TypeDef.h:239 */
enum hmx_text_type { HMX_TEXT_LUMA = 0, HMX_TEXT_CHROMA = 1, HMX_TEXT_CHROMA_U = 2, HMX_TEXT_CHROMA_V = 3 };
enum hmx_slice_type { HMX_B_SLICE = 0, HMX_P_SLICE = 1, HMX_I_SLICE = 2 };

/* ------------------------------------------------------------------------------------------------
 * Context: replaces the process globals g_uiBitDepth/g_uiBitIncrement/g_uiIBDI_MAX
 * (TLibCommon/TComRom.cpp:445-448) and the scratch members of the three classes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int bit_depth; /* internal bit depth B = g_uiBitDepth + g_uiBitIncrement (8 or 10 in the cfgs) */
  int device;    /* HIP device ordinal */
  void *stream;  /* hipStream_t to launch on; NULL = the library creates its own */
  int ctu_size;  /* g_uiMaxCUWidth, 64 in every shipped cfg */
} hmx_config;

int hmx_create(const hmx_config *cfg, hmx_ctx **out);
void hmx_destroy(hmx_ctx *ctx);
const char *hmx_last_error(const hmx_ctx *ctx);
int hmx_sync(hmx_ctx *ctx);
/* Tuning knobs (the reference has none: these select between schedules of the SAME arithmetic for A/B runs and
 * cross-checks).  Each is read once from the environment variable of the same name in hmx_create; hmx_set_option
 * changes one afterwards, value NULL restores the default.  HMX_INTRA_SCHEDULE = packed (default) | level | wave,
 * HMX_INTRA_ACROSS, HMX_INTRA_STREAMS, HMX_PIPELINE_CONV, HMX_GRAPH (level schedules), HMX_PACK_SLOTS4 = 16 | 64, HMX_PACK_SLOTS8 = 8 | 16,
 * HMX_PACK_GROUP, HMX_PACK_WAVES, HMX_PACK_SLEEP0, HMX_PACK_SLEEP1 (packed schedule), HMX_RDOQ_LANE (RDOQ: every block
 * through the one-lane-per-block kernel), HMX_PLAN_ROWS (device plan builder: rows of the level table per picture to start with), HMX_PLAN_STREAMS (1: its luma and chroma
 * level walks one after the other instead of side by side on two streams). */
int hmx_set_option(hmx_ctx *ctx, const char *name, const char *value);
/* device memory + timing plumbing so that callers need no HIP headers */
int hmx_malloc(hmx_ctx *ctx, size_t bytes, void **dptr);
int hmx_free(hmx_ctx *ctx, void *dptr);
int hmx_upload(hmx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int hmx_download(hmx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int hmx_memset(hmx_ctx *ctx, void *dst_dev, int value, size_t bytes);
int hmx_event_create(hmx_ctx *ctx, void **ev);
int hmx_event_record(hmx_ctx *ctx, void *ev); /* on the context's stream */
int hmx_event_elapsed_ms(hmx_ctx *ctx, void *ev_start, void *ev_stop, float *ms); /* syncs ev_stop */
int hmx_event_destroy(hmx_ctx *ctx, void *ev);

/* ------------------------------------------------------------------------------------------------
 * Quantiser state.  TComTrQuant::setQPforQuant (TLibCommon/TComTrQuant.cpp:192-222) writes the
 * hidden member m_cQP (QpParam, TComTrQuant.h:79-113); here it returns the value.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int qp, per, rem, bits;
} hmx_qp;
hmx_qp hmx_setQPforQuant(int qpy, int text_type, int qp_bd_offset, int chroma_qp_offset);

/* What xQuant reads through TComDataCU / TComSlice / TComPPS (TComTrQuant.cpp:1121-1267) */
typedef struct {
  hmx_qp qp;          /* m_cQP */
  int per_base;       /* cQpBase.m_iPer from getSliceQpBase() (ADAPTIVE_QP_SELECTION); -1 = qp.per */
  int slice_type;     /* hmx_slice_type: rounding 171 (I) / 85 */
  int sign_hide;      /* PPS getSignHideFlag() */
  int is_intra;       /* pcCU->isIntra(uiAbsPartIdx) */
  int dir_mode;       /* luma/chroma intra direction used by getCoefScanIdx (TComDataCU.cpp:4014) */
} hmx_quant_param;

/* ------------------------------------------------------------------------------------------------
 * Scalar drop-ins, TComTrQuant
 * ---------------------------------------------------------------------------------------------- */
/* xT  (TComTrQuant.cpp:1542): strided Pel residual -> Int[w*h]; w == h in {4,8,16,32} */
int hmx_xT(hmx_ctx *ctx, unsigned mode, const hmx_pel *resi, unsigned stride, int32_t *coef, int w, int h);
/* xIT (TComTrQuant.cpp:1583) */
int hmx_xIT(hmx_ctx *ctx, unsigned mode, const int32_t *coef, hmx_pel *resi, unsigned stride, int w, int h);
/* xTransformSkip / xITransformSkip (TComTrQuant.cpp:1622, 1667) */
int hmx_xTransformSkip(hmx_ctx *ctx, const hmx_pel *resi, unsigned stride, int32_t *coef, int w, int h);
int hmx_xITransformSkip(hmx_ctx *ctx, const int32_t *coef, hmx_pel *resi, unsigned stride, int w, int h);
/* xQuant, flat path + signBitHidingHDQ (TComTrQuant.cpp:1102-1270, 977-1100); ac_sum accumulates */
int hmx_xQuant(hmx_ctx *ctx, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *ac_sum,
               int text_type, const hmx_quant_param *qp);
/* xDeQuant with a scaling list (TComTrQuant.cpp:1311-1342, getUseScalingList()): dequant_coef[w*h] = the table
 * getDequantCoeff(scalingListType, m_cQP.m_iRem, log2(w) - 2, SCALING_LIST_SQT) that setScalingListDec built from the slice's lists
 * (:2773-2793, 2852-2873: header handling, stays in HM).  Host pointers like the other scalar drop-ins. */
int hmx_xDeQuant_scaled(hmx_ctx *ctx, const hmx_coeff *src, int32_t *dst, int w, int h, const hmx_qp *qp, const int32_t *dequant_coef);
/* The pArlDes output of xQuant (ADAPTIVE_QP_SELECTION, m_bUseAdaptQpSelect): arl[n] = (|src[n]| * quantScale + round) >> (iQBits - 7),
 * what TEncSlice's adaptive QP selection accumulates.  rdoq_form = 0: the flat branch (TComTrQuant.cpp:1229-1249; iQBits from
 * qp->per_base, the slice's base QP); rdoq_form = 1: as xRateDistOptQuant writes it (:1757-1765, 1886-1891; iQBits from qp->qp.per,
 * the product limited to MAX_INT - (1 << (iQBits - 1)) first).  It depends on the coefficients alone: call it beside hmx_xQuant /
 * hmx_xRateDistOptQuant when the encoder runs with AdaptiveQpSelection. */
int hmx_arlCoeff(hmx_ctx *ctx, const int32_t *src, int32_t *arl, int w, int h, int text_type, const hmx_quant_param *qp, int rdoq_form,
                 const int32_t *quant_coef /* getQuantCoeff table of a scaling list, or NULL */);
/* The quantisers with a scaling list (getUseScalingList()): the per-position tables HM's setScalingList built for the block's list
 * type, QP remainder and size are INPUTS (building them from the slice's lists is header handling, TComTrQuant.cpp:2747-2847, 2953-
 * 2977) -- quant_coef = getQuantCoeff(...), err_scale = getErrScaleCoeff(...), w*h entries each, row-major.
 * hmx_xQuant_scaled: xQuant's flat branch (:1215, 1244-1255) + signBitHidingHDQ.
 * hmx_xRateDistOptQuant_scaled: xRateDistOptQuant (:1759-1762, 1882-1883). */
int hmx_xQuant_scaled(hmx_ctx *ctx, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *ac_sum, int text_type,
                      const hmx_quant_param *qp, const int32_t *quant_coef);
/* xRateDistOptQuant (TComTrQuant.cpp:1719-2305), the quantiser transformNxN selects when RDOQ is on
 * (:1122-1128; every shipped cfg).  It reads CABAC bit estimates that TEncSbac::estBit leaves in
 * m_pcEstBitsSbac for the block's size and texture type (estBitsSbacStruct, TComTrQuant.h:59-72; same
 * field order here, 1/32768 bit) and the Lagrange multiplier m_dLambda: both are inputs. */
typedef struct hmx_est_bits {
  int32_t significantCoeffGroupBits[2][2];
  int32_t significantBits[42][2];
  int32_t lastXBits[32];
  int32_t lastYBits[32];
  int32_t greaterOneBits[24][2];
  int32_t levelAbsBits[6][2];
  int32_t blockCbpBits[15][2];
  int32_t blockRootCbpBits[4][2];
  int32_t scanZigzag[2];
  int32_t scanNonZigzag[2];
} hmx_est_bits;
typedef struct hmx_rdoq_param {
  hmx_qp qp;       /* m_cQP */
  int sign_hide;   /* PPS getSignHideFlag() */
  int is_intra;    /* pcCU->isIntra(uiAbsPartIdx) */
  int dir_mode;    /* intra direction used by getCoefScanIdx */
  int root_cbf;    /* 1: inter luma block with transform index 0 -> blockRootCbpBits[0] (:2139-2144) */
  int cbf_ctx;     /* otherwise the index into blockCbpBits, texture offset included (:2147-2150) */
  double lambda;   /* m_dLambda */
} hmx_rdoq_param;
int hmx_xRateDistOptQuant(hmx_ctx *ctx, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *abs_sum,
                          int text_type, const hmx_rdoq_param *rp, const hmx_est_bits *est);
int hmx_xRateDistOptQuant_scaled(hmx_ctx *ctx, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *abs_sum, int text_type,
                                 const hmx_rdoq_param *rp, const hmx_est_bits *est, const int32_t *quant_coef, const double *err_scale);
/* xDeQuant, flat path (TComTrQuant.cpp:1272-1355) */
int hmx_xDeQuant(hmx_ctx *ctx, const hmx_coeff *src, int32_t *dst, int w, int h, const hmx_qp *qp);
/* transformNxN (TComTrQuant.cpp:1373-1426): uiMode is derived as the reference does
 * (luma && intra -> dir_mode, else REG_DCT) */
int hmx_transformNxN(hmx_ctx *ctx, const hmx_pel *resi, unsigned stride, hmx_coeff *level, unsigned w,
                     unsigned h, uint32_t *abs_sum, int text_type, const hmx_quant_param *qp,
                     int use_transform_skip, int trans_quant_bypass);
/* invtransformNxN (TComTrQuant.cpp:1428-1450); scaling lists are off in every shipped cfg */
int hmx_invtransformNxN(hmx_ctx *ctx, int trans_quant_bypass, int text_type, unsigned mode, hmx_pel *resi,
                        unsigned stride, const hmx_coeff *level, unsigned w, unsigned h, const hmx_qp *qp,
                        int use_transform_skip);

/* ------------------------------------------------------------------------------------------------
 * Scalar drop-ins, TComPattern / TComPrediction (intra)
 * ---------------------------------------------------------------------------------------------- */
/* initAdiPattern / initAdiPatternChroma (TLibCommon/TComPattern.cpp:213-366): rec = sample (0,0) of a
 * HOST reconstruction plane; (x,y,n) = block position/size in samples of that plane; neighbour
 * availability is derived geometrically for one slice / one tile / no constrained intra pred
 * (TComDataCU.cpp:1221-1735).  adi receives the reference layout: (2n+1)^2 Int border buffer and,
 * for luma, the [1 2 1]-smoothed copy behind it.  adi must hold 2*(2n+1)^2 ints. */
int hmx_initAdiPattern(hmx_ctx *ctx, const hmx_pel *rec, int stride, int x, int y, int n, int is_chroma,
                       int pic_w_luma, int pic_h_luma, int32_t *adi);
/* predIntraLumaAng / predIntraChromaAng (TLibCommon/TComPrediction.cpp:338-386) */
int hmx_predIntraLumaAng(hmx_ctx *ctx, const int32_t *adi, unsigned dir_mode, hmx_pel *pred, unsigned stride,
                         int w, int h);
int hmx_predIntraChromaAng(hmx_ctx *ctx, const int32_t *adi, unsigned dir_mode, hmx_pel *pred, unsigned stride,
                           int w, int h);
/* The protected building blocks of the two wrappers (TComPrediction.cpp:129-167, 190-336, 689-730), named by
 * the north star.  `adi` is ONE (2w+1) x (2w+1) border buffer (raw or smoothed: the reference's callers choose
 * by passing a pointer; the reference's pSrc is its cell (1,1)).  xPredIntraAng: dir_mode 1 = DC from the
 * sides flagged available (no edge smoothing: xDCPredFiltering is the wrapper's), 2..34 angular, `filter` =
 * bFilter (edge filter of the pure vertical / horizontal modes). */
int hmx_predIntraGetPredValDC(hmx_ctx *ctx, const int32_t *adi, int w, int h, int above, int left, hmx_pel *dc);
int hmx_xPredIntraPlanar(hmx_ctx *ctx, const int32_t *adi, hmx_pel *pred, unsigned stride, int w, int h);
int hmx_xPredIntraAng(hmx_ctx *ctx, const int32_t *adi, hmx_pel *pred, unsigned stride, int w, int h, unsigned dir_mode,
                      int above, int left, int filter);

/* ------------------------------------------------------------------------------------------------
 * Scalar drop-ins, TComRdCost (TLibCommon/TComRdCost.cpp): calcHAD :404-450 (same argument order) and
 * getDistPart(piCur, iCurStride, piOrg, iOrgStride, w, h, false, DF_SSE) :452-478 -> xGetSSE* :1313-1657
 * (the IBDI_DISTORTION 0 variant: per-sample (diff^2) >> 2*bitIncrement).  w, h <= 64.
 * ---------------------------------------------------------------------------------------------- */
int hmx_calcHAD(hmx_ctx *ctx, const hmx_pel *pi0, int stride0, const hmx_pel *pi1, int stride1, int w, int h,
                uint32_t *satd);
int hmx_getSSE(hmx_ctx *ctx, const hmx_pel *cur, int cur_stride, const hmx_pel *org, int org_stride, int w, int h,
               uint32_t *sse);

/* ------------------------------------------------------------------------------------------------
 * Scalar drop-ins, TComInterpolationFilter (TLibCommon/TComInterpolationFilter.cpp:323-415) and
 * TComYuv::addAvg (TLibCommon/TComYuv.cpp:520-581).  src must be readable 3 (luma) / 1 (chroma)
 * samples before and 4 / 2 after the block in the filtered direction, as in the reference.
 * ---------------------------------------------------------------------------------------------- */
int hmx_filterHorLuma(hmx_ctx *ctx, const hmx_pel *src, int src_stride, int16_t *dst, int dst_stride, int w,
                      int h, int frac, int is_last);
int hmx_filterVerLuma(hmx_ctx *ctx, const hmx_pel *src, int src_stride, int16_t *dst, int dst_stride, int w,
                      int h, int frac, int is_first, int is_last);
int hmx_filterHorChroma(hmx_ctx *ctx, const hmx_pel *src, int src_stride, int16_t *dst, int dst_stride, int w,
                        int h, int frac, int is_last);
int hmx_filterVerChroma(hmx_ctx *ctx, const hmx_pel *src, int src_stride, int16_t *dst, int dst_stride, int w,
                        int h, int frac, int is_first, int is_last);
int hmx_addAvg(hmx_ctx *ctx, const hmx_pel *src0, int s0_stride, const hmx_pel *src1, int s1_stride,
               hmx_pel *dst, int dst_stride, int w, int h);
/* TComPrediction::xPredInterLumaBlk / xPredInterChromaBlk (TLibCommon/TComPrediction.cpp:554-585, :599-642) for one block:
 * ref = the reference plane at the block's own position (refPic->getLumaAddr(cuAddr, zorder + partAddr); the reference
 * adds the vector's integer part itself), mv in quarter-pel luma units, bi = the block is one half of a bi-prediction
 * (output = the 14-bit intermediate, not clipped).  w, h: LUMA size of the block (the chroma form halves it, like the
 * reference); dst stride in elements.  The planes must be readable 3 (1) samples before and 4 (2) after the moved block. */
int hmx_xPredInterLumaBlk(hmx_ctx *ctx, const hmx_pel *ref, int ref_stride, int mv_hor, int mv_ver, int w, int h, hmx_pel *dst,
                          int dst_stride, int bi);
int hmx_xPredInterChromaBlk(hmx_ctx *ctx, const hmx_pel *ref, int ref_stride, int mv_hor, int mv_ver, int w, int h, hmx_pel *dst,
                            int dst_stride, int bi);

/* ------------------------------------------------------------------------------------------------
 * Batched device path
 * ---------------------------------------------------------------------------------------------- */
/* One transform / prediction block.  8 bytes, identical on host and device. */
typedef struct {
  uint16_t x, y;  /* position in samples of its plane */
  uint8_t log2n;  /* 2..5; 6 = the 64x64 luma prediction unit of a 64x64 coding unit, prediction entry points only */
  uint8_t plane;  /* 0 Y, 1 Cb, 2 Cr */
  uint8_t mode;   /* intra prediction mode 0..34 (also selects DST for 4x4 luma and the scan) */
  uint8_t flags;  /* HMX_TU_* */
} hmx_tu;
#define HMX_TU_TRANSFORM_SKIP 1u
#define HMX_TU_INTER 2u /* non-intra CU: REG_DCT, diagonal scan, no DST */
#define HMX_TU_CBF_CTX(ctx) (((unsigned)(ctx) & 15u) << 4) /* bits 4..7: context of the coded-block flag (hmx_set_rdoq) */

typedef struct {
  int pic_w, pic_h;      /* luma size of the picture */
  int qp;                /* CU QP (pcCU->getQP(0)); one QP per call */
  int chroma_qp_offset;  /* PPS cb/cr offset */
  int slice_type;        /* hmx_slice_type */
  int sign_hide;
} hmx_pic_param;

/* A picture in HBM: three Pel planes; plane[i] points at sample (0,0); stride in elements. */
typedef struct {
  hmx_pel *plane[3];
  int stride[3];
} hmx_pic;
/* Quantised levels.  stride[p] > 0: plane geometry, level of sample (x,y) of plane p at
 * plane[p][y*stride[p]+x].  stride[p] == 0 (whole-picture calls only): the reference's own coefficient
 * layout (TComDataCU::m_pcTrCoeffY/Cb/Cr, TLibCommon/TComDataCU.cpp:117-141): CTU blocks in raster order,
 * C*C ints each (C = CTU size in that plane, planes padded to whole CTUs); inside a CTU the N x N block
 * of the transform block whose first 4x4 unit is (ux,uy) starts at 16 * Zorder(ux,uy), row-major. */
typedef struct {
  hmx_coeff *plane[3];
  int stride[3];
} hmx_levels;

/* A list of blocks resident on the device.  Built once from a HOST array (the library buckets the
 * blocks by size: one launch per size class) and reused by any number of batch calls. */
typedef struct hmx_tu_list hmx_tu_list;
int hmx_tu_list_create(hmx_ctx *ctx, const hmx_tu *tus, int n, hmx_tu_list **list);
void hmx_tu_list_destroy(hmx_ctx *ctx, hmx_tu_list *list);

/* transformNxN over a list of independent blocks (HOT LOOP B' of ENC/TEncSearch.cpp:4784-4990):
 * residual planes -> level planes; d_abs_sum[i] = uiAbsSum of block i in the caller's order
 * (device pointer, may be NULL). */
int hmx_batch_transformNxN(hmx_ctx *ctx, const hmx_tu_list *list, const hmx_pic *resi, const hmx_levels *lev,
                           uint32_t *d_abs_sum, const hmx_pic_param *pp);
/* The inter residual path in one pass (ENC/TEncSearch.cpp:4526-4990): residual = org - pred
 * (TComYuv::subtract, TLibCommon/TComYuv.cpp:461-518) fused into transformNxN. */
int hmx_batch_residual_transformNxN(hmx_ctx *ctx, const hmx_tu_list *list, const hmx_pic *org, const hmx_pic *pred,
                                    const hmx_levels *lev, uint32_t *d_abs_sum, const hmx_pic_param *pp);
/* invtransformNxN over a list of blocks (DEC/TDecCu.cpp:791-831): levels -> residual planes `out`;
 * when pred != NULL the reconstruction Clip(pred + resi) is written instead (TComYuv::addClip). */
int hmx_batch_invtransformNxN(hmx_ctx *ctx, const hmx_tu_list *list, const hmx_levels *lev, const hmx_pic *pred,
                              const hmx_pic *out, const hmx_pic_param *pp);
/* RDOQ over a list of blocks (host list): Int coefficients in `coef` (plane geometry, e.g. the output of the
 * xT drop-in or of a batch transform without quantisation) -> levels in `lev`; d_abs_sum[i] (device, may be
 * NULL) receives block i's absolute sum.  side[i] carries what the reference reads from the CU for block i
 * and which bit-estimate table (est[side[i].est_idx], host array) applies.  8x8 and larger blocks: a wave shares 8 / 4 / 1
 * blocks in LDS, coefficient groups walked in parallel (thevc_amd/csrc/hmx_rdoq_core.h); 4x4 blocks: one lane per block.
 * Coefficients are what xT / xTransformSkip produce: |c| <= 32768 (levels of 8x8 and larger blocks travel as 16-bit words
 * inside the kernel).  The block list and the tables stay resident on the device while the arguments repeat. */
typedef struct hmx_rdoq_side {
  uint16_t est_idx;
  uint8_t root_cbf, cbf_ctx;
} hmx_rdoq_side;
int hmx_batch_xRateDistOptQuant(hmx_ctx *ctx, const hmx_tu *tus, const hmx_rdoq_side *side, int n, const hmx_levels *coef,
                                const hmx_levels *lev, uint32_t *d_abs_sum, const hmx_pic_param *pp,
                                const hmx_est_bits *est, int n_est, double lambda_luma, double lambda_chroma);
/* Intra prediction of a list of blocks from a reconstructed picture (all neighbours are read
 * from rec as it is: the caller guarantees the dependency order).  Output in plane geometry of
 * `pred`.  HOT LOOP A shape (ENC/TEncSearch.cpp:2534): d_modes != NULL evaluates modes[0..n_modes) for
 * every block from ONE reference gather and writes candidate k at element offset
 * k * mode_plane_elems[plane] of the pred planes. */
int hmx_batch_predIntra(hmx_ctx *ctx, const hmx_tu_list *list, const hmx_pic *rec, const hmx_pic *pred,
                        const hmx_pic_param *pp, const uint8_t *d_modes, int n_modes,
                        const size_t mode_plane_elems[3]);

/* The mode pre-selection of estIntraPredQT (TLibEncoder/TEncSearch.cpp:2509-2540) without writing the
 * candidates: for every block and every mode of d_modes (NULL: the block's own mode, n_modes = 1) predict
 * from `rec` and cost against `org` with TComRdCost::calcHAD (TLibCommon/TComRdCost.cpp:404-450: Hadamard
 * SATD over 8x8 sub-blocks, 4x4 for 4x4 blocks).  d_satd[(block index in the list's creation order) * n_modes
 * + k] (device).  The 35 predictions stay in registers: 35x less write traffic than hmx_batch_predIntra. */
int hmx_batch_predIntra_cost(hmx_ctx *ctx, const hmx_tu_list *list, const hmx_pic *rec, const hmx_pic *org,
                             const hmx_pic_param *pp, const uint8_t *d_modes, int n_modes, uint32_t *d_satd);
/* Whole-picture all-intra reconstruction from decisions: for every block, refs <- recon, predict,
 * residual, T, Q, IQ, IT, recon (ENC/TEncSearch.cpp:1006-1165 with RDOQ off; DEC/TDecCu.cpp:469-687 for
 * the decode direction).  Blocks are given per picture in coding order on the HOST; the library
 * derives the dependency schedule (CTU diagonals x in-CTU levels) and launches one kernel per CTU
 * diagonal over all pictures of the batch. */
typedef struct hmx_intra_plan hmx_intra_plan;
/* The neighbour units (bits as in the availability flags of initAdiPattern: below-left bottom first, left, corner, above,
 * above-right; units of 4 luma / 2 chroma samples) whose reconstruction the prediction of a block of n_samples x n_samples
 * with this mode can depend on, given which units are available: what the mode reads (TComPrediction.cpp:179-290, 689-730),
 * one sample wider where the smoothed reference line is used, and the units the padding of unavailable ones copies from.
 * The dependency order of a plan follows these bits, not the availability flags (a horizontal mode does not wait for the
 * block above-right).  Pure host function (no device work); exported for the test that holds it against the oracle. */
unsigned long long hmx_intra_dependency_mask(int n_samples, int is_luma, int mode, unsigned long long avail);
/* The availability flags of initAdiPattern for a block at luma position (x, y) of luma size size_luma (a chroma block: its
 * luma-scaled position and size), CTU size 64: bit u in the bNeighborFlags order, units of four luma samples.  closed_form 0:
 * the unit-by-unit rule (TComPattern.cpp:607-786); 1: the closed form the device plan builder uses.  Pure host functions,
 * exported for the test that holds one against the other. */
unsigned long long hmx_intra_avail_mask(int x, int y, int size_luma, int pic_w, int pic_h, int closed_form);
int hmx_intra_plan_create(hmx_ctx *ctx, const hmx_tu *tus, int n_tu, const hmx_pic_param *pp,
                          hmx_intra_plan **plan);
/* The plans of n_pics pictures (picture i: tus[i][0 .. n_tu[i])): the dependency analysis -- host work, about 43 ms per
 * 2160p picture on one core -- runs on as many host threads as the machine has (at most 32), the uploads follow.
 * Same plans as n_pics calls of hmx_intra_plan_create; on an error no plan is left behind. */
int hmx_intra_plan_create_multi(hmx_ctx *ctx, const hmx_tu *const *tus, const int *n_tu, int n_pics, const hmx_pic_param *pp,
                                hmx_intra_plan **plans);
/* The same plans analysed ON THE DEVICE, for pipelines whose every batch brings new decisions (the host analysis above is
 * ~550x a picture's share of a whole-picture call).  d_tus (DEVICE): the decision lists of n_pics pictures back to back,
 * picture i = d_tus[offsets[i] .. offsets[i+1]) (offsets: HOST array of n_pics + 1 entries), each list in coding order --
 * CTUs in raster order, the blocks of a CTU together, in the order the reference codes them (TEncSearch.cpp:1394-1700).
 * The dependency levels are a wavefront over the CTU diagonals with one lane per (picture, CTU, plane) walking its CTU's
 * blocks; the bucket order is a counting sort.  plans[i] equals what hmx_intra_plan_create builds from the same list,
 * entry for entry; such plans serve the packed schedule (default) and the level schedule, not the CTU-wave schedule.
 * The lists may be freed or overwritten when the call returns.  The plans of one call share their device tables: destroy
 * each with hmx_intra_plan_destroy (the memory is kept for the next call of this context). */
int hmx_intra_plan_create_device(hmx_ctx *ctx, const hmx_tu *d_tus, const uint32_t *offsets, int n_pics, const hmx_pic_param *pp,
                                 hmx_intra_plan **plans);
void hmx_intra_plan_destroy(hmx_ctx *ctx, hmx_intra_plan *plan);
/* The plans of a batch at once (waits ONCE for the context's stream: hmx_intra_plan_destroy does so per plan). */
void hmx_intra_plan_destroy_many(hmx_ctx *ctx, hmx_intra_plan *const *plans, int n);
/* The two tables of a plan that the packed and the level schedule run on, copied to the HOST (for tests and tools that hold a
 * device-built plan against a host-built one): blocks[n_blocks] = the blocks sorted by (dependency level, size, code path,
 * coding index), 16 bytes each: the hmx_tu followed by a 64-bit mask over the 4n+1 neighbour units in initAdiPattern's order: the
 * units the block's mode READS among the available ones, closed under the padding rule (hmx_intra_dependency_mask of the availability);
 * levels[n_levels] = per dependency level {uint32 start[4], count[4]} by transform size (4, 8, 16, 32), starts into blocks.
 * Either pointer may be NULL. */
int hmx_intra_plan_download(hmx_ctx *ctx, const hmx_intra_plan *plan, void *blocks, void *levels);
/* Size of the dependency schedules of a plan: blocks, picture-wide dependency levels (= launches of
 * the level schedule) and CTU diagonals (= launches of the wave schedule). */
int hmx_intra_plan_info(const hmx_intra_plan *plan, int *n_blocks, int *n_levels, int *n_diagonals);
/* One dependency level of the level schedule: its block counts by transform size (4, 8, 16, 32) and
 * the wavefronts one picture contributes to that level's launch. */
int hmx_intra_plan_level(const hmx_intra_plan *plan, int level, uint32_t counts[4], uint32_t *n_waves);
/* How the LAST whole-picture call was issued: schedule 0 = CTU-diagonal waves (k_intra_wave), 1 = levels,
 * one picture per wave (k_intra_level), 2 = levels across pictures (k_intra_level_across); stream_groups =
 * picture groups whose launches ran concurrently on separate streams (launches per level). */
int hmx_last_call_shape(const hmx_ctx *ctx, int *schedule, int *stream_groups);
/* Optional stage timing of whole-picture calls (HIP events on the context's stream): layout conversion
 * in, dependency chain, layout conversion out, of the LAST call issued after hmx_set_timing(ctx, 1). */
int hmx_set_timing(hmx_ctx *ctx, int enable);
int hmx_last_call_timing(hmx_ctx *ctx, float *to_tiled_ms, float *chain_ms, float *from_tiled_ms);
/* Packed schedule: the part of chain_ms the LAST timed call spent building its schedule tables on the device (0 when it re-used
 * the tables of an earlier call with the same pictures and plans). */
int hmx_last_call_tables_ms(hmx_ctx *ctx, float *ms);
/* Which schedule a whole-picture call with n_pics pictures uses: 1 = level, 0 = wave. */
int hmx_intra_schedule_for(const hmx_ctx *ctx, int n_pics);
/* n_pics pictures share one plan (same block structure); org/rec/lev are arrays of n_pics entries. */
int hmx_frame_intra_encode(hmx_ctx *ctx, const hmx_intra_plan *plan, int n_pics, const hmx_pic *org,
                           const hmx_pic *rec, const hmx_levels *lev);
int hmx_frame_intra_decode(hmx_ctx *ctx, const hmx_intra_plan *plan, int n_pics, const hmx_pic *rec,
                           const hmx_levels *lev);
/* The general form: picture i follows plans[i] (every picture its own block structure and modes, as in a
 * real stream); all plans of one call share picture size, QP and slice settings. */
int hmx_frame_intra_encode_multi(hmx_ctx *ctx, const hmx_intra_plan *const *plans, int n_pics, const hmx_pic *org,
                                 const hmx_pic *rec, const hmx_levels *lev);
/* The same for the intra blocks of an INTER picture: the plan lists only the intra-coded blocks, and they are
 * reconstructed ONTO what `rec` already holds -- the inter-coded blocks (motion compensation + residual) of the same
 * picture, which the decoder has reconstructed before any later coding unit predicts from them
 * (DEC/TDecCu.cpp:384-446 xDecompressCU: xReconInter / xReconIntraQT per coding unit; an intra block only ever reads
 * samples of coding units that precede it, so "all inter blocks first, then the intra blocks in dependency order"
 * gives the same picture).  The reconstruction planes are brought into the working pool first. */
int hmx_frame_intra_decode_onto(hmx_ctx *ctx, const hmx_intra_plan *plan, int n_pics, const hmx_pic *rec,
                                const hmx_levels *lev);
/* Encoder side of the same (ENC/TEncSearch.cpp:1006-1390 for the intra coding units the encoder chose inside an inter
 * picture): prediction from what `rec` holds, residual against `org`, T, flat Q + sign hiding -> levels, IQ, IT,
 * reconstruction of the plan's blocks onto `rec`. */
int hmx_frame_intra_encode_onto(hmx_ctx *ctx, const hmx_intra_plan *plan, int n_pics, const hmx_pic *org, const hmx_pic *rec,
                                const hmx_levels *lev);
int hmx_frame_intra_decode_multi(hmx_ctx *ctx, const hmx_intra_plan *const *plans, int n_pics, const hmx_pic *rec,
                                 const hmx_levels *lev);

/* RDOQ as the quantiser of the whole-picture encode calls (the encoder's default: TEncSearch::xIntraCodingLumaBlk /
 * ChromaBlk call transformNxN, which runs xRateDistOptQuant when m_useRDOQ, TComTrQuant.cpp:1395-1404).  What the
 * encoder takes from its live state is an input, per picture: the bit estimates m_pcEstBitsSbac as estBit leaves them
 * for [luma, chroma] x [4x4, 8x8, 16x16, 32x32] (TEncSbac.cpp:1507-1667), and m_dLambda for luma and chroma blocks
 * (setLambda per component, TEncSlice.cpp:380-395).  The context of a block's coded-block flag travels in
 * hmx_tu::flags, bits 4..7 (= getCtxQtCbf: the transform depth for luma, + NUM_QT_CBF_CTX for chroma).
 * n_pics = the pictures of the following calls, or 1 = one set for every picture.  pics = NULL turns RDOQ off again.
 * Transform-skip blocks keep the flat quantiser (TComTrQuant.cpp:1121-1122 with TransformSkipFast, which every shipped
 * cfg that enables transform skip sets); packed schedule only, packing groups of at most 2 pictures (HMX_PACK_GROUP; the
 * tables of a group wait in LDS).  Levels travel as 16-bit words inside the kernel: a call whose QP, bit depth and block sizes allow
 * |level| = (32768 * quantScale) >> qbits > 32767 (e.g. 10-bit, QP < 5 with 32x32 blocks) is refused with HMX_ERR_ARG. */
typedef struct hmx_rdoq_pic {
  hmx_est_bits est[8]; /* [luma, chroma][log2(size) - 2] */
  double lambda_luma, lambda_chroma;
} hmx_rdoq_pic;
int hmx_set_rdoq(hmx_ctx *ctx, const hmx_rdoq_pic *pics, int n_pics);
/* Distortion next to the chain: the encoder calls TComRdCost::getDistPart(rec, org, DF_SSE) right after every
 * reconstruction (TLibEncoder/TEncSearch.cpp:1163, :1381).  After hmx_set_sse_output(ctx, sse, n) the whole-picture
 * ENCODE calls (any of the hmx_frame_intra_encode* entry points, packed schedule) with at most n pictures also write,
 * for picture i and plane p, xGetSSE(org, rec) of every block -- sum of (org - rec)^2 >> 2 * (B - 8),
 * TLibCommon/TComRdCost.cpp:1313-1657 -- into sse[i].plane[p][u], u = the block's first 4x4 unit in the reference's
 * partition order (CTU blocks in raster order over the CTU-padded plane, Z-order inside a CTU: 1/16 of the block's
 * offset in the stride-0 level layout).  Device arrays of (CTU-padded plane samples / 16) entries; sse == NULL
 * switches the output off.  The sums are formed in the kernel that reconstructs the block. */
typedef struct {
  uint32_t *plane[3];
} hmx_sse;
int hmx_set_sse_output(hmx_ctx *ctx, const hmx_sse *sse, int n_pics);

/* Pictures RESIDENT in the library's working layout: CTU blocks in raster order, 4x4 tiles in
 * Z-order inside a CTU, groups of up to 64 pictures interleaved 8x8 quad by quad.  The whole-picture entry points above
 * take pictures in the reference's plane geometry (TComPicYuv) and convert them into and out of this layout around
 * every call; an application that keeps its pictures here pays that once, or never: hmx_yuv_unpack_resident writes a
 * file's frame straight into a pool picture, hmx_yuv_pack_resident reads one back, hmx_tpool_import / _export convert
 * from / to planes (e.g. a reconstruction that motion compensation needs with the reference's margins).
 * A pool holds n_pics pictures of one size; picture indices are positions in the pool. */
typedef struct hmx_tpool hmx_tpool;
int hmx_tpool_create(hmx_ctx *ctx, int pic_w, int pic_h, int n_pics, hmx_tpool **pool);
void hmx_tpool_destroy(hmx_ctx *ctx, hmx_tpool *pool);
int hmx_tpool_import(hmx_ctx *ctx, hmx_tpool *pool, int first, int n, const hmx_pic *src);       /* planes -> pictures first..first+n-1 */
int hmx_tpool_export(hmx_ctx *ctx, const hmx_tpool *pool, int first, int n, const hmx_pic *dst); /* and back */
/* TVideoIOYuv::read / write (TLibVideoIO/TVideoIOYuv.cpp:226-480) on a pool picture; arguments as hmx_yuv_unpack /
 * hmx_yuv_pack with the picture size taken from the pool (w_full x h_full = the pool's picture size). */
int hmx_yuv_unpack_resident(hmx_ctx *ctx, const void *d_file, int file_bits, hmx_tpool *pool, int index, int pad_x, int pad_y);
int hmx_yuv_pack_resident(hmx_ctx *ctx, const hmx_tpool *pool, int index, int crop_right, int crop_bottom, int file_bits, void *d_file);
/* hmx_frame_intra_encode_multi / _decode_multi on pictures 0..n_pics-1 of resident pools: no layout conversion inside the
 * call.  plan_stride 1: picture i follows plans[i]; 0: every picture follows plans[0].  org and rec must be pools of the
 * plans' picture size and of EQUAL picture count (the count fixes how many pictures form a group of the layout). */
int hmx_frame_intra_encode_resident(hmx_ctx *ctx, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                    const hmx_tpool *org, hmx_tpool *rec, const hmx_levels *lev);
int hmx_frame_intra_decode_resident(hmx_ctx *ctx, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                    hmx_tpool *rec, const hmx_levels *lev);

/* TComPrediction::motionCompensation for one prediction unit (TComPrediction.cpp:410-552 -> xPredInterUni / xPredInterBi
 * -> the two functions above -> TComYuv::addAvg), the call TEncCu.cpp:1299 and TDecCu.cpp:452 make per coding unit.
 * ref0 / ref1: the reference pictures of list 0 / 1 as HOST planes (plane[i] at sample (0,0), margins readable), NULL =
 * list unused; mv0 / mv1: {hor, ver} in quarter-pel, clipped (hmx_clipMv); (x, y, w, h): the unit in luma samples;
 * dst: HOST planes, plane[i] at the unit's first sample of that plane (a TComYuv part address). */
int hmx_motionCompensation(hmx_ctx *ctx, const hmx_pic *ref0, const int *mv0, const hmx_pic *ref1, const int *mv1, int x, int y,
                           int w, int h, const hmx_pic *dst);

/* Motion compensation of a list of PUs against reference pictures resident in HBM with the
 * reference's margin layout (TLibCommon/TComPicYuv.cpp:82-94): motionCompensation -> xPredInterUni/Bi ->
 * xPredInterLumaBlk/ChromaBlk -> addAvg (TLibCommon/TComPrediction.cpp:410-642). MVs already clipped. */
typedef struct {
  uint16_t x, y;       /* luma position */
  uint8_t w, h;        /* luma size */
  uint8_t ref0, ref1;  /* index into refs[], 255 = list unused */
  int16_t mv0x, mv0y, mv1x, mv1y; /* quarter-pel */
} hmx_pu;
int hmx_batch_motionCompensation(hmx_ctx *ctx, const hmx_pu *d_pus, int n, const hmx_pic *refs, int n_refs,
                                 const hmx_pic *dst);
/* The encoder's sub-pel refinement fan-out (HOT LOOP C): TEncSearch::xPatternSearchFracDIF (TLibEncoder/TEncSearch.cpp:
 * 4480-4514) builds the half- and quarter-sample planes around a unit's integer vector (xExtDIFUpSamplingH / Q, :5982-6165:
 * 2 + 4, then 2 + 8 interpolation calls) and costs nine candidates per stage (xPatternRefinement :711-760) with
 * TComRdCost::xGetHADs (UseHADME, TComRdCost.cpp:2186-2283) or xGetSAD (:488-516).  Here: for every unit i of `pus` (HOST
 * array; ref0 and mv0 = the INTEGER vector in quarter samples, a multiple of 4) and every candidate k the distortion between
 * the original block and the two-stage prediction (filterHorLuma isLast = false, then filterVerLuma isFirst = false, isLast =
 * true -- what the reference's m_filteredBlock planes hold, zero fractions included) at mv0 + offs[k] quarter samples
 * (offs: host array of n_cand {dx, dy} pairs, |dx|, |dy| <= 3; n_cand <= 49: both stages of the reference's search, or the
 * whole 7 x 7 neighbourhood at once): d_cost[i * n_cand + k] (device) = sum over the unit's 8x8 (4x4 when a side is not a
 * multiple of 8) sub-blocks of the Hadamard sum (use_had) or the SAD, >> (bit depth - 8).  The vector-bits term of
 * xPatternRefinement (TComRdCost::getCost(x, y)) and the choice among candidates stay with the caller.
 * refs / org: device pictures; the references with the reference's margins. */
int hmx_batch_subpel_cost(hmx_ctx *ctx, const hmx_pu *pus, int n, const hmx_pic *refs, int n_refs, const hmx_pic *org,
                          const int8_t *offs, int n_cand, int use_had, uint32_t *d_cost);

/* Deblocking filter, the application part (TLibCommon/TComLoopFilter.cpp:571-922: xEdgeFilterLuma, xEdgeFilterChroma,
 * the pel filters, the strong/weak decision), in place on a reconstructed picture whose size
 * is a multiple of 8.  Maps (device) hold one entry per 4x4 luma unit in raster order: d_bs_ver[u] = boundary
 * strength 0..2 of the vertical edge on the unit's LEFT side, d_bs_hor[u] = of the horizontal edge on its TOP side
 * (what xGetBoundaryStrengthSingle :444 leaves in m_aapucBS; 0 on the picture boundary); d_qp[u] = the unit's luma
 * QP (TComDataCU::getQP); d_no_filter[u] != 0 (may be NULL) keeps a unit's samples: IPCM with
 * pcm_loop_filter_disable (:609-614).  (The reference's lossless flags are ORed into variables that stay set for
 * the rest of a CU's edge, :616-617; that CU-shaped stickiness is not reproduced.)  Edges on the 8x8 luma grid
 * are filtered, chroma on its own 8x8 grid for strength 2; all vertical edges, then all horizontal ones. */
/* The boundary strengths of the 8x8-grid edges, xGetBoundaryStrengthSingle (TComLoopFilter.cpp:444-569).  d_units[u]
 * = what the function reads of a 4x4 partition: intra flag, luma cbf of its transform block and, per reference
 * list, a picture id (< 0: list unused; equal ids = same picture) and the motion vector.  d_edge_ver / d_edge_hor[u]:
 * 0 = the unit's left / top side is not a filtered edge (m_aapbEdgeFilter), 1 = filtered edge, 3 = filtered edge
 * that is also a transform-block or coding-block edge (the value xSetEdgefilterTU / xSetEdgefilterMultiple leave
 * in m_aapucBS).  Horizontal edges on a CTU boundary read the P side's motion at the compressed position
 * (g_motionRefer, TComRom.cpp:221-258).  Outputs feed hmx_deblock_picture. */
typedef struct hmx_dbk_unit {
  uint8_t intra, cbf;
  int8_t ref[2];
  int16_t mv[2][2]; /* [list][x, y], quarter-pel */
} hmx_dbk_unit;
int hmx_deblock_strengths(hmx_ctx *ctx, const hmx_dbk_unit *d_units, const uint8_t *d_edge_ver, const uint8_t *d_edge_hor,
                          int pic_w, int pic_h, int is_b_slice, uint8_t *d_bs_ver, uint8_t *d_bs_hor);
int hmx_deblock_picture(hmx_ctx *ctx, const hmx_pic *rec, int pic_w, int pic_h, const uint8_t *d_bs_ver,
                        const uint8_t *d_bs_hor, const int8_t *d_qp, const uint8_t *d_no_filter, int beta_offset_div2,
                        int tc_offset_div2);

/* Sample adaptive offset, the application part (TLibCommon/TComSampleAdaptiveOffset.cpp:781-1240: SAOProcess,
 * processSaoUnitAll, processSaoCuOrg), from the deblocked picture `in` to `out` (different
 * buffers: the reference's line buffers exist to keep the unfiltered neighbours an in-place pass destroys).
 * d_params (device): [component][CTU in raster order], merge flags already resolved: type -1 = off, 0..3 = edge
 * offset classes (horizontal, vertical, 135, 45 degrees), 4 = band offset from band `band` of 32; four offsets,
 * scaled by << (bit depth - min(bit depth, 10)).  n_lcu = CTUs of the picture. */
typedef struct hmx_sao_lcu {
  int8_t type;
  uint8_t band;
  int8_t offset[4];
} hmx_sao_lcu;
int hmx_sao_picture(hmx_ctx *ctx, const hmx_pic *in, const hmx_pic *out, int pic_w, int pic_h, const hmx_sao_lcu *d_params,
                    int n_lcu);

/* Planar 4:2:0 YUV frames, the format either side of the path (TLibVideoIO/TVideoIOYuv.cpp:226-480, SURVEY.md
 * 8f rank 4).  d_file (device) holds one frame as the file does: 8-bit or 16-bit little-endian samples, Y then
 * Cb then Cr.  unpack = TVideoIOYuv::read: the file's (w_full - pad_x) x (h_full - pad_y) samples are padded to
 * the right and below by replication and every sample is scaled from file_bits to the context's bit depth
 * (<< when deeper; (v + half) >> s clipped to [0, 2^bits - 1] when shallower).  pack = TVideoIOYuv::write:
 * scaled the other way, the top-left (w - crop_right) x (h - crop_bottom) samples stored.
 * hmx_yuv_frame_bytes(w, h, file_bits) = size of a frame of w x h samples. */
size_t hmx_yuv_frame_bytes(int w, int h, int file_bits);
int hmx_yuv_unpack(hmx_ctx *ctx, const void *d_file, int file_bits, const hmx_pic *dst, int w_full, int h_full, int pad_x,
                   int pad_y);
int hmx_yuv_pack(hmx_ctx *ctx, const hmx_pic *src, int w, int h, int crop_right, int crop_bottom, int file_bits, void *d_file);

/* One picture's motion compensation in a multi-picture call. */
typedef struct hmx_mc_job {
  const hmx_pu *d_pus; /* device */
  int n_pus;
  const hmx_pic *refs; /* host array [n_refs]; hmx_pu::ref0/ref1 index it */
  int n_refs;
  const hmx_pic *dst;
  int pic_w, pic_h;    /* luma picture size; when > 0 the PUs are scattered into a 4x4-cell map and the
                          prediction runs one lane per cell of the picture (full waves whatever the PU sizes);
                          0: one wave per PU */
} hmx_mc_job;
/* The batch entry points over SEVERAL pictures in one launch each (grid.y = picture): pictures at the
 * same position of different intra-period segments are independent, so a random-access
 * encoder issues one call per GOP position, not one per picture.  The block list (decisions) is shared
 * by the pictures of a call; planes and levels are arrays of n_pics entries.  d_abs_sum, if not NULL,
 * holds n_pics * (blocks of the list) sums, picture-major. */
int hmx_batch_motionCompensation_multi(hmx_ctx *ctx, int n_jobs, const hmx_mc_job *jobs);
int hmx_batch_residual_transformNxN_multi(hmx_ctx *ctx, const hmx_tu_list *list, int n_pics, const hmx_pic *org,
                                          const hmx_pic *pred, const hmx_levels *lev, uint32_t *d_abs_sum,
                                          const hmx_pic_param *pp);
/* The encoder's inter block chain in one pass per block (xEstimateResidualQT's transformNxN followed by
 * invtransformNxN and TComYuv::addClip, TLibEncoder/TEncSearch.cpp:4890-4990): residual org - pred, T, Q
 * (levels out), IQ, IT, rec = Clip(pred + resi).  Same results as hmx_batch_residual_transformNxN_multi
 * followed by hmx_batch_invtransformNxN_multi. */
int hmx_batch_residual_transform_recon_multi(hmx_ctx *ctx, const hmx_tu_list *list, int n_pics, const hmx_pic *org,
                                             const hmx_pic *pred, const hmx_levels *lev, const hmx_pic *rec,
                                             uint32_t *d_abs_sum, const hmx_pic_param *pp);
/* The same with the distortion the encoder takes right behind it (getDistPart(rec, org, DF_SSE), TEncSearch.cpp:4990 ->
 * TComRdCost::xGetSSE*, TComRdCost.cpp:1313-1657): d_sse[picture * blocks + i] = sum of (org - rec)^2 >> 2 * (B - 8) over
 * block i (the caller's order, like d_abs_sum; device, may be NULL), formed in the pass that reconstructs the block. */
int hmx_batch_residual_transform_recon_sse_multi(hmx_ctx *ctx, const hmx_tu_list *list, int n_pics, const hmx_pic *org,
                                                 const hmx_pic *pred, const hmx_levels *lev, const hmx_pic *rec,
                                                 uint32_t *d_abs_sum, uint32_t *d_sse, const hmx_pic_param *pp);
int hmx_batch_invtransformNxN_multi(hmx_ctx *ctx, const hmx_tu_list *list, int n_pics, const hmx_levels *lev,
                                    const hmx_pic *pred, const hmx_pic *out, const hmx_pic_param *pp);
int hmx_pic_extend_border_multi(hmx_ctx *ctx, int n_pics, const hmx_pic *pics, int pic_w, int pic_h, int margin_x,
                                int margin_y);
/* TComPicYuv::extendPicBorder (TLibCommon/TComPicYuv.cpp:248-286); margins in luma samples */
int hmx_pic_extend_border(hmx_ctx *ctx, const hmx_pic *pic, int pic_w, int pic_h, int margin_x, int margin_y);
/* TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3505-3517), host helper */
void hmx_clipMv(int *mvx, int *mvy, int cu_x, int cu_y, int pic_w, int pic_h, int ctu_size);

#ifdef __cplusplus
}
#endif
#endif

/*
 * hmx_oracle.h -- CPU restatement of the HM block hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for libhmx (the HIP product).  It is a from-scratch plain-C
 * restatement of the arithmetic of the reference (fr34k8/thevc = HM 7.2/8-dev) for the path
 * named in BASELINE.json: integer DCT/DST + inverse, flat quantisation (+ sign-bit hiding),
 * de-quantisation, transform skip, intra reference-sample preparation, intra prediction
 * (angular / DC / planar), MC interpolation filters, bi-pred average and picture border
 * extension.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it;
 * the product (thevc_amd/, include/) never links or calls it.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors of its own
 * (SURVEY.md section 4), so the oracle is pinned against the reference ITSELF, compiled from
 * /root/reference by oracle/build_ref.sh into oracle/_ref/libhmref.so: tests/test_oracle_vs_ref.py
 * runs randomized differential checks of every function below against the reference's own
 * functions, and tests/golden/ holds vectors generated from the reference for the GPU box.
 *
 * Conventions: Pel = int16_t, TCoeff = int32_t, strides in elements, B = internal bit depth
 * (g_uiBitDepth + g_uiBitIncrement of the reference, COM/TComRom.cpp:445-448).
 * "COM/" below = /root/reference/source/Lib/TLibCommon/.
 */
#ifndef HMX_ORACLE_H
#define HMX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HMO_REG_DCT 65535u /* COM/TypeDef.h:239 */
#define HMO_SCAN_ZIGZAG 0
#define HMO_SCAN_HOR 1
#define HMO_SCAN_VER 2
#define HMO_SCAN_DIAG 3

/* ---- tables (COM/TComRom.cpp:293-405, 564-698) ---- */
void hmo_dct_matrix(int N, int16_t *T);            /* N x N row-major, N in {4,8,16,32} */
void hmo_dst_matrix(int16_t *T);                   /* 4 x 4 */
const uint32_t *hmo_scan(int scan_idx, int log2n); /* scan position -> raster index, N*N entries */
int hmo_quant_scale(int rem);
int hmo_inv_quant_scale(int rem);
int hmo_chroma_scale(int idx); /* g_aucChromaScale[58] */

/* ---- transforms (COM/TComTrQuant.cpp:417-972, 1542-1704) ---- */
void hmo_fwd_pass(const int16_t *src, int16_t *dst, int N, int shift, int line, int use_dst);
void hmo_inv_pass(const int16_t *src, int16_t *dst, int N, int shift, int line, int use_dst);
void hmo_xTrMxN(const int16_t *block, int16_t *coeff, int N, unsigned mode, int B);
void hmo_xITrMxN(const int16_t *coeff, int16_t *block, int N, unsigned mode, int B);
void hmo_xT(unsigned mode, const int16_t *resi, int stride, int32_t *coef, int N, int B);
void hmo_xIT(unsigned mode, const int32_t *coef, int16_t *resi, int stride, int N, int B);
void hmo_xTransformSkip(const int16_t *resi, int stride, int32_t *coef, int N, int B);
void hmo_xITransformSkip(const int32_t *coef, int16_t *resi, int stride, int N, int B);

/* ---- quantisation (COM/TComTrQuant.cpp:192-222, 977-1355; COM/TComTrQuant.h:79-113) ---- */
typedef struct {
  int qp, per, rem;
} hmo_qp;
hmo_qp hmo_setQPforQuant(int qpy, int is_chroma, int qp_bd_offset, int chroma_qp_offset);

typedef struct {
  int per;         /* m_cQP.m_iPer (dequant) */
  int rem;         /* m_cQP.m_iRem */
  int per_qbits;   /* cQpBase.m_iPer: ADAPTIVE_QP_SELECTION derives iQBits from the slice base QP
                      (COM/TComTrQuant.cpp:1169-1231); equal to per when there is no delta QP */
  int intra_slice; /* I_SLICE -> rounding 171, else 85 */
  int sign_hide;   /* PPS SignHideFlag */
  int scan_idx;    /* result of getCoefScanIdx (COM/TComDataCU.cpp:4014); ZIGZAG is mapped to DIAG */
} hmo_quant_cfg;

int hmo_coef_scan_idx(int N, int is_luma, int is_intra, int dir_mode);
void hmo_xQuant(const int32_t *src, int32_t *dst, int N, int B, const hmo_quant_cfg *cfg,
                uint32_t *ac_sum);
void hmo_xDeQuant(const int32_t *src, int32_t *dst, int N, int B, int per, int rem);
/* xDeQuant's scaling-list branch (COM/TComTrQuant.cpp:1311-1342); coef[N*N] = getDequantCoeff(list, rem, size) */
void hmo_xDeQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, int per, const int32_t *coef);
/* pArlDes of xQuant (rdoq = 0, COM/TComTrQuant.cpp:1229-1249) / of xRateDistOptQuant (rdoq = 1, :1764-1765, 1886-1891) */
void hmo_arlCoeff(const int32_t *src, int32_t *arl, int N, int B, const hmo_quant_cfg *cfg, int rdoq, const int32_t *qtab);
/* xQuant's flat branch with a scaling list: qtab[N*N] = getQuantCoeff(list, rem, size) (COM/TComTrQuant.cpp:1215, 1244); NULL = flat */
void hmo_xQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, const hmo_quant_cfg *cfg, uint32_t *ac_sum, const int32_t *qtab);

/* Rate-distortion optimised quantisation, xRateDistOptQuant (COM/TComTrQuant.cpp:1719-2305) with its
 * helpers (:2315-2735), flat scaling (setErrScaleCoeff :2794-2818), as compiled in the reference
 * (REMOVE_NSQT, REMOVAL_8x2_2x8_CG, REMOVE_NUM_GREATER1, COEF_REMAIN_BIN_REDUCTION 3, C1FLAG_NUMBER 8,
 * C2FLAG_NUMBER 1).  Bit estimates are an INPUT: the table TEncSbac::estBit prepares for the block's size
 * and texture type (estBitsSbacStruct, COM/TComTrQuant.h:59-72, same field order, 1/32768 bit units). */
typedef struct {
  int32_t sig_cg[2][2];    /* significantCoeffGroupBits */
  int32_t sig[42][2];      /* significantBits */
  int32_t last_x[32], last_y[32];
  int32_t greater1[24][2]; /* m_greaterOneBits */
  int32_t greater2[6][2];  /* m_levelAbsBits */
  int32_t cbf[15][2];      /* blockCbpBits */
  int32_t root_cbf[4][2];  /* blockRootCbpBits */
  int32_t scan_zigzag[2], scan_nonzigzag[2];
} hmo_est_bits;
typedef struct {
  int per, rem;      /* m_cQP of the block's texture type */
  int is_luma;       /* eTType == TEXT_LUMA */
  int is_intra;      /* pcCU->isIntra */
  int scan_idx;      /* getCoefScanIdx (HMO_SCAN_*; ZIGZAG is treated as DIAG) */
  int root_cbf;      /* 1: inter luma block with transform index 0 -> blockRootCbpBits[0] (:2139-2144) */
  int cbf_ctx;       /* otherwise: index into blockCbpBits, texture offset included (:2147-2150) */
  int sign_hide;     /* PPS SignHideFlag */
  double lambda;     /* m_dLambda */
} hmo_rdoq_cfg;
void hmo_xRateDistOptQuant(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg,
                           const hmo_est_bits *est, uint32_t *abs_sum);
/* the same with a scaling list: qtab = getQuantCoeff, estab = getErrScaleCoeff of (list, rem, size), per position (NULL = flat) */
void hmo_xRateDistOptQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg, const hmo_est_bits *est,
                                  uint32_t *abs_sum, const int32_t *qtab, const double *estab);
/* transformNxN / invtransformNxN without the TComDataCU plumbing (COM/TComTrQuant.cpp:1373-1450) */
void hmo_transformNxN(const int16_t *resi, int stride, int32_t *level, int N, int B, unsigned mode,
                      int transform_skip, int bypass, const hmo_quant_cfg *cfg, uint32_t *abs_sum);
void hmo_invtransformNxN(int bypass, unsigned mode, int16_t *resi, int stride, const int32_t *level,
                         int N, int B, int per, int rem, int transform_skip);

/* ---- intra reference samples (COM/TComPattern.cpp:213-605, COM/TComDataCU.cpp:1221-1735) ---- */
/* flags: 4*n+1 entries (n = size_luma/4), ordered below-left(bottom first) .. left .. corner ..
 * above .. above-right, exactly the bNeighborFlags order of COM/TComPattern.cpp:240-245.
 * Geometry rule for one slice, one tile, no constrained intra pred.  Returns #available. */
int hmo_intra_avail(int x, int y, int size_luma, int pic_w, int pic_h, int ctu, uint8_t *flags);
/* rec points at the block origin inside a plane; unit = 4 (luma) or 2 (chroma) samples per flag;
 * adi receives the (2N+1)x(2N+1) border buffer of fillReferenceSamples (row 0 and column 0). */
void hmo_fillReferenceSamples(const int16_t *rec, int stride, const uint8_t *flags, int n_avail,
                              int unit, int N, int B, int32_t *adi);
/* luma: writes the [1 2 1] smoothed copy at adi + (2N+1)^2 (COM/TComPattern.cpp:265-306) */
void hmo_filterAdi(int32_t *adi, int N);
int hmo_use_filtered_refs(int mode, int log2n); /* getPredictorPtr decision */

/* ---- intra prediction (COM/TComPrediction.cpp:129-386, 689-730, 1010-1029) ---- */
void hmo_xPredIntraAng(const int32_t *src, int src_stride, int16_t *dst, int dst_stride, int N,
                       int mode, int filter_edge, int B);
/* predIntraGetPredValDC (COM/TComPrediction.cpp:129-167): src points at buffer cell (1,1) */
int16_t hmo_predIntraGetPredValDC(const int32_t *src, int src_stride, int N, int above, int left);
void hmo_xPredIntraPlanar(const int32_t *src, int src_stride, int16_t *dst, int dst_stride, int N);
void hmo_xDCPredFiltering(const int32_t *src, int src_stride, int16_t *dst, int dst_stride, int N);
void hmo_predIntraLumaAng(const int32_t *adi, int mode, int16_t *dst, int dst_stride, int N, int B);
void hmo_predIntraChromaAng(const int32_t *adi, int mode, int16_t *dst, int dst_stride, int N,
                            int B);

/* ---- distortion (COM/TComRdCost.cpp): calcHAD :404-450 (Hadamard SATD over 8x8 sub-blocks when both sizes are
 *      multiples of 8, else 4x4; xCalcHADs4x4 :1684, xCalcHADs8x8 :1778), xGetSSE* :1313-1657 (the IBDI_DISTORTION 0 variant) ---- */
uint32_t hmo_calcHAD(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int B);
uint32_t hmo_getSSE(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int B);

/* ---- inter prediction (COM/TComInterpolationFilter.cpp, COM/TComPrediction.cpp:554-642,
 *      COM/TComYuv.cpp:401-581, COM/TComPicYuv.cpp:248-286) ---- */
void hmo_filterHorLuma(const int16_t *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                       int is_last, int B);
void hmo_filterVerLuma(const int16_t *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                       int is_first, int is_last, int B);
void hmo_filterHorChroma(const int16_t *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                         int is_last, int B);
void hmo_filterVerChroma(const int16_t *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                         int is_first, int is_last, int B);
/* ref points at the co-located block origin in the reference plane (before the MV offset) */
void hmo_predInterLumaBlk(const int16_t *ref, int ref_stride, int mvx, int mvy, int w, int h,
                          int16_t *dst, int dst_stride, int bi, int B);
void hmo_predInterChromaBlk(const int16_t *ref, int ref_stride, int mvx, int mvy, int w, int h,
                            int16_t *dst, int dst_stride, int bi, int B); /* w,h = luma size */
void hmo_addAvg(const int16_t *s0, int s0s, const int16_t *s1, int s1s, int16_t *dst, int ds, int w,
                int h, int B);
void hmo_addClip(const int16_t *pred, int ps, const int16_t *resi, int rs, int16_t *dst, int ds,
                 int w, int h, int B);
void hmo_subtract(const int16_t *org, int os, const int16_t *pred, int ps, int16_t *dst, int ds,
                  int w, int h);
void hmo_extendPicBorder(int16_t *org, int stride, int w, int h, int mx, int my);
void hmo_clipMv(int *mvx, int *mvy, int cu_x, int cu_y, int pic_w, int pic_h, int ctu);

/* ---- frame-level drivers used as the whole-picture oracle ---- */
typedef struct {
  uint16_t x, y;   /* sample position of the block inside its plane */
  uint8_t log2n;   /* block size in samples of its plane */
  uint8_t plane;   /* 0 Y, 1 Cb, 2 Cr */
  uint8_t mode;    /* intra prediction mode 0..34 */
  uint8_t flags;   /* bit0 transform skip */
} hmo_tu;

typedef struct {
  int pic_w, pic_h; /* luma size */
  int ctu;          /* 64 */
  int B;
  int qp;           /* slice/CU luma QP */
  int chroma_qp_offset;
  int sign_hide;
  int inter_slice;  /* != 0: the blocks belong to a P/B slice (quantiser rounding 85 instead of 171, :1142) */
} hmo_frame_cfg;

/* Encoder-side all-intra reconstruction of one picture from decisions (HOT LOOP B of
 * ENC/TEncSearch.cpp:1006-1165 / 1167-1390 with RDOQ=0): for each TU in list order,
 * refs <- recon, pred, resi = org - pred, T, Q, IQ, IT, recon = Clip(pred + resi').
 * planes: pointers to sample (0,0) of each plane; strides in elements.
 * level[p]: dense TCoeff plane of the same geometry as plane p, stride = plane width. */
void hmo_intra_frame_encode(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                            const int16_t *const org[3], const int org_stride[3],
                            int16_t *const rec[3], const int rec_stride[3], int32_t *const level[3]);
/* Decoder-side (DEC/TDecCu.cpp:469-687): levels + modes -> recon. */
void hmo_intra_frame_encode_rdoq(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                                 const int16_t *const org[3], const int org_stride[3],
                                 int16_t *const rec[3], const int rec_stride[3],
                                 int32_t *const level[3], const hmo_est_bits *est /* [luma, chroma][4 sizes] */,
                                 const double *lambda /* [luma, chroma] */);
void hmo_intra_frame_decode(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                            int16_t *const rec[3], const int rec_stride[3],
                            const int32_t *const level[3]);

typedef struct {
  uint16_t x, y;   /* luma position of the PU */
  uint8_t w, h;    /* luma size */
  uint8_t ref0, ref1; /* reference picture slot, 255 = unused list */
  int16_t mv0x, mv0y, mv1x, mv1y; /* quarter-pel, already clipped */
} hmo_pu;

/* motionCompensation for a list of PUs of one picture (COM/TComPrediction.cpp:410-642). */
void hmo_mc_frame(const hmo_pu *pus, int n_pu, int B, const int16_t *const *ref_planes /*[nref*3]*/,
                  const int *ref_strides /*[3]*/, int16_t *const dst[3], const int dst_stride[3]);

/* ---- deblocking filter, application part (COM/TComLoopFilter.cpp:571-922: xEdgeFilterLuma, xEdgeFilterChroma,
 *      xPelFilterLuma/Chroma, xUseStrongFiltering, xCalcDP/DQ, tables :54-62).  Boundary strengths are an INPUT
 *      (xGetBoundaryStrengthSingle :444 derives them from modes, cbf and motion): per 4x4 luma unit in raster
 *      order, bs_ver[u] = strength of the vertical edge on the unit's LEFT side, bs_hor[u] = of the horizontal
 *      edge on its TOP side (0 = no edge); only edges on the 8x8 luma grid are filtered (chroma: on its own
 *      8x8 grid, strength 2 only).  qp[u] = the unit's luma QP (TComDataCU::getQP); no_filter[u] != 0 marks
 *      IPCM-with-filter-disabled or lossless units (may be NULL).  All vertical edges of the picture, then all
 *      horizontal ones (loopFilterPic :153-201). ---- */
/* Boundary strengths, xGetBoundaryStrengthSingle (COM/TComLoopFilter.cpp:444-569), for the edges of the 8x8 grid.
 * units[u]: what the function reads of the partition: intra flag, luma cbf of its transform block, and per
 * reference list a picture id (< 0: list unused; equal ids = same picture) and the motion vector (quarter-pel).
 * edge_ver / edge_hor[u]: 0 = the unit's left / top side is not a filtered edge (m_aapbEdgeFilter), 1 = filtered
 * edge, 3 = filtered edge that is also a transform-block or coding-block edge (the value xSetEdgefilterTU /
 * xSetEdgefilterMultiple leave in m_aapucBS before the strength is computed).  Horizontal edges on a CTU boundary
 * read the P side's motion at the compressed position (g_motionRefer, COM/TComRom.cpp:221-258). */
typedef struct {
  uint8_t intra, cbf;
  int8_t ref[2];
  int16_t mv[2][2]; /* [list][x, y] */
} hmo_dbk_unit;
void hmo_deblock_strengths(const hmo_dbk_unit *units, const uint8_t *edge_ver, const uint8_t *edge_hor, int pic_w, int pic_h,
                           int ctu, int is_b_slice, uint8_t *bs_ver, uint8_t *bs_hor);
void hmo_deblock_picture(int16_t *const planes[3], const int strides[3], int pic_w, int pic_h, int B, const uint8_t *bs_ver,
                         const uint8_t *bs_hor, const int8_t *qp, const uint8_t *no_filter, int beta_offset_div2,
                         int tc_offset_div2);

/* ---- sample adaptive offset, application (COM/TComSampleAdaptiveOffset.cpp:781-1240: processSaoCuOrg,
 *      processSaoUnitAll, tables :92-116, :176-215).  Per CTU and component: type -1 = off, 0..3 = edge offset
 *      classes (horizontal, vertical, 135, 45 degrees), 4 = band offset starting at band `band` (of 32); four
 *      offsets (scaled by << (B - min(B, 10))).  The reference filters in place with line buffers that keep the
 *      unfiltered neighbours: the same as filtering from `in` to `out`.  Samples whose class neighbour lies outside
 *      the picture are copied. ---- */
typedef struct {
  int8_t type;
  uint8_t band;
  int8_t offset[4];
} hmo_sao_lcu;
void hmo_sao_picture(const int16_t *const in[3], int16_t *const out[3], const int strides[3], int pic_w, int pic_h, int B, int ctu,
                     const hmo_sao_lcu *const params[3]);

/* ---- planar 4:2:0 YUV frames as the reference reads and writes them (VIO/TVideoIOYuv.cpp:226-480):
 *      8-bit or 16-bit little-endian samples, Y then Cb then Cr; on read the active area is padded to the
 *      right and below by replication and the whole padded plane is scaled to the internal bit depth
 *      (<< when deeper; (v + half) >> s clipped to [0, 2^bits - 1] when shallower, CLIP_TO_709_RANGE 0);
 *      on write the planes are scaled the other way and the top-left (w - crop_right) x (h - crop_bottom)
 *      samples are stored ---- */
void hmo_yuv_unpack(const uint8_t *file, int file_bits, int internal_bits, int w_full, int h_full, int pad_x, int pad_y,
                    int16_t *const planes[3], const int strides[3]);
void hmo_yuv_pack(const int16_t *const planes[3], const int strides[3], int w, int h, int crop_right, int crop_bottom,
                  int internal_bits, int file_bits, uint8_t *file);

#ifdef __cplusplus
}
#endif
#endif

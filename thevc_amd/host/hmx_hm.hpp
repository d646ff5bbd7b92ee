// hmx_hm.hpp -- C++ host mirror of the reference's block-kernel classes over the C-ABI (include/hmx.h).
//
// Same member names, argument order and meaning as the reference (HM: TLibCommon/TComTrQuant.h,
// TComPrediction.h, TComPattern.h, TComInterpolationFilter.h), so TEncSearch / TEncCu / TDecCu call
// sites keep their shape.  What the reference reads through TComDataCU / TComSlice / globals is
// explicit state here (setBlockState(), the context).  Every call goes to libhmx (HIP); nothing is
// computed on the host.  Errors: the reference returns Void and asserts; these wrappers throw.
#pragma once
#include <stdexcept>
#include <string>

#include "hmx.h"

namespace hmx_hm {

typedef short Pel;
typedef int TCoeff;
typedef int Int;
typedef unsigned UInt;
typedef bool Bool;
typedef double Double;
enum TextType { TEXT_LUMA = 0, TEXT_CHROMA = 1, TEXT_CHROMA_U = 2, TEXT_CHROMA_V = 3 };
static const UInt REG_DCT = 65535;

class Context { // replaces g_uiBitDepth / g_uiBitIncrement / g_uiIBDI_MAX (TComRom.cpp:445-448)
public:
  explicit Context(int bitDepth, int device = 0, int ctuSize = 64) {
    hmx_config cfg = {bitDepth, device, nullptr, ctuSize};
    if (hmx_create(&cfg, &m_ctx) != HMX_OK) throw std::runtime_error("hmx_create failed (no HIP device or bad config)");
    m_bitDepth = bitDepth;
  }
  ~Context() { hmx_destroy(m_ctx); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  hmx_ctx *get() const { return m_ctx; }
  int bitDepth() const { return m_bitDepth; }
  void check(int rc, const char *what) const {
    if (rc != HMX_OK) throw std::runtime_error(std::string(what) + ": " + hmx_last_error(m_ctx));
  }

private:
  hmx_ctx *m_ctx = nullptr;
  int m_bitDepth = 8;
};

// TComTrQuant (TComTrQuant.h:115-316)
class TComTrQuant {
public:
  explicit TComTrQuant(Context &c) : m_c(c) {
    m_qp.qp = hmx_setQPforQuant(0, HMX_TEXT_LUMA, 0, 0);
    m_qp.per_base = -1;
    m_qp.slice_type = HMX_I_SLICE;
    m_qp.sign_hide = 1;
    m_qp.is_intra = 1;
    m_qp.dir_mode = 1;
  }
  // setQPforQuant (TComTrQuant.cpp:192): same arguments; the result is kept like m_cQP
  void setQPforQuant(Int qpy, TextType eTxtType, Int qpBdOffset, Int chromaQPOffset) {
    m_qp.qp = hmx_setQPforQuant(qpy, eTxtType, qpBdOffset, chromaQPOffset);
  }
  // what xQuant reads through pcCU / the slice / the PPS (TComTrQuant.cpp:1121-1267)
  void setBlockState(Bool isIntra, UInt dirMode, Int sliceType, Bool signHideFlag, Int sliceQpBasePer = -1) {
    m_qp.is_intra = isIntra;
    m_qp.dir_mode = (int)dirMode;
    m_qp.slice_type = sliceType;
    m_qp.sign_hide = signHideFlag;
    m_qp.per_base = sliceQpBasePer;
  }
  // transformNxN (TComTrQuant.cpp:1373); pcCU is replaced by setBlockState()
  void transformNxN(Pel *pcResidual, UInt uiStride, TCoeff *rpcCoeff, UInt uiWidth, UInt uiHeight, UInt &uiAbsSum,
                    TextType eTType, Bool useTransformSkip = false, Bool transQuantBypass = false) {
    uint32_t s = 0;
    m_c.check(hmx_transformNxN(m_c.get(), pcResidual, uiStride, rpcCoeff, uiWidth, uiHeight, &s, eTType, &m_qp,
                               useTransformSkip, transQuantBypass),
              "transformNxN");
    uiAbsSum = s;
  }
  // invtransformNxN (TComTrQuant.cpp:1428); scalingListType is accepted and ignored (lists are off)
  void invtransformNxN(Bool transQuantBypass, TextType eText, UInt uiMode, Pel *rpcResidual, UInt uiStride,
                       TCoeff *pcCoeff, UInt uiWidth, UInt uiHeight, Int /*scalingListType*/,
                       Bool useTransformSkip = false) {
    m_c.check(hmx_invtransformNxN(m_c.get(), transQuantBypass, eText, uiMode, rpcResidual, uiStride, pcCoeff, uiWidth,
                                  uiHeight, &m_qp.qp, useTransformSkip),
              "invtransformNxN");
  }
  // invRecurTransformNxN (TComTrQuant.cpp:1452-1529): the walk over the transform quadtree of an inter CU.
  // What the reference asks pcCU becomes a view over the per-partition arrays TComDataCU stores (one entry per
  // 4x4 luma partition in z-order): transform index, coded-block flags of THIS texture, transform-skip flags.
  struct CuTransformTree {
    const unsigned char *trIdx;         // getTransformIdx
    const unsigned char *cbf;           // getCbf(eTxt): bit d = flag at transform depth d
    const unsigned char *transformSkip; // getTransformSkip(eTxt), may be null
    UInt cuDepth;                       // getDepth
    UInt maxCuWidth;                    // SPS getMaxCUWidth (64)
    UInt numPartInLCU;                  // getPic()->getNumPartInCU() (256 for a 64x64 LCU of 4x4 partitions)
    UInt totalNumPart;                  // getTotalNumPart(): partitions of this CU
    Bool transquantBypass;
  };
  void invRecurTransformNxN(const CuTransformTree &cu, UInt uiAbsPartIdx, TextType eTxt, Pel *rpcResidual, UInt uiAddr,
                            UInt uiStride, UInt uiWidth, UInt uiHeight, UInt uiMaxTrMode, UInt uiTrMode, TCoeff *rpcCoeff) {
    if (!((cu.cbf[uiAbsPartIdx] >> uiTrMode) & 1)) return; // nothing coded below this node
    if (uiTrMode == cu.trIdx[uiAbsPartIdx]) {               // a leaf (convertTransIdx is the identity, TComDataCU.cpp:3520)
      const UInt depth = cu.cuDepth + uiTrMode;
      if (eTxt != TEXT_LUMA && (cu.maxCuWidth >> depth) == 4) {
        // four 4x4 luma leaves share one 4x4 chroma block, carried by the first of them (:1467-1476)
        const UInt quarter = cu.numPartInLCU >> ((depth - 1) << 1);
        if (uiAbsPartIdx % quarter) return;
        uiWidth <<= 1;
        uiHeight <<= 1;
      }
      const Bool ts = cu.transformSkip && cu.transformSkip[uiAbsPartIdx];
      invtransformNxN(cu.transquantBypass, eTxt, HMX_REG_DCT, rpcResidual + uiAddr, uiStride, rpcCoeff, uiWidth, uiHeight, 0, ts);
      return;
    }
    const UInt half_w = uiWidth >> 1, half_h = uiHeight >> 1, parts = cu.totalNumPart >> ((uiTrMode + 1) << 1);
    for (UInt q = 0; q < 4; q++) // z-order: coefficients and partitions advance together
      invRecurTransformNxN(cu, uiAbsPartIdx + q * parts, eTxt, rpcResidual, uiAddr + (q & 1) * half_w + (q >> 1) * half_h * uiStride,
                           uiStride, half_w, half_h, uiMaxTrMode, uiTrMode + 1, rpcCoeff + q * half_w * half_h);
  }
  // private members of the reference, named by the north star
  void xT(UInt uiMode, Pel *piBlkResi, UInt uiStride, Int *psCoeff, Int iWidth, Int iHeight) {
    m_c.check(hmx_xT(m_c.get(), uiMode, piBlkResi, uiStride, psCoeff, iWidth, iHeight), "xT");
  }
  void xIT(UInt uiMode, Int *plCoef, Pel *pResidual, UInt uiStride, Int iWidth, Int iHeight) {
    m_c.check(hmx_xIT(m_c.get(), uiMode, plCoef, pResidual, uiStride, iWidth, iHeight), "xIT");
  }
  void xQuant(Int *pSrc, TCoeff *pDes, Int iWidth, Int iHeight, UInt &uiAcSum, TextType eTType) {
    uint32_t s = uiAcSum;
    m_c.check(hmx_xQuant(m_c.get(), pSrc, pDes, iWidth, iHeight, &s, eTType, &m_qp), "xQuant");
    uiAcSum = s;
  }
  void xDeQuant(const TCoeff *pSrc, Int *pDes, Int iWidth, Int iHeight, Int /*scalingListType*/) {
    m_c.check(hmx_xDeQuant(m_c.get(), pSrc, pDes, iWidth, iHeight, &m_qp.qp), "xDeQuant");
  }
  // xRateDistOptQuant (TComTrQuant.cpp:1719): m_pcEstBitsSbac and m_dLambda are members here too (setLambda,
  // TComTrQuant.h:155; the table is filled by the entropy coder's estBit); transform index and cbf context, which
  // the reference reads from pcCU, come through setRdoqBlockState()
  hmx_est_bits *m_pcEstBitsSbac = &m_estBits;
  void setLambda(Double dLambda) { m_dLambda = dLambda; }
  void setRdoqBlockState(Bool rootCbf, Int cbfCtx) {
    m_rootCbf = rootCbf;
    m_cbfCtx = cbfCtx;
  }
  void xRateDistOptQuant(Int *plSrcCoeff, TCoeff *piDstCoeff, UInt uiWidth, UInt uiHeight, UInt &uiAbsSum, TextType eTType) {
    hmx_rdoq_param rp{m_qp.qp, m_qp.sign_hide, m_qp.is_intra, m_qp.dir_mode, m_rootCbf, m_cbfCtx, m_dLambda};
    uint32_t s = uiAbsSum;
    m_c.check(hmx_xRateDistOptQuant(m_c.get(), plSrcCoeff, piDstCoeff, (int)uiWidth, (int)uiHeight, &s, eTType, &rp, m_pcEstBitsSbac),
              "xRateDistOptQuant");
    uiAbsSum = s;
  }
  const hmx_qp &qp() const { return m_qp.qp; }

private:
  Context &m_c;
  hmx_quant_param m_qp;
  hmx_est_bits m_estBits{};
  Double m_dLambda = 1.0;
  int m_rootCbf = 0, m_cbfCtx = 0;
};

// TComPattern + TComPrediction, intra part (TComPattern.cpp:213-366, TComPrediction.cpp:338-386)
class TComPrediction {
public:
  explicit TComPrediction(Context &c) : m_c(c) {}
  // initAdiPattern: the CU walk is replaced by the block geometry inside the reconstructed plane
  void initAdiPattern(const Pel *recPlane, Int stride, Int x, Int y, Int size, Bool chroma, Int picW, Int picH,
                      Int *piAdiBuf) {
    m_c.check(hmx_initAdiPattern(m_c.get(), recPlane, stride, x, y, size, chroma, picW, picH, piAdiBuf), "initAdiPattern");
  }
  void predIntraLumaAng(const Int *piAdiBuf, UInt uiDirMode, Pel *piPred, UInt uiStride, Int iWidth, Int iHeight) {
    m_c.check(hmx_predIntraLumaAng(m_c.get(), piAdiBuf, uiDirMode, piPred, uiStride, iWidth, iHeight), "predIntraLumaAng");
  }
  void predIntraChromaAng(const Int *piSrc, UInt uiDirMode, Pel *piPred, UInt uiStride, Int iWidth, Int iHeight) {
    m_c.check(hmx_predIntraChromaAng(m_c.get(), piSrc, uiDirMode, piPred, uiStride, iWidth, iHeight), "predIntraChromaAng");
  }
  // xPredInterLumaBlk / xPredInterChromaBlk (TComPrediction.cpp:554-642): refBlock = refPic->getLumaAddr(cuAddr, zorder + partAddr),
  // the TComMv as its two components, dst = dstPic->getLumaAddr(partAddr) with dstPic's stride
  void xPredInterLumaBlk(const Pel *refBlock, Int refStride, Int mvHor, Int mvVer, Int width, Int height, Pel *dst, Int dstStride, Bool bi) {
    m_c.check(hmx_xPredInterLumaBlk(m_c.get(), refBlock, refStride, mvHor, mvVer, width, height, dst, dstStride, bi), "xPredInterLumaBlk");
  }
  void xPredInterChromaBlk(const Pel *refBlock, Int refStride, Int mvHor, Int mvVer, Int width, Int height, Pel *dst, Int dstStride, Bool bi) {
    m_c.check(hmx_xPredInterChromaBlk(m_c.get(), refBlock, refStride, mvHor, mvVer, width, height, dst, dstStride, bi), "xPredInterChromaBlk");
  }
  // motionCompensation (TComPrediction.cpp:410-552) of one prediction unit: the reference pictures of the two lists (NULL = unused),
  // their vectors, the unit's luma rectangle, the prediction planes at the unit's first sample
  void motionCompensation(const hmx_pic *ref0, const Int mv0[2], const hmx_pic *ref1, const Int mv1[2], Int x, Int y, Int width, Int height,
                          const hmx_pic *pred) {
    m_c.check(hmx_motionCompensation(m_c.get(), ref0, mv0, ref1, mv1, x, y, width, height, pred), "motionCompensation");
  }

private:
  Context &m_c;
};

// TComRdCost, the two distortion entry points next to the path (TComRdCost.cpp:404-478)
class TComRdCost {
public:
  explicit TComRdCost(Context &c) : m_c(c) {}
  UInt calcHAD(Pel *pi0, Int iStride0, Pel *pi1, Int iStride1, Int iWidth, Int iHeight) {
    uint32_t v = 0;
    m_c.check(hmx_calcHAD(m_c.get(), pi0, iStride0, pi1, iStride1, iWidth, iHeight, &v), "calcHAD");
    return v;
  }
  // getDistPart(..., bWeighted = false, DF_SSE)
  UInt getDistPart(Pel *piCur, Int iCurStride, Pel *piOrg, Int iOrgStride, UInt uiBlkWidth, UInt uiBlkHeight) {
    uint32_t v = 0;
    m_c.check(hmx_getSSE(m_c.get(), piCur, iCurStride, piOrg, iOrgStride, (int)uiBlkWidth, (int)uiBlkHeight, &v), "getDistPart");
    return v;
  }

private:
  Context &m_c;
};

// TComInterpolationFilter (TComInterpolationFilter.cpp:323-415): identical signatures
class TComInterpolationFilter {
public:
  explicit TComInterpolationFilter(Context &c) : m_c(c) {}
  void filterHorLuma(Pel *src, Int srcStride, short *dst, Int dstStride, Int width, Int height, Int frac, Bool isLast) {
    m_c.check(hmx_filterHorLuma(m_c.get(), src, srcStride, dst, dstStride, width, height, frac, isLast), "filterHorLuma");
  }
  void filterVerLuma(Pel *src, Int srcStride, short *dst, Int dstStride, Int width, Int height, Int frac, Bool isFirst,
                     Bool isLast) {
    m_c.check(hmx_filterVerLuma(m_c.get(), src, srcStride, dst, dstStride, width, height, frac, isFirst, isLast), "filterVerLuma");
  }
  void filterHorChroma(Pel *src, Int srcStride, short *dst, Int dstStride, Int width, Int height, Int frac, Bool isLast) {
    m_c.check(hmx_filterHorChroma(m_c.get(), src, srcStride, dst, dstStride, width, height, frac, isLast), "filterHorChroma");
  }
  void filterVerChroma(Pel *src, Int srcStride, short *dst, Int dstStride, Int width, Int height, Int frac, Bool isFirst,
                       Bool isLast) {
    m_c.check(hmx_filterVerChroma(m_c.get(), src, srcStride, dst, dstStride, width, height, frac, isFirst, isLast), "filterVerChroma");
  }

private:
  Context &m_c;
};

} // namespace hmx_hm

#!/usr/bin/env python3
"""oracle/ref_shim_edit.py UNIT -- TEST INFRASTRUCTURE.  Reads one translation unit of the reference where it lies under
/root/reference, replaces the BODIES of its hot-path members by the libhmx calls of INTEGRATION.md section 3 (signatures
untouched) and writes the result to stdout for `g++ -x c++ -` (oracle/build_ref_shim.sh): nothing of the reference is
stored in the repository, and no header, library or generated file of it is substituted.
UNIT = TComTrQuant | TComPrediction | TComInterpolationFilter | TComYuv | TComRdCost (encoder shim only); a second argument
`enc` adds the bodies of the members only the encoder calls (ENC_BODIES)."""
import os
import re
import sys

SRC = os.path.join(os.environ.get("REF_ROOT", "/root/reference"), "source", "Lib", "TLibCommon")

BODIES = {
    "TComTrQuant": {
        # inverse transform, de-quantiser (flat path; every shipped cfg has ScalingList 0), inverse transform skip
        "TComTrQuant::xIT": "HMX_SHIM_CHECK(hmx_xIT(hmx_shim_ctx(), uiMode, plCoef, pResidual, uiStride, iWidth, iHeight));",
        "TComTrQuant::xITransformSkip": "HMX_SHIM_CHECK(hmx_xITransformSkip(hmx_shim_ctx(), plCoef, pResidual, uiStride, width, height));",
        "TComTrQuant::xDeQuant": """hmx_qp q = { m_cQP.m_iQP, m_cQP.m_iPer, m_cQP.m_iRem, m_cQP.m_iBits };
  if (iWidth > (Int)m_uiMaxTrSize) iWidth = iHeight = m_uiMaxTrSize; /* :1290-1294 */
  if (getUseScalingList()) /* :1311-1342: the table setScalingListDec built for this list type, remainder and size */
    HMX_SHIM_CHECK(hmx_xDeQuant_scaled(hmx_shim_ctx(), pSrc, pDes, iWidth, iHeight, &q,
                                       getDequantCoeff(scalingListType, m_cQP.m_iRem, g_aucConvertToBit[iWidth], SCALING_LIST_SQT)));
  else
    HMX_SHIM_CHECK(hmx_xDeQuant(hmx_shim_ctx(), pSrc, pDes, iWidth, iHeight, &q));""",
    },
    "TComPrediction": {
        # bAbove / bLeft are always true in this tree (TComPattern.cpp: the padded reference line replaces them)
        "TComPrediction::predIntraLumaAng": "assert(bAbove && bLeft); HMX_SHIM_CHECK(hmx_predIntraLumaAng(hmx_shim_ctx(), m_piYuvExt, uiDirMode, piPred, uiStride, iWidth, iHeight));",
        "TComPrediction::predIntraChromaAng": "assert(bAbove && bLeft); HMX_SHIM_CHECK(hmx_predIntraChromaAng(hmx_shim_ctx(), piSrc, uiDirMode, piPred, uiStride, iWidth, iHeight));",
    },
    "TComInterpolationFilter": {
        "TComInterpolationFilter::filterHorLuma": "HMX_SHIM_CHECK(hmx_filterHorLuma(hmx_shim_ctx(), src, srcStride, dst, dstStride, width, height, frac, isLast));",
        "TComInterpolationFilter::filterVerLuma": "HMX_SHIM_CHECK(hmx_filterVerLuma(hmx_shim_ctx(), src, srcStride, dst, dstStride, width, height, frac, isFirst, isLast));",
        "TComInterpolationFilter::filterHorChroma": "HMX_SHIM_CHECK(hmx_filterHorChroma(hmx_shim_ctx(), src, srcStride, dst, dstStride, width, height, frac, isLast));",
        "TComInterpolationFilter::filterVerChroma": "HMX_SHIM_CHECK(hmx_filterVerChroma(hmx_shim_ctx(), src, srcStride, dst, dstStride, width, height, frac, isFirst, isLast));",
    },
    "TComYuv": {
        "TComYuv::addAvg": """HMX_SHIM_CHECK(hmx_addAvg(hmx_shim_ctx(), pcYuvSrc0->getLumaAddr(iPartUnitIdx), pcYuvSrc0->getStride(), pcYuvSrc1->getLumaAddr(iPartUnitIdx), pcYuvSrc1->getStride(), getLumaAddr(iPartUnitIdx), getStride(), iWidth, iHeight));
  HMX_SHIM_CHECK(hmx_addAvg(hmx_shim_ctx(), pcYuvSrc0->getCbAddr(iPartUnitIdx), pcYuvSrc0->getCStride(), pcYuvSrc1->getCbAddr(iPartUnitIdx), pcYuvSrc1->getCStride(), getCbAddr(iPartUnitIdx), getCStride(), iWidth >> 1, iHeight >> 1));
  HMX_SHIM_CHECK(hmx_addAvg(hmx_shim_ctx(), pcYuvSrc0->getCrAddr(iPartUnitIdx), pcYuvSrc0->getCStride(), pcYuvSrc1->getCrAddr(iPartUnitIdx), pcYuvSrc1->getCStride(), getCrAddr(iPartUnitIdx), getCStride(), iWidth >> 1, iHeight >> 1));""",
    },
}


# The ENCODER shim (oracle/build_ref_enc_shim.sh, `ref_shim_edit.py UNIT enc`) replaces, on top of the bodies above, the members only
# the encoder calls: the forward transform and transform skip, the quantiser in both its forms (xQuant's flat branch -> hmx_xQuant,
# xRateDistOptQuant -> hmx_xRateDistOptQuant with the encoder's LIVE bit-estimate table and multiplier, TEncSearch.cpp:1101), the
# Hadamard cost of the intra mode pre-selection (TEncSearch.cpp:2534-2537) and the per-block motion compensation.
_SCAN_DIR = """UInt hmxScan = pcCU->getCoefScanIdx(uiAbsPartIdx, %(w)s, eTType == TEXT_LUMA, pcCU->isIntra(uiAbsPartIdx));
  const int hmxDir = hmxScan == 1 ? 26 : hmxScan == 2 ? 10 : 0; /* a mode that selects the same scan (getCoefScanIdx resolved chroma DM already) */"""
ENC_BODIES = {
    "TComTrQuant": {
        "TComTrQuant::xT": "HMX_SHIM_CHECK(hmx_xT(hmx_shim_ctx(), uiMode, piBlkResi, uiStride, psCoeff, iWidth, iHeight));",
        "TComTrQuant::xTransformSkip": "HMX_SHIM_CHECK(hmx_xTransformSkip(hmx_shim_ctx(), piBlkResi, uiStride, psCoeff, width, height));",
        "TComTrQuant::xQuant": """Bool useRDOQForTransformSkip = !(m_useTansformSkipFast && pcCU->getTransformSkip(uiAbsPartIdx,eTType));
  if ( m_bUseRDOQ && (eTType == TEXT_LUMA || RDOQ_CHROMA) && useRDOQForTransformSkip) { /* TComTrQuant.cpp:1121-1128 */
    xRateDistOptQuant( pcCU, pSrc, pDes, pArlDes, iWidth, iHeight, uiAcSum, eTType, uiAbsPartIdx );
    return;
  }
  """ + _SCAN_DIR % {"w": "iWidth"} + """
  const Int *hmxQ = getUseScalingList() /* :1212-1215: the table setScalingList built for this list type, remainder and size */
      ? getQuantCoeff((pcCU->isIntra(uiAbsPartIdx) ? 0 : 3) + g_eTTable[(Int)eTType], m_cQP.m_iRem, g_aucConvertToBit[iWidth], SCALING_LIST_SQT) : NULL;
  QpParam cQpBase; /* iQBits of the flat branch comes from the slice's BASE QP (:1162-1193, ADAPTIVE_QP_SELECTION) */
  {
    Int qpBDOffset = (eTType == TEXT_LUMA) ? pcCU->getSlice()->getSPS()->getQpBDOffsetY() : pcCU->getSlice()->getSPS()->getQpBDOffsetC();
    Int qpScaled = pcCU->getSlice()->getSliceQpBase();
    if (eTType == TEXT_LUMA) qpScaled += qpBDOffset;
    else {
      qpScaled = Clip3(-qpBDOffset, 57, qpScaled);
      qpScaled = qpScaled < 0 ? qpScaled + qpBDOffset : g_aucChromaScale[qpScaled] + qpBDOffset;
    }
    cQpBase.setQpParam(qpScaled);
  }
  hmx_quant_param p;
  p.qp.qp = m_cQP.m_iQP, p.qp.per = m_cQP.m_iPer, p.qp.rem = m_cQP.m_iRem, p.qp.bits = m_cQP.m_iBits;
  p.per_base = cQpBase.m_iPer;
  p.slice_type = pcCU->getSlice()->getSliceType();
  p.sign_hide = pcCU->getSlice()->getPPS()->getSignHideFlag();
  p.is_intra = pcCU->isIntra(uiAbsPartIdx);
  p.dir_mode = hmxDir;
  uint32_t ac = uiAcSum;
  if (hmxQ) HMX_SHIM_CHECK(hmx_xQuant_scaled(hmx_shim_ctx(), pSrc, pDes, iWidth, iHeight, &ac, eTType, &p, hmxQ));
  else HMX_SHIM_CHECK(hmx_xQuant(hmx_shim_ctx(), pSrc, pDes, iWidth, iHeight, &ac, eTType, &p));
  uiAcSum = ac;
  if (m_bUseAdaptQpSelect) HMX_SHIM_CHECK(hmx_arlCoeff(hmx_shim_ctx(), pSrc, pArlDes, iWidth, iHeight, eTType, &p, 0, hmxQ)); /* :1246-1249 */""",
        "TComTrQuant::xRateDistOptQuant": """static_assert(sizeof(estBitsSbacStruct) == sizeof(hmx_est_bits), "estBitsSbacStruct and hmx_est_bits share one layout");
  """ + _SCAN_DIR % {"w": "uiWidth"} + """
  const Int hmxList = (pcCU->isIntra(uiAbsPartIdx) ? 0 : 3) + g_eTTable[(Int)eTType];
  const Int *hmxQ = getUseScalingList() ? getQuantCoeff(hmxList, m_cQP.m_iRem, g_aucConvertToBit[uiWidth], SCALING_LIST_SQT) : NULL;       /* :1760 */
  const double *hmxE = getUseScalingList() ? getErrScaleCoeff(hmxList, g_aucConvertToBit[uiWidth], m_cQP.m_iRem, SCALING_LIST_SQT) : NULL; /* :1759 */
  hmx_rdoq_param p;
  p.qp.qp = m_cQP.m_iQP, p.qp.per = m_cQP.m_iPer, p.qp.rem = m_cQP.m_iRem, p.qp.bits = m_cQP.m_iBits;
  p.sign_hide = pcCU->getSlice()->getPPS()->getSignHideFlag();
  p.is_intra = pcCU->isIntra(uiAbsPartIdx);
  p.dir_mode = hmxDir;
  p.root_cbf = !pcCU->isIntra(uiAbsPartIdx) && eTType == TEXT_LUMA && pcCU->getTransformIdx(uiAbsPartIdx) == 0; /* :2104-2109 */
  p.cbf_ctx = (eTType ? TEXT_CHROMA : eTType) * NUM_QT_CBF_CTX + pcCU->getCtxQtCbf(uiAbsPartIdx, eTType, pcCU->getTransformIdx(uiAbsPartIdx));
  p.lambda = m_dLambda; /* what setLambda / selectLambda left for this component */
  uint32_t s = 0;
  if (hmxQ) HMX_SHIM_CHECK(hmx_xRateDistOptQuant_scaled(hmx_shim_ctx(), plSrcCoeff, piDstCoeff, uiWidth, uiHeight, &s, eTType, &p,
                                                        reinterpret_cast<const hmx_est_bits *>(m_pcEstBitsSbac), hmxQ, hmxE));
  else HMX_SHIM_CHECK(hmx_xRateDistOptQuant(hmx_shim_ctx(), plSrcCoeff, piDstCoeff, uiWidth, uiHeight, &s, eTType, &p,
                                            reinterpret_cast<const hmx_est_bits *>(m_pcEstBitsSbac)));
  uiAbsSum = s;
  if (m_bUseAdaptQpSelect) { /* :1886-1891 */
    hmx_quant_param a;
    a.qp = p.qp, a.per_base = -1, a.slice_type = pcCU->getSlice()->getSliceType(), a.sign_hide = p.sign_hide, a.is_intra = p.is_intra, a.dir_mode = p.dir_mode;
    HMX_SHIM_CHECK(hmx_arlCoeff(hmx_shim_ctx(), plSrcCoeff, piArlDstCoeff, uiWidth, uiHeight, eTType, &a, 1, hmxQ));
  }""",
    },
    "TComPrediction": {
        "TComPrediction::xPredInterLumaBlk": """Pel *ref = refPic->getLumaAddr(cu->getAddr(), cu->getZorderIdxInCU() + partAddr);
  HMX_SHIM_CHECK(hmx_xPredInterLumaBlk(hmx_shim_ctx(), ref, refPic->getStride(), mv->getHor(), mv->getVer(), width, height,
                                       dstPic->getLumaAddr(partAddr), dstPic->getStride(), bi));""",
        "TComPrediction::xPredInterChromaBlk": """Pel *refCb = refPic->getCbAddr(cu->getAddr(), cu->getZorderIdxInCU() + partAddr), *refCr = refPic->getCrAddr(cu->getAddr(), cu->getZorderIdxInCU() + partAddr);
  HMX_SHIM_CHECK(hmx_xPredInterChromaBlk(hmx_shim_ctx(), refCb, refPic->getCStride(), mv->getHor(), mv->getVer(), width, height,
                                         dstPic->getCbAddr(partAddr), dstPic->getCStride(), bi));
  HMX_SHIM_CHECK(hmx_xPredInterChromaBlk(hmx_shim_ctx(), refCr, refPic->getCStride(), mv->getHor(), mv->getVer(), width, height,
                                         dstPic->getCrAddr(partAddr), dstPic->getCStride(), bi));""",
    },
    "TComRdCost": {
        "TComRdCost::calcHAD": """uint32_t satd = 0;
  HMX_SHIM_CHECK(hmx_calcHAD(hmx_shim_ctx(), pi0, iStride0, pi1, iStride1, iWidth, iHeight, &satd));
  return satd;""",
    },
}


# `ref_shim_edit.py TComTrQuant rdoqtap`: the UNMODIFIED xRateDistOptQuant with a recorder at its entry (oracle/ref_rdoq_tap.h: an
# object whose destructor writes the call's inputs -- the live bit-estimate table, the multiplier, the coefficients, what the
# function reads of the CU -- and its outputs to the file named by HMX_RDOQ_TAP).  tests/golden/make_rdoq_enc_tap.py turns a
# run of the reference encoder built this way into the fixture tests/golden/rdoq_enc_tap.npz.
TAP_PREFIX = {
    "TComTrQuant::xRateDistOptQuant": """HmxRdoqTap hmx_tap_(m_pcEstBitsSbac, sizeof(estBitsSbacStruct), m_dLambda, m_cQP.m_iQP, m_cQP.m_iPer, m_cQP.m_iRem, m_cQP.m_iBits,
                      pcCU->getSlice()->getPPS()->getSignHideFlag(), pcCU->isIntra(uiAbsPartIdx),
                      (int)pcCU->getCoefScanIdx(uiAbsPartIdx, uiWidth, eTType == TEXT_LUMA, pcCU->isIntra(uiAbsPartIdx)),
                      !pcCU->isIntra(uiAbsPartIdx) && eTType == TEXT_LUMA && pcCU->getTransformIdx(uiAbsPartIdx) == 0,
                      (eTType ? TEXT_CHROMA : eTType) * NUM_QT_CBF_CTX + pcCU->getCtxQtCbf(uiAbsPartIdx, eTType, pcCU->getTransformIdx(uiAbsPartIdx)),
                      (int)eTType, pcCU->getQP(uiAbsPartIdx), pcCU->getSlice()->getPOC(), plSrcCoeff, piDstCoeff, (int)uiWidth, &uiAbsSum);""",
}


def prefix_body(text, name, code):
    m = re.search(r"^(?:Void|UInt)\s+" + re.escape(name) + r"\s*\(", text, re.M)
    if not m:
        raise SystemExit(f"ref_shim_edit: {name} not found")
    i = text.index("{", text.index(")", m.end()))
    return text[:i + 1] + "\n  " + code + "\n" + text[i + 1:]


def replace_body(text, name, body):
    m = re.search(r"^(?:Void|UInt)\s+" + re.escape(name) + r"\s*\(", text, re.M)
    if not m:
        raise SystemExit(f"ref_shim_edit: {name} not found")
    i = text.index("{", text.index(")", m.end()))
    depth, j = 0, i
    while True:
        c = text[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return text[:i] + "{\n  " + body + "\n}" + text[j + 1:]


def main():
    unit = sys.argv[1]
    text = open(os.path.join(SRC, unit + ".cpp")).read()
    if unit == "TComTrQuant":  # the two pre-standard for-scope uses g++ rejects (oracle/build_ref.sh makes the same edit in its stream)
        text = text.replace("for (Int iCGScanPos = uiCGNum-1;", "Int iCGScanPos; for (iCGScanPos = uiCGNum-1;")
        text = text.replace("for ( Int scanPos = 0; scanPos < iBestLastIdxP1; scanPos++ )", "Int scanPos; for ( scanPos = 0; scanPos < iBestLastIdxP1; scanPos++ )")
    if sys.argv[2:] == ["rdoqtap"]:
        for name, code in TAP_PREFIX.items():
            text = prefix_body(text, name, code)
        sys.stdout.write(text)
        return
    bodies = dict(BODIES.get(unit, {}))
    if sys.argv[2:] == ["enc"]:
        bodies.update(ENC_BODIES.get(unit, {}))
    for name, body in bodies.items():
        text = replace_body(text, name, body)
    text += f'\nstatic HmxShimReport g_hmx_shim_report = {{"{unit}"}};\n'
    sys.stdout.write(text)


if __name__ == "__main__":
    main()

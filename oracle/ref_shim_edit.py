#!/usr/bin/env python3
"""oracle/ref_shim_edit.py UNIT -- TEST INFRASTRUCTURE.  Reads one translation unit of the reference where it lies under
/root/reference, replaces the BODIES of its hot-path members by the libhmx calls of INTEGRATION.md section 3 (signatures
untouched) and writes the result to stdout for `g++ -x c++ -` (oracle/build_ref_shim.sh): nothing of the reference is
stored in the repository, and no header, library or generated file of it is substituted.
UNIT = TComTrQuant | TComPrediction | TComInterpolationFilter | TComYuv"""
import os
import re
import sys

SRC = os.path.join(os.environ.get("REF_ROOT", "/root/reference"), "source", "Lib", "TLibCommon")

BODIES = {
    "TComTrQuant": {
        # inverse transform, de-quantiser (flat path; every shipped cfg has ScalingList 0), inverse transform skip
        "TComTrQuant::xIT": "HMX_SHIM_CHECK(hmx_xIT(hmx_shim_ctx(), uiMode, plCoef, pResidual, uiStride, iWidth, iHeight));",
        "TComTrQuant::xITransformSkip": "HMX_SHIM_CHECK(hmx_xITransformSkip(hmx_shim_ctx(), plCoef, pResidual, uiStride, width, height));",
        "TComTrQuant::xDeQuant": """if (getUseScalingList()) { fprintf(stderr, "libhmx shim: scaling lists are outside the built path\\n"); exit(EXIT_FAILURE); }
  hmx_qp q = { m_cQP.m_iQP, m_cQP.m_iPer, m_cQP.m_iRem, m_cQP.m_iBits };
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


def replace_body(text, name, body):
    m = re.search(r"^Void\s+" + re.escape(name) + r"\s*\(", text, re.M)
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
    for name, body in BODIES[unit].items():
        text = replace_body(text, name, body)
    text += f'\nstatic HmxShimReport g_hmx_shim_report = {{"{unit}"}};\n'
    sys.stdout.write(text)


if __name__ == "__main__":
    main()

// TEST INFRASTRUCTURE (see oracle/README in hmx_oracle.h's header): a tap on the compiled REFERENCE DECODER.
//
//   hm_decision_tap <bitstream.bin> <out.hmxd>
//
// drives the reference's decoder library (TDecTop, built by oracle/build_ref_apps.sh from the sources where they
// lie under /root/reference) over an Annex-B stream and writes, for every decoded picture, what SURVEY.md 8f rank 4
// calls the per-CTU decision list: the transform blocks of the picture in decoding order (position, size, plane,
// resolved intra mode, transform-skip flag), the parsed levels in the reference's own per-CTU coefficient layout,
// and the reference's reconstruction of that picture.  tests/golden/make_stream_golden.py turns the file into a
// fixture; the parity tests then reconstruct the picture from the decisions (oracle on the CPU, libhmx on the
// GPU) and compare with what the reference decoder itself produced.
//
// File layout (little endian, int32 unless noted):
//   magic "HMXD", version 4, n_pictures (patched at the end)
//   per picture: poc, width, height, bit_depth, slice_qp, ctu, slice_type (0 B, 1 P, 2 I), n_tu, n_pu, n_cu,
//                n_tu x { u16 x, u16 y, u8 log2n, u8 plane, u8 mode, u8 flags }      (= hmx_tu, include/hmx.h;
//                         flags bit 0 transform skip, bit 1 block of an inter coding unit, bit 7 (luma blocks) coded
//                         block flag: what the deblocking filter's boundary strength asks of a transform block)
//                n_pu x { u16 x, u16 y, u16 cu_x, u16 cu_y, u8 w, u8 h, i16 poc0, i16 poc1, i16 mv0x, mv0y, mv1x, mv1y }   prediction units
//                         of the inter coding units, merge / skip resolved; pocN = POC of the reference picture
//                         of list N, -32768 = list unused; vectors as decoded (TComDataCU::clipMv, which motion
//                         compensation applies relative to the coding unit's origin cu_x, cu_y, is left to the reader)
//                n_cu x { u16 x, u16 y, u8 log2size, u8 intra, u8 skipped, u8 0 }   coding units (their edges are
//                         deblocking edges also where a unit has no transform block)
//                3 planes x levels  (CTUs in raster order, ctu*ctu ints each (chroma: /4), TComDataCU::m_pcTrCoeff*)
//                3 components x n_ctu x { i8 type, u8 band, i8 offset[4] }  SAO as the decoder parsed it, merges
//                                         resolved (= hmx_sao_lcu; type -1 everywhere when SAO is off for the component)
//                deblocking: int32 disabled (slice flag), beta_offset_div2, tc_offset_div2
//                3 planes x reconstruction (int16, w x h, no margins)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComRom.h"
#include "TLibDecoder/AnnexBread.h"
#include "TLibDecoder/NALread.h"
#define private public /* TDecTop::m_pcPic: the picture under decoding, before TDecGop::filterPicture compresses its motion */
#include "TLibDecoder/TDecTop.h"
#undef private

// The decoder library reports a picture-digest mismatch through a flag its application owns (decmain.cpp).
bool g_md5_mismatch = false;

namespace {

struct Tu {
  uint16_t x, y;
  uint8_t log2n, plane, mode, flags;
};
const uint8_t kTransformSkip = 1, kInter = 2, kCbf = 0x80; // HMX_TU_TRANSFORM_SKIP, HMX_TU_INTER; tap-only: luma cbf
struct Cu {
  uint16_t x, y;
  uint8_t log2size, intra, skipped, pad;
};
struct Pu {
  uint16_t x, y, cu_x, cu_y;
  uint8_t w, h;
  int16_t poc0, poc1, mv0x, mv0y, mv1x, mv1y;
};

void put32(FILE *f, int32_t v) { fwrite(&v, 4, 1, f); }

// Transform blocks of the coding unit that starts at partition `part` of CTU `cu`, in the order
// TDecCu::xIntraRecQT visits them (DEC/TDecCu.cpp:663-687): luma block, then Cb and Cr of the same node; a
// 4x4 luma node carries no chroma except the first of its four siblings, which carries the parent's 4x4 chroma.
void walk(TComDataCU *cu, unsigned part, unsigned depth, unsigned cu_part, int ctu_x, int ctu_y, int ctu, bool inter, std::vector<Tu> &out) {
  const unsigned n_part = cu->getPic()->getNumPartInCU();
  const unsigned leaf_depth = cu->getDepth(cu_part) + cu->getTransformIdx(part);
  const unsigned span = n_part >> (2 * depth); // partitions covered by a node at this depth
  if (depth < leaf_depth) {
    for (unsigned q = 0; q < 4; q++) walk(cu, part + q * (span >> 2), depth + 1, cu_part, ctu_x, ctu_y, ctu, inter, out);
    return;
  }
  const int size = ctu >> depth, log2n = g_aucConvertToBit[size] + 2;
  const unsigned raster = g_auiZscanToRaster[part];
  const int per_row = ctu / 4; // 4x4 units per CTU row (minimum partition is 4x4 at depth 4)
  const int x = ctu_x + (int)(raster % per_row) * 4, y = ctu_y + (int)(raster / per_row) * 4;
  const int luma_mode = inter ? 0 : cu->getLumaIntraDir(part);
  Tu t;
  t.x = (uint16_t)x, t.y = (uint16_t)y, t.log2n = (uint8_t)log2n, t.plane = 0, t.mode = (uint8_t)luma_mode;
  t.flags = (cu->getTransformSkip(part, TEXT_LUMA) ? kTransformSkip : 0) | (inter ? kInter : 0) |
            (cu->getCbf(part, TEXT_LUMA, leaf_depth - cu->getDepth(cu_part)) ? kCbf : 0);
  out.push_back(t);
  // chroma of this node
  unsigned cpart = part;
  int clog = log2n - 1;
  if (log2n == 2) {
    const unsigned parent_span = span << 2;
    if (part % parent_span) return; // not the first of the four 4x4 siblings
    clog = 2;
  }
  int cmode = inter ? 0 : cu->getChromaIntraDir(cu_part);
  if (!inter && cmode == DM_CHROMA_IDX) cmode = cu->getLumaIntraDir(cu_part); // derived from the CU's first luma mode (:576-580)
  for (int pl = 1; pl <= 2; pl++) {
    Tu c;
    c.x = (uint16_t)(x >> 1), c.y = (uint16_t)(y >> 1), c.log2n = (uint8_t)clog, c.plane = (uint8_t)pl, c.mode = (uint8_t)cmode;
    c.flags = (cu->getTransformSkip(cpart, pl == 1 ? TEXT_CHROMA_U : TEXT_CHROMA_V) ? kTransformSkip : 0) | (inter ? kInter : 0);
    out.push_back(c);
  }
}

// Prediction units of the inter coding unit of `size` samples at (x, y) whose first partition is `part`
// (TComDataCU::getPartIndexAndSize, COM/TComDataCU.cpp:3180-3250: the eight partition shapes).
void units_of(TComDataCU *cu, unsigned part, int x, int y, int size, int ctu_x, int ctu_y, int ctu, std::vector<Pu> &out) {
  const int q = size / 4, h2 = size / 2;
  int n = 1, rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0}, rw[4] = {size, 0, 0, 0}, rh[4] = {size, 0, 0, 0};
  switch (cu->getPartitionSize(part)) {
  case SIZE_2Nx2N: break;
  case SIZE_2NxN: n = 2, rh[0] = rh[1] = h2, rw[1] = size, ry[1] = h2; break;
  case SIZE_Nx2N: n = 2, rw[0] = rw[1] = h2, rh[1] = size, rx[1] = h2; break;
  case SIZE_NxN:
    n = 4;
    for (int i = 0; i < 4; i++) rw[i] = rh[i] = h2, rx[i] = (i & 1) * h2, ry[i] = (i >> 1) * h2;
    break;
  case SIZE_2NxnU: n = 2, rh[0] = q, rw[1] = size, rh[1] = size - q, ry[1] = q; break;
  case SIZE_2NxnD: n = 2, rh[0] = size - q, rw[1] = size, rh[1] = q, ry[1] = size - q; break;
  case SIZE_nLx2N: n = 2, rw[0] = q, rh[1] = size, rw[1] = size - q, rx[1] = q; break;
  case SIZE_nRx2N: n = 2, rw[0] = size - q, rh[1] = size, rw[1] = q, rx[1] = size - q; break;
  default: break;
  }
  TComSlice *slice = cu->getSlice();
  for (int i = 0; i < n; i++) {
    const int px = x + rx[i], py = y + ry[i];
    const unsigned idx = g_auiRasterToZscan[((py - ctu_y) / 4) * (ctu / 4) + (px - ctu_x) / 4]; // the unit's first partition
    Pu u;
    u.x = (uint16_t)px, u.y = (uint16_t)py, u.cu_x = (uint16_t)x, u.cu_y = (uint16_t)y, u.w = (uint8_t)rw[i], u.h = (uint8_t)rh[i];
    int16_t *poc[2] = {&u.poc0, &u.poc1}, *mvx[2] = {&u.mv0x, &u.mv1x}, *mvy[2] = {&u.mv0y, &u.mv1y};
    for (int l = 0; l < 2; l++) {
      TComCUMvField *mf = cu->getCUMvField(l ? REF_PIC_LIST_1 : REF_PIC_LIST_0);
      const int ri = mf->getRefIdx((Int)idx);
      *poc[l] = ri < 0 ? (int16_t)-32768 : (int16_t)slice->getRefPOC(l ? REF_PIC_LIST_1 : REF_PIC_LIST_0, ri);
      *mvx[l] = ri < 0 ? 0 : (int16_t)mf->getMv((Int)idx).getHor();
      *mvy[l] = ri < 0 ? 0 : (int16_t)mf->getMv((Int)idx).getVer();
    }
    out.push_back(u);
  }
}

// The decisions of a picture whose slices have all been decoded; called BEFORE the loop filters run, because
// TDecGop::filterPicture ends with TComPic::compressMotion (:289), which overwrites the motion of every 16x16 area
// with that of its first unit.
int collect(TComPic *pic, std::vector<Tu> &tus, std::vector<Pu> &pus, std::vector<Cu> &cus) {
  TComPicYuv *rec = pic->getPicYuvRec();
  const int w = rec->getWidth(), h = rec->getHeight(), ctu = (int)g_uiMaxCUWidth;
  const unsigned n_ctu = pic->getNumCUsInFrame(), per_row = pic->getFrameWidthInCU(), n_part = pic->getNumPartInCU();
  for (unsigned a = 0; a < n_ctu; a++) {
    TComDataCU *cu = pic->getCU(a);
    const int cx = (int)(a % per_row) * ctu, cy = (int)(a / per_row) * ctu;
    for (unsigned part = 0; part < n_part;) {
      const unsigned depth = cu->getDepth(part), span = n_part >> (2 * depth);
      const unsigned raster = g_auiZscanToRaster[part];
      const int x = cx + (int)(raster % (ctu / 4)) * 4, y = cy + (int)(raster / (ctu / 4)) * 4;
      if (x < w && y < h) { // coding units outside the picture are not coded
        if (cu->getIPCMFlag(part) || cu->isLosslessCoded(part)) {
          fprintf(stderr, "hm_decision_tap: PCM / lossless coding units are outside the path\n");
          return 1;
        }
        const bool inter = cu->getPredictionMode(part) != MODE_INTRA;
        Cu cc = {(uint16_t)x, (uint16_t)y, (uint8_t)(g_aucConvertToBit[ctu >> depth] + 2), (uint8_t)!inter, (uint8_t)cu->getSkipFlag(part), 0};
        cus.push_back(cc);
        if (inter) units_of(cu, part, x, y, ctu >> depth, cx, cy, ctu, pus);
        // an inter coding unit without residual has no transform blocks (its reconstruction is its prediction)
        if (!inter || cu->getQtRootCbf(part)) walk(cu, part, depth, part, cx, cy, ctu, inter, tus);
      }
      part += span;
    }
  }
  return 0;
}

int dump_picture(FILE *f, TComPic *pic, const std::vector<Tu> &tus, const std::vector<Pu> &pus, const std::vector<Cu> &cus) {
  TComPicYuv *rec = pic->getPicYuvRec();
  const int w = rec->getWidth(), h = rec->getHeight(), ctu = (int)g_uiMaxCUWidth;
  const int B = (int)(g_uiBitDepth + g_uiBitIncrement);
  const unsigned n_ctu = pic->getNumCUsInFrame();
  put32(f, pic->getPOC()), put32(f, w), put32(f, h), put32(f, B), put32(f, pic->getSlice(0)->getSliceQp()), put32(f, ctu);
  put32(f, (int32_t)pic->getSlice(0)->getSliceType()), put32(f, (int32_t)tus.size()), put32(f, (int32_t)pus.size());
  put32(f, (int32_t)cus.size());
  fwrite(tus.data(), sizeof(Tu), tus.size(), f);
  fwrite(pus.data(), sizeof(Pu), pus.size(), f);
  fwrite(cus.data(), sizeof(Cu), cus.size(), f);
  for (int pl = 0; pl < 3; pl++) {
    const size_t per_ctu = (size_t)ctu * ctu >> (pl ? 2 : 0);
    for (unsigned a = 0; a < n_ctu; a++) {
      TComDataCU *cu = pic->getCU(a);
      const TCoeff *c = pl == 0 ? cu->getCoeffY() : pl == 1 ? cu->getCoeffCb() : cu->getCoeffCr();
      fwrite(c, sizeof(TCoeff), per_ctu, f);
    }
  }
  // SAO parameters of the picture (TDecGop::filterPicture :243-262 hands exactly these to SAOProcess)
  TComSlice *slice = pic->getSlice(0);
  SAOParam *sp = pic->getPicSym()->getSaoParam();
  const bool sao_on = slice->getSPS()->getUseSAO() && slice->getSaoEnabledFlag() && sp;
  for (int c = 0; c < 3; c++) {
    const bool on = sao_on && (c == 0 ? slice->getSaoEnabledFlag() : slice->getSaoEnabledFlagChroma());
    for (unsigned a = 0; a < n_ctu; a++) {
      int8_t q[6] = {-1, 0, 0, 0, 0, 0};
      if (on) {
        const SaoLcuParam &L = sp->saoLcuParam[c][a];
        q[0] = (int8_t)L.typeIdx, q[1] = (int8_t)(uint8_t)L.subTypeIdx;
        for (int i = 0; i < 4; i++) q[2 + i] = (int8_t)L.offset[i];
      }
      fwrite(q, 1, 6, f);
    }
  }
  put32(f, slice->getLoopFilterDisable() ? 1 : 0), put32(f, slice->getLoopFilterBetaOffset()),
      put32(f, slice->getLoopFilterTcOffset());
  for (int pl = 0; pl < 3; pl++) {
    const Pel *p = pl == 0 ? rec->getLumaAddr() : pl == 1 ? rec->getCbAddr() : rec->getCrAddr();
    const int s = pl ? rec->getCStride() : rec->getStride(), pw = w >> (pl ? 1 : 0), ph = h >> (pl ? 1 : 0);
    for (int r = 0; r < ph; r++) fwrite(p + (size_t)r * s, sizeof(Pel), pw, f);
  }
  return 0;
}

} // namespace

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s <bitstream> <out.hmxd>\n", argv[0]);
    return 2;
  }
  std::ifstream in(argv[1], std::ifstream::in | std::ifstream::binary);
  if (!in) {
    fprintf(stderr, "hm_decision_tap: cannot open %s\n", argv[1]);
    return 2;
  }
  // all NAL units first: the decoder wants the first slice of a new picture a second time once it has closed
  // the previous picture (TDecTop::decode returns true and has consumed nothing), which is an index step here
  std::vector<std::vector<uint8_t> > units;
  {
    InputByteStream bs(in);
    while (!!in) {
      std::vector<uint8_t> u;
      AnnexBStats stats = AnnexBStats();
      byteStreamNALUnit(bs, u, stats);
      if (!u.empty()) units.push_back(u);
    }
  }
  FILE *f = fopen(argv[2], "wb");
  if (!f) return 2;
  fwrite("HMXD", 1, 4, f);
  put32(f, 4), put32(f, 0);
  TDecTop dec;
  dec.create();
  dec.init();
  dec.setPictureDigestEnabled(true);
  int skip = 0, last_display = -1, n_pics = 0, rc = 0;
  auto close_picture = [&]() {
    TComPic *cur = dec.m_pcPic;
    if (!cur) return;
    std::vector<Tu> tus;
    std::vector<Pu> pus;
    std::vector<Cu> cus;
    rc |= collect(cur, tus, pus, cus);
    UInt poc = 0;
    TComList<TComPic *> *list = NULL;
    dec.executeDeblockAndAlf(poc, list, skip, last_display); // loop filters; the picture's samples are final after this
    if (!list || rc) return;
    rc |= dump_picture(f, cur, tus, pus, cus);
    n_pics++;
  };
  for (size_t i = 0; i < units.size() && !rc;) {
    std::vector<uint8_t> bytes = units[i]; // read() rewrites its buffer
    InputNALUnit nalu;
    read(nalu, bytes);
    if (dec.decode(nalu, skip, last_display))
      close_picture(); // the unit at i opens the next picture: feed it again
    else
      i++;
  }
  if (!rc) close_picture();
  fseek(f, 8, SEEK_SET);
  put32(f, n_pics);
  fclose(f);
  dec.deletePicBuffer();
  dec.destroy();
  fprintf(stderr, "hm_decision_tap: %d picture(s)%s\n", n_pics, g_md5_mismatch ? ", PICTURE DIGEST MISMATCH" : "");
  return rc || g_md5_mismatch;
}

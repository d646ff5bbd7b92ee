// hmx_inter.hip: interpolation filters, motion compensation, sub-pel cost fan-out, border extension -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#include "hmx_host.h"

// =============================================================================================
// Interpolation (TComInterpolationFilter.cpp), addAvg, motion compensation, border extension
// =============================================================================================
__device__ __forceinline__ int luma_tap(int frac, int t) {
  constexpr signed char k[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1},
                                   {0, 1, -5, 17, 58, -10, 4, -1}};
  return k[frac][t];
}
__device__ __forceinline__ int chroma_tap(int frac, int t) {
  constexpr signed char k[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                   {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
  return k[frac][t];
}

// One output sample of filterHor*/filterVer* incl. the frac == 0 filterCopy cases (:91-244).
// src points at the sample co-located with the output; step = 1 (horizontal) or the stride.
template <int NTAP>
__device__ __forceinline__ int interp_sample(const short *src, int step, int frac, bool first, bool last, int B) {
  const int head = 14 - B, maxv = (1 << B) - 1;
  if (frac == 0) {
    int v = src[0];
    if (first == last) return v;
    if (first) return wrap16(wrap16(v << head) - 8192);
    int off = wrap16(8192 + (head ? (1 << (head - 1)) : 0));
    return clip3(0, maxv, wrap16((v + off) >> head));
  }
  int shift = 6, offset;
  if (last) {
    shift += first ? 0 : head;
    offset = (1 << (shift - 1)) + (first ? 0 : 8192 << 6);
  } else {
    shift -= first ? head : 0;
    offset = first ? -(8192 << shift) : 0;
  }
  int sum = 0;
#pragma unroll
  for (int t = 0; t < NTAP; t++) sum += src[(t - (NTAP / 2 - 1)) * step] * (NTAP == 8 ? luma_tap(frac, t) : chroma_tap(frac, t));
  int v = wrap16((sum + offset) >> shift); // narrowed to Short before the clip (:232-236)
  return last ? clip3(0, maxv, v) : v;
}

__global__ void k_filter(const short *src, int ss, short *dst, int ds, int w, int h, int frac, int chroma, int vertical,
                         int first, int last, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  int r = i / w, col = i % w;
  const short *p = src + (size_t)r * ss + col;
  int step = vertical ? ss : 1;
  dst[(size_t)r * ds + col] =
      (short)(chroma ? interp_sample<4>(p, step, frac, first, last, B) : interp_sample<8>(p, step, frac, first, last, B));
}

static int filter_scalar(hmx_ctx *c, const hmx_pel *src, int ss, int16_t *dst, int ds, int w, int h, int frac, int chroma,
                         int vertical, int first, int last) {
  if (!c || !src || !dst || w <= 0 || h <= 0 || w > 128 || h > 128 || frac < 0 || frac >= (chroma ? 8 : 4))
    return fail(c, HMX_ERR_ARG, "filter: bad argument");
  const int before = frac ? (chroma ? 1 : 3) : 0, after = frac ? (chroma ? 2 : 4) : 0;
  const int ww = w + (vertical ? 0 : before + after), wh = h + (vertical ? before + after : 0);
  Scratch s{c};
  short *d_in = s.take<short>((size_t)ww * wh), *d_out = s.take<short>((size_t)w * h);
  const hmx_pel *h0 = src - (vertical ? (ptrdiff_t)before * ss : before);
  int r = up2d(c, d_in, h0, 2, ww, wh, ss);
  if (r) return r;
  const short *d_org = d_in + (vertical ? before * ww : before);
  hipLaunchKernelGGL(k_filter, dim3((w * h + 255) / 256), dim3(256), 0, c->stream, d_org, ww, d_out, w, w, h, frac, chroma,
                     vertical, first, last, c->cfg.bit_depth);
  HIPCHK(c, hipGetLastError());
  return down2d(c, dst, ds, d_out, 2, w, h);
}
extern "C" int hmx_filterHorLuma(hmx_ctx *c, const hmx_pel *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                                 int is_last) {
  return filter_scalar(c, src, ss, dst, ds, w, h, frac, 0, 0, 1, is_last);
}
extern "C" int hmx_filterVerLuma(hmx_ctx *c, const hmx_pel *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                                 int is_first, int is_last) {
  return filter_scalar(c, src, ss, dst, ds, w, h, frac, 0, 1, is_first, is_last);
}
extern "C" int hmx_filterHorChroma(hmx_ctx *c, const hmx_pel *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                                   int is_last) {
  return filter_scalar(c, src, ss, dst, ds, w, h, frac, 1, 0, 1, is_last);
}
extern "C" int hmx_filterVerChroma(hmx_ctx *c, const hmx_pel *src, int ss, int16_t *dst, int ds, int w, int h, int frac,
                                   int is_first, int is_last) {
  return filter_scalar(c, src, ss, dst, ds, w, h, frac, 1, 1, is_first, is_last);
}

// ---- scalar drop-ins of the inter prediction of ONE block (host pointers) ----
// xPredInterLumaBlk / xPredInterChromaBlk (TComPrediction.cpp:554-642): the window the filters reach goes up once, the
// one or two filter stages run on the device (the reference's three cases: horizontal only, vertical only, horizontal
// into the 14-bit intermediate then vertical), the block comes back.  w, h: the block IN ITS PLANE.
static int pred_inter_blk(hmx_ctx *c, const hmx_pel *ref, int ref_stride, int mvx, int mvy, int w, int h, hmx_pel *dst, int dst_stride, int bi,
                          int chroma, short *d_keep = nullptr) {
  if (!c || !ref || (!dst && !d_keep) || w <= 0 || h <= 0 || w > 64 || h > 64) return fail(c, HMX_ERR_ARG, "xPredInterBlk: bad argument");
  const int fb = chroma ? 3 : 2, fm = (1 << fb) - 1, xf = mvx & fm, yf = mvy & fm;
  const int before = chroma ? 1 : 3, after = chroma ? 2 : 4, ww = w + before + after, wh = h + before + after;
  Scratch s{c};
  short *d_in = s.take<short>((size_t)ww * wh), *d_tmp = s.take<short>((size_t)w * wh), *d_out = d_keep ? d_keep : s.take<short>((size_t)w * h);
  const hmx_pel *h0 = ref + (ptrdiff_t)((mvy >> fb) - before) * ref_stride + ((mvx >> fb) - before);
  int r = up2d(c, d_in, h0, 2, ww, wh, ref_stride);
  if (r) return r;
  const short *d_blk = d_in + before * ww + before; // the block's first sample inside the window
  const int B = c->cfg.bit_depth, last = !bi;
  const dim3 g1((w * h + 255) / 256), g2((w * wh + 255) / 256), blk(256);
  if (yf == 0) {
    hipLaunchKernelGGL(k_filter, g1, blk, 0, c->stream, d_blk, ww, d_out, w, w, h, xf, chroma, 0, 1, last, B);
  } else if (xf == 0) {
    hipLaunchKernelGGL(k_filter, g1, blk, 0, c->stream, d_blk, ww, d_out, w, w, h, yf, chroma, 1, 1, last, B);
  } else { // rows -before .. h+after-1 through the horizontal stage (isLast = false), then the vertical one (isFirst = false)
    hipLaunchKernelGGL(k_filter, g2, blk, 0, c->stream, d_in + before, ww, d_tmp, w, w, wh, xf, chroma, 0, 1, 0, B);
    hipLaunchKernelGGL(k_filter, g1, blk, 0, c->stream, d_tmp + before * w, w, d_out, w, w, h, yf, chroma, 1, 0, last, B);
  }
  HIPCHK(c, hipGetLastError());
  return d_keep ? HMX_OK : down2d(c, dst, dst_stride, d_out, 2, w, h);
}
extern "C" int hmx_xPredInterLumaBlk(hmx_ctx *c, const hmx_pel *ref, int ref_stride, int mv_hor, int mv_ver, int w, int h, hmx_pel *dst,
                                     int dst_stride, int bi) {
  return pred_inter_blk(c, ref, ref_stride, mv_hor, mv_ver, w, h, dst, dst_stride, bi, 0);
}
extern "C" int hmx_xPredInterChromaBlk(hmx_ctx *c, const hmx_pel *ref, int ref_stride, int mv_hor, int mv_ver, int w, int h, hmx_pel *dst,
                                       int dst_stride, int bi) {
  if ((w & 1) || (h & 1)) return fail(c, HMX_ERR_ARG, "hmx_xPredInterChromaBlk: odd luma size");
  return pred_inter_blk(c, ref, ref_stride, mv_hor, mv_ver, w >> 1, h >> 1, dst, dst_stride, bi, 1);
}
__global__ void k_addavg(const short *a, const short *b, short *d, int n, int B);
// motionCompensation of ONE prediction unit (TComPrediction.cpp:410-552): xPredInterUni per used list (isLast = uni-prediction),
// TComYuv::addAvg when both lists are used.  ref0 / ref1: planes of the reference pictures (plane[i] at sample (0,0), margins
// readable), NULL = list unused; (x, y, w, h): the unit in luma samples; dst: plane[i] at the unit's first sample.
extern "C" int hmx_motionCompensation(hmx_ctx *c, const hmx_pic *ref0, const int *mv0, const hmx_pic *ref1, const int *mv1, int x, int y, int w,
                                      int h, const hmx_pic *dst) {
  if (!c || !dst || (!ref0 && !ref1) || (ref0 && !mv0) || (ref1 && !mv1) || w <= 0 || h <= 0 || w > 64 || h > 64 || (w & 1) || (h & 1))
    return fail(c, HMX_ERR_ARG, "hmx_motionCompensation: bad argument");
  const bool bi = ref0 && ref1;
  for (int p = 0; p < 3; p++) {
    const int ch = p ? 1 : 0, pw = w >> ch, ph = h >> ch;
    short *d_pred[2] = {nullptr, nullptr};
    if (bi) { // both 14-bit intermediates stay on the device (the tail of the scratch area), addAvg there
      d_pred[0] = reinterpret_cast<short *>(c->d_scratch + c->scratch_bytes) - 2 * 64 * 64;
      d_pred[1] = d_pred[0] + 64 * 64;
    }
    for (int l = 0; l < 2; l++) {
      const hmx_pic *rp = l ? ref1 : ref0;
      const int *mv = l ? mv1 : mv0;
      if (!rp) continue;
      const hmx_pel *r0 = rp->plane[p] + (ptrdiff_t)(y >> ch) * rp->stride[p] + (x >> ch);
      int r = pred_inter_blk(c, r0, rp->stride[p], mv[0], mv[1], pw, ph, dst->plane[p], dst->stride[p], bi, ch, bi ? d_pred[l] : nullptr);
      if (r) return r;
    }
    if (bi) {
      Scratch s{c};
      short *d_out = s.take<short>((size_t)pw * ph);
      hipLaunchKernelGGL(k_addavg, dim3((pw * ph + 255) / 256), dim3(256), 0, c->stream, d_pred[0], d_pred[1], d_out, pw * ph, c->cfg.bit_depth);
      HIPCHK(c, hipGetLastError());
      int r = down2d(c, dst->plane[p], dst->stride[p], d_out, 2, pw, ph);
      if (r) return r;
    }
  }
  return HMX_OK;
}

// ---- the encoder's sub-pel refinement fan-out (HOT LOOP C) ----
// xPatternSearchFracDIF (TEncSearch.cpp:4480-4514) makes the half- and quarter-sample planes of a prediction unit
// (xExtDIFUpSamplingH / Q, :5982-6165: filterHorLuma(frac x, isLast = false) into the 14-bit intermediate, then
// filterVerLuma(frac y, isFirst = false, isLast = true), zero fractions included) and costs nine candidates per stage
// (xPatternRefinement, :711-760) with xGetHADs / xGetSAD (TComRdCost.cpp:2186-2283, :488-516).  The sample a plane holds
// at a candidate's position depends on the position alone, so the fan-out is: for every unit and every candidate
// displacement (integer vector + up to 3 quarter samples either way) the distortion of the displaced two-stage
// prediction against the original.  One thread = one 8x8 (4x4) sub-block of one unit at one candidate: column by column
// the horizontal stage of the 15 (11) rows it needs, the vertical stage, the difference; then the Hadamard sum of the
// sub-block (rounded per sub-block as the reference does) or its SAD, added to the unit's candidate.
struct SubpelArgs {
  const hmx_pu *pus;
  const uint32_t *first; // [n + 1] prefix of sub-blocks per unit
  int n;
  PlanesDev refs[4];
  PlanesDev org;
  const signed char *offs; // [n_cand][2]
  int n_cand, use_had, B;
  uint32_t *cost; // [n][n_cand]
};
__global__ __launch_bounds__(64) void k_subpel_cost(SubpelArgs A) {
  const uint32_t sb = blockIdx.x * blockDim.x + threadIdx.x;
  const int cand = blockIdx.y;
  if (sb >= A.first[A.n]) return;
  int lo = 0, hi = A.n; // the unit this sub-block belongs to
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (A.first[mid] <= sb) lo = mid;
    else hi = mid;
  }
  const hmx_pu pu = A.pus[lo];
  const int w = pu.w, h = pu.h, n = (w % 8 == 0 && h % 8 == 0) ? 8 : 4, bw = w / n, k = (int)(sb - A.first[lo]);
  const int bx = pu.x + (k % bw) * n, by = pu.y + (k / bw) * n;
  const int mvx = pu.mv0x + A.offs[2 * cand], mvy = pu.mv0y + A.offs[2 * cand + 1];
  const int xf = mvx & 3, yf = mvy & 3, B = A.B, head = 14 - B, maxv = (1 << B) - 1;
  const PlanesDev &R = A.refs[pu.ref0 < 4 ? pu.ref0 : 0];
  const short *ref = R.p[0] + (ptrdiff_t)(by + (mvy >> 2)) * R.s[0] + bx + (mvx >> 2);
  const short *org = A.org.p[0] + (size_t)by * A.org.s[0] + bx;
  int d[64];
  for (int c = 0; c < n; c++) {
    int t[15]; // horizontal stage of rows -3 .. n+3 of this column (isFirst = true, isLast = false)
    for (int r = 0; r < n + 7; r++) t[r] = interp_sample<8>(ref + (ptrdiff_t)(r - 3) * R.s[0] + c, 1, xf, true, false, B);
    for (int r = 0; r < n; r++) {
      int v;
      if (yf == 0) { // filterCopy, last only (:124-145)
        const int off = wrap16(8192 + (head ? (1 << (head - 1)) : 0));
        v = clip3(0, maxv, wrap16((t[r + 3] + off) >> head));
      } else {
        const int shift = 6 + head, offset = (1 << (shift - 1)) + (8192 << 6);
        int sum = 0;
        for (int q = 0; q < 8; q++) sum += t[r + q] * luma_tap(yf, q);
        v = clip3(0, maxv, wrap16((sum + offset) >> shift));
      }
      d[r * 8 + c] = org[(size_t)r * A.org.s[0] + c] - v;
    }
  }
  int sum = 0;
  if (!A.use_had) {
    for (int r = 0; r < n; r++)
      for (int c = 0; c < n; c++) sum += abs(d[r * 8 + c]);
  } else if (n == 8) {
    for (int r = 0; r < 8; r++) wht_regs<8>(d + r * 8);
    for (int c = 0; c < 8; c++) {
      int col[8];
      for (int r = 0; r < 8; r++) col[r] = d[r * 8 + c];
      wht_regs<8>(col);
      for (int r = 0; r < 8; r++) sum += abs(col[r]);
    }
    sum = (sum + 2) >> 2;
  } else {
    for (int r = 0; r < 4; r++) wht_regs<4>(d + r * 8);
    for (int c = 0; c < 4; c++) {
      int col[4];
      for (int r = 0; r < 4; r++) col[r] = d[r * 8 + c];
      wht_regs<4>(col);
      for (int r = 0; r < 4; r++) sum += abs(col[r]);
    }
    sum = (sum + 1) >> 1;
  }
  atomicAdd(&A.cost[(size_t)lo * A.n_cand + cand], (unsigned)sum);
}
__global__ void k_shift_u32(uint32_t *v, size_t n, int sh) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] >>= sh;
}
extern "C" int hmx_batch_subpel_cost(hmx_ctx *c, const hmx_pu *pus, int n, const hmx_pic *refs, int n_refs, const hmx_pic *org,
                                     const int8_t *offs, int n_cand, int use_had, uint32_t *d_cost) {
  if (!c || !pus || n <= 0 || !refs || n_refs <= 0 || n_refs > 4 || !org || !offs || n_cand <= 0 || n_cand > 49 || !d_cost)
    return fail(c, HMX_ERR_ARG, "hmx_batch_subpel_cost: bad argument");
  std::vector<uint32_t> first((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) {
    const int w = pus[i].w, h = pus[i].h;
    if (w <= 0 || h <= 0 || w > 64 || h > 64 || ((w | h) & 3) || pus[i].ref0 >= n_refs || ((pus[i].mv0x | pus[i].mv0y) & 3))
      return fail(c, HMX_ERR_ARG, "hmx_batch_subpel_cost: unit size not a multiple of 4, reference index, or a vector that is not integer");
    const int nb = (w % 8 == 0 && h % 8 == 0) ? 8 : 4;
    first[i + 1] = first[i] + (uint32_t)((w / nb) * (h / nb));
  }
  for (int k = 0; k < 2 * n_cand; k++)
    if (offs[k] < -3 || offs[k] > 3) return fail(c, HMX_ERR_ARG, "hmx_batch_subpel_cost: candidate further than 3 quarter samples");
  SubpelArgs A{};
  const size_t pu_bytes = sizeof(hmx_pu) * (size_t)n, first_bytes = sizeof(uint32_t) * ((size_t)n + 1);
  // unit list and prefix through the argument arena (they are the caller's host arrays)
  A.pus = static_cast<const hmx_pu *>(arena_push(c, pus, pu_bytes));
  A.first = static_cast<const uint32_t *>(arena_push(c, first.data(), first_bytes));
  A.offs = static_cast<const signed char *>(arena_push(c, offs, (size_t)2 * n_cand));
  if (!A.pus || !A.first || !A.offs) return fail(c, HMX_ERR_NOMEM, "argument arena (unit list too long: split the call)");
  A.n = n;
  for (int r = 0; r < n_refs; r++) A.refs[r] = to_dev(&refs[r]);
  A.org = to_dev(org);
  A.n_cand = n_cand;
  A.use_had = use_had;
  A.B = c->cfg.bit_depth;
  A.cost = d_cost;
  HIPCHK(c, hipMemsetAsync(d_cost, 0, sizeof(uint32_t) * (size_t)n * n_cand, c->stream));
  hipLaunchKernelGGL(k_subpel_cost, dim3((first[n] + 63) / 64, (unsigned)n_cand), dim3(64), 0, c->stream, A);
  if (c->cfg.bit_depth > 8) // xGetHADs / xGetSAD return uiSum >> g_uiBitIncrement
    hipLaunchKernelGGL(k_shift_u32, dim3((unsigned)(((size_t)n * n_cand + 255) / 256)), dim3(256), 0, c->stream, d_cost, (size_t)n * n_cand, c->cfg.bit_depth - 8);
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

__device__ __forceinline__ int add_avg(int a, int b, int B) { // TComYuv.cpp:539-540
  const int sh = 15 - B, off = (1 << (sh - 1)) + 2 * 8192;
  return clip3(0, (1 << B) - 1, (a + b + off) >> sh);
}
__global__ void k_addavg(const short *a, const short *b, short *d, int n, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = (short)add_avg(a[i], b[i], B);
}
extern "C" int hmx_addAvg(hmx_ctx *c, const hmx_pel *s0, int s0s, const hmx_pel *s1, int s1s, hmx_pel *dst, int ds, int w,
                          int h) {
  if (!c || !s0 || !s1 || !dst || w <= 0 || h <= 0 || w > 128 || h > 128) return fail(c, HMX_ERR_ARG, "hmx_addAvg: bad argument");
  Scratch s{c};
  short *da = s.take<short>((size_t)w * h), *db = s.take<short>((size_t)w * h), *dd = s.take<short>((size_t)w * h);
  int r = up2d(c, da, s0, 2, w, h, s0s);
  if (!r) r = up2d(c, db, s1, 2, w, h, s1s);
  if (r) return r;
  hipLaunchKernelGGL(k_addavg, dim3((w * h + 255) / 256), dim3(256), 0, c->stream, da, db, dd, w * h, c->cfg.bit_depth);
  HIPCHK(c, hipGetLastError());
  return down2d(c, dst, ds, dd, 2, w, h);
}

// ---- motionCompensation over PU lists ----
struct McJob { // one picture: its prediction units, its reference pictures, its destination planes
  const hmx_pu *pus;
  int n, ref_off; // refs of this job start at McArgs::refs[ref_off]
  PlanesDev dst;
  int *map;       // cell -> PU index (-1: none), cw x ch cells of 4x4 luma samples; NULL: one wave per PU
  int cw, ch;
};
struct McArgs {
  const McJob *jobs;     // [grid.y]
  const PlanesDev *refs; // all jobs' reference tables, back to back
  int B;
};

// ---- the prediction of one cell, on packed 16-bit pairs ----
// A cell is 4x4 luma samples (2x2 chroma) of one PU.  Its reference window is read row by row with
// DWORD-ALIGNED wide loads (x4 + x2 / x3 per row: tools/loadshape_probe.hip measures 41 cycles per wave-row
// against 105 for the same loads at a 2-byte-aligned address and 194 for twelve 16-bit loads), the samples stay
// packed two per register as they lie in memory, and both filter stages run on v_dot2_i32_i16 (two taps per
// instruction, full rate).  With p = 1 when the window starts on the odd half of a dword, output c of a
// row starts at sample p + c of the loaded registers d[]:
//   p + c even:  pairs d[(p+c)/2 + j] with tap pairs (t0,t1)(t2,t3)...                 NTAP/2 products
//   p + c odd:   pairs d[(p+c-1)/2 + j] with the taps moved up by one, (0,t0)(t1,t2)...(t7,0)   NTAP/2+1
// Both cases are written as NTAP/2+1 products on d[c/2 + j] with a tap set chosen by (fraction, p) -- T0 for
// even c, T1 for odd c, one of them padded with a zero pair -- so no lane ever re-aligns samples and the
// lanes of a wave (different PUs, fractions and parities) run the same instructions.  Every cell takes the
// two-stage route (horizontal into the 14-bit intermediate, then vertical), a zero fraction being the filter
// {0,..,64,..,0}: with the reference's offsets and shifts that is bit-identical to its one-stage and copy
// cases (xPredInterLumaBlk :554-601 -- for a first-and-last stage (sum + 32) >> 6 ==
// ((sum >> (6-head)) + (1 << (head-1))) >> head because 8192 << (6-head) is a multiple of the first shift).
template <int NTAP>
__device__ __forceinline__ int tap_pair(int frac, int k) { // the pair (t[k], t[k+1]); taps outside 0..NTAP-1 are 0
  auto tap = [&](int t) { return (t < 0 || t >= NTAP) ? 0 : (NTAP == 8 ? luma_tap(frac, t) : chroma_tap(frac, t)); };
  return (tap(k) & 0xffff) | (tap(k + 1) << 16);
}
// table[frac][p][set][j]: set 0 = T0 (even c), set 1 = T1 (odd c); rows padded to 12 / 8 registers
constexpr int kLumaRow = 12, kChromaRow = 8;
constexpr int kTapTable = 4 * 2 * kLumaRow + 8 * 2 * kChromaRow;
template <int NTAP>
__device__ __forceinline__ int tap_table_entry(int frac, int p, int i) {
  constexpr int NO = NTAP / 2 + 1;
  if (i >= 2 * NO) return 0;
  const int set = i / NO, j = i % NO;
  // p + c even (set == p): even pairs from register 0 when c is even, from register 1 when c is odd
  if (set == p) return set == 0 ? tap_pair<NTAP>(frac, 2 * j) : tap_pair<NTAP>(frac, 2 * j - 2);
  return tap_pair<NTAP>(frac, 2 * j - 1);
}
__device__ __forceinline__ void fill_tap_table(int *lds, int tid, int nthreads) {
  for (int i = tid; i < kTapTable; i += nthreads) {
    if (i < 8 * kLumaRow) lds[i] = tap_table_entry<8>(i / (2 * kLumaRow), (i / kLumaRow) & 1, i % kLumaRow);
    else {
      const int k = i - 8 * kLumaRow;
      lds[i] = tap_table_entry<4>(k / (2 * kChromaRow), (k / kChromaRow) & 1, k % kChromaRow);
    }
  }
  __syncthreads();
}
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(int pair, int taps, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2v, pair), __builtin_bit_cast(s2v, taps), acc, false);
}

// Prediction of one list for a W x H cell whose first sample is `ref` in the reference plane:
// xPredInterLumaBlk / ChromaBlk (:554-642) restricted to the cell.  The reference's two-stage
// filtering is position-wise (every output is the vertical filter of horizontally filtered rows), so
// cutting a PU into cells gives the same samples.  Rows go to emit(r, v[W]) as they are finished.
template <int NTAP, int W, int H, typename Emit>
__device__ __forceinline__ void mc_cell(const int *lds_taps, const short *ref, int rs, int mvx, int mvy, bool bi, int B, Emit emit) {
  constexpr int SH = NTAP == 8 ? 2 : 3, MASK = (1 << SH) - 1, HALF = NTAP / 2, R = H + NTAP - 1;
  constexpr int ND = (W + NTAP + 1) / 2;        // registers per window row: 12 / 6 samples, W + NTAP of them used
  constexpr int NE = NTAP / 2, NO = NTAP / 2 + 1, NP = (R + 1) / 2, ROW = NTAP == 8 ? kLumaRow : kChromaRow;
  typedef __attribute__((address_space(1))) const short gpel; // the table pointer is generic to the compiler: no FLAT loads
  typedef __attribute__((address_space(1))) const int gword;
  const gpel *win = (const gpel *)ref + (mvx >> SH) - (HALF - 1) + (ptrdiff_t)((mvy >> SH) - (HALF - 1)) * rs;
  const int p = (int)(((uintptr_t)win >> 1) & 1); // window starts on the odd half of a dword: start one sample earlier
  win -= p; // (with an odd stride every other row is still 2-byte aligned: the same samples, merely slower loads)
  const int *base = lds_taps + (NTAP == 8 ? 0 : 8 * kLumaRow);
  const int *tx = base + ((mvx & MASK) * 2 + p) * ROW, *ty = base + (mvy & MASK) * 2 * ROW;
  int t0[NO], t1[NO], ey[NE], oy[NO];
#pragma unroll
  for (int j = 0; j < NO; j++) t0[j] = tx[j], t1[j] = tx[NO + j], oy[j] = ty[NO + j];
#pragma unroll
  for (int j = 0; j < NE; j++) ey[j] = ty[j];
  const int head = 14 - B, maxv = (1 << B) - 1;
  // stage 1, first and not last (:206-221): shift 6 - head, offset -(8192 << shift); narrowed to 16 bits
  const int sh1 = 6 - head, off1 = -(8192 << sh1);
  int P[NP][W]; // the intermediate, rows 2k and 2k+1 packed per column
  int lo[W];
  // The rows are fetched four at a time, one chunk ahead of the arithmetic: a wave spends its life waiting for
  // window rows (25 us per wave against 2 us of VALU issue when every row was loaded where it is used), so the
  // loads of the next chunk are in flight while this one is filtered.
  constexpr int CH = 4, NCH = (R + CH - 1) / CH;
  int buf[2][CH][ND];
  auto fetch = [&](int k, int (&dst)[CH][ND]) {
#pragma unroll
    for (int i = 0; i < CH; i++)
      if (k * CH + i < R) __builtin_memcpy(dst[i], (gword *)(win + (ptrdiff_t)(k * CH + i) * rs), ND * 4);
  };
  fetch(0, buf[0]);
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    if (k + 1 < NCH) fetch(k + 1, buf[(k + 1) & 1]);
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const int r = k * CH + i;
      if (r < R) {
        const int *d = buf[k & 1][i];
#pragma unroll
        for (int c = 0; c < W; c++) {
          int s = off1;
#pragma unroll
          for (int j = 0; j < NO; j++) s = dot2(d[c / 2 + j], c % 2 ? t1[j] : t0[j], s);
          s >>= sh1;
          if (r % 2 == 0) lo[c] = s;
          else P[r / 2][c] = (int)__builtin_amdgcn_perm((unsigned)s, (unsigned)lo[c], 0x05040100u);
        }
      }
    }
  }
  if (R % 2)
#pragma unroll
    for (int c = 0; c < W; c++) P[NP - 1][c] = lo[c] & 0xffff;
  // stage 2, not first: last -> shift 6 + head, offset (1 << (shift-1)) + (8192 << 6), clipped; else shift 6
  const bool last = !bi;
  const int sh2 = last ? 6 + head : 6, off2 = last ? (1 << (5 + head)) + (8192 << 6) : 0;
#pragma unroll
  for (int r = 0; r < H; r++) {
    int v[W];
#pragma unroll
    for (int c = 0; c < W; c++) {
      int s = off2;
      if (r % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NE; j++) s = dot2(P[r / 2 + j][c], ey[j], s);
      } else {
#pragma unroll
        for (int j = 0; j < NO; j++) s = dot2(P[r / 2 + j][c], oy[j], s);
      }
      const int w16 = wrap16(s >> sh2);
      v[c] = last ? clip3(0, maxv, w16) : w16;
    }
    emit(r, v);
  }
}

// Prediction of one plane's W x H cell from both lists (+ addAvg) into dst.  Pass one runs for every lane: the
// only list of a uni-predicted PU (final samples, stored) or list 0 of a bi-predicted one (14-bit samples, kept
// packed two per register); pass two runs list 1 for the bi-predicted lanes and stores addAvg rows.
template <int NTAP, int W, int H>
__device__ __forceinline__ void mc_cell_plane(const int *lds_taps, const McArgs &A, const McJob &J, const hmx_pu &u, int pl, int x, int y) {
  typedef __attribute__((address_space(1))) short gpel;
  const bool bi = u.ref0 != 255 && u.ref1 != 255, first1 = u.ref0 == 255;
  gpel *d = (gpel *)J.dst.p[pl] + (size_t)y * J.dst.s[pl] + x;
  const int ds = J.dst.s[pl];
  unsigned keep[H * W / 2];
  {
    const PlanesDev &R = A.refs[J.ref_off + (first1 ? u.ref1 : u.ref0)];
    mc_cell<NTAP, W, H>(lds_taps, R.p[pl] + (ptrdiff_t)y * R.s[pl] + x, R.s[pl], first1 ? u.mv1x : u.mv0x, first1 ? u.mv1y : u.mv0y,
                        bi, A.B, [&](int r, const int *v) {
                          short row[W];
#pragma unroll
                          for (int c = 0; c < W; c++) row[c] = (short)v[c];
#pragma unroll
                          for (int c = 0; c < W; c += 2) keep[(r * W + c) / 2] = (unsigned)(unsigned short)row[c] | ((unsigned)(unsigned short)row[c + 1] << 16);
                          if (!bi) __builtin_memcpy(d + (size_t)r * ds, row, W * 2); // one 8-byte (4-byte) store per row
                        });
  }
  if (bi) {
    const PlanesDev &R = A.refs[J.ref_off + u.ref1];
    mc_cell<NTAP, W, H>(lds_taps, R.p[pl] + (ptrdiff_t)y * R.s[pl] + x, R.s[pl], u.mv1x, u.mv1y, true, A.B, [&](int r, const int *v) {
      short row[W];
#pragma unroll
      for (int c = 0; c < W; c += 2) {
        const unsigned k = keep[(r * W + c) / 2];
        row[c] = (short)add_avg((int)(short)(k & 0xffff), v[c], A.B);
        row[c + 1] = (short)add_avg((int)(short)(k >> 16), v[c + 1], A.B);
      }
      __builtin_memcpy(d + (size_t)r * ds, row, W * 2);
    });
  }
}

// Two ways to hand cells to lanes.  With the picture size known (hmx_mc_job::pic_w/pic_h) a scatter pass
// writes each PU's index into a cell map and the prediction kernel runs one lane per cell of the PICTURE:
// every wave is full whatever the PU sizes.  Without it, one wave per PU, its lanes looping over the
// PU's cells (an 8x4 PU keeps 2 of 64 lanes busy).  A cell reads its (W+7) / (W+3) window rows straight
// from the margin-extended reference planes (the caches absorb the overlap between neighbouring
// cells); nothing is staged, nothing synchronises after the tap table is in LDS.
__global__ __launch_bounds__(256) void k_mc_map(McArgs A) { // 16 threads per PU, one per row of its cells (PUs are <= 64 high)
  const McJob J = A.jobs[blockIdx.y];
  const int pi = blockIdx.x * 16 + (threadIdx.x >> 4), r = threadIdx.x & 15;
  if (pi >= J.n) return;
  const hmx_pu u = J.pus[pi];
  if (u.ref0 == 255 && u.ref1 == 255) return;
  const int cw = u.w >> 2, rows = u.h >> 2;
  for (int rr = r; rr < rows; rr += 16) { // one pass for every legal PU
    const int cy = (u.y >> 2) + rr;
    if (cy >= J.ch) break;
    for (int i = 0; i < cw; i++)
      if ((u.x >> 2) + i < J.cw) J.map[(size_t)cy * J.cw + (u.x >> 2) + i] = pi;
  }
}
__global__ __launch_bounds__(256) void k_mc_cells(McArgs A) {
  __shared__ int taps[kTapTable];
  fill_tap_table(taps, threadIdx.x, 256);
  const McJob J = A.jobs[blockIdx.y];
  // A lane owns two vertically adjacent cells (4 x 8 luma samples); a wave 8 x 8 such pairs = 32 x 64 samples, a
  // workgroup 64 x 128: the window rows of the cells of one PU fall into the same cache lines of the same load
  // instruction, and rows shared by vertical neighbours are fetched by the same wave.  When both cells belong to
  // ONE PU (every PU at least 8 high does that) they are predicted as one 4 x 8 cell -- 15 window rows instead of
  // 2 x 11, one horizontal pass over them; otherwise each cell on its own.
  const int tiles_x = (J.cw + 15) >> 4, tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cx = (tile % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7), cy = (tile / tiles_x) * 32 + (wave >> 1) * 16 + (lane >> 3) * 2;
  if (cx >= J.cw || cy >= J.ch) return;
  const int pi0 = J.map[cy * J.cw + cx], pi1 = cy + 1 < J.ch ? J.map[(cy + 1) * J.cw + cx] : -1;
  const int x = cx << 2, y = cy << 2;
  if (pi0 >= 0 && pi0 == pi1) {
    const hmx_pu u = J.pus[pi0];
    mc_cell_plane<8, 4, 8>(taps, A, J, u, 0, x, y);
    mc_cell_plane<4, 2, 4>(taps, A, J, u, 1, x >> 1, y >> 1);
    mc_cell_plane<4, 2, 4>(taps, A, J, u, 2, x >> 1, y >> 1);
  } else {
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
      const int pi = k ? pi1 : pi0;
      if (pi < 0) continue;
      const hmx_pu u = J.pus[pi];
      mc_cell_plane<8, 4, 4>(taps, A, J, u, 0, x, y + 4 * k);
      mc_cell_plane<4, 2, 2>(taps, A, J, u, 1, x >> 1, (y >> 1) + 2 * k);
      mc_cell_plane<4, 2, 2>(taps, A, J, u, 2, x >> 1, (y >> 1) + 2 * k);
    }
  }
}
__global__ __launch_bounds__(64) void k_mc(McArgs A) {
  __shared__ int taps[kTapTable];
  fill_tap_table(taps, threadIdx.x, 64);
  const McJob J = A.jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.n) return; // jobs of one call may differ in length
  const hmx_pu u = J.pus[blockIdx.x];
  if (u.ref0 == 255 && u.ref1 == 255) return;
  const int cw = u.w >> 2, cells = cw * (u.h >> 2);
  for (int i = threadIdx.x; i < cells; i += 64) {
    const int x = u.x + ((i % cw) << 2), y = u.y + ((i / cw) << 2);
    mc_cell_plane<8, 4, 4>(taps, A, J, u, 0, x, y);
    mc_cell_plane<4, 2, 2>(taps, A, J, u, 1, x >> 1, y >> 1);
    mc_cell_plane<4, 2, 2>(taps, A, J, u, 2, x >> 1, y >> 1);
  }
}

extern "C" int hmx_batch_motionCompensation_multi(hmx_ctx *c, int n_jobs, const hmx_mc_job *jobs) {
  if (!c || !jobs || n_jobs <= 0 || n_jobs > 65535) return fail(c, HMX_ERR_ARG, "hmx_batch_motionCompensation_multi: bad argument");
  std::vector<McJob> hj(n_jobs);
  std::vector<PlanesDev> hr;
  int max_n = 0;
  size_t map_cells = 0, max_cells = 0, max_tiles = 0;
  bool mapped = true;
  for (int i = 0; i < n_jobs; i++) {
    const hmx_mc_job &j = jobs[i];
    if (j.n_pus < 0 || (j.n_pus > 0 && !j.d_pus) || !j.refs || j.n_refs <= 0 || j.n_refs > 16 || !j.dst || j.pic_w < 0 || j.pic_h < 0)
      return fail(c, HMX_ERR_ARG, "hmx_batch_motionCompensation_multi: bad job");
    hj[i] = McJob{j.d_pus, j.n_pus, (int)hr.size(), to_dev(j.dst), nullptr, (j.pic_w + 3) / 4, (j.pic_h + 3) / 4};
    for (int k = 0; k < j.n_refs; k++) hr.push_back(to_dev(&j.refs[k]));
    max_n = std::max(max_n, j.n_pus);
    const size_t cells = (size_t)hj[i].cw * hj[i].ch;
    mapped = mapped && cells > 0;
    map_cells += cells;
    max_cells = std::max(max_cells, cells);
    max_tiles = std::max(max_tiles, (size_t)((hj[i].cw + 15) / 16) * ((hj[i].ch + 31) / 32)); // 64 x 128 luma samples
  }
  if (!max_n) return HMX_OK;
  if (mapped) { // cell maps of all jobs, back to back, in a grow-only scratch buffer
    if (map_cells > c->mcmap_cap) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipFree(c->d_mcmap);
      c->d_mcmap = nullptr;
      c->mcmap_cap = 0;
      if (hipMalloc((void **)&c->d_mcmap, map_cells * sizeof(int)) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc cell map");
      c->mcmap_cap = map_cells;
    }
    HIPCHK(c, hipMemsetAsync(c->d_mcmap, 0xff, map_cells * sizeof(int), c->stream));
    size_t off = 0;
    for (int i = 0; i < n_jobs; i++) {
      hj[i].map = c->d_mcmap + off;
      off += (size_t)hj[i].cw * hj[i].ch;
    }
  }
  // both tables in one copy
  const size_t jb = (sizeof(McJob) * hj.size() + 255) & ~(size_t)255;
  std::vector<char> blob(jb + sizeof(PlanesDev) * hr.size());
  memcpy(blob.data(), hj.data(), sizeof(McJob) * hj.size());
  memcpy(blob.data() + jb, hr.data(), sizeof(PlanesDev) * hr.size());
  char *d = static_cast<char *>(arena_push(c, blob.data(), blob.size()));
  if (!d) return fail(c, HMX_ERR_NOMEM, "argument arena");
  McArgs A;
  A.jobs = reinterpret_cast<const McJob *>(d);
  A.refs = reinterpret_cast<const PlanesDev *>(d + jb);
  A.B = c->cfg.bit_depth;
  if (mapped) {
    hipLaunchKernelGGL(k_mc_map, dim3((unsigned)((max_n + 15) / 16), (unsigned)n_jobs), dim3(256), 0, c->stream, A);
    hipLaunchKernelGGL(k_mc_cells, dim3((unsigned)max_tiles, (unsigned)n_jobs), dim3(256), 0, c->stream, A);
  } else {
    hipLaunchKernelGGL(k_mc, dim3((unsigned)max_n, (unsigned)n_jobs), dim3(64), 0, c->stream, A);
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

extern "C" int hmx_batch_motionCompensation(hmx_ctx *c, const hmx_pu *d_pus, int n, const hmx_pic *refs, int n_refs,
                                            const hmx_pic *dst) {
  if (!c || !d_pus || !refs || !dst || n_refs <= 0 || n_refs > 16) return fail(c, HMX_ERR_ARG, "hmx_batch_motionCompensation: bad argument");
  if (n <= 0) return HMX_OK;
  const hmx_mc_job j{d_pus, n, refs, n_refs, dst, 0, 0}; // picture size unknown here: one wave per PU
  return hmx_batch_motionCompensation_multi(c, 1, &j);
}

// ---- extendPicBorder (TComPicYuv.cpp:248-286): every margin sample is the nearest picture sample, so
// one launch covers all margins of all planes of all pictures (no left/right-then-up/down ordering) ----
__global__ __launch_bounds__(256) void k_border1(const PlanesDev *pics, int pic_w, int pic_h, int mx, int my) {
  const int pl = blockIdx.y % 3;
  const PlanesDev &D = pics[blockIdx.y / 3];
  const int sh = pl ? 1 : 0, w = pic_w >> sh, h = pic_h >> sh, bx = mx >> sh, by = my >> sh;
  const int ww = w + 2 * bx, band = ww * by, side = h * bx;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int x, y;
  if (i < band) { // above
    y = -1 - i / ww, x = i % ww - bx;
  } else if ((i -= band) < band) { // below
    y = h + i / ww, x = i % ww - bx;
  } else if ((i -= band) < side) { // left
    y = i / bx, x = -1 - i % bx;
  } else if ((i -= side) < side) { // right
    y = i / bx, x = w + i % bx;
  } else
    return;
  short *p = D.p[pl];
  const int s = D.s[pl];
  p[(ptrdiff_t)y * s + x] = p[(ptrdiff_t)min(max(y, 0), h - 1) * s + min(max(x, 0), w - 1)];
}
// The same with four samples per thread (one 8-byte store): for plane widths and margins that are multiples of
// four samples a group never straddles the picture edge, so it is either a copy of four picture samples
// (above / below the picture) or one edge sample four times.
__global__ __launch_bounds__(256) void k_border(const PlanesDev *pics, int pic_w, int pic_h, int mx, int my) {
  typedef __attribute__((address_space(1))) short gpel;
  const int pl = blockIdx.y % 3;
  const PlanesDev &D = pics[blockIdx.y / 3];
  const int sh = pl ? 1 : 0, w = pic_w >> sh, h = pic_h >> sh, bx = mx >> sh, by = my >> sh;
  const int gw = (w + 2 * bx) >> 2, gb = bx >> 2, band = gw * by, side = h * gb;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int x, y;
  if (i < band) { // above
    y = -1 - i / gw, x = ((i % gw) << 2) - bx;
  } else if ((i -= band) < band) { // below
    y = h + i / gw, x = ((i % gw) << 2) - bx;
  } else if ((i -= band) < side) { // left
    y = i / gb, x = ((i % gb) << 2) - bx;
  } else if ((i -= side) < side) { // right
    y = i / gb, x = w + ((i % gb) << 2);
  } else
    return;
  gpel *p = (gpel *)wave_uniform(D.p[pl]);
  const int s = wave_uniform(D.s[pl]);
  gpel *src = p + (ptrdiff_t)min(max(y, 0), h - 1) * s;
  short v[4];
  if (x >= 0 && x < w) {
    __builtin_memcpy(v, src + x, 8);
  } else {
    v[0] = v[1] = v[2] = v[3] = src[x < 0 ? 0 : w - 1];
  }
  __builtin_memcpy(p + (ptrdiff_t)y * s + x, v, 8);
}
extern "C" int hmx_pic_extend_border_multi(hmx_ctx *c, int n_pics, const hmx_pic *pics, int pic_w, int pic_h, int mx, int my) {
  if (!c || !pics || n_pics <= 0 || n_pics > 21845 || mx < 0 || my < 0) return fail(c, HMX_ERR_ARG, "hmx_pic_extend_border_multi: bad argument");
  if (!mx && !my) return HMX_OK;
  std::vector<PlanesDev> t(n_pics);
  for (int i = 0; i < n_pics; i++) t[i] = to_dev(&pics[i]);
  const PlanesDev *d = static_cast<const PlanesDev *>(arena_push(c, t.data(), sizeof(PlanesDev) * n_pics));
  if (!d) return fail(c, HMX_ERR_NOMEM, "argument arena");
  const long long total = 2LL * (pic_w + 2 * mx) * my + 2LL * pic_h * mx; // luma margin samples (chroma has fewer)
  if (pic_w % 8 == 0 && mx % 8 == 0) // chroma width and margin are then multiples of four as well
    hipLaunchKernelGGL(k_border, dim3((unsigned)((total / 4 + 255) / 256), (unsigned)n_pics * 3), dim3(256), 0, c->stream, d, pic_w, pic_h, mx, my);
  else
    hipLaunchKernelGGL(k_border1, dim3((unsigned)((total + 255) / 256), (unsigned)n_pics * 3), dim3(256), 0, c->stream, d, pic_w, pic_h, mx, my);
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}
extern "C" int hmx_pic_extend_border(hmx_ctx *c, const hmx_pic *pic, int pic_w, int pic_h, int mx, int my) {
  if (!c || !pic) return fail(c, HMX_ERR_ARG, "hmx_pic_extend_border: bad argument");
  return hmx_pic_extend_border_multi(c, 1, pic, pic_w, pic_h, mx, my);
}


// hmx_rdoq_core.h -- TComTrQuant::xRateDistOptQuant (TLibCommon/TComTrQuant.cpp:1719-2305) decomposed so that the lanes of
// a wave can share one block, with every floating-point operation in the reference's order.
//
// What makes the reference sequential is small: (a) the running context state (c1, c2, Rice parameter, context set) --
// but it is RESET at every coefficient-group boundary, and all that crosses the boundary is one bit (did the previous
// group code a level above 1?); (b) the significance contexts of a group depend on whether its right and lower
// neighbour groups are coded (2 bits); (c) the sums of costs (d64BaseCost, d64BlockUncodedCost) are running doubles, and
// the group-level "zero the whole group?" and the last-position search compare against them.
// So:  1. every coefficient group is walked for all 2 x 4 values of (carry, neighbour pattern) IN PARALLEL (rdoq_walk_cg
//         with a SpecSink: a walk is 16 coefficients with the full per-coefficient decision), keeping per variant the 16
//         cost terms, four group sums, the non-zero mask and the carry it would hand on;
//      2. ONE lane walks the groups in the reference's order (rdoq_resolve): it knows carry and pattern by then, picks the
//         variant, adds its 16 terms to the running cost one by one (the reference's additions, in its order) and takes the
//         group-level decision;
//      3. with (carry, pattern) of every group known the groups are walked once more in parallel (FullSink) for what the
//         rest needs of the chosen variant only: levels, per-coefficient costs, the rate deltas of sign hiding;
//      4. last position (rdoq_phase_b, one lane, usually a few steps), final levels, sign-bit hiding per group in parallel.
// The walk is ONE function for steps 1 and 3, the serial parts are plain restatements: bit-exactness against the oracle is
// checked on the CPU too (tests/test_rdoq_core.py compiles this header with g++ and runs the decomposition lane by lane).
// No HIP dependency: HMX_HD is __host__ __device__ under hipcc and nothing under g++.  Fused multiply-add is off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HMX_HD __host__ __device__ __forceinline__
#else
#define HMX_HD inline
#endif
#pragma clang fp contract(off)

namespace hmx {

struct EstBitsDev { // estBitsSbacStruct (TComTrQuant.h:59-72), 1/32768 bit
  int sig_cg[2][2];
  int sig[42][2];
  int last_x[32], last_y[32];
  int greater1[24][2];
  int greater2[6][2];
  int cbf[15][2];
  int root_cbf[4][2];
  int scan_zigzag[2], scan_nonzigzag[2];
};

HMX_HD int rdoq_base_level(unsigned c1i, unsigned c2i) { return c1i < 8 ? (2 + (c2i < 1)) : 1; }

// The bits of the three contexts a coefficient can touch, fetched together at its start: a walk is a chain of decisions, and
// every table entry fetched where it is used would be a round trip of its own (the tables wait in LDS on the device).
struct RdoqCtxBits {
  int sig[2], g1[2], g2[2]; // significantBits[ctx_sig], greaterOneBits[ctx1], levelAbsBits[ctx2]
};
HMX_HD RdoqCtxBits rdoq_ctx_bits(const EstBitsDev &e, unsigned ctx_sig, unsigned ctx1, unsigned ctx2) {
  RdoqCtxBits b;
  b.sig[0] = e.sig[ctx_sig][0], b.sig[1] = e.sig[ctx_sig][1];
  b.g1[0] = e.greater1[ctx1][0], b.g1[1] = e.greater1[ctx1][1];
  b.g2[0] = e.greater2[ctx2][0], b.g2[1] = e.greater2[ctx2][1];
  return b;
}

// rate of |level| beyond the significance flag as a COST (xGetICRateCost :2508)
HMX_HD double rdoq_level_cost(const RdoqCtxBits &e, double lambda, unsigned lvl, unsigned rice, unsigned c1i, unsigned c2i) {
  double rate = 32768;
  const unsigned base = (unsigned)rdoq_base_level(c1i, c2i);
  if (lvl >= base) {
    unsigned sym = lvl - base, len;
    if (sym < (3u << rice)) {
      len = sym >> rice;
      rate += (double)((len + 1 + rice) << 15);
    } else {
      len = rice;
      sym -= 3u << rice;
      while (sym >= (1u << len)) sym -= 1u << (len++);
      rate += (double)((3 + len + 1 - rice + len) << 15);
    }
    if (c1i < 8) {
      rate += e.g1[1];
      if (c2i < 1) rate += e.g2[1];
    }
  } else if (lvl == 1) {
    rate += e.g1[0];
  } else {
    rate += e.g1[1];
    rate += e.g2[0];
  }
  return lambda * rate;
}
HMX_HD double rdoq_level_cost(const EstBitsDev &e, double lambda, unsigned lvl, unsigned ctx1, unsigned ctx2, unsigned rice, unsigned c1i,
                              unsigned c2i) {
  return rdoq_level_cost(rdoq_ctx_bits(e, 0, ctx1, ctx2), lambda, lvl, rice, c1i, c2i);
}

// integer rate of |level| (xGetICRate :2577): only for the sign-hiding deltas
HMX_HD int rdoq_level_rate(const RdoqCtxBits &e, unsigned lvl, unsigned rice, unsigned c1i, unsigned c2i) {
  int rate = 0;
  const unsigned base = (unsigned)rdoq_base_level(c1i, c2i);
  if (lvl >= base) {
    unsigned sym = lvl - base;
    const unsigned max_vlc = rice == 0 ? 7u : rice == 1 ? 14u : rice == 2 ? 26u : rice == 3 ? 46u : 78u;
    const unsigned pre_max = 8u - rice;
    if (sym > max_vlc) {
      const unsigned a = sym - max_vlc;
      int egs = 1;
      for (unsigned m = 2; a >= m; m <<= 1) egs += 2;
      rate += egs << 15;
      sym = sym < max_vlc + 1 ? sym : max_vlc + 1;
    }
    const unsigned pre = ((sym >> rice) & 0xffffu) + 1;
    const unsigned bins = (pre < pre_max ? pre : pre_max) + rice;
    rate += (int)((bins & 0xffffu) << 15);
    if (c1i < 8) {
      rate += e.g1[1];
      if (c2i < 1) rate += e.g2[1];
    }
  } else if (lvl == 0) {
    return 0;
  } else if (lvl == 1) {
    rate += e.g1[0];
  } else {
    rate += e.g1[1];
    rate += e.g2[0];
  }
  return rate;
}
HMX_HD int rdoq_level_rate(const EstBitsDev &e, unsigned lvl, unsigned ctx1, unsigned ctx2, unsigned rice, unsigned c1i, unsigned c2i) {
  return rdoq_level_rate(rdoq_ctx_bits(e, 0, ctx1, ctx2), lvl, rice, c1i, c2i);
}

// significance context (getSigCtxInc :2349, REMOVAL_8x2_2x8_CG branch); scan_idx 0 = diagonal
HMX_HD int rdoq_sig_ctx(int pattern, int scan_idx, int px, int py, int log2n, bool is_luma) {
  if (px + py == 0) return 0;
  if (log2n == 2) {
    const unsigned long long map4 = 0x8877886654325410ull; // {0,1,4,5, 2,3,4,5, 6,6,8,8, 7,7,8,8}, one nibble each
    return (int)((map4 >> (4 * (4 * py + px))) & 15);
  }
  const int offset = log2n == 3 ? (scan_idx == 0 ? 9 : 15) : (is_luma ? 21 : 12);
  const int sx = px & 3, sy = py & 3;
  int cnt;
  if (pattern == 0)
    cnt = sx + sy <= 2 ? (sx + sy == 0 ? 2 : 1) : 0;
  else if (pattern == 1)
    cnt = sy <= 1 ? (sy == 0 ? 2 : 1) : 0;
  else if (pattern == 2)
    cnt = sx <= 1 ? (sx == 0 ? 2 : 1) : 0;
  else
    cnt = 2;
  return ((is_luma && ((px >> 2) + (py >> 2)) > 0) ? 3 : 0) + offset + cnt;
}

HMX_HD unsigned rdoq_group_idx(unsigned p) { // g_uiGroupIdx
  return p < 4 ? p : p < 6 ? 4 : p < 8 ? 5 : p < 12 ? 6 : p < 16 ? 7 : p < 24 ? 8 : 9;
}
HMX_HD double rdoq_last_cost(const EstBitsDev &e, double lambda, unsigned px, unsigned py) {
  const unsigned cx = rdoq_group_idx(px), cy = rdoq_group_idx(py);
  double cost = e.last_x[cx] + e.last_y[cy];
  if (cx > 3) cost += 32768.0 * ((cx - 2) >> 1);
  if (cy > 3) cost += 32768.0 * ((cy - 2) >> 1);
  return lambda * cost;
}

// ---------------------------------------------------------------------------------------------------------------------
struct RdoqConst { // what a block's walk needs besides the bit estimates
  int lg;        // log2 of the block size
  int scan_idx;  // 0 diagonal, 1 horizontal, 2 vertical
  int is_luma;
  int qbits, q;
  int root_cbf, cbf_ctx, sign_hide;
  double lambda, err_scale;
  long long rd_factor;
};

// one coefficient: |coef| * q clamped as the reference's Int arithmetic leaves it (:1811-1813), and the cost of coding it as zero
HMX_HD void rdoq_prep(int coef, const RdoqConst &C, int &ld, double &cz) {
  const long long wide = (long long)(coef < 0 ? -(long long)coef : (long long)coef) * C.q, cap = 2147483647ll - (1ll << (C.qbits - 1));
  ld = (int)(wide < cap ? wide : cap);
  const double e0 = (double)ld;
  cz = e0 * e0 * C.err_scale;
}
HMX_HD unsigned rdoq_max_level(int ld, int qbits) { return (unsigned)((ld + (1 << (qbits - 1))) >> qbits); }

struct RdoqCgSums { // what the group-level decision needs of a walked group
  double s_sig, s_sig0, s_coded, s_uncoded;
  unsigned short nz_mask;    // bit k: scan entry k of the group got a non-zero level
  unsigned char nnz_before0; // non-zero levels at k = 1..15
  unsigned char carry_out;   // c1 == 0 at the end of the group: the next group's context set moves up by one
};

// Walk scan entries 15..0 of coefficient group cg (TComTrQuant.cpp:1786-1990 for one group).  bp(sp): block position (raster)
// of scan position sp; pattern: 0..3 (right | lower << 1), ignored for 4x4 blocks; carry_in: the previous group's carry
// (ignored in the group of the last position, whose state starts there).
// in(k, bp): the coefficient of entry k of the group (block position bp).  Sink: add(k, term) the term the running cost receives
// for entry k; pos(k, bp, level, cc, cs, rate_up, rate_down, sig_delta, delta_u, cz) everything else about it.
// Per entry the loads come first, all together -- the next entry's coefficient and the bits of this entry's three contexts --
// then nothing but arithmetic.
template <typename BpFn, typename InFn, typename Sink>
HMX_HD RdoqCgSums rdoq_walk_cg_in(const RdoqConst &C, const EstBitsDev &E, int cg, BpFn bp_of, InFn in, int pattern, int carry_in, int last_pos,
                                  Sink &sink) {
  const int lg = C.lg, N = 1 << lg, qbits = C.qbits;
  const bool is_luma = C.is_luma != 0;
  const double lambda = C.lambda, err_scale = C.err_scale;
  if (lg == 2) pattern = -1;
  unsigned ctx_set = ((cg == 0 || !is_luma) ? 0u : 2u) + (carry_in ? 1u : 0u), rice = 0, c1i = 0, c2i = 0;
  int c1 = 1, c2 = 0;
  RdoqCgSums S;
  S.s_sig = 0, S.s_sig0 = 0, S.s_coded = 0, S.s_uncoded = 0;
  S.nz_mask = 0, S.nnz_before0 = 0;
  unsigned bp = bp_of(cg * 16 + 15);
  int coef = in(15, bp);
  for (int k = 15; k >= 0; k--) {
    const int sp = cg * 16 + k;
    const unsigned bp_k = bp;
    const int coef_k = coef;
    if (k > 0) { // the next entry, under way while this one is decided
      bp = bp_of(sp - 1);
      coef = in(k - 1, bp);
    }
    if (sp == last_pos) ctx_set = (sp < 16 || !is_luma) ? 0u : 2u; // the state starts here: c1 = 1, c2 = 0, no carry
    const bool is_last = sp == last_pos;
    const unsigned ctx1 = 4 * ctx_set + (unsigned)c1, ctx2 = ctx_set + (unsigned)c2;
    unsigned ctx_sig = 0;
    if (!is_last) ctx_sig = (unsigned)rdoq_sig_ctx(pattern, C.scan_idx, (int)(bp_k & (unsigned)(N - 1)), (int)(bp_k >> lg), lg, is_luma);
    const RdoqCtxBits B = rdoq_ctx_bits(E, ctx_sig, ctx1, ctx2);
    int ld;
    double cz;
    rdoq_prep(coef_k, C, ld, cz);
    const unsigned max_lvl = rdoq_max_level(ld, qbits);
    int out_level = (int)max_lvl;
    double cc = 0, cs = 0;
    int r_up = 0, r_down = 0, sdel = 0, du = 0;
    if (sp > last_pos) { // above the last position: the coefficient is zero and costs its distortion
      sink.add(k, cz);
    } else {
      unsigned best = 0;
      double sig1 = 0;
      bool decided = false;
      if (!is_last && max_lvl < 3) {
        cs = lambda * B.sig[0];
        cc = cz + cs;
        if (max_lvl == 0) decided = true;
      } else {
        cc = 1.7e+308;
      }
      if (!decided) {
        if (!is_last) sig1 = lambda * B.sig[1];
        const unsigned lo = max_lvl > 1 ? max_lvl - 1 : 1;
        for (int l = (int)max_lvl; l >= (int)lo; l--) {
          const double d = (double)(ld - (l << qbits));
          double cst = d * d * err_scale + rdoq_level_cost(B, lambda, (unsigned)l, rice, c1i, c2i);
          cst += sig1;
          if (cst < cc) {
            best = (unsigned)l;
            cc = cst;
            cs = sig1;
          }
        }
      }
      if (!is_last) sdel = B.sig[1] - B.sig[0];
      du = (ld - ((int)best << qbits)) >> (qbits - 8);
      if (best > 0) {
        const int now = rdoq_level_rate(B, best, rice, c1i, c2i);
        r_up = rdoq_level_rate(B, best + 1, rice, c1i, c2i) - now;
        r_down = rdoq_level_rate(B, best - 1, rice, c1i, c2i) - now;
      } else {
        r_up = B.g1[0];
      }
      out_level = (int)best;
      sink.add(k, cc);
      if (best >= (unsigned)rdoq_base_level(c1i, c2i) && best > 3u * (1u << rice)) rice = rice + 1 < 4u ? rice + 1 : 4u;
      if (best >= 1) c1i++;
      if (best > 1) {
        c1 = 0;
        c2 += (c2 < 2);
        c2i++;
      } else if (c1 < 3 && c1 > 0 && best) {
        c1++;
      }
    }
    sink.pos(k, bp_k, out_level, cc, cs, r_up, r_down, sdel, du, cz);
    S.s_sig += cs;
    if (k == 0) S.s_sig0 = cs;
    if (out_level) {
      S.nz_mask = (unsigned short)(S.nz_mask | (1u << k));
      S.s_coded += cc - cs;
      S.s_uncoded += cz;
      if (k != 0) S.nnz_before0++;
    }
  }
  S.carry_out = (unsigned char)(c1 == 0);
  return S;
}
struct RdoqBlockIn { // the coefficients from a block in memory: src[row * stride + col]
  const int *src;
  int stride, lg;
  HMX_HD int operator()(int, unsigned bp) const { return src[(bp >> lg) * (unsigned)stride + (bp & ((1u << lg) - 1u))]; }
};
template <typename BpFn, typename Sink>
HMX_HD RdoqCgSums rdoq_walk_cg(const RdoqConst &C, const EstBitsDev &E, int cg, BpFn bp_of, const int *src, int stride, int pattern, int carry_in,
                               int last_pos, Sink &sink) {
  return rdoq_walk_cg_in(C, E, cg, bp_of, RdoqBlockIn{src, stride, C.lg}, pattern, carry_in, last_pos, sink);
}

struct RdoqSpec { // one (group, carry, pattern) variant of step 1
  double add[16];
  RdoqCgSums S;
};
struct RdoqSpecSink {
  RdoqSpec *o;
  HMX_HD void add(int k, double v) { o->add[k] = v; }
  HMX_HD void pos(int, unsigned, int, double, double, int, int, int, int, double) {}
};
struct RdoqFullSink { // step 3: everything about the chosen variant, scan order
  int *lev;
  double *cost_coded, *cost_sig;
  int *rate_up, *rate_down, *sig_delta, *delta_u;
  int base; // cg * 16
  HMX_HD void add(int, double) {}
  HMX_HD void pos(int k, unsigned, int level, double cc, double cs, int ru, int rd, int sd, int du, double) {
    const int sp = base + k;
    lev[sp] = level, cost_coded[sp] = cc, cost_sig[sp] = cs;
    rate_up[sp] = ru, rate_down[sp] = rd, sig_delta[sp] = sd, delta_u[sp] = du;
  }
};

struct RdoqRun { // the running quantities of the serial pass
  double base, uncoded;
  unsigned long long cg_flag; // by group position gy * G + gx
  unsigned long long zeroed;  // by group index in scan order: the group-level decision zeroed it
};

// Step 2, ONE lane: groups n_cg-1 .. 0 in the reference's order, in pieces so that the caller decides where the 8 variants
// of a group and the costs of zero come from.
HMX_HD void rdoq_run_init(RdoqRun &R) { R.base = 0, R.uncoded = 0, R.cg_flag = 0, R.zeroed = 0; }
// a coefficient above the last position's group: it stays zero and costs its distortion
HMX_HD void rdoq_resolve_above(RdoqRun &R, double cz) {
  R.uncoded += cz;
  R.base += cz;
}
// Which variants [carry * 4 + right + 2 * lower] can group cg still take when the groups above round_top are resolved (their
// flags in R.cg_flag, `carry` the carry of group round_top + 1) and the groups round_top .. cg + 1 are walked in the same round?
// What the same round leaves open, the candidates often close: a group none of whose coefficients can get a level
// (gmax == 0) stays uncoded, and a group none of whose coefficients can get a level above 1 (gmax <= 1) hands on no carry
// (c1 only drops to 0 behind a level above 1).  gmax_of(cg) -> min(highest candidate level of the group, 2);
// cg_at(gy, gx) -> scan index of the group at that position.  Returns a bit per variant; never empty.
template <typename GmaxFn, typename CgAtFn>
HMX_HD unsigned rdoq_variant_mask(const RdoqConst &C, int cg, int last_cg, int round_top, unsigned g, const RdoqRun &R, int carry, GmaxFn gmax_of,
                                  CgAtFn cg_at) {
  const int G = (1 << C.lg) >> 2;
  const unsigned gx = g & 255u, gy = g >> 8;
  unsigned cset, rset = 1u, lset = 1u; // bit 0: the value 0 is possible, bit 1: the value 1
  if (cg == last_cg) cset = 1u;        // the state starts inside this group: no carry
  else if (cg == round_top) cset = 1u << carry;
  else cset = gmax_of(cg + 1) <= 1 ? 1u : 3u;
  if (gx < (unsigned)G - 1) {
    const int n = cg_at((int)gy, (int)gx + 1);
    rset = n > round_top ? 1u << ((R.cg_flag >> (gy * G + gx + 1)) & 1) : (gmax_of(n) == 0 ? 1u : 3u);
  }
  if (gy < (unsigned)G - 1) {
    const int n = cg_at((int)gy + 1, (int)gx);
    lset = n > round_top ? 1u << ((R.cg_flag >> ((gy + 1) * G + gx)) & 1) : (gmax_of(n) == 0 ? 1u : 3u);
  }
  unsigned mask = 0;
  for (int v = 0; v < 8; v++)
    if (((cset >> (v >> 2)) & 1) && ((rset >> (v & 1)) & 1) && ((lset >> ((v >> 1) & 1)) & 1)) mask |= 1u << v;
  return mask;
}
// group cg <= last_cg.  gpos = gx | gy << 8; spec_of(v) = the group's variant v = carry * 4 + pattern; cz_of(k) the cost of zero of
// entry k; carry: the previous group's (in/out).  Returns the variant taken; cg_sig = cost_cg_sig[cg] as the reference keeps it.
template <typename SpecFn, typename CzFn>
HMX_HD int rdoq_resolve_group_fn(const RdoqConst &C, const EstBitsDev &E, int cg, int last_cg, unsigned g, SpecFn spec_of, CzFn cz_of, RdoqRun &R,
                                 int &carry, double &cg_sig) {
  const int N = 1 << C.lg, G = N >> 2;
  const double lambda = C.lambda;
  cg_sig = 0;
  const unsigned gx = g & 255u, gy = g >> 8, gpos = gy * (unsigned)G + gx;
  const unsigned right = gx < (unsigned)G - 1 ? (unsigned)((R.cg_flag >> (gy * G + gx + 1)) & 1) : 0u;
  const unsigned lower = gy < (unsigned)G - 1 ? (unsigned)((R.cg_flag >> ((gy + 1) * G + gx)) & 1) : 0u;
  const int v = (cg == last_cg ? 0 : carry * 4) + (int)(right + (lower << 1));
  const RdoqSpec &S = spec_of(v);
#pragma unroll
  for (int k = 15; k >= 0; k--) { // (unrolled: the 32 loads go out together, the two chains of additions follow)
    R.uncoded += cz_of(k);
    R.base += S.add[k];
  }
  double s_sig = S.S.s_sig;
  if (S.S.nz_mask) R.cg_flag |= 1ull << gpos;
  // the group-level decision (:1992-2089); last_cg >= 0 here
  if (cg) {
    const unsigned cctx = (right || lower) ? 1u : 0u; // the neighbours' flags have not changed since the pattern was formed
    if (!((R.cg_flag >> gpos) & 1)) {
      R.base += lambda * E.sig_cg[cctx][0] - s_sig;
      cg_sig = lambda * E.sig_cg[cctx][0];
    } else if (cg < last_cg) {
      if (S.S.nnz_before0 == 0) {
        R.base -= S.S.s_sig0;
        s_sig -= S.S.s_sig0;
      }
      double zero_cost = R.base;
      R.base += lambda * E.sig_cg[cctx][1];
      zero_cost += lambda * E.sig_cg[cctx][0];
      cg_sig = lambda * E.sig_cg[cctx][1];
      zero_cost += S.S.s_uncoded;
      zero_cost -= S.S.s_coded;
      zero_cost -= s_sig;
      if (zero_cost < R.base) {
        R.cg_flag &= ~(1ull << gpos);
        R.base = zero_cost;
        cg_sig = lambda * E.sig_cg[cctx][0];
        R.zeroed |= 1ull << cg;
      }
    }
  } else {
    R.cg_flag |= 1ull << gpos;
  }
  carry = S.S.carry_out;
  return v;
}
template <typename CzFn>
HMX_HD int rdoq_resolve_group(const RdoqConst &C, const EstBitsDev &E, int cg, int last_cg, unsigned g, const RdoqSpec *spec8, CzFn cz_of, RdoqRun &R,
                              int &carry, double &cg_sig) {
  return rdoq_resolve_group_fn(C, E, cg, last_cg, g, [spec8](int v) -> const RdoqSpec & { return spec8[v]; }, cz_of, R, carry, cg_sig);
}
// the whole pass over arrays: spec[cg * 8 + carry * 4 + pattern], cz in scan order.  gpos_of(cg) -> gx | gy << 8.
// sel[cg] = the variant taken (0xff above the last position's group).
template <typename GposFn>
HMX_HD void rdoq_resolve(const RdoqConst &C, const EstBitsDev &E, int n_cg, int last_cg, GposFn gpos_of, const double *cz, const RdoqSpec *spec,
                         unsigned char *sel, double *cost_cg_sig, RdoqRun &R) {
  rdoq_run_init(R);
  int carry = 0;
  for (int cg = n_cg - 1; cg >= 0; cg--) {
    if (cg > last_cg) {
      cost_cg_sig[cg] = 0;
      for (int k = 15; k >= 0; k--) rdoq_resolve_above(R, cz[cg * 16 + k]);
      sel[cg] = 0xff;
      continue;
    }
    const double *cz16 = cz + cg * 16;
    sel[cg] = (unsigned char)rdoq_resolve_group(C, E, cg, last_cg, gpos_of(cg), spec + cg * 8, [cz16](int k) { return cz16[k]; }, R, carry, cost_cg_sig[cg]);
  }
}

// after step 3: a group the group-level decision zeroed keeps no level, and its coefficients cost their distortion (:2076-2086)
HMX_HD void rdoq_apply_zeroed_cg(int cg, const double *cz, int *lev, double *cost_coded, double *cost_sig) {
  for (int k = 15; k >= 0; k--) {
    const int sp = cg * 16 + k;
    if (lev[sp]) {
      lev[sp] = 0;
      cost_coded[sp] = cz[sp];
      cost_sig[sp] = 0;
    }
  }
}

// Step 4a, one lane: the last significant position (:2132-2200), in pieces: the running state, the entry of a group, one
// position of a coded group (k = 15..0, only sp <= last_pos).
struct RdoqLast {
  double base, best_cost;
  int best_last_p1;
  int found;
};
HMX_HD void rdoq_last_init(const RdoqConst &C, const EstBitsDev &E, const RdoqRun &R, RdoqLast &T) {
  const double lambda = C.lambda;
  T.base = R.base;
  if (C.root_cbf) {
    T.best_cost = R.uncoded + lambda * E.root_cbf[0][0];
    T.base += lambda * E.root_cbf[0][1];
  } else {
    T.best_cost = R.uncoded + lambda * E.cbf[C.cbf_ctx][0];
    T.base += lambda * E.cbf[C.cbf_ctx][1];
  }
  T.best_last_p1 = 0, T.found = 0;
}
HMX_HD void rdoq_last_group(RdoqLast &T, double cost_cg_sig) { T.base -= cost_cg_sig; } // every group visited before the search ends
HMX_HD void rdoq_last_pos(const RdoqConst &C, const EstBitsDev &E, RdoqLast &T, int sp, unsigned bp, int lv, double cost_coded, double cost_sig,
                          double cz) {
  if (lv) {
    const unsigned py = bp >> C.lg, px = bp & (unsigned)((1 << C.lg) - 1);
    const double lc = C.scan_idx == 2 ? rdoq_last_cost(E, C.lambda, py, px) : rdoq_last_cost(E, C.lambda, px, py);
    const double total = T.base + lc - cost_sig;
    if (total < T.best_cost) {
      T.best_last_p1 = sp + 1;
      T.best_cost = total;
    }
    if (lv > 1) {
      T.found = 1;
      return;
    }
    T.base -= cost_coded;
    T.base += cz;
  } else {
    T.base -= cost_sig;
  }
}
// the search as a sink of the walk: the chosen variant of a coded group walked again, nothing stored
struct RdoqLastSink {
  const RdoqConst &C;
  const EstBitsDev &E;
  RdoqLast &T;
  int base_sp, last_pos;
  HMX_HD void add(int, double) {}
  HMX_HD void pos(int k, unsigned bp, int level, double cc, double cs, int, int, int, int, double cz) {
    const int sp = base_sp + k;
    if (T.found || sp > last_pos) return;
    rdoq_last_pos(C, E, T, sp, bp, level, cc, cs, cz);
  }
};
// the search over arrays.  Returns best_last_p1; R.base / R.uncoded as rdoq_resolve left them.
template <typename GposFn, typename BpFn>
HMX_HD int rdoq_phase_b(const RdoqConst &C, const EstBitsDev &E, int last_pos, int last_cg, GposFn gpos_of, BpFn bp_of, const RdoqRun &Rin,
                        const double *cz, const int *lev, const double *cost_coded, const double *cost_sig, const double *cost_cg_sig) {
  const int G = (1 << C.lg) >> 2;
  RdoqLast T;
  rdoq_last_init(C, E, Rin, T);
  for (int cg = last_cg; cg >= 0 && !T.found; cg--) {
    const unsigned g = gpos_of(cg), gpos = (g >> 8) * (unsigned)G + (g & 255u);
    rdoq_last_group(T, cost_cg_sig[cg]);
    if (!((Rin.cg_flag >> gpos) & 1)) continue;
    for (int k = 15; k >= 0 && !T.found; k--) {
      const int sp = cg * 16 + k;
      if (sp > last_pos) continue;
      rdoq_last_pos(C, E, T, sp, bp_of(sp), lev[sp], cost_coded[sp], cost_sig[sp], cz[sp]);
    }
  }
  return T.best_last_p1;
}

// Step 4c, one group of the FINAL signed levels (scan order): rate-aware sign-bit hiding (:2203-2304), in pieces: what the
// levels alone decide, the cost of changing one position (n from the start position down to 0), the change.
struct RdoqHide {
  long long min_cost;
  int min_pos, final_change;
  int first, lastnz, start; // start: the first position the search looks at (lastnz in the block's top group, else 15)
  unsigned signbit;
  int top;                  // first_nz_group
};
// lev_of(n): final signed level of entry n.  false: nothing to hide in this group.
template <typename LevFn>
HMX_HD bool rdoq_hide_begin(bool first_nz_group, LevFn lev_of, RdoqHide &H) {
  // one pass over the 16 levels (their loads go out together): which are non-zero, which negative, their sum -- the levels
  // outside [first, lastnz] are zero, so the sum over that range is the sum over the group
  unsigned nzm = 0, ngm = 0;
  int sum = 0;
#pragma unroll
  for (int n = 0; n < 16; n++) {
    const int l = lev_of(n);
    nzm |= (l != 0 ? 1u : 0u) << n;
    ngm |= (l < 0 ? 1u : 0u) << n;
    sum += l;
  }
  if (!nzm) return false;
  const int first = __builtin_ctz(nzm), lastnz = 31 - __builtin_clz(nzm);
  if (lastnz - first < 4) return false;
  H.signbit = (ngm >> first) & 1u;
  if (H.signbit == (unsigned)(sum & 1)) return false;
  H.min_cost = 0x7fffffffffffffffll, H.min_pos = -1, H.final_change = 0;
  H.first = first, H.lastnz = lastnz, H.top = first_nz_group, H.start = first_nz_group ? lastnz : 15;
  return true;
}
// entry n <= H.start, in the order start..0.  neg_n: the unquantised coefficient is negative.
HMX_HD void rdoq_hide_pos(const RdoqConst &C, RdoqHide &H, int n, int lv, unsigned neg_n, int ru, int rate_down, int sig_delta, int du) {
  const long long rd_factor = C.rd_factor, kMax = 0x7fffffffffffffffll;
  long long cur;
  int change;
  const int alv = lv < 0 ? -lv : lv;
  if (lv != 0) {
    const long long up = rd_factor * (-du) + ru;
    long long down = rd_factor * (du) + rate_down - (alv == 1 ? ((1 << 15) + sig_delta) : 0);
    if (H.top && H.lastnz == n && alv == 1) down -= (4 << 15);
    if (up < down) {
      cur = up;
      change = 1;
    } else {
      change = -1;
      cur = (n == H.first && alv == 1) ? kMax : down;
    }
  } else {
    const int adu = du < 0 ? -du : du;
    cur = rd_factor * (-(long long)adu) + (1 << 15) + ru + sig_delta;
    change = 1;
    if (n < H.first && neg_n != H.signbit) cur = kMax;
  }
  if (cur < H.min_cost) {
    H.min_cost = cur;
    H.final_change = change;
    H.min_pos = n;
  }
}
// the change to apply at H.min_pos (0: none): added to the signed level
HMX_HD int rdoq_hide_change(const RdoqHide &H, unsigned neg_min_pos) { return H.min_pos < 0 ? 0 : (neg_min_pos ? -H.final_change : H.final_change); }
// the search as a sink of the walk; lev_of(n) / neg16 as above
template <typename LevFn>
struct RdoqHideSink {
  const RdoqConst &C;
  RdoqHide &H;
  LevFn lev_of;
  unsigned neg16;
  HMX_HD void add(int, double) {}
  HMX_HD void pos(int k, unsigned, int, double, double, int ru, int rd, int sd, int du, double) {
    if (k <= H.start) rdoq_hide_pos(C, H, k, lev_of(k), (neg16 >> k) & 1u, ru, rd, sd, du);
  }
};
// over arrays: lev16 = the group's 16 entries.  first_nz_group: no group above this one holds a level.  neg16 bit k: the
// unquantised coefficient is negative.
HMX_HD void rdoq_phase_c_cg(const RdoqConst &C, bool first_nz_group, int *lev16, unsigned neg16, const int *rate_up, const int *rate_down,
                            const int *sig_delta, const int *delta_u) {
  RdoqHide H;
  if (!rdoq_hide_begin(first_nz_group, [lev16](int n) { return lev16[n]; }, H)) return;
  for (int n = H.start; n >= 0; n--) rdoq_hide_pos(C, H, n, lev16[n], (neg16 >> n) & 1u, rate_up[n], rate_down[n], sig_delta[n], delta_u[n]);
  if (H.min_pos >= 0) lev16[H.min_pos] += rdoq_hide_change(H, (neg16 >> H.min_pos) & 1u);
}

} // namespace hmx

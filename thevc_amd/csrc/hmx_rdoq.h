// hmx_rdoq.h - rate-distortion optimised quantisation on the device.
//
// TComTrQuant::xRateDistOptQuant (TLibCommon/TComTrQuant.cpp:1719-2305) with xGetCodedLevel :2446,
// xGetICRateCost :2508, xGetICRate :2577, xGetRateLast :2652, getSigCtxInc :2349, calcPatternSigCtx :2315,
// getSigCoeffGroupCtxInc :2707, flat error scale setErrScaleCoeff :2794, as the reference is compiled
// (REMOVE_NSQT, REMOVAL_8x2_2x8_CG, REMOVE_NUM_GREATER1, COEF_REMAIN_BIN_REDUCTION 3, C1FLAG_NUMBER 8,
// C2FLAG_NUMBER 1).  Same three phases as the oracle's restatement:
//   A. reverse scan: per coefficient the cheaper of {quantised level, one below, (zero)} in D + lambda*R under
//      the running c1/c2/Rice context state; per coefficient group the decision to zero the whole group;
//   B. the last significant position;
//   C. sign-bit hiding with rate-aware costs.
// Two forms.  k_rdoq: ONE LANE PER BLOCK, the reference's sequential walk with its per-position records in a global
// workspace interleaved by lane (4x4 blocks of the batch entry points, and the cross-check of the other form).
// rdoq_wave_tiles: the lanes of a wave SHARE the blocks it holds in LDS (hmx_rdoq_core.h's decomposition), nothing per
// coefficient stored, nothing outside the LDS -- the quantiser of the whole-picture chain and of 8x8 and larger blocks of
// the batch entry points.
//
// Costs are IEEE doubles evaluated in the reference's order.  Fused multiply-add would round differently:
// contraction is switched off for this file, and the two quotients (error scale, sign-hiding factor) are
// formed on the host.
#pragma once
#include "hmx_device.h"
#include "hmx_kernels.h"
#include "hmx_rdoq_core.h"

#pragma clang fp contract(off)

namespace hmx {

struct RdoqBlock { // one block of a launch
  const int *src;  // coefficients (Int), row stride src_stride
  int *dst;        // levels out
  int src_stride, dst_stride;
  uint32_t *abs_sum; // may be NULL
  unsigned char log2n, is_luma, scan_idx /* 0 diag, 1 hor, 2 ver */, root_cbf, cbf_ctx, plane_type /* 0 luma, 1 chroma QP */;
  unsigned short est_idx;
};

struct RdoqArgs {
  const RdoqBlock *blocks;
  int n;
  const EstBitsDev *est; // tables, RdoqBlock::est_idx selects
  // workspace of k_rdoq (one lane per block), n_threads = T lanes: doubles [3 * 1024 + 64] * T, ints [4 * 1024] * T
  double *wd;
  int *wi;
  int T;
  int bit_depth, sign_hide;
  int per[2], rem[2], q[2];
  double lambda[2];
  double err_scale[2][4]; // [plane type][log2n - 2]: 2^15 * 2^(-2 tshift) / q / q / 2^(2 inc)
  long long rd_factor[2]; // (Int64)(invq * invq * 2^(2 per) / lambda / 16 / 2^(2 inc) + 0.5)
  // scaling list (hmx_xRateDistOptQuant_scaled; k_rdoq only): getQuantCoeff / getErrScaleCoeff of the launch's ONE block per position
  // (row-major N x N), NULL = the flat values above
  const int *qtab;
  const double *estab;
};

__device__ __forceinline__ unsigned rdoq_scan_pos(int log2n, int scan_idx, int sp) {
  return log2n == 2 ? kScan4.t[scan_idx][sp] : log2n == 3 ? kScan8.t[scan_idx][sp] : log2n == 4 ? kScan16.t[scan_idx][sp] : kScan32.t[scan_idx][sp];
}

#ifdef HMX_RDOQ_KERNELS // non-template kernels: emitted by the one translation unit that launches them (hmx_scalar.hip)
__global__ __launch_bounds__(64) void k_rdoq(RdoqArgs A) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= A.n) return;
  const RdoqBlock K = A.blocks[tid];
  const EstBitsDev &E = A.est[K.est_idx];
  const size_t T = (size_t)A.T;
  // workspace records of this lane
  double *cost_coded = A.wd + tid, *cost_sig = A.wd + 1024 * T + tid, *cost_zero = A.wd + 2048 * T + tid, *cost_cg_sig = A.wd + 3072 * T + tid;
  int *rate_up = A.wi + tid, *rate_down = A.wi + 1024 * T + tid, *sig_delta = A.wi + 2048 * T + tid, *delta_u = A.wi + 3072 * T + tid;
#define WS(a, i) a[(size_t)(i) * T]
  const int lg = K.log2n, N = 1 << lg, nn = N * N, G = N >> 2, n_cg = nn >> 4;
  const int pt = K.plane_type, B = A.bit_depth;
  const int tshift = 15 - B - lg, qbits = 14 + A.per[pt] + tshift;
  const int q_flat = A.q[pt], scan_idx = K.scan_idx;
  const bool is_luma = K.is_luma;
  const double lambda = A.lambda[pt], err_flat = A.err_scale[pt][lg - 2];
  const int *src = K.src;
  int *dst = K.dst;
  const int ss = K.src_stride, ds = K.dst_stride;
#define SRC(bp) src[((bp) >> lg) * ss + ((bp) & (N - 1))]
#define DST(bp) dst[((bp) >> lg) * ds + ((bp) & (N - 1))]
  for (int i = 0; i < nn; i++) {
    WS(cost_coded, i) = 0;
    WS(cost_sig, i) = 0;
    WS(rate_up, i) = 0;
    WS(rate_down, i) = 0;
    WS(sig_delta, i) = 0;
    WS(delta_u, i) = 0;
    DST(i) = 0;
  }
  for (int i = 0; i < 64; i++) WS(cost_cg_sig, i) = 0;
  unsigned long long cg_flag = 0; // bit gpos
  double uncoded = 0, base = 0;
  int last_pos = -1, last_cg = -1;
  unsigned ctx_set = 0, rice = 0, c1i = 0, c2i = 0;
  int c1 = 1, c2 = 0;
  uint32_t abs_sum = 0;

  // ---- phase A ----
  for (int cg = n_cg - 1; cg >= 0; cg--) {
    const unsigned p0 = rdoq_scan_pos(lg, scan_idx, cg * 16), gx = (p0 & (N - 1)) >> 2, gy = (p0 >> lg) >> 2, gpos = gy * G + gx;
    const unsigned right = gx < (unsigned)G - 1 ? (unsigned)((cg_flag >> (gy * G + gx + 1)) & 1) : 0u;
    const unsigned lower = gy < (unsigned)G - 1 ? (unsigned)((cg_flag >> ((gy + 1) * G + gx)) & 1) : 0u;
    const int pattern = N == 4 ? -1 : (int)(right + (lower << 1));
    double s_sig = 0, s_sig0 = 0, s_coded = 0, s_uncoded = 0;
    int nnz_before0 = 0;
    for (int k = 15; k >= 0; k--) {
      const int sp = cg * 16 + k;
      const unsigned bp = rdoq_scan_pos(lg, scan_idx, sp);
      const int q = A.qtab ? A.qtab[bp] : q_flat;                 // TComTrQuant.cpp:1882-1883: per position under a scaling list
      const double err_scale = A.estab ? A.estab[bp] : err_flat;
      const long long wide = (long long)abs(SRC(bp)) * q, cap = 2147483647ll - (1ll << (qbits - 1));
      const int ld = (int)(wide < cap ? wide : cap);
      const unsigned max_lvl = (unsigned)((ld + (1 << (qbits - 1))) >> qbits);
      const double e0 = (double)ld;
      const double cz = e0 * e0 * err_scale;
      WS(cost_zero, sp) = cz;
      uncoded += cz;
      int out_level = (int)max_lvl;
      double cc = 0, cs = 0; // cost_coded / cost_sig of this position
      if (max_lvl > 0 && last_pos < 0) {
        last_pos = sp;
        ctx_set = (sp < 16 || !is_luma) ? 0 : 2;
        last_cg = cg;
      }
      if (last_pos >= 0) {
        const unsigned ctx1 = 4 * ctx_set + (unsigned)c1, ctx2 = ctx_set + (unsigned)c2;
        const bool is_last = sp == last_pos;
        unsigned ctx_sig = 0;
        if (!is_last) ctx_sig = (unsigned)rdoq_sig_ctx(pattern, scan_idx, (int)(bp & (N - 1)), (int)(bp >> lg), lg, is_luma);
        unsigned best = 0;
        double sig1 = 0;
        bool decided = false;
        if (!is_last && max_lvl < 3) {
          cs = lambda * E.sig[ctx_sig][0];
          cc = cz + cs;
          if (max_lvl == 0) decided = true;
        } else {
          cc = 1.7e+308;
        }
        if (!decided) {
          if (!is_last) sig1 = lambda * E.sig[ctx_sig][1];
          const unsigned lo = max_lvl > 1 ? max_lvl - 1 : 1;
          for (int l = (int)max_lvl; l >= (int)lo; l--) {
            const double d = (double)(ld - (l << qbits));
            double cst = d * d * err_scale + rdoq_level_cost(E, lambda, (unsigned)l, ctx1, ctx2, rice, c1i, c2i);
            cst += sig1;
            if (cst < cc) {
              best = (unsigned)l;
              cc = cst;
              cs = sig1;
            }
          }
        }
        if (!is_last) WS(sig_delta, bp) = E.sig[ctx_sig][1] - E.sig[ctx_sig][0];
        WS(delta_u, bp) = (ld - ((int)best << qbits)) >> (qbits - 8);
        if (best > 0) {
          const int now = rdoq_level_rate(E, best, ctx1, ctx2, rice, c1i, c2i);
          WS(rate_up, bp) = rdoq_level_rate(E, best + 1, ctx1, ctx2, rice, c1i, c2i) - now;
          WS(rate_down, bp) = rdoq_level_rate(E, best - 1, ctx1, ctx2, rice, c1i, c2i) - now;
        } else {
          WS(rate_up, bp) = E.greater1[ctx1][0];
        }
        out_level = (int)best;
        base += cc;
        if (best >= (unsigned)rdoq_base_level(c1i, c2i) && best > 3u * (1u << rice)) rice = min(rice + 1, 4u);
        if (best >= 1) c1i++;
        if (best > 1) {
          c1 = 0;
          c2 += (c2 < 2);
          c2i++;
        } else if (c1 < 3 && c1 > 0 && best) {
          c1++;
        }
        if ((sp & 15) == 0 && sp > 0) {
          c2 = 0;
          rice = 0;
          c1i = 0;
          c2i = 0;
          ctx_set = (sp == 16 || !is_luma) ? 0 : 2;
          if (c1 == 0) ctx_set++;
          c1 = 1;
        }
      } else {
        base += cz;
      }
      WS(cost_coded, sp) = cc;
      WS(cost_sig, sp) = cs;
      DST(bp) = out_level;
      s_sig += cs;
      if (k == 0) s_sig0 = cs;
      if (out_level) {
        cg_flag |= 1ull << gpos;
        s_coded += cc - cs;
        s_uncoded += cz;
        if (k != 0) nnz_before0++;
      }
    }
    if (last_cg >= 0) {
      if (cg) {
        const unsigned r2 = gx < (unsigned)G - 1 ? (unsigned)((cg_flag >> (gy * G + gx + 1)) & 1) : 0u;
        const unsigned l2 = gy < (unsigned)G - 1 ? (unsigned)((cg_flag >> ((gy + 1) * G + gx)) & 1) : 0u;
        const unsigned cctx = (r2 || l2) ? 1u : 0u;
        if (!((cg_flag >> gpos) & 1)) {
          base += lambda * E.sig_cg[cctx][0] - s_sig;
          WS(cost_cg_sig, cg) = lambda * E.sig_cg[cctx][0];
        } else if (cg < last_cg) {
          if (nnz_before0 == 0) {
            base -= s_sig0;
            s_sig -= s_sig0;
          }
          double zero_cost = base;
          base += lambda * E.sig_cg[cctx][1];
          zero_cost += lambda * E.sig_cg[cctx][0];
          WS(cost_cg_sig, cg) = lambda * E.sig_cg[cctx][1];
          zero_cost += s_uncoded;
          zero_cost -= s_coded;
          zero_cost -= s_sig;
          if (zero_cost < base) {
            cg_flag &= ~(1ull << gpos);
            base = zero_cost;
            WS(cost_cg_sig, cg) = lambda * E.sig_cg[cctx][0];
            for (int k = 15; k >= 0; k--) {
              const int sp = cg * 16 + k;
              const unsigned bp = rdoq_scan_pos(lg, scan_idx, sp);
              if (DST(bp)) {
                DST(bp) = 0;
                WS(cost_coded, sp) = WS(cost_zero, sp);
                WS(cost_sig, sp) = 0;
              }
            }
          }
        }
      } else {
        cg_flag |= 1ull << gpos;
      }
    }
  }
  if (last_pos < 0) {
    if (K.abs_sum) *K.abs_sum = 0;
    return;
  }

  // ---- phase B: last position ----
  double best_cost;
  if (K.root_cbf) {
    best_cost = uncoded + lambda * E.root_cbf[0][0];
    base += lambda * E.root_cbf[0][1];
  } else {
    best_cost = uncoded + lambda * E.cbf[K.cbf_ctx][0];
    base += lambda * E.cbf[K.cbf_ctx][1];
  }
  int best_last_p1 = 0;
  bool found = false;
  for (int cg = last_cg; cg >= 0 && !found; cg--) {
    const unsigned p0 = rdoq_scan_pos(lg, scan_idx, cg * 16), gpos = ((p0 >> lg) >> 2) * G + ((p0 & (N - 1)) >> 2);
    base -= WS(cost_cg_sig, cg);
    if (!((cg_flag >> gpos) & 1)) continue;
    for (int k = 15; k >= 0; k--) {
      const int sp = cg * 16 + k;
      if (sp > last_pos) continue;
      const unsigned bp = rdoq_scan_pos(lg, scan_idx, sp);
      const int lv = DST(bp);
      if (lv) {
        const unsigned py = bp >> lg, px = bp & (N - 1);
        const double lc = scan_idx == 2 ? rdoq_last_cost(E, lambda, py, px) : rdoq_last_cost(E, lambda, px, py);
        const double total = base + lc - WS(cost_sig, sp);
        if (total < best_cost) {
          best_last_p1 = sp + 1;
          best_cost = total;
        }
        if (lv > 1) {
          found = true;
          break;
        }
        base -= WS(cost_coded, sp);
        base += WS(cost_zero, sp);
      } else {
        base -= WS(cost_sig, sp);
      }
    }
  }
  for (int sp = 0; sp < best_last_p1; sp++) {
    const unsigned bp = rdoq_scan_pos(lg, scan_idx, sp);
    const int l = DST(bp);
    abs_sum += (uint32_t)l;
    DST(bp) = SRC(bp) < 0 ? -l : l;
  }
  for (int sp = best_last_p1; sp <= last_pos; sp++) DST(rdoq_scan_pos(lg, scan_idx, sp)) = 0;
  if (K.abs_sum) *K.abs_sum = abs_sum;

  // ---- phase C: sign-bit hiding with rate-aware costs (:2203-2304) ----
  if (!(A.sign_hide && abs_sum >= 2)) return;
  const long long rd_factor = A.rd_factor[pt];
  const long long kMax = 0x7fffffffffffffffll;
  int seen_last = -1;
  for (int sub = (nn - 1) >> 4; sub >= 0; sub--) {
    const int o = sub << 4;
    int first = 16, lastnz = -1, sum = 0;
    for (int n = 15; n >= 0; n--)
      if (DST(rdoq_scan_pos(lg, scan_idx, n + o))) {
        lastnz = n;
        break;
      }
    for (int n = 0; n < 16; n++)
      if (DST(rdoq_scan_pos(lg, scan_idx, n + o))) {
        first = n;
        break;
      }
    for (int n = first; n <= lastnz; n++) sum += DST(rdoq_scan_pos(lg, scan_idx, n + o));
    if (lastnz >= 0 && seen_last == -1) seen_last = 1;
    if (lastnz - first >= 4) {
      const unsigned signbit = DST(rdoq_scan_pos(lg, scan_idx, o + first)) > 0 ? 0u : 1u;
      if (signbit != (unsigned)(sum & 1)) {
        long long min_cost = kMax, cur = kMax;
        int min_pos = -1, final_change = 0, change = 0;
        for (int n = (seen_last == 1 ? lastnz : 15); n >= 0; n--) {
          const unsigned bp = rdoq_scan_pos(lg, scan_idx, n + o);
          const int lv = DST(bp), du = WS(delta_u, bp), ru = WS(rate_up, bp);
          if (lv != 0) {
            const long long up = rd_factor * (-du) + ru;
            long long down = rd_factor * (du) + WS(rate_down, bp) - (abs(lv) == 1 ? ((1 << 15) + WS(sig_delta, bp)) : 0);
            if (seen_last == 1 && lastnz == n && abs(lv) == 1) down -= (4 << 15);
            if (up < down) {
              cur = up;
              change = 1;
            } else {
              change = -1;
              cur = (n == first && abs(lv) == 1) ? kMax : down;
            }
          } else {
            cur = rd_factor * (-(long long)(abs(du))) + (1 << 15) + ru + WS(sig_delta, bp);
            change = 1;
            if (n < first) {
              const unsigned s = SRC(bp) >= 0 ? 0u : 1u;
              if (s != signbit) cur = kMax;
            }
          }
          if (cur < min_cost) {
            min_cost = cur;
            final_change = change;
            min_pos = (int)bp;
          }
        }
        // (the reference's test of the flat quantiser coefficient against +-32768, :2290, never fires)
        if (min_pos >= 0) {
          if (SRC(min_pos) >= 0)
            DST(min_pos) += final_change;
          else
            DST(min_pos) -= final_change;
        }
      }
    }
    if (seen_last == 1) seen_last = 0;
  }
#undef WS
#undef SRC
#undef DST
}
#endif // HMX_RDOQ_KERNELS

// ---------------------------------------------------------------------------------------------------------------------
// RDOQ as the quantiser of the whole-picture chain (k_intra_packed): the second decomposition of hmx_rdoq_core.h over the
// SL blocks a chain wave holds in LDS -- NOTHING per coefficient is stored, and nothing leaves the LDS:
//   * the bit-estimate tables of the wave-item's pictures and size class and its scan tables are staged in LDS once per
//     (picture group, size class) -- a walk is a chain of dependent table lookups, and from global memory every one of them
//     was a round trip (measured: 70-210 us per round of walks);
//   * |coef| * q and the cost of zero are recomputed from the coefficient in the tile wherever they are needed;
//   * the variants of 64 / 8 groups at a time ("a round" = one lane per (block, group, carry, pattern)) go through a 64-entry
//     buffer and are resolved by the block's resolving lane before the next round;
//   * last position and sign hiding walk the chosen variant of a group AGAIN with a sink that does their arithmetic
//     (RdoqLastSink, RdoqHideSink): a walk costs a few microseconds, an array per coefficient costs the LDS.
// tests/native/rdoq_core_host.cpp runs exactly these steps on the CPU against the oracle.
// On entry Ls[b].tile[row][col] holds block b's Int coefficients (fwd_tq_block without the quantiser) and Ls[b].line[0..4] =
// {active, picture, is_luma, scan_idx, cbf_ctx}, line[9] = its table in W.est, line[10] = root_cbf; on exit the tile holds the levels.
// What the encoder takes from its live state is an input of the call (hmx_set_rdoq): per picture the bit estimates for
// [luma, chroma][4 sizes] and lambda for luma / chroma blocks.
// ---------------------------------------------------------------------------------------------------------------------
struct RdoqChain {
  const EstBitsDev *est;      // [picture][luma, chroma][log2n - 2]
  const double *lambda;       // [picture][luma, chroma]
  const long long *rd_factor; // [picture][luma, chroma]
  int pic_mul;                // 0: one set of tables for every picture of the call, 1: a set per picture
  int n_pics;
  double err_scale[2][4];
};
constexpr int kRdoqMaxGroup = 2; // pictures per packing group with RDOQ: 2 x 2 tables of 1016 bytes in LDS
template <int NEST>
struct RdoqWaveLdsT {
  union {
    RdoqSpec spec[64]; // the round's variants: [block][group of the round][carry * 4 + pattern]
    double cz[1024];   // before the rounds: the costs of zero above the last position's group
    struct {
      short lev[1024]; // after them: the levels, scan order (as walked, then cut at the last position and signed)
      struct Rec {     // and what the last-position search needs of the upper groups of every block, [block][last_cg - cg]
        double cs[16], cc[16];
      } rec[32];
    } a;
    char lane4[7680];  // 4x4 blocks, one per lane: the chain's Lane4Lds (its RDOQ needs no buffer)
  } u;
  double cgs[64];            // cost_cg_sig [block][group]
  double lam[8];             // per block of the wave-item: lambda and the sign-hiding factor of its picture and plane type
  long long rdf[8];          // (fetched once per block: a load from global memory per phase otherwise)
  double rcz[8][16];         // the costs of zero of the round's groups, left by the lane of variant 0 for the resolving lane
  unsigned short scan[1024]; // the size class's scan tables: [scan_idx][position] (32x32: the diagonal scan only)
  unsigned char ginv[192];   // and their inverse for groups: [scan_idx][gy * G + gx] -> scan index of the group
  unsigned short task[64];   // the round's lanes: group | variant << 6 | (cost-of-zero row + 1) << 9; 0xffff: none
  unsigned ginfo[16];        // the round's groups [block][rank]: group | variant mask << 8 | first lane of the block's share << 16
  unsigned char sel[64];     // variant taken [block][group]
  unsigned long long zeroed[8], cg_flag[8];
  EstBitsDev est[NEST]; // the chain: [picture of the group][luma, chroma] for the wave-item's size class; line[9] of a block selects
  int key;              // ((group << 2) | size class) + 1 of what est / scan hold
};
typedef RdoqWaveLdsT<2 * kRdoqMaxGroup> RdoqWaveLds; // of the whole-picture chain
template <typename WL>
__device__ __forceinline__ void rdoq_stage_scan(WL &W, int s, int lane) { // the scan tables of size class s = log2n - 2
  // the walks are bound by the instructions they issue: a table lookup, not the arithmetic of the scan (2 KB of LDS)
  if (s == 3) {
    for (int i = lane; i < 1024; i += 64) W.scan[i] = (unsigned short)kScan32.t[0][i];
  } else if (s > 0) {
    const int nn = 16 << (2 * s);
    for (int i = lane; i < 3 * nn; i += 64) W.scan[i] = (unsigned short)rdoq_scan_pos(s + 2, i / nn, i % nn);
  }
  if (s > 0) { // straight from the constant tables: nothing here waits for the stores above
    const int lg = s + 2, gl = s, n_cg = 1 << (2 * gl), n_sc = s == 3 ? 1 : 3;
    for (int i = lane; i < n_sc * n_cg; i += 64) {
      const int sc = i / n_cg, cg = i - sc * n_cg;
      const unsigned p0 = rdoq_scan_pos(lg, sc, cg * 16);
      W.ginv[sc * 64 + (((p0 >> lg) >> 2) << gl) + ((p0 & ((1u << lg) - 1u)) >> 2)] = (unsigned char)cg;
    }
  }
}
// stage the tables of (picture group g, size class s = log2n - 2); I = pictures per group
__device__ __forceinline__ void rdoq_stage_tables(RdoqWaveLds &W, const RdoqChain &RC, int g, int s, int I, int lane) {
  const int key = ((g << 2) | s) + 1;
  if (W.key == key) return; // wave-uniform
  wave_sync();
  constexpr int kWords = (int)(sizeof(EstBitsDev) / sizeof(int));
  for (int i = lane; i < I * 2 * kWords; i += 64) {
    const int tb = i / kWords, w = i - tb * kWords;
    const int pic = min(g * I + (tb >> 1), RC.n_pics - 1) * RC.pic_mul;
    reinterpret_cast<int *>(&W.est[tb])[w] = reinterpret_cast<const int *>(&RC.est[((size_t)pic * 2 + (tb & 1)) * 4 + s])[w];
  }
  rdoq_stage_scan(W, s, lane);
  if (lane == 0) W.key = key;
  wave_sync();
}
template <int N, typename WL>
__device__ __forceinline__ RdoqConst rdoq_chain_const(const TuLds<N> &L, const RdoqChain &RC, const PicDev &P, const WL &W, int b) {
  constexpr int LG = N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 2;
  const int luma = L.line[2], pt = luma ? 0 : 1;
  RdoqConst C;
  C.lg = LG, C.scan_idx = L.line[3], C.is_luma = luma;
  C.q = pt ? P.qd[1].q : P.qd[0].q;
  C.qbits = 14 + (pt ? P.qd[1].per_qbits : P.qd[0].per_qbits) + (15 - P.bit_depth - LG);
  C.root_cbf = L.line[10], C.cbf_ctx = L.line[4], C.sign_hide = P.sign_hide;
  C.lambda = W.lam[b], C.err_scale = RC.err_scale[pt][LG - 2], C.rd_factor = W.rdf[b];
  return C;
}
template <int N>
struct RdoqTileIn { // entry of a group from the block's coefficient tile
  const TuLds<N> *L;
  const RdoqConst *C;
  __device__ __forceinline__ int operator()(int, unsigned bp) const {
    constexpr int LG = N == 8 ? 3 : N == 16 ? 4 : 5;
    return L->tile[bp >> LG][bp & (N - 1)];
  }
};
struct RdoqSpecCzSink { // a variant of a round; variant 0 also leaves the group's costs of zero
  RdoqSpec *o;
  double *cz16; // NULL for the other variants
  __device__ __forceinline__ void add(int k, double v) { o->add[k] = v; }
  __device__ __forceinline__ void pos(int k, unsigned, int, double, double, int, int, int, int, double cz) {
    if (cz16) cz16[k] = cz;
  }
};
struct RdoqRecordSink { // the chosen variant of a group: its levels as walked, and (upper groups) the search's two costs per entry
  short *lev;           // the group's 16 entries
  double *cs, *cc;      // NULL: no record
  __device__ __forceinline__ void add(int, double) {}
  __device__ __forceinline__ void pos(int k, unsigned, int level, double c_coded, double c_sig, int, int, int, int, double) {
    lev[k] = (short)level;
    if (cs) cs[k] = c_sig, cc[k] = c_coded;
  }
};
#ifdef HMX_PACK_PROFILE
static __device__ unsigned long long g_rdoq_prof[40]; // [log2n - 2][step 0..8, calls], 10 ns units
#define RQ_T(i)                                                         \
  if (lane == 0) {                                                      \
    const unsigned long long now_ = wall_clock64();                     \
    atomicAdd(&g_rdoq_prof[(LG - 2) * 10 + (i)], now_ - tprev_);        \
    tprev_ = now_;                                                      \
  }
#define RQ_T0 unsigned long long tprev_ = wall_clock64()
#define RQ_COUNT if (lane == 0) atomicAdd(&g_rdoq_prof[(LG - 2) * 10 + 9], 1ull)
#else
#define RQ_T(i)
#define RQ_T0
#define RQ_COUNT
#endif
template <int N, int SL, typename WL>
__device__ __forceinline__ void rdoq_wave_tiles(TuLds<N> *Ls, WL &W, const RdoqChain &RC, const PicDev &P, int lane) {
  constexpr int LG = N == 8 ? 3 : N == 16 ? 4 : 5, NN = N * N, NCG = NN / 16, G = N / 4;
  constexpr int GPR = 8 / SL; // groups of a block per round: 64 lanes = SL blocks x GPR groups x 8 variants
  static_assert(SL * NCG <= 64 && SL * NN <= 1024 && SL * GPR * 8 == 64, "one lane per group, one round per 64 variants");
  RQ_T0;
  // line[5] = last position, [6] = last position + 1 after the search, [7] = sum of levels, [8] = highest group with a level
  constexpr int GM0 = N == 8 ? 12 : 32; // line[GM0 + cg]: min(the group's highest candidate level, 2)
  static_assert(GM0 + NCG <= 4 * N + 2, "the block's scratch line holds the groups' candidate classes");
  if (lane < SL * NCG) Ls[lane / NCG].line[GM0 + lane % NCG] = 0;
  if (lane < SL) {
    TuLds<N> &L = Ls[lane];
    L.line[5] = -1, L.line[6] = 0, L.line[7] = 0, L.line[8] = -1;
    const int idx = L.line[1] * RC.pic_mul * 2 + (L.line[2] ? 0 : 1);
    W.lam[lane] = RC.lambda[idx], W.rdf[lane] = RC.rd_factor[idx];
  }
  wave_sync();
  auto scan_of = [&](int scan_idx, int sp) -> unsigned { return W.scan[(N == 32 ? 0 : scan_idx * NN) + sp]; };
  for (int t = lane; t < SL * NN; t += 64) { // the last position with a non-zero candidate
    const int b = t / NN, sp = t - b * NN;
    TuLds<N> &L = Ls[b];
    if (!L.line[0]) continue;
    const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, b);
    const unsigned bp = scan_of(C.scan_idx, sp);
    int l;
    double z;
    rdoq_prep(L.tile[bp >> LG][bp & (N - 1)], C, l, z);
    const unsigned m = rdoq_max_level(l, C.qbits);
    if (m > 0) {
      atomicMax(&L.line[5], sp);
      atomicMax(&L.line[GM0 + (sp >> 4)], (int)(m < 2u ? m : 2u));
    }
  }
  wave_sync();
  for (int t = lane; t < SL * NN; t += 64) { // the costs of zero above the last position's group, for the resolving lane
    const int b = t / NN, sp = t - b * NN;
    TuLds<N> &L = Ls[b];
    if (!L.line[0] || L.line[5] < 0 || sp < ((L.line[5] >> 4) + 1) * 16) continue;
    const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, b);
    const unsigned bp = scan_of(C.scan_idx, sp);
    int l;
    double z;
    rdoq_prep(L.tile[bp >> LG][bp & (N - 1)], C, l, z);
    W.u.cz[t] = z;
  }
  wave_sync();
  RQ_T(0);
  // ---- the resolving lane of block b is lane b: its running state stays in its registers
  const bool resolver = lane < SL && Ls[lane < SL ? lane : 0].line[0] && Ls[lane < SL ? lane : 0].line[5] >= 0;
  TuLds<N> &LR = Ls[lane < SL ? lane : 0];
  RdoqConst CR;
  RdoqRun R;
  int carry = 0;
  const int my_last_pos = LR.line[5], my_last_cg = my_last_pos >> 4;
  if (resolver) {
    CR = rdoq_chain_const<N>(LR, RC, P, W, lane);
    rdoq_run_init(R);
    for (int cg = NCG - 1; cg > my_last_cg; cg--) { // (a group's 16 loads go out together)
      const double *z = &W.u.cz[lane * NN + cg * 16];
#pragma unroll
      for (int k = 15; k >= 0; k--) rdoq_resolve_above(R, z[k]);
    }
  }
  wave_sync(); // the buffer changes hands
  RQ_T(1);
  const EstBitsDev &ER = W.est[LR.line[9]];
  // Rounds composed at run time.  A group gets one lane per variant it can STILL take (rdoq_variant_mask: what earlier rounds
  // resolved is known; what the same round leaves open the candidates often close -- a group without a candidate level stays
  // uncoded, a group without a candidate above 1 hands on no carry), a block as many groups per round as its share of the 64
  // lanes holds (at most 16 / SL).  The block's resolving lane plans, every lane walks, the resolving lane resolves.
  constexpr int SLOTS = 64 / SL, GMAXB = 16 / SL, RCZB = 8 / SL;
  static_assert(GPR == RCZB, "cost-of-zero rows per block");
  int next = resolver ? my_last_cg : -1;
  if constexpr (SL == 1) {
    // 32x32: one block, one planning lane -- composing rounds of up to 16 groups costs more than it saves (measured: 206 us
    // against 188 us of rounds per block): fixed rounds of 8 groups x 8 variants
    const int rounds = __shfl(resolver ? (my_last_cg + GPR) / GPR : 0, 0, 64);
    for (int r = 0; r < rounds; r++) {
      { // one lane per (group of the round, carry, pattern)
        const int j = lane >> 3, v = lane & 7;
        TuLds<N> &L = Ls[0];
        const int last_pos = L.line[5], cg = (last_pos >> 4) - r * GPR - j;
        if (cg >= 0) {
          const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, 0);
          const EstBitsDev &E = W.est[L.line[9]];
          const int scan_idx = C.scan_idx;
          auto bp_of = [&](int sp) { return scan_of(scan_idx, sp); };
          RdoqSpecCzSink sink{&W.u.spec[lane], v == 0 ? W.rcz[lane >> 3] : nullptr};
          const RdoqCgSums S = rdoq_walk_cg_in(C, E, cg, bp_of, RdoqTileIn<N>{&L, &C}, v & 3, v >> 2, last_pos, sink);
          W.u.spec[lane].S = S;
        }
      }
      wave_sync();
      if (resolver) {
        for (int j = 0; j < GPR; j++) {
          const int cg = my_last_cg - r * GPR - j;
          if (cg < 0) break;
          const unsigned p0 = scan_of(CR.scan_idx, cg * 16), g = ((p0 & (unsigned)(N - 1)) >> 2) | (((p0 >> LG) >> 2) << 8);
          const double *rcz = W.rcz[j];
          auto cz_of = [rcz](int k) { return rcz[k]; };
          double cg_sig;
          const int v = rdoq_resolve_group(CR, ER, cg, my_last_cg, g, &W.u.spec[j * 8], cz_of, R, carry, cg_sig);
          W.sel[cg] = (unsigned char)v;
          W.cgs[cg] = cg_sig;
        }
      }
      wave_sync();
    }
    next = -1;
  }
  for (; SL > 1;) {
    W.task[lane] = 0xffffu;
    wave_sync();
    int ng = 0;
    if (next >= 0) {
      const int scan_idx = CR.scan_idx;
      auto gmax_of = [&](int cg) { return LR.line[GM0 + cg]; };
      auto cg_at = [&](int gy, int gx) { return (int)W.ginv[(N == 32 ? 0 : scan_idx * 64) + gy * G + gx]; };
      int used = 0;
      for (int cg = next; cg >= 0 && ng < GMAXB; cg--) {
        const unsigned p0 = scan_of(scan_idx, cg * 16), g = ((p0 & (unsigned)(N - 1)) >> 2) | (((p0 >> LG) >> 2) << 8);
        const unsigned mask = rdoq_variant_mask(CR, cg, my_last_cg, next, g, R, carry, gmax_of, cg_at);
        const int cnt = __builtin_popcount(mask);
        if (used + cnt > SLOTS) break;
        W.ginfo[lane * GMAXB + ng] = (unsigned)cg | (mask << 8) | ((unsigned)used << 16);
        for (int v = 0, rank = 0; v < 8; v++)
          if ((mask >> v) & 1) {
            W.task[lane * SLOTS + used + rank] = (unsigned short)(cg | (v << 6) | ((rank == 0 && ng < RCZB ? ng + 1 : 0) << 9));
            rank++;
          }
        used += cnt;
        ng++;
      }
    }
    if (!__ballot(ng > 0)) break; // every block is through
    wave_sync();
    { // one lane per (block, group of the round, variant it can take)
      const unsigned t = W.task[lane];
      if (t != 0xffffu) {
        const int b = lane / SLOTS, cg = (int)(t & 63u), v = (int)((t >> 6) & 7u), row = (int)((t >> 9) & 15u);
        TuLds<N> &L = Ls[b];
        const int last_pos = L.line[5];
        const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, b);
        const EstBitsDev &E = W.est[L.line[9]];
        const int scan_idx = C.scan_idx;
        auto bp_of = [&](int sp) { return scan_of(scan_idx, sp); };
        RdoqSpecCzSink sink{&W.u.spec[lane], row ? W.rcz[b * RCZB + row - 1] : nullptr};
        const RdoqCgSums S = rdoq_walk_cg_in(C, E, cg, bp_of, RdoqTileIn<N>{&L, &C}, v & 3, v >> 2, last_pos, sink);
        W.u.spec[lane].S = S;
      }
    }
    wave_sync();
    for (int q = 0; q < ng; q++) { // the resolving lane: the round's groups in the reference's order
      const unsigned info = W.ginfo[lane * GMAXB + q], mask = (info >> 8) & 255u;
      const int cg = (int)(info & 255u);
      const RdoqSpec *b0 = &W.u.spec[lane * SLOTS + (int)(info >> 16)];
      const int scan_idx = CR.scan_idx;
      const unsigned p0 = scan_of(scan_idx, cg * 16), g = ((p0 & (unsigned)(N - 1)) >> 2) | (((p0 >> LG) >> 2) << 8);
      auto spec_of = [b0, mask](int v) -> const RdoqSpec & { return b0[__builtin_popcount(mask & ((1u << v) - 1u))]; };
      const double *rcz = W.rcz[lane * RCZB + (q < RCZB ? q : 0)];
      const bool have_cz = q < RCZB;
      auto cz_of = [&](int k) { // the round's first groups find their costs of zero in LDS, the others form them again
        if (have_cz) return rcz[k];
        const unsigned bp = scan_of(scan_idx, cg * 16 + k);
        int l;
        double z;
        rdoq_prep(LR.tile[bp >> LG][bp & (N - 1)], CR, l, z);
        return z;
      };
      double cg_sig;
      const int v = rdoq_resolve_group_fn(CR, ER, cg, my_last_cg, g, spec_of, cz_of, R, carry, cg_sig);
      W.sel[lane * NCG + cg] = (unsigned char)v;
      W.cgs[lane * NCG + cg] = cg_sig;
      next = cg - 1;
    }
    wave_sync();
  }
  RQ_T(2);
  if (resolver) W.zeroed[lane] = R.zeroed, W.cg_flag[lane] = R.cg_flag;
  wave_sync(); // and the buffer changes hands again: levels and the search's records
  constexpr int REC = 32 / SL; // records per block: its REC upper groups (8x8: all four)
  if (lane < SL * NCG) { // one lane per (block, group): the chosen variant walked again
    const int b = lane / NCG, cg = lane - b * NCG;
    TuLds<N> &L = Ls[b];
    short *out = &W.u.a.lev[b * NN + cg * 16];
    const int last_pos = L.line[5], last_cg = last_pos >> 4;
    bool walked = false;
    if (L.line[0] && last_pos >= 0 && cg <= last_cg && !((W.zeroed[b] >> cg) & 1)) {
      const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, b);
      const EstBitsDev &E = W.est[L.line[9]];
      const int scan_idx = C.scan_idx;
      auto bp_of = [&](int sp) { return scan_of(scan_idx, sp); };
      const unsigned p0 = scan_of(scan_idx, cg * 16), gpos = ((p0 >> LG) >> 2) * (unsigned)G + ((p0 & (unsigned)(N - 1)) >> 2);
      const bool rec = last_cg - cg < REC && ((W.cg_flag[b] >> gpos) & 1);
      auto &RR = W.u.a.rec[b * REC + (rec ? last_cg - cg : 0)];
      RdoqRecordSink sink{out, rec ? RR.cs : nullptr, rec ? RR.cc : nullptr};
      const int v = W.sel[b * NCG + cg];
      rdoq_walk_cg_in(C, E, cg, bp_of, RdoqTileIn<N>{&L, &C}, v & 3, v >> 2, last_pos, sink);
      walked = true;
    }
    if (!walked) {
#pragma unroll
      for (int k = 0; k < 16; k += 4) *reinterpret_cast<uint2 *>(out + k) = make_uint2(0u, 0u);
    }
  }
  wave_sync();
  RQ_T(3);
  if (resolver) { // the last position (:2132-2200): from the top until a level above 1 ends the search, over the records
    RdoqLast T;
    rdoq_last_init(CR, ER, R, T);
    const int scan_idx = CR.scan_idx;
    auto bp_of = [&](int sp) { return scan_of(scan_idx, sp); };
    for (int cg = my_last_cg; cg >= 0 && !T.found; cg--) {
      const unsigned p0 = scan_of(scan_idx, cg * 16), gpos = ((p0 >> LG) >> 2) * (unsigned)G + ((p0 & (unsigned)(N - 1)) >> 2);
      rdoq_last_group(T, W.cgs[lane * NCG + cg]);
      if (!((R.cg_flag >> gpos) & 1)) continue;
      if (my_last_cg - cg < REC) {
        const auto &RR = W.u.a.rec[lane * REC + (my_last_cg - cg)];
        const short *l16 = &W.u.a.lev[lane * NN + cg * 16];
        for (int k4 = 12; k4 >= 0 && !T.found; k4 -= 4) { // four entries' records at a time: one LDS round trip, not four
          int lv4[4];
          double cc4[4], cs4[4];
#pragma unroll
          for (int i = 0; i < 4; i++) lv4[i] = l16[k4 + i], cc4[i] = RR.cc[k4 + i], cs4[i] = RR.cs[k4 + i];
#pragma unroll
          for (int i = 3; i >= 0; i--) {
            const int sp = cg * 16 + k4 + i;
            if (sp > my_last_pos || T.found) continue;
            const int lv = lv4[i];
            unsigned bp = 0;
            double cz = 0;
            if (lv) {
              int l;
              bp = bp_of(sp);
              rdoq_prep(LR.tile[bp >> LG][bp & (N - 1)], CR, l, cz);
            }
            rdoq_last_pos(CR, ER, T, sp, bp, lv, cc4[i], cs4[i], cz);
          }
        }
      } else { // deeper than the records reach (rare): the group walked once more with the search as the sink
        RdoqLastSink sink{CR, ER, T, cg * 16, my_last_pos};
        const int v = W.sel[lane * NCG + cg];
        rdoq_walk_cg_in(CR, ER, cg, bp_of, RdoqTileIn<N>{&LR, &CR}, v & 3, v >> 2, my_last_pos, sink);
      }
    }
    LR.line[6] = T.best_last_p1;
  }
  wave_sync();
  if (lane < SL * NCG) { // one lane per (block, group): levels cut at the last position and signed; their sum; the highest group
    const int b = lane / NCG, cg = lane - b * NCG;
    TuLds<N> &L = Ls[b];
    short *l16 = &W.u.a.lev[b * NN + cg * 16];
    if (L.line[0] && L.line[5] >= 0 && cg <= (L.line[5] >> 4)) {
      const int blp1 = L.line[6], scan_idx = L.line[3];
      int sum = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int sp = cg * 16 + k, l = sp < blp1 ? (int)l16[k] : 0;
        const unsigned bp = scan_of(scan_idx, sp);
        sum += l;
        l16[k] = (short)(L.tile[bp >> LG][bp & (N - 1)] < 0 ? -l : l);
      }
      if (sum) {
        atomicAdd(&L.line[7], sum);
        atomicMax(&L.line[8], cg);
      }
    }
  }
  wave_sync();
  RQ_T(4);
  if (P.sign_hide && lane < SL * NCG) { // one lane per (block, group): rate-aware sign hiding, the group walked a third time
    const int b = lane / NCG, cg = lane - b * NCG;
    TuLds<N> &L = Ls[b];
    short *l16 = &W.u.a.lev[b * NN + cg * 16];
    if (L.line[0] && L.line[5] >= 0 && L.line[7] >= 2 && cg <= L.line[8]) {
      auto lev_of = [l16](int n) { return (int)l16[n]; };
      RdoqHide H;
      if (rdoq_hide_begin(cg == L.line[8], lev_of, H)) { // implies the group holds a level: it was walked above
        const RdoqConst C = rdoq_chain_const<N>(L, RC, P, W, b);
        const EstBitsDev &E = W.est[L.line[9]];
        const int scan_idx = C.scan_idx;
        auto bp_of = [&](int sp) { return scan_of(scan_idx, sp); };
        unsigned neg = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const unsigned bp = bp_of(cg * 16 + k);
          neg |= (L.tile[bp >> LG][bp & (N - 1)] < 0 ? 1u : 0u) << k;
        }
        RdoqHideSink<decltype(lev_of)> sink{C, H, lev_of, neg};
        const int v = W.sel[b * NCG + cg];
        rdoq_walk_cg_in(C, E, cg, bp_of, RdoqTileIn<N>{&L, &C}, v & 3, v >> 2, L.line[5], sink);
        if (H.min_pos >= 0) l16[H.min_pos] = (short)(l16[H.min_pos] + rdoq_hide_change(H, (neg >> H.min_pos) & 1u));
      }
    }
  }
  wave_sync();
  RQ_T(5);
  for (int t = lane; t < SL * NN; t += 64) { // the levels take the coefficients' place in the tile
    const int b = t / NN, sp = t - b * NN;
    TuLds<N> &L = Ls[b];
    if (!L.line[0]) continue;
    const unsigned bp = scan_of(L.line[3], sp);
    L.tile[bp >> LG][bp & (N - 1)] = L.line[5] < 0 ? 0 : (int)W.u.a.lev[t];
  }
  wave_sync();
  RQ_T(6);
  RQ_COUNT;
}

// The same routine behind hmx_batch_xRateDistOptQuant: a wave takes SL consecutive blocks of one size from the (sorted) list,
// brings their coefficients and tables into LDS, and writes levels and sums back.  RC.lambda / RC.rd_factor: [luma, chroma].
template <int N, int SL>
__global__ __launch_bounds__(64) void k_rdoq_tiles(RdoqArgs A, RdoqChain RC, PicDev P) {
  constexpr int LG = N == 8 ? 3 : N == 16 ? 4 : 5, NN = N * N;
  __shared__ TuLds<N> tiles[SL];
  __shared__ RdoqWaveLdsT<SL> W;
  const int lane = threadIdx.x, first = blockIdx.x * SL;
  rdoq_stage_scan(W, LG - 2, lane);
  constexpr int kWords = (int)(sizeof(EstBitsDev) / sizeof(int));
  for (int b = 0; b < SL; b++) {
    const bool active = first + b < A.n;
    const RdoqBlock K = A.blocks[active ? first + b : first];
    TuLds<N> &L = tiles[b];
    if (lane == 0)
      L.line[0] = active, L.line[1] = 0, L.line[2] = K.is_luma, L.line[3] = K.scan_idx, L.line[4] = K.cbf_ctx, L.line[9] = b, L.line[10] = K.root_cbf;
    if (!active) continue;
    for (int i = lane; i < kWords; i += 64) reinterpret_cast<int *>(&W.est[b])[i] = reinterpret_cast<const int *>(&A.est[K.est_idx])[i];
    for (int i = lane; i < NN; i += 64) L.tile[i >> LG][i & (N - 1)] = K.src[(i >> LG) * K.src_stride + (i & (N - 1))];
  }
  wave_sync();
  rdoq_wave_tiles<N, SL>(tiles, W, RC, P, lane);
  for (int b = 0; b < SL; b++) {
    if (first + b >= A.n) break;
    const RdoqBlock K = A.blocks[first + b];
    TuLds<N> &L = tiles[b];
    for (int i = lane; i < NN; i += 64) K.dst[(i >> LG) * K.dst_stride + (i & (N - 1))] = L.tile[i >> LG][i & (N - 1)];
    if (lane == 0 && K.abs_sum) *K.abs_sum = (uint32_t)L.line[7];
  }
}

// One 4x4 block in ONE lane (the lane-per-block 4x4 chain): a single coefficient group, so no variants -- the walk with a
// sink that keeps the two running sums, then the same re-walks as above.  The lane's coefficients (scan order) and levels
// live in its LDS rows (c16: 16 ints; l16: 16 shorts), nothing in private arrays.  Returns the levels in l16, scan order.
struct RdoqSumSink { // resolve of a single group: the terms arrive in the order the reference adds them
  RdoqRun &R;
  HMX_HD void add(int, double v) { R.base += v; }
  HMX_HD void pos(int, unsigned, int, double, double, int, int, int, int, double cz) { R.uncoded += cz; }
};
struct RdoqLaneIn {
  const int *c16;
  const RdoqConst *C;
  __device__ __forceinline__ int operator()(int k, unsigned) const { return c16[k]; }
};
struct RdoqLaneLevelSink {
  short *out;
  const int *c16;
  int blp1, sum;
  __device__ __forceinline__ void add(int, double) {}
  __device__ __forceinline__ void pos(int k, unsigned, int level, double, double, int, int, int, int, double) {
    const int l = k < blp1 ? level : 0;
    sum += l;
    out[k] = (short)(c16[k] < 0 ? -l : l);
  }
};
template <typename WL>
__device__ __forceinline__ void rdoq_lane_4x4(const int *c16, short *l16, int pic, int est_slot, bool luma, int scan_idx, int cbf_ctx,
                                              const RdoqChain &RC, const WL &W, const PicDev &P) {
  const int pt = luma ? 0 : 1;
  pic *= RC.pic_mul;
  RdoqConst C;
  C.lg = 2, C.scan_idx = scan_idx, C.is_luma = luma;
  C.q = pt ? P.qd[1].q : P.qd[0].q;
  C.qbits = 14 + (pt ? P.qd[1].per_qbits : P.qd[0].per_qbits) + (15 - P.bit_depth - 2);
  C.root_cbf = 0, C.cbf_ctx = cbf_ctx, C.sign_hide = P.sign_hide;
  C.lambda = RC.lambda[pic * 2 + pt], C.err_scale = RC.err_scale[pt][0], C.rd_factor = RC.rd_factor[pic * 2 + pt];
  const EstBitsDev &E = W.est[est_slot];
  auto bp_of = [scan_idx](int sp) { // raster position of scan entry sp of a 4x4 block
    const unsigned dg = (unsigned)((0xfbe7ad369c258140ull >> (4 * sp)) & 15), ver = (unsigned)(((sp & 3) << 2) | (sp >> 2));
    return scan_idx == 1 ? (unsigned)sp : (scan_idx == 2 ? ver : dg);
  };
  const RdoqLaneIn in{c16, &C};
  int last_pos = -1;
  for (int sp = 0; sp < 16; sp++) {
    int l;
    double z;
    rdoq_prep(c16[sp], C, l, z);
    if (rdoq_max_level(l, C.qbits) > 0) last_pos = sp;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) reinterpret_cast<int *>(l16)[k] = 0; // (rows of 13 ints: 4-byte aligned)
  if (last_pos < 0) return;
  RdoqRun R;
  rdoq_run_init(R);
  {
    RdoqSumSink sink{R};
    rdoq_walk_cg_in(C, E, 0, bp_of, in, 0, 0, last_pos, sink);
  }
  R.cg_flag = 1; // group 0 always counts as coded (:2088)
  RdoqLast T;
  rdoq_last_init(C, E, R, T);
  rdoq_last_group(T, 0.0);
  {
    RdoqLastSink sink{C, E, T, 0, last_pos};
    rdoq_walk_cg_in(C, E, 0, bp_of, in, 0, 0, last_pos, sink);
  }
  RdoqLaneLevelSink lsink{l16, c16, T.best_last_p1, 0};
  rdoq_walk_cg_in(C, E, 0, bp_of, in, 0, 0, last_pos, lsink);
  if (C.sign_hide && lsink.sum >= 2) {
    auto lev_of = [l16](int n) { return (int)l16[n]; };
    RdoqHide H;
    if (rdoq_hide_begin(true, lev_of, H)) {
      unsigned neg = 0;
      for (int k = 0; k < 16; k++) neg |= (c16[k] < 0 ? 1u : 0u) << k;
      RdoqHideSink<decltype(lev_of)> sink{C, H, lev_of, neg};
      rdoq_walk_cg_in(C, E, 0, bp_of, in, 0, 0, last_pos, sink);
      if (H.min_pos >= 0) l16[H.min_pos] = (short)(l16[H.min_pos] + rdoq_hide_change(H, (neg >> H.min_pos) & 1u));
    }
  }
}

} // namespace hmx

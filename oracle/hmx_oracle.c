/*
 * hmx_oracle.c -- CPU restatement of the HM block hot path.  TEST INFRASTRUCTURE ONLY
 * (see hmx_oracle.h for scope, rules of use and parity status: PINNED against the reference
 * compiled into oracle/_ref/, tests/test_oracle_vs_ref.py and tests/golden/).
 *
 * Written from the arithmetic, not transliterated: transforms are stated as the matrix products
 * the reference's partial butterflies factorise (exact in 32-bit integers), reference samples
 * as one linear line of 4N+1 samples, prediction in a "main/side" frame with a final transpose.
 * "COM/" = /root/reference/source/Lib/TLibCommon/.
 */
#include "hmx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 64

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return l;
}

/* ------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------- */

/* Column 0 of the 32-point core matrix (COM/TComRom.cpp:341-377, g_aiT32[k][0]).  Entry p is the
 * integer chosen for cos(p*pi/64); every other entry of every DCT size is +-one of these, found
 * by folding the angle (2n+1)*k*pi/64 into the first quadrant. */
static const int16_t kCos64[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                   61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};

void hmo_dct_matrix(int N, int16_t *T) {
  int step = 32 / N; /* T_N[k][n] = T_32[k*32/N][n] (COM/TComRom.cpp:303-340 are sub-samplings) */
  for (int k = 0; k < N; k++)
    for (int n = 0; n < N; n++) {
      int p = ((2 * n + 1) * k * step) & 127, sign = 1;
      if (p > 64) p = 128 - p;
      if (p > 32) {
        p = 64 - p;
        sign = -1;
      }
      T[k * N + n] = (int16_t)(sign * kCos64[p]);
    }
}

/* 4x4 DST-VII, g_as_DST_MAT_4 (COM/TComRom.cpp:399-405): round(128 * 2/3 * sin((2k+1)(n+1)pi/9)). */
void hmo_dst_matrix(int16_t *T) {
  static const int16_t s[5] = {0, 29, 55, 74, 84}; /* 85.33*sin(j*20deg), j = 0..4 */
  for (int k = 0; k < 4; k++)
    for (int n = 0; n < 4; n++) {
      int j = ((2 * k + 1) * (n + 1)) % 18, sign = 1; /* angle j*20deg, period 360deg */
      if (j > 9) {
        j = 18 - j;
        sign = -1;
      }
      if (j > 4) j = 9 - j;
      T[k * 4 + n] = (int16_t)(sign * s[j]);
    }
}

int hmo_quant_scale(int rem) {
  static const int q[6] = {26214, 23302, 20560, 18396, 16384, 14564}; /* COM/TComRom.cpp:293-296 */
  return q[rem];
}
int hmo_inv_quant_scale(int rem) {
  static const int q[6] = {40, 45, 51, 57, 64, 72}; /* COM/TComRom.cpp:298-301 */
  return q[rem];
}
int hmo_chroma_scale(int idx) { /* g_aucChromaScale[58], COM/TComRom.cpp:380-386 */
  static const uint8_t mid[13] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37};
  if (idx < 30) return idx;
  if (idx >= 43) return idx - 6;
  return mid[idx - 30];
}

/* Scan tables g_auiSigLastScan[scan][log2N-1] (COM/TComRom.cpp:564-698 with REMOVAL_8x2_2x8_CG):
 * coefficient groups of 4x4, groups ordered like the samples inside a group. */
static uint32_t *g_scan[4][6];

static void diag_order(int W, int *out) { /* up-right diagonal: x ascending, y descending */
  int c = 0;
  for (int d = 0; d <= 2 * W - 2; d++)
    for (int x = (d < W ? 0 : d - W + 1); x <= d && x < W; x++) out[c++] = (d - x) * W + x;
}

static void build_scan(int scan_idx, int log2n) {
  int N = 1 << log2n, G = N / 4;
  uint32_t *t = (uint32_t *)malloc(sizeof(uint32_t) * N * N);
  int cg[64], in[16];
  if (scan_idx == HMO_SCAN_DIAG) {
    diag_order(G, cg);
    diag_order(4, in);
  } else if (scan_idx == HMO_SCAN_HOR) {
    for (int i = 0; i < G * G; i++) cg[i] = i;
    for (int i = 0; i < 16; i++) in[i] = i;
  } else { /* VER: column-major groups, column-major inside */
    for (int gx = 0, c = 0; gx < G; gx++)
      for (int gy = 0; gy < G; gy++) cg[c++] = gy * G + gx;
    for (int x = 0, c = 0; x < 4; x++)
      for (int y = 0; y < 4; y++) in[c++] = y * 4 + x;
  }
  for (int g = 0; g < G * G; g++) {
    int gy = cg[g] / G, gx = cg[g] % G;
    for (int i = 0; i < 16; i++) {
      int y = in[i] / 4, x = in[i] % 4;
      t[g * 16 + i] = (uint32_t)((gy * 4 + y) * N + gx * 4 + x);
    }
  }
  g_scan[scan_idx][log2n] = t;
}

const uint32_t *hmo_scan(int scan_idx, int log2n) {
  if (scan_idx == HMO_SCAN_ZIGZAG) scan_idx = HMO_SCAN_DIAG; /* COM/TComTrQuant.cpp:1135-1138 */
  if (!g_scan[scan_idx][log2n]) build_scan(scan_idx, log2n);
  return g_scan[scan_idx][log2n];
}

/* ------------------------------------------------------------------------------------------
 * Transforms.  One 1-D pass of the reference = (matrix * rows) with transposed store:
 *   forward  dst[k*line + j] = wrap16((sum_n M[k][n]*src[j*N + n] + rnd) >> shift)    (:417-795)
 *   inverse  dst[j*N + n]    = clip16((sum_k M[k][n]*src[k*line + j] + rnd) >> shift)
 * ---------------------------------------------------------------------------------------- */
static void matrix_for(int N, int use_dst, int16_t *M) {
  if (use_dst)
    hmo_dst_matrix(M);
  else
    hmo_dct_matrix(N, M);
}

void hmo_fwd_pass(const int16_t *src, int16_t *dst, int N, int shift, int line, int use_dst) {
  int16_t M[32 * 32];
  matrix_for(N, use_dst, M);
  int rnd = 1 << (shift - 1);
  for (int j = 0; j < line; j++)
    for (int k = 0; k < N; k++) {
      int acc = 0;
      for (int n = 0; n < N; n++) acc += M[k * N + n] * src[j * N + n];
      dst[k * line + j] = (int16_t)(uint16_t)((acc + rnd) >> shift); /* plain store to short: wraps */
    }
}

void hmo_inv_pass(const int16_t *src, int16_t *dst, int N, int shift, int line, int use_dst) {
  int16_t M[32 * 32];
  matrix_for(N, use_dst, M);
  int rnd = 1 << (shift - 1);
  for (int j = 0; j < line; j++)
    for (int n = 0; n < N; n++) {
      int acc = 0;
      for (int k = 0; k < N; k++) acc += M[k * N + n] * src[k * line + j];
      dst[j * N + n] = (int16_t)clip3(-32768, 32767, (acc + rnd) >> shift);
    }
}

/* COM/TComTrQuant.cpp:803-885: shift_1st = log2N - 1 + (B - 8), shift_2nd = log2N + 6;
 * INTRA_TRANS_SIMP: 4x4 uses the DST in both passes whenever mode != REG_DCT. */
void hmo_xTrMxN(const int16_t *block, int16_t *coeff, int N, unsigned mode, int B) {
  int16_t tmp[32 * 32];
  int lg = ilog2(N), dst4 = (N == 4 && mode != HMO_REG_DCT);
  hmo_fwd_pass(block, tmp, N, lg - 1 + (B - 8), N, dst4);
  hmo_fwd_pass(tmp, coeff, N, lg + 6, N, dst4);
}

/* COM/TComTrQuant.cpp:892-972: shift_1st = 7, shift_2nd = 12 - (B - 8). */
void hmo_xITrMxN(const int16_t *coeff, int16_t *block, int N, unsigned mode, int B) {
  int16_t tmp[32 * 32];
  int dst4 = (N == 4 && mode != HMO_REG_DCT);
  hmo_inv_pass(coeff, tmp, N, 7, N, dst4);
  hmo_inv_pass(tmp, block, N, 12 - (B - 8), N, dst4);
}

void hmo_xT(unsigned mode, const int16_t *resi, int stride, int32_t *coef, int N, int B) {
  int16_t blk[32 * 32], out[32 * 32];
  for (int r = 0; r < N; r++) memcpy(blk + r * N, resi + r * stride, sizeof(int16_t) * N);
  hmo_xTrMxN(blk, out, N, mode, B);
  for (int i = 0; i < N * N; i++) coef[i] = out[i];
}

void hmo_xIT(unsigned mode, const int32_t *coef, int16_t *resi, int stride, int N, int B) {
  int16_t in[32 * 32], blk[32 * 32];
  for (int i = 0; i < N * N; i++) in[i] = (int16_t)(uint16_t)coef[i]; /* (short) cast, :1602 */
  hmo_xITrMxN(in, blk, N, mode, B);
  for (int r = 0; r < N; r++) memcpy(resi + r * stride, blk + r * N, sizeof(int16_t) * N);
}

/* COM/TComTrQuant.cpp:1622-1704, shift = 15 - B - log2N */
void hmo_xTransformSkip(const int16_t *resi, int stride, int32_t *coef, int N, int B) {
  int sh = 15 - B - ilog2(N);
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) {
      int v = resi[r * stride + c];
      coef[r * N + c] = sh >= 0 ? (int32_t)((uint32_t)v << sh) : (v + (1 << (-sh - 1))) >> (-sh);
    }
}

void hmo_xITransformSkip(const int32_t *coef, int16_t *resi, int stride, int N, int B) {
  int sh = 15 - B - ilog2(N);
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) {
      int v = coef[r * N + c];
      int o = sh > 0 ? (v + (1 << (sh - 1))) >> sh : (int32_t)((uint32_t)v << (-sh));
      resi[r * stride + c] = (int16_t)(uint16_t)o;
    }
}

/* ------------------------------------------------------------------------------------------
 * Quantisation
 * ---------------------------------------------------------------------------------------- */
hmo_qp hmo_setQPforQuant(int qpy, int is_chroma, int qp_bd_offset, int chroma_qp_offset) {
  int q; /* COM/TComTrQuant.cpp:192-222 (CHROMA_QP_EXTENSION) */
  if (!is_chroma)
    q = qpy + qp_bd_offset;
  else {
    q = clip3(-qp_bd_offset, 57, qpy + chroma_qp_offset);
    q = (q < 0) ? q + qp_bd_offset : hmo_chroma_scale(q) + qp_bd_offset;
  }
  hmo_qp r = {q, q / 6, q % 6};
  return r;
}

int hmo_coef_scan_idx(int N, int is_luma, int is_intra, int dir_mode) {
  if (!is_intra) return HMO_SCAN_ZIGZAG; /* COM/TComDataCU.cpp:4014-4063 */
  int multi = is_luma ? (N == 4 || N == 8) : (N == 4 || N == 2);
  if (!multi) return HMO_SCAN_ZIGZAG;
  if (abs(dir_mode - 26) < 5) return HMO_SCAN_HOR;
  if (abs(dir_mode - 10) < 5) return HMO_SCAN_VER;
  return HMO_SCAN_ZIGZAG;
}

/* signBitHidingHDQ (COM/TComTrQuant.cpp:977-1100), per 16-coefficient group from the last one. */
static void sign_bit_hiding(int32_t *q, const int32_t *c, const uint32_t *scan, const int *deltaU,
                            int N) {
  int seen_last_group = 0; /* becomes 1 after the first (highest) group holding a non-zero */
  for (int g = (N * N - 1) >> 4; g >= 0; g--) {
    const uint32_t *s = scan + (g << 4);
    int first = 16, last = -1, sum = 0;
    for (int n = 0; n < 16; n++)
      if (q[s[n]]) {
        if (first == 16) first = n;
        last = n;
      }
    for (int n = first; n <= last; n++) sum += q[s[n]];
    int is_last_group = (last >= 0 && !seen_last_group);
    if (last - first >= 4) {
      int signbit = q[s[first]] > 0 ? 0 : 1;
      if (signbit != (sum & 1)) {
        int best_cost = 0x7fffffff, best_pos = -1, best_chg = 0;
        for (int n = is_last_group ? last : 15; n >= 0; n--) {
          int p = (int)s[n], cost = 0x7fffffff, chg = 0;
          /* every branch assigns a cost; a MAX_INT cost can never win the strict '<' below,
           * so the change value attached to it is never consumed */
          if (q[p] != 0) {
            if (deltaU[p] > 0) {
              cost = -deltaU[p];
              chg = 1;
            } else if (n == first && abs(q[p]) == 1) {
              cost = 0x7fffffff;
            } else {
              cost = deltaU[p];
              chg = -1;
            }
          } else if (n < first) {
            int this_sign = c[p] >= 0 ? 0 : 1;
            if (this_sign != signbit) {
              cost = 0x7fffffff;
            } else {
              cost = -deltaU[p];
              chg = 1;
            }
          } else {
            cost = -deltaU[p];
            chg = 1;
          }
          if (cost < best_cost) {
            best_cost = cost;
            best_chg = chg;
            best_pos = p;
          }
        }
        if (q[best_pos] == 32767 || q[best_pos] == -32768) best_chg = -1;
        if (c[best_pos] >= 0)
          q[best_pos] += best_chg;
        else
          q[best_pos] -= best_chg;
      }
    }
    if (is_last_group) seen_last_group = 1;
  }
}

/* Flat quantiser of xQuant (COM/TComTrQuant.cpp:1130-1267). */
/* qtab != NULL: the per-position table getQuantCoeff(list, rem, size) of a scaling list (COM/TComTrQuant.cpp:1215, 1244) -- an INPUT,
 * HM's setScalingList builds it (:2747-2771, 2826-2847, 2953-2977); NULL: the flat scale */
void hmo_xQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, const hmo_quant_cfg *cfg,
                       uint32_t *ac_sum, const int32_t *qtab) {
  int lg = ilog2(N), tshift = 15 - B - lg;
  int qbits = 14 + cfg->per_qbits + tshift;
  int64_t add = (int64_t)(cfg->intra_slice ? 171 : 85) << (qbits - 9);
  int q = hmo_quant_scale(cfg->rem);
  int deltaU[32 * 32];
  uint32_t sum = *ac_sum;
  for (int i = 0; i < N * N; i++) {
    int64_t t = (int64_t)abs(src[i]) * (qtab ? qtab[i] : q);
    int lvl = (int)((t + add) >> qbits);
    deltaU[i] = (int)((t - ((int64_t)lvl << qbits)) >> (qbits - 8));
    sum += (uint32_t)lvl;
    dst[i] = clip3(-32768, 32767, src[i] < 0 ? -lvl : lvl);
  }
  *ac_sum = sum;
  if (cfg->sign_hide && sum >= 2) sign_bit_hiding(dst, src, hmo_scan(cfg->scan_idx, lg), deltaU, N);
}
void hmo_xQuant(const int32_t *src, int32_t *dst, int N, int B, const hmo_quant_cfg *cfg, uint32_t *ac_sum) {
  hmo_xQuant_scaled(src, dst, N, B, cfg, ac_sum, NULL);
}

/* The pArlDes output of the quantiser (ADAPTIVE_QP_SELECTION; what TEncSlice's adaptive QP selection accumulates): the coefficient
 * scaled like a level but with ARL_C_PRECISION = 7 more fractional bits.  Flat path COM/TComTrQuant.cpp:1229-1249 (iQBits from the
 * slice's BASE QP, cQpBase); RDOQ path :1757, 1764-1765, 1886-1891 (iQBits from m_cQP, the product limited first). */
void hmo_arlCoeff(const int32_t *src, int32_t *arl, int N, int B, const hmo_quant_cfg *cfg, int rdoq, const int32_t *qtab) {
  const int lg = ilog2(N), tshift = 15 - B - lg;
  const int qbits = 14 + (rdoq ? cfg->per : cfg->per_qbits) + tshift, qbits_c = qbits - 7;
  const int q = hmo_quant_scale(cfg->rem);
  for (int i = 0; i < N * N; i++) {
    const int64_t t = (int64_t)abs(src[i]) * (qtab ? qtab[i] : q);
    if (rdoq) {
      const int64_t lim = (int64_t)2147483647 - ((int64_t)1 << (qbits - 1));
      const int ld = (int)(t < lim ? t : lim);
      arl[i] = (ld + (1 << (qbits_c - 1))) >> qbits_c;
    } else {
      arl[i] = (int)((t + ((int64_t)1 << (qbits_c - 1))) >> qbits_c);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Rate-distortion optimised quantisation (COM/TComTrQuant.cpp:1719-2305)
 *
 * Restated as three phases over per-scan-position records:
 *   A. reverse scan: candidate level per coefficient (the quantised value or one below, whichever
 *      costs less in D + lambda*R under the running c1/c2/Rice context state), per-group decision to
 *      zero a whole coefficient group;
 *   B. choice of the last significant position (scan backwards while levels are <= 1);
 *   C. sign-bit hiding with rate-aware costs.
 * All costs are doubles evaluated in the reference's operation order (no fused multiply-add).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const hmo_est_bits *e;
  double lambda;
} rdoq_rates;

static int rdoq_base_level(unsigned c1i, unsigned c2i) { return c1i < 8 ? (2 + (c2i < 1)) : 1; }

/* rate of coding |level| beyond the significance flag, as a double-valued COST (xGetICRateCost :2508) */
static double rdoq_level_cost(const rdoq_rates *r, unsigned lvl, unsigned ctx1, unsigned ctx2, unsigned rice, unsigned c1i,
                              unsigned c2i) {
  double rate = 32768; /* the sign: one equiprobable bin */
  unsigned base = (unsigned)rdoq_base_level(c1i, c2i);
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
      rate += r->e->greater1[ctx1][1];
      if (c2i < 1) rate += r->e->greater2[ctx2][1];
    }
  } else if (lvl == 1) {
    rate += r->e->greater1[ctx1][0];
  } else { /* lvl == 2 */
    rate += r->e->greater1[ctx1][1];
    rate += r->e->greater2[ctx2][0];
  }
  return r->lambda * rate;
}

/* integer rate of |level| (xGetICRate :2577): used only for the sign-hiding deltas */
static int rdoq_level_rate(const rdoq_rates *r, unsigned lvl, unsigned ctx1, unsigned ctx2, unsigned rice, unsigned c1i,
                           unsigned c2i) {
  static const unsigned range[5] = {7, 14, 26, 46, 78}, prefix[5] = {8, 7, 6, 5, 4};
  int rate = 0;
  unsigned base = (unsigned)rdoq_base_level(c1i, c2i);
  if (lvl >= base) {
    unsigned sym = lvl - base, max_vlc = range[rice];
    if (sym > max_vlc) {
      unsigned a = sym - max_vlc;
      int egs = 1;
      for (unsigned m = 2; a >= m; m <<= 1) egs += 2;
      rate += egs << 15;
      sym = sym < max_vlc + 1 ? sym : max_vlc + 1;
    }
    unsigned pre = (uint16_t)(sym >> rice) + 1;
    unsigned bins = (pre < prefix[rice] ? pre : prefix[rice]) + rice;
    rate += (int)((uint16_t)bins << 15);
    if (c1i < 8) {
      rate += r->e->greater1[ctx1][1];
      if (c2i < 1) rate += r->e->greater2[ctx2][1];
    }
  } else if (lvl == 0) {
    return 0;
  } else if (lvl == 1) {
    rate += r->e->greater1[ctx1][0];
  } else {
    rate += r->e->greater1[ctx1][1];
    rate += r->e->greater2[ctx2][0];
  }
  return rate;
}

/* significance context of a coefficient (getSigCtxInc :2349, REMOVAL_8x2_2x8_CG branch) */
static int rdoq_sig_ctx(int pattern, int scan_idx, int px, int py, int log2n, int is_luma) {
  static const int map4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
  if (px + py == 0) return 0;
  if (log2n == 2) return map4[4 * py + px];
  int offset = log2n == 3 ? (scan_idx == HMO_SCAN_DIAG ? 9 : 15) : (is_luma ? 21 : 12);
  int sx = px & 3, sy = py & 3, cnt;
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

static double rdoq_last_cost(const rdoq_rates *r, unsigned px, unsigned py) { /* xGetRateLast :2652 */
  static const unsigned grp[32] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
  unsigned cx = grp[px], cy = grp[py];
  double cost = r->e->last_x[cx] + r->e->last_y[cy];
  if (cx > 3) cost += 32768.0 * ((cx - 2) >> 1);
  if (cy > 3) cost += 32768.0 * ((cy - 2) >> 1);
  return r->lambda * cost;
}

/* qtab / estab != NULL: getQuantCoeff / getErrScaleCoeff of a scaling list per position (COM/TComTrQuant.cpp:1759-1762, 1882-1883) */
static void rdoq_tables(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg, const hmo_est_bits *est,
                        uint32_t *abs_sum, const int32_t *qtab, const double *estab);
void hmo_xRateDistOptQuant(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg,
                           const hmo_est_bits *est, uint32_t *abs_sum) {
  rdoq_tables(src, dst, N, B, cfg, est, abs_sum, NULL, NULL);
}
void hmo_xRateDistOptQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg, const hmo_est_bits *est,
                                  uint32_t *abs_sum, const int32_t *qtab, const double *estab) {
  rdoq_tables(src, dst, N, B, cfg, est, abs_sum, qtab, estab);
}
static void rdoq_tables(const int32_t *src, int32_t *dst, int N, int B, const hmo_rdoq_cfg *cfg, const hmo_est_bits *est,
                        uint32_t *abs_sum, const int32_t *qtab, const double *estab) {
  const int lg = ilog2(N), nn = N * N, G = N / 4, n_cg = nn >> 4;
  const int tshift = 15 - B - lg, qbits = 14 + cfg->per + tshift, inc = B - 8;
  const int q_flat = hmo_quant_scale(cfg->rem);
  const int scan_idx = cfg->scan_idx == HMO_SCAN_ZIGZAG ? HMO_SCAN_DIAG : cfg->scan_idx; /* :1770-1774 */
  const uint32_t *scan = hmo_scan(scan_idx, lg);
  const rdoq_rates R = {est, cfg->lambda};
  /* flat error scale (setErrScaleCoeff :2794-2818) */
  double err_flat = (double)(1 << 15);
  err_flat = err_flat * ldexp(1.0, -2 * tshift); /* pow(2.0, -2.0 * iTransformShift): an exact power of two */
  err_flat = err_flat / (double)q_flat / (double)q_flat / (double)(1 << (2 * inc));

  static double cost_coded[1024], cost_sig[1024], cost_zero[1024], cost_cg_sig[64];
  static int rate_up[1024], rate_down[1024], sig_delta[1024], delta_u[1024];
  unsigned cg_flag[64];
  memset(cost_coded, 0, sizeof(double) * nn);
  memset(cost_sig, 0, sizeof(double) * nn);
  memset(rate_up, 0, sizeof(int) * nn);
  memset(rate_down, 0, sizeof(int) * nn);
  memset(sig_delta, 0, sizeof(int) * nn);
  memset(delta_u, 0, sizeof(int) * nn);
  memset(cost_cg_sig, 0, sizeof(cost_cg_sig));
  memset(cg_flag, 0, sizeof(cg_flag));
  for (int i = 0; i < nn; i++) dst[i] = 0;

  double uncoded = 0, base = 0;
  int last_pos = -1, last_cg = -1;
  unsigned ctx_set = 0, rice = 0, c1i = 0, c2i = 0;
  int c1 = 1, c2 = 0;

  /* ---- phase A ---- */
  for (int cg = n_cg - 1; cg >= 0; cg--) {
    /* the group's position in the grid of groups: from its first scan entry */
    const unsigned p0 = scan[cg * 16], gx = (p0 & (N - 1)) >> 2, gy = (p0 >> lg) >> 2, gpos = gy * G + gx;
    double s_sig = 0, s_sig0 = 0, s_coded = 0, s_uncoded = 0;
    int nnz_before0 = 0;
    int pattern = -1; /* calcPatternSigCtx :2315 */
    if (N != 4) {
      unsigned right = gx < (unsigned)G - 1 ? cg_flag[gy * G + gx + 1] != 0 : 0;
      unsigned lower = gy < (unsigned)G - 1 ? cg_flag[(gy + 1) * G + gx] != 0 : 0;
      pattern = (int)(right + (lower << 1));
    }
    for (int k = 15; k >= 0; k--) {
      const int sp = cg * 16 + k;
      const unsigned bp = scan[sp];
      const int q = qtab ? qtab[bp] : q_flat;
      const double err_scale = estab ? estab[bp] : err_flat;
      int64_t wide = (int64_t)abs(src[bp]) * q;
      const int64_t cap = (int64_t)2147483647 - ((int64_t)1 << (qbits - 1));
      const int ld = (int)(wide < cap ? wide : cap); /* "level double": |c| * q */
      const unsigned max_lvl = (unsigned)((ld + (1 << (qbits - 1))) >> qbits);
      const double e0 = (double)ld;
      cost_zero[sp] = e0 * e0 * err_scale;
      uncoded += cost_zero[sp];
      dst[bp] = (int32_t)max_lvl;
      if (max_lvl > 0 && last_pos < 0) {
        last_pos = sp;
        ctx_set = (sp < 16 || !cfg->is_luma) ? 0 : 2;
        last_cg = cg;
      }
      if (last_pos >= 0) {
        const unsigned ctx1 = 4 * ctx_set + (unsigned)c1, ctx2 = ctx_set + (unsigned)c2;
        const int is_last = sp == last_pos;
        unsigned ctx_sig = 0;
        if (!is_last) ctx_sig = (unsigned)rdoq_sig_ctx(pattern, scan_idx, (int)(bp & (N - 1)), (int)(bp >> lg), lg, cfg->is_luma);
        /* best level among {max_lvl, max_lvl - 1 (>= 1)} and, when allowed, zero (xGetCodedLevel :2446) */
        unsigned best = 0;
        double sig1 = 0;
        int decided = 0;
        if (!is_last && max_lvl < 3) {
          cost_sig[sp] = R.lambda * est->sig[ctx_sig][0];
          cost_coded[sp] = cost_zero[sp] + cost_sig[sp];
          if (max_lvl == 0) decided = 1;
        } else {
          cost_coded[sp] = 1.7e+308;
        }
        if (!decided) {
          if (!is_last) sig1 = R.lambda * est->sig[ctx_sig][1];
          const unsigned lo = max_lvl > 1 ? max_lvl - 1 : 1;
          for (int l = (int)max_lvl; l >= (int)lo; l--) {
            const double d = (double)(ld - (l << qbits));
            double cst = d * d * err_scale + rdoq_level_cost(&R, (unsigned)l, ctx1, ctx2, rice, c1i, c2i);
            cst += sig1;
            if (cst < cost_coded[sp]) {
              best = (unsigned)l;
              cost_coded[sp] = cst;
              cost_sig[sp] = sig1;
            }
          }
        }
        if (!is_last) sig_delta[bp] = est->sig[ctx_sig][1] - est->sig[ctx_sig][0];
        delta_u[bp] = (ld - ((int)best << qbits)) >> (qbits - 8);
        if (best > 0) {
          const int now = rdoq_level_rate(&R, best, ctx1, ctx2, rice, c1i, c2i);
          rate_up[bp] = rdoq_level_rate(&R, best + 1, ctx1, ctx2, rice, c1i, c2i) - now;
          rate_down[bp] = rdoq_level_rate(&R, best - 1, ctx1, ctx2, rice, c1i, c2i) - now;
        } else {
          rate_up[bp] = est->greater1[ctx1][0];
        }
        dst[bp] = (int32_t)best;
        base += cost_coded[sp];
        /* context state for the next (lower) scan position (:1952-2003) */
        if (best >= (unsigned)rdoq_base_level(c1i, c2i) && best > 3u * (1u << rice)) rice = rice + 1 < 4 ? rice + 1 : 4;
        if (best >= 1) c1i++;
        if (best > 1) {
          c1 = 0;
          c2 += (c2 < 2);
          c2i++;
        } else if (c1 < 3 && c1 > 0 && best) {
          c1++;
        }
        if ((sp % 16 == 0) && sp > 0) {
          c2 = 0;
          rice = 0;
          c1i = 0;
          c2i = 0;
          ctx_set = (sp == 16 || !cfg->is_luma) ? 0 : 2;
          if (c1 == 0) ctx_set++;
          c1 = 1;
        }
      } else {
        base += cost_zero[sp];
      }
      s_sig += cost_sig[sp];
      if (k == 0) s_sig0 = cost_sig[sp];
      if (dst[bp]) {
        cg_flag[gpos] = 1;
        s_coded += cost_coded[sp] - cost_sig[sp];
        s_uncoded += cost_zero[sp];
        if (k != 0) nnz_before0++;
      }
    }
    /* whole-group decision (:2022-2086) */
    if (last_cg >= 0) {
      if (cg) {
        unsigned right = gx < (unsigned)G - 1 ? cg_flag[gy * G + gx + 1] != 0 : 0;
        unsigned lower = gy < (unsigned)G - 1 ? cg_flag[(gy + 1) * G + gx] != 0 : 0;
        const unsigned cctx = right || lower; /* getSigCoeffGroupCtxInc :2707 */
        if (cg_flag[gpos] == 0) {
          base += R.lambda * est->sig_cg[cctx][0] - s_sig;
          cost_cg_sig[cg] = R.lambda * est->sig_cg[cctx][0];
        } else if (cg < last_cg) {
          if (nnz_before0 == 0) {
            base -= s_sig0;
            s_sig -= s_sig0;
          }
          double zero_cost = base;
          base += R.lambda * est->sig_cg[cctx][1];
          zero_cost += R.lambda * est->sig_cg[cctx][0];
          cost_cg_sig[cg] = R.lambda * est->sig_cg[cctx][1];
          zero_cost += s_uncoded;
          zero_cost -= s_coded;
          zero_cost -= s_sig;
          if (zero_cost < base) {
            cg_flag[gpos] = 0;
            base = zero_cost;
            cost_cg_sig[cg] = R.lambda * est->sig_cg[cctx][0];
            for (int k = 15; k >= 0; k--) {
              const int sp = cg * 16 + k;
              const unsigned bp = scan[sp];
              if (dst[bp]) {
                dst[bp] = 0;
                cost_coded[sp] = cost_zero[sp];
                cost_sig[sp] = 0;
              }
            }
          }
        }
      } else {
        cg_flag[gpos] = 1;
      }
    }
  }
  if (last_pos < 0) return;

  /* ---- phase B: last position ---- */
  double best_cost;
  if (cfg->root_cbf) {
    best_cost = uncoded + R.lambda * est->root_cbf[0][0];
    base += R.lambda * est->root_cbf[0][1];
  } else {
    best_cost = uncoded + R.lambda * est->cbf[cfg->cbf_ctx][0];
    base += R.lambda * est->cbf[cfg->cbf_ctx][1];
  }
  int best_last_p1 = 0, found = 0;
  for (int cg = last_cg; cg >= 0 && !found; cg--) {
    const unsigned p0 = scan[cg * 16], gpos = ((p0 >> lg) >> 2) * G + ((p0 & (N - 1)) >> 2);
    base -= cost_cg_sig[cg];
    if (!cg_flag[gpos]) continue;
    for (int k = 15; k >= 0; k--) {
      const int sp = cg * 16 + k;
      if (sp > last_pos) continue;
      const unsigned bp = scan[sp];
      if (dst[bp]) {
        const unsigned py = bp >> lg, px = bp & (N - 1);
        const double lc = scan_idx == HMO_SCAN_VER ? rdoq_last_cost(&R, py, px) : rdoq_last_cost(&R, px, py);
        const double total = base + lc - cost_sig[sp];
        if (total < best_cost) {
          best_last_p1 = sp + 1;
          best_cost = total;
        }
        if (dst[bp] > 1) {
          found = 1;
          break;
        }
        base -= cost_coded[sp];
        base += cost_zero[sp];
      } else {
        base -= cost_sig[sp];
      }
    }
  }
  for (int sp = 0; sp < best_last_p1; sp++) {
    const unsigned bp = scan[sp];
    const int l = dst[bp];
    *abs_sum += (uint32_t)l;
    dst[bp] = src[bp] < 0 ? -l : l;
  }
  for (int sp = best_last_p1; sp <= last_pos; sp++) dst[scan[sp]] = 0;

  /* ---- phase C: sign-bit hiding with rate-aware costs (:2203-2304) ---- */
  if (!(cfg->sign_hide && *abs_sum >= 2)) return;
  const int invq = hmo_inv_quant_scale(cfg->rem);
  const int64_t rd_factor =
      (int64_t)((double)invq * (double)invq * (double)(1 << (2 * cfg->per)) / cfg->lambda / 16 / (double)(1 << (2 * inc)) + 0.5);
  int seen_last = -1;
  for (int sub = (nn - 1) >> 4; sub >= 0; sub--) {
    const int o = sub << 4;
    int first = 16, lastnz = -1, sum = 0;
    for (int n = 15; n >= 0; n--)
      if (dst[scan[n + o]]) {
        lastnz = n;
        break;
      }
    for (int n = 0; n < 16; n++)
      if (dst[scan[n + o]]) {
        first = n;
        break;
      }
    for (int n = first; n <= lastnz; n++) sum += dst[scan[n + o]];
    if (lastnz >= 0 && seen_last == -1) seen_last = 1;
    if (lastnz - first >= 4) {
      const unsigned signbit = dst[scan[o + first]] > 0 ? 0 : 1;
      if (signbit != (unsigned)(sum & 1)) {
        int64_t min_cost = INT64_MAX, cur = INT64_MAX;
        int min_pos = -1, final_change = 0, change = 0;
        for (int n = (seen_last == 1 ? lastnz : 15); n >= 0; n--) {
          const unsigned bp = scan[n + o];
          if (dst[bp] != 0) {
            const int64_t up = rd_factor * (-delta_u[bp]) + rate_up[bp];
            int64_t down = rd_factor * (delta_u[bp]) + rate_down[bp] - (abs(dst[bp]) == 1 ? ((1 << 15) + sig_delta[bp]) : 0);
            if (seen_last == 1 && lastnz == n && abs(dst[bp]) == 1) down -= (4 << 15);
            if (up < down) {
              cur = up;
              change = 1;
            } else {
              change = -1;
              cur = (n == first && abs(dst[bp]) == 1) ? INT64_MAX : down;
            }
          } else {
            cur = rd_factor * (-(abs(delta_u[bp]))) + (1 << 15) + rate_up[bp] + sig_delta[bp];
            change = 1;
            if (n < first) {
              const unsigned s = src[bp] >= 0 ? 0 : 1;
              if (s != signbit) cur = INT64_MAX;
            }
          }
          if (cur < min_cost) {
            min_cost = cur;
            final_change = change;
            min_pos = (int)bp;
          }
        }
        /* the reference tests the flat quantiser coefficient (never +-32768) here (:2290): no effect */
        if (src[min_pos] >= 0)
          dst[min_pos] += final_change;
        else
          dst[min_pos] -= final_change;
      }
    }
    if (seen_last == 1) seen_last = 0;
  }
}

/* Flat de-quantiser (COM/TComTrQuant.cpp:1343-1354). */
void hmo_xDeQuant(const int32_t *src, int32_t *dst, int N, int B, int per, int rem) {
  int shift = 20 - 14 - (15 - B - ilog2(N));
  int add = 1 << (shift - 1), scale = hmo_inv_quant_scale(rem) << per;
  for (int i = 0; i < N * N; i++) {
    /* the reference multiplies in 32-bit Int; keep two's-complement wrap for extreme QPs */
    int32_t l = clip3(-32768, 32767, src[i]);
    int32_t v = (int32_t)((uint32_t)l * (uint32_t)scale + (uint32_t)add);
    dst[i] = clip3(-32768, 32767, v >> shift);
  }
}

/* xDeQuant with a scaling list (COM/TComTrQuant.cpp:1311-1342): coef = the per-position table HM's setScalingListDec built for
 * (list type, QP remainder, size) -- scaling-list entry times g_invQuantScales[rem] (:2860-2861, 2979-3003).  The table is an INPUT:
 * building it from the slice's lists is header handling that stays in HM. */
void hmo_xDeQuant_scaled(const int32_t *src, int32_t *dst, int N, int B, int per, const int32_t *coef) {
  const int lg = ilog2(N), shift = 20 - 14 - (15 - B - lg) + 4;
  const int bit_range = 12 + lg + B - per < 15 ? 12 + lg + B - per : 15, limit = 1 << bit_range;
  for (int i = 0; i < N * N; i++) {
    if (shift > per) {
      const int32_t l = clip3(-32768, 32767, src[i]);
      const int32_t v = (int32_t)((uint32_t)l * (uint32_t)coef[i] + (uint32_t)(1 << (shift - per - 1))); /* 32-bit Int in the reference */
      dst[i] = clip3(-32768, 32767, v >> (shift - per));
    } else {
      const int32_t l = clip3(-limit, limit - 1, src[i]);
      const int32_t v = (int32_t)(((uint32_t)l * (uint32_t)coef[i]) << (per - shift));
      dst[i] = clip3(-32768, 32767, v);
    }
  }
}

void hmo_transformNxN(const int16_t *resi, int stride, int32_t *level, int N, int B, unsigned mode,
                      int transform_skip, int bypass, const hmo_quant_cfg *cfg, uint32_t *abs_sum) {
  int32_t tmp[32 * 32];
  *abs_sum = 0;
  if (bypass) { /* :1388-1399 */
    for (int r = 0; r < N; r++)
      for (int c = 0; c < N; c++) {
        level[r * N + c] = resi[r * stride + c];
        *abs_sum += (uint32_t)abs(resi[r * stride + c]);
      }
    return;
  }
  if (transform_skip)
    hmo_xTransformSkip(resi, stride, tmp, N, B);
  else
    hmo_xT(mode, resi, stride, tmp, N, B);
  hmo_xQuant(tmp, level, N, B, cfg, abs_sum);
}

void hmo_invtransformNxN(int bypass, unsigned mode, int16_t *resi, int stride, const int32_t *level,
                         int N, int B, int per, int rem, int transform_skip) {
  int32_t tmp[32 * 32];
  if (bypass) { /* :1430-1440 */
    for (int r = 0; r < N; r++)
      for (int c = 0; c < N; c++) resi[r * stride + c] = (int16_t)(uint16_t)level[r * N + c];
    return;
  }
  hmo_xDeQuant(level, tmp, N, B, per, rem);
  if (transform_skip)
    hmo_xITransformSkip(tmp, resi, stride, N, B);
  else
    hmo_xIT(mode, tmp, resi, stride, N, B);
}

/* ------------------------------------------------------------------------------------------
 * Distortion.  The sum of |coefficients| of a 2-D Hadamard transform does not depend on the order or
 * the signs of the transform's rows, so the butterflies below are the plain Walsh-Hadamard ones.
 * ---------------------------------------------------------------------------------------- */
static void wht(int *v, int n, int stride) {
  for (int half = 1; half < n; half <<= 1)
    for (int i = 0; i < n; i += 2 * half)
      for (int j = i; j < i + half; j++) {
        int a = v[j * stride], b = v[(j + half) * stride];
        v[j * stride] = a + b;
        v[(j + half) * stride] = a - b;
      }
}
static uint32_t had_block(const int16_t *org, int so, const int16_t *cur, int sc, int n) {
  int d[64], sum = 0;
  for (int r = 0; r < n; r++)
    for (int k = 0; k < n; k++) d[r * n + k] = org[r * so + k] - cur[r * sc + k];
  for (int r = 0; r < n; r++) wht(d + r * n, n, 1);
  for (int k = 0; k < n; k++) wht(d + k, n, n);
  for (int i = 0; i < n * n; i++) sum += abs(d[i]);
  return (uint32_t)(n == 8 ? (sum + 2) >> 2 : (sum + 1) >> 1); /* :1865, :1747 */
}
uint32_t hmo_calcHAD(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int B) {
  const int n = (w % 8 == 0 && h % 8 == 0) ? 8 : 4;
  uint32_t sum = 0;
  for (int y = 0; y < h; y += n)
    for (int x = 0; x < w; x += n) sum += had_block(org + y * so + x, so, cur + y * sc + x, sc, n);
  return sum >> (B - 8);
}
uint32_t hmo_getSSE(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int B) {
  /* the variant compiled with IBDI_DISTORTION 0 (COM/TComRdCost.cpp:1313-1657): per-sample square, then >> 2*inc */
  const unsigned shift = (unsigned)(B - 8) << 1;
  uint32_t sum = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int t = org[y * so + x] - cur[y * sc + x];
      sum += (uint32_t)((t * t) >> shift);
    }
  return sum;
}

/* ------------------------------------------------------------------------------------------
 * Intra reference samples
 * ---------------------------------------------------------------------------------------- */
static inline unsigned zorder(unsigned cx, unsigned cy) { /* g_auiRasterToZscan for a unit */
  unsigned z = 0;
  for (int b = 0; b < 8; b++) z |= ((cx >> b) & 1u) << (2 * b) | ((cy >> b) & 1u) << (2 * b + 1);
  return z;
}

int hmo_intra_avail(int x, int y, int size, int pic_w, int pic_h, int ctu, uint8_t *f) {
  int n = size / 4, U = ctu / 4;
  int cx = (x % ctu) / 4, cy = (y % ctu) / 4;
  int ctu_col = x / ctu, ctu_cols = (pic_w + ctu - 1) / ctu;
  int cnt = 0;
  /* corner, above, left: inside the picture <=> already coded (COM/TComDataCU.cpp:1221-1432) */
  f[2 * n] = (uint8_t)(x > 0 && y > 0);
  for (int i = 0; i < n; i++) f[2 * n + 1 + i] = (uint8_t)(y > 0);
  for (int i = 0; i < n; i++) f[2 * n - 1 - i] = (uint8_t)(x > 0);
  /* above-right (getPUAboveRightAdi, :1664-1735): Z-order decides inside the CTU */
  int rx = cx + n - 1;
  for (int o = 1; o <= n; o++) {
    int a;
    if (x + size - 4 + 4 * o >= pic_w)
      a = 0;
    else if (rx + o < U)
      a = cy > 0 ? (zorder(rx, cy) > zorder(rx + o, cy - 1)) : (y > 0);
    else
      a = (cy == 0) && y > 0 && ctu_col < ctu_cols - 1;
    f[3 * n + o] = (uint8_t)a;
  }
  /* below-left (getPUBelowLeftAdi, :1597-1662) */
  int by = cy + n - 1;
  for (int o = 1; o <= n; o++) {
    int a;
    if (y + size - 4 + 4 * o >= pic_h)
      a = 0;
    else if (by + o < U)
      a = cx > 0 ? (zorder(cx, by) > zorder(cx - 1, by + o)) : (x > 0);
    else
      a = 0;
    f[n - o] = (uint8_t)a;
  }
  for (int i = 0; i < 4 * n + 1; i++) cnt += f[i];
  return cnt;
}

/* One linear line L[0..4N]: L[0] = lowest below-left sample .. L[2N-1] = top-most left sample,
 * L[2N] = corner, L[2N+1..4N] = above and above-right, left to right. */
static inline int16_t line_sample(const int16_t *rec, int stride, int N, int p) {
  if (p < 2 * N) return rec[(2 * N - 1 - p) * stride - 1];
  if (p == 2 * N) return rec[-stride - 1];
  return rec[-stride + (p - 2 * N - 1)];
}

static void build_ref_line(const int16_t *rec, int stride, const uint8_t *flags, int n_avail,
                           int unit, int N, int B, int32_t *L) {
  int n = N / unit, units = 4 * n + 1, len = 4 * N + 1;
  int dc = 1 << (B - 1);
  if (n_avail == 0) { /* COM/TComPattern.cpp:374-385 */
    for (int p = 0; p < len; p++) L[p] = dc;
    return;
  }
  /* unit u covers samples [lo(u), hi(u)] of the line */
#define U_LO(u) ((u) < 2 * n ? (u) * unit : ((u) == 2 * n ? 2 * N : 2 * N + 1 + ((u)-2 * n - 1) * unit))
#define U_HI(u) ((u) < 2 * n ? (u) * unit + unit - 1 : ((u) == 2 * n ? 2 * N : 2 * N + ((u)-2 * n) * unit))
  for (int u = 0; u < units; u++)
    if (flags[u])
      for (int p = U_LO(u); p <= U_HI(u); p++) L[p] = line_sample(rec, stride, N, p);
  if (n_avail == units) return;
  /* padding (:512-547): a leading unavailable run copies the first sample of the first
   * available unit; every later unavailable unit copies the sample just before it. */
  int u = 0;
  if (!flags[0]) {
    int nx = 1;
    while (nx < units && !flags[nx]) nx++;
    int v = L[U_LO(nx)];
    for (; u < nx; u++)
      for (int p = U_LO(u); p <= U_HI(u); p++) L[p] = v;
  }
  for (; u < units; u++)
    if (!flags[u]) {
      int v = L[U_LO(u) - 1];
      for (int p = U_LO(u); p <= U_HI(u); p++) L[p] = v;
    }
#undef U_LO
#undef U_HI
}

static void line_to_adi(const int32_t *L, int32_t *adi, int N) {
  int W = 2 * N + 1;
  for (int i = 0; i <= 2 * N; i++) adi[i] = L[2 * N + i];
  for (int i = 1; i <= 2 * N; i++) adi[i * W] = L[2 * N - i];
}
static void adi_to_line(const int32_t *adi, int32_t *L, int N) {
  int W = 2 * N + 1;
  for (int i = 0; i <= 2 * N; i++) L[2 * N + i] = adi[i];
  for (int i = 1; i <= 2 * N; i++) L[2 * N - i] = adi[i * W];
}

void hmo_fillReferenceSamples(const int16_t *rec, int stride, const uint8_t *flags, int n_avail,
                              int unit, int N, int B, int32_t *adi) {
  int32_t L[4 * MAXN + 1];
  build_ref_line(rec, stride, flags, n_avail, unit, N, B, L);
  line_to_adi(L, adi, N);
}

void hmo_filterAdi(int32_t *adi, int N) { /* COM/TComPattern.cpp:265-306 */
  int32_t L[4 * MAXN + 1], F[4 * MAXN + 1];
  int len = 4 * N + 1;
  adi_to_line(adi, L, N);
  F[0] = L[0];
  F[len - 1] = L[len - 1];
  for (int i = 1; i < len - 1; i++) F[i] = (L[i - 1] + 2 * L[i] + L[i + 1] + 2) >> 2;
  line_to_adi(F, adi + (2 * N + 1) * (2 * N + 1), N);
}

int hmo_use_filtered_refs(int mode, int log2n) { /* COM/TComPattern.cpp:49-56, 577-605 */
  static const int thr[5] = {10, 7, 1, 0, 10};
  if (mode == 1) return 0;
  int dh = abs(mode - 10), dv = abs(mode - 26);
  return (dh < dv ? dh : dv) > thr[log2n - 2];
}

/* ------------------------------------------------------------------------------------------
 * Intra prediction.  src points at buffer cell (1,1) like the reference (ptrSrc+sw+1), so
 * top(k) = src[k - stride - 1] (k=0 is the corner) and left(k) = src[(k-1)*stride - 1].
 * ---------------------------------------------------------------------------------------- */
int16_t hmo_predIntraGetPredValDC(const int32_t *src, int ss, int N, int above, int left) {
  int sum = 0;
  if (above)
    for (int i = 0; i < N; i++) sum += src[i - ss];
  if (left)
    for (int i = 0; i < N; i++) sum += src[i * ss - 1];
  if (above && left) return (int16_t)((sum + N) / (2 * N));
  if (above || left) return (int16_t)((sum + N / 2) / N);
  return (int16_t)src[-1]; /* the default value the reference samples were filled with */
}

void hmo_xPredIntraAng(const int32_t *src, int ss, int16_t *dst, int ds, int N, int mode,
                       int filter_edge, int B) {
  static const int ang_tab[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
  static const int inv_tab[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
  int maxv = (1 << B) - 1;
  if (mode < 2) { /* DC (both neighbours flagged available by initAdiPattern, :247-248) */
    int sum = 0;
    for (int i = 0; i < N; i++) sum += src[i - ss] + src[i * ss - 1];
    int16_t dc = (int16_t)((sum + N) / (2 * N));
    for (int r = 0; r < N; r++)
      for (int c = 0; c < N; c++) dst[r * ds + c] = dc;
    return;
  }
  int ver = mode >= 18;
  int idx = ver ? mode - 26 : -(mode - 10);
  int angle = (idx < 0 ? -1 : 1) * ang_tab[abs(idx)], inv_angle = inv_tab[abs(idx)];
  int16_t main_buf[3 * MAXN + 2], side_buf[3 * MAXN + 2];
  int16_t *rm = main_buf + MAXN, *rs = side_buf + MAXN; /* index 0 = corner */
  int cnt = angle < 0 ? N : 2 * N;
  for (int k = 0; k <= cnt; k++) {
    int16_t a = (int16_t)src[k - ss - 1], l = (int16_t)src[(k - 1) * ss - 1];
    rm[k] = ver ? a : l;
    rs[k] = ver ? l : a;
  }
  if (angle < 0) { /* project the side reference onto the main one (:235-255) */
    int acc = 128, lim = (N * angle) >> 5;
    for (int k = -1; k > lim; k--) {
      acc += inv_angle;
      rm[k] = rs[acc >> 8];
    }
  }
  int16_t P[MAXN][MAXN];
  if (angle == 0) {
    for (int k = 0; k < N; k++)
      for (int l = 0; l < N; l++) P[k][l] = rm[l + 1];
    if (filter_edge)
      for (int k = 0; k < N; k++) P[k][0] = (int16_t)clip3(0, maxv, P[k][0] + ((rs[k + 1] - rs[0]) >> 1));
  } else {
    int pos = 0;
    for (int k = 0; k < N; k++) {
      pos += angle;
      int di = pos >> 5, df = pos & 31;
      for (int l = 0; l < N; l++) {
        int i = l + di + 1;
        P[k][l] = df ? (int16_t)(((32 - df) * rm[i] + df * rm[i + 1] + 16) >> 5) : rm[i];
      }
    }
  }
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) dst[r * ds + c] = ver ? P[r][c] : P[c][r];
}

void hmo_xPredIntraPlanar(const int32_t *src, int ss, int16_t *dst, int ds, int N) {
  int lg = ilog2(N); /* closed form of the accumulators at :689-730 */
  int tr = src[N - ss], bl = src[N * ss - 1];
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) {
      int left = src[r * ss - 1], top = src[c - ss];
      int hor = (left << lg) + N + (c + 1) * (tr - left);
      int vert = (top << lg) + (r + 1) * (bl - top);
      dst[r * ds + c] = (int16_t)((hor + vert) >> (lg + 1));
    }
}

void hmo_xDCPredFiltering(const int32_t *src, int ss, int16_t *dst, int ds, int N) {
  dst[0] = (int16_t)((src[-ss] + src[-1] + 2 * dst[0] + 2) >> 2);
  for (int c = 1; c < N; c++) dst[c] = (int16_t)((src[c - ss] + 3 * dst[c] + 2) >> 2);
  for (int r = 1; r < N; r++) dst[r * ds] = (int16_t)((src[r * ss - 1] + 3 * dst[r * ds] + 2) >> 2);
}

void hmo_predIntraLumaAng(const int32_t *adi, int mode, int16_t *dst, int ds, int N, int B) {
  int W = 2 * N + 1;
  const int32_t *p = adi + (hmo_use_filtered_refs(mode, ilog2(N)) ? W * W : 0) + W + 1;
  if (mode == 0)
    hmo_xPredIntraPlanar(p, W, dst, ds, N);
  else {
    hmo_xPredIntraAng(p, W, dst, ds, N, mode, 1, B);
    if (mode == 1) hmo_xDCPredFiltering(p, W, dst, ds, N); /* any size: :361-364 */
  }
}

void hmo_predIntraChromaAng(const int32_t *adi, int mode, int16_t *dst, int ds, int N, int B) {
  int W = 2 * N + 1;
  if (mode == 0)
    hmo_xPredIntraPlanar(adi + W + 1, W, dst, ds, N);
  else
    hmo_xPredIntraAng(adi + W + 1, W, dst, ds, N, mode, 0, B);
}

/* ------------------------------------------------------------------------------------------
 * Interpolation (COM/TComInterpolationFilter.cpp)
 * ---------------------------------------------------------------------------------------- */
static const int16_t kLuma[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0},
                                    {-1, 4, -10, 58, 17, -5, 1, 0},
                                    {-1, 4, -11, 40, 40, -11, 4, -1},
                                    {0, 1, -5, 17, 58, -10, 4, -1}};
static const int16_t kChroma[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2},
                                      {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6},
                                      {-2, 16, 54, -4}, {-2, 10, 58, -2}};

static void filter_copy(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, int first,
                        int last, int B) { /* :91-146 */
  int sh = 14 - B, maxv = (1 << B) - 1;
  for (int r = 0; r < h; r++, s += ss, d += ds)
    for (int c = 0; c < w; c++) {
      if (first == last)
        d[c] = s[c];
      else if (first)
        d[c] = (int16_t)((int16_t)(uint16_t)(s[c] << sh) - 8192);
      else {
        int16_t off = (int16_t)(8192 + (sh ? (1 << (sh - 1)) : 0));
        int16_t v = (int16_t)(uint16_t)((s[c] + off) >> sh);
        d[c] = (int16_t)clip3(0, maxv, v);
      }
    }
}

static void fir(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, const int16_t *tap,
                int ntap, int vertical, int first, int last, int B) { /* :163-244 */
  int step = vertical ? ss : 1, head = 14 - B, shift = 6, offset, maxv = (1 << B) - 1;
  if (last) {
    shift += first ? 0 : head;
    offset = (1 << (shift - 1)) + (first ? 0 : 8192 << 6);
  } else {
    shift -= first ? head : 0;
    offset = first ? -(8192 << shift) : 0;
  }
  s -= (ntap / 2 - 1) * step;
  for (int r = 0; r < h; r++, s += ss, d += ds)
    for (int c = 0; c < w; c++) {
      int sum = 0;
      for (int t = 0; t < ntap; t++) sum += s[c + t * step] * tap[t];
      int16_t v = (int16_t)(uint16_t)((sum + offset) >> shift); /* narrowed BEFORE the clip */
      d[c] = last ? (int16_t)clip3(0, maxv, v) : v;
    }
}

void hmo_filterHorLuma(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, int frac,
                       int last, int B) {
  if (!frac)
    filter_copy(s, ss, d, ds, w, h, 1, last, B);
  else
    fir(s, ss, d, ds, w, h, kLuma[frac], 8, 0, 1, last, B);
}
void hmo_filterVerLuma(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, int frac,
                       int first, int last, int B) {
  if (!frac)
    filter_copy(s, ss, d, ds, w, h, first, last, B);
  else
    fir(s, ss, d, ds, w, h, kLuma[frac], 8, 1, first, last, B);
}
void hmo_filterHorChroma(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, int frac,
                         int last, int B) {
  if (!frac)
    filter_copy(s, ss, d, ds, w, h, 1, last, B);
  else
    fir(s, ss, d, ds, w, h, kChroma[frac], 4, 0, 1, last, B);
}
void hmo_filterVerChroma(const int16_t *s, int ss, int16_t *d, int ds, int w, int h, int frac,
                         int first, int last, int B) {
  if (!frac)
    filter_copy(s, ss, d, ds, w, h, first, last, B);
  else
    fir(s, ss, d, ds, w, h, kChroma[frac], 4, 1, first, last, B);
}

/* COM/TComPrediction.cpp:554-585 */
void hmo_predInterLumaBlk(const int16_t *ref, int rs, int mvx, int mvy, int w, int h, int16_t *dst,
                          int ds, int bi, int B) {
  ref += (mvx >> 2) + (mvy >> 2) * rs;
  int xf = mvx & 3, yf = mvy & 3;
  if (yf == 0)
    hmo_filterHorLuma(ref, rs, dst, ds, w, h, xf, !bi, B);
  else if (xf == 0)
    hmo_filterVerLuma(ref, rs, dst, ds, w, h, yf, 1, !bi, B);
  else {
    int16_t tmp[(64 + 7) * 64];
    hmo_filterHorLuma(ref - 3 * rs, rs, tmp, 64, w, h + 7, xf, 0, B);
    hmo_filterVerLuma(tmp + 3 * 64, 64, dst, ds, w, h, yf, 0, !bi, B);
  }
}

/* COM/TComPrediction.cpp:599-642; one chroma plane per call, w/h are the LUMA block size */
void hmo_predInterChromaBlk(const int16_t *ref, int rs, int mvx, int mvy, int w, int h,
                            int16_t *dst, int ds, int bi, int B) {
  ref += (mvx >> 3) + (mvy >> 3) * rs;
  int xf = mvx & 7, yf = mvy & 7, cw = w >> 1, ch = h >> 1;
  if (yf == 0)
    hmo_filterHorChroma(ref, rs, dst, ds, cw, ch, xf, !bi, B);
  else if (xf == 0)
    hmo_filterVerChroma(ref, rs, dst, ds, cw, ch, yf, 1, !bi, B);
  else {
    int16_t tmp[(32 + 3) * 32];
    hmo_filterHorChroma(ref - rs, rs, tmp, 32, cw, ch + 3, xf, 0, B);
    hmo_filterVerChroma(tmp + 32, 32, dst, ds, cw, ch, yf, 0, !bi, B);
  }
}

void hmo_addAvg(const int16_t *s0, int s0s, const int16_t *s1, int s1s, int16_t *dst, int ds, int w,
                int h, int B) { /* COM/TComYuv.cpp:520-581 */
  int sh = 15 - B, off = (1 << (sh - 1)) + 2 * 8192, maxv = (1 << B) - 1;
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++)
      dst[r * ds + c] = (int16_t)clip3(0, maxv, (s0[r * s0s + c] + s1[r * s1s + c] + off) >> sh);
}

void hmo_addClip(const int16_t *pred, int ps, const int16_t *resi, int rs, int16_t *dst, int ds,
                 int w, int h, int B) { /* COM/TComYuv.cpp:401-459 */
  int maxv = (1 << B) - 1;
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++)
      dst[r * ds + c] = (int16_t)clip3(0, maxv, pred[r * ps + c] + resi[r * rs + c]);
}

void hmo_subtract(const int16_t *org, int os, const int16_t *pred, int ps, int16_t *dst, int ds,
                  int w, int h) { /* COM/TComYuv.cpp:461-518 */
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) dst[r * ds + c] = (int16_t)(org[r * os + c] - pred[r * ps + c]);
}

void hmo_extendPicBorder(int16_t *org, int stride, int w, int h, int mx, int my) {
  for (int y = 0; y < h; y++) { /* COM/TComPicYuv.cpp:259-286 */
    int16_t *row = org + y * stride;
    for (int x = 1; x <= mx; x++) {
      row[-x] = row[0];
      row[w - 1 + x] = row[w - 1];
    }
  }
  for (int y = 1; y <= my; y++) {
    memcpy(org - y * stride - mx, org - mx, sizeof(int16_t) * (w + 2 * mx));
    memcpy(org + (h - 1 + y) * stride - mx, org + (h - 1) * stride - mx, sizeof(int16_t) * (w + 2 * mx));
  }
}

void hmo_clipMv(int *mvx, int *mvy, int cu_x, int cu_y, int pic_w, int pic_h, int ctu) {
  int hmax = (pic_w + 8 - cu_x - 1) << 2, hmin = (-ctu - 8 - cu_x + 1) << 2; /* TComDataCU.cpp:3505 */
  int vmax = (pic_h + 8 - cu_y - 1) << 2, vmin = (-ctu - 8 - cu_y + 1) << 2;
  *mvx = clip3(hmin, hmax, *mvx);
  *mvy = clip3(vmin, vmax, *mvy);
}

/* ------------------------------------------------------------------------------------------
 * Frame drivers
 * ---------------------------------------------------------------------------------------- */
static void tu_predict(const hmo_frame_cfg *cfg, const hmo_tu *t, int16_t *rec, int rs,
                       int16_t *pred /* N*N dense */) {
  int N = 1 << t->log2n, chroma = t->plane != 0;
  uint8_t flags[4 * 16 + 1];
  int32_t adi[2 * (2 * MAXN + 1) * (2 * MAXN + 1)];
  int lx = t->x << chroma, ly = t->y << chroma, lsize = N << chroma;
  int navail = hmo_intra_avail(lx, ly, lsize, cfg->pic_w, cfg->pic_h, cfg->ctu, flags);
  hmo_fillReferenceSamples(rec + t->y * rs + t->x, rs, flags, navail, chroma ? 2 : 4, N, cfg->B, adi);
  if (!chroma) {
    hmo_filterAdi(adi, N);
    hmo_predIntraLumaAng(adi, t->mode, pred, N, N, cfg->B);
  } else
    hmo_predIntraChromaAng(adi, t->mode, pred, N, N, cfg->B);
}

static void tu_quant_cfg(const hmo_frame_cfg *cfg, const hmo_tu *t, hmo_qp *qp, hmo_quant_cfg *qc) {
  int chroma = t->plane != 0, N = 1 << t->log2n;
  *qp = hmo_setQPforQuant(cfg->qp, chroma, 6 * (cfg->B - 8), chroma ? cfg->chroma_qp_offset : 0);
  qc->per = qp->per;
  qc->rem = qp->rem;
  qc->per_qbits = qp->per;
  qc->intra_slice = !cfg->inter_slice;
  qc->sign_hide = cfg->sign_hide;
  qc->scan_idx = hmo_coef_scan_idx(N, !chroma, 1, t->mode);
}

void hmo_intra_frame_encode(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                            const int16_t *const org[3], const int org_stride[3],
                            int16_t *const rec[3], const int rec_stride[3],
                            int32_t *const level[3]) {
  int16_t pred[32 * 32], resi[32 * 32];
  int32_t lvl[32 * 32];
  for (int i = 0; i < n_tu; i++) {
    const hmo_tu *t = &tus[i];
    int p = t->plane, N = 1 << t->log2n, ts = t->flags & 1;
    int pw = p ? cfg->pic_w / 2 : cfg->pic_w;
    hmo_qp qp;
    hmo_quant_cfg qc;
    tu_predict(cfg, t, rec[p], rec_stride[p], pred);
    tu_quant_cfg(cfg, t, &qp, &qc);
    hmo_subtract(org[p] + t->y * org_stride[p] + t->x, org_stride[p], pred, N, resi, N, N, N);
    unsigned tmode = p ? HMO_REG_DCT : t->mode;
    uint32_t abs_sum;
    hmo_transformNxN(resi, N, lvl, N, cfg->B, tmode, ts, 0, &qc, &abs_sum);
    if (abs_sum)
      hmo_invtransformNxN(0, tmode, resi, N, lvl, N, cfg->B, qp.per, qp.rem, ts);
    else
      memset(resi, 0, sizeof(resi)); /* ENC/TEncSearch.cpp:1128-1138 */
    hmo_addClip(pred, N, resi, N, rec[p] + t->y * rec_stride[p] + t->x, rec_stride[p], N, N, cfg->B);
    for (int r = 0; r < N; r++) memcpy(level[p] + (t->y + r) * pw + t->x, lvl + r * N, sizeof(int32_t) * N);
  }
}

/* The same with the quantiser every shipped cfg selects (RDOQ : 1): xRateDistOptQuant for every block that is not a
 * transform-skip block (TSFast: those keep the flat quantiser, COM/TComTrQuant.cpp:1121-1128), as xIntraCodingLumaBlk /
 * ChromaBlk (ENC/TEncSearch.cpp:1006-1390) reach it.  What the encoder takes from its live state is an input here: est[8] =
 * the bit estimates TEncSbac::estBit gives for [luma, chroma][4x4 .. 32x32], lambda[2] = m_dLambda for luma / chroma blocks
 * (selectLambda), the block's cbf context (getCtxQtCbf + texture offset) in bits 4..7 of hmo_tu::flags. */
void hmo_intra_frame_encode_rdoq(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                                 const int16_t *const org[3], const int org_stride[3],
                                 int16_t *const rec[3], const int rec_stride[3],
                                 int32_t *const level[3], const hmo_est_bits *est, const double *lambda) {
  int16_t pred[32 * 32], resi[32 * 32];
  int32_t lvl[32 * 32], coef[32 * 32];
  for (int i = 0; i < n_tu; i++) {
    const hmo_tu *t = &tus[i];
    int p = t->plane, N = 1 << t->log2n, ts = t->flags & 1;
    int pw = p ? cfg->pic_w / 2 : cfg->pic_w;
    hmo_qp qp;
    hmo_quant_cfg qc;
    tu_predict(cfg, t, rec[p], rec_stride[p], pred);
    tu_quant_cfg(cfg, t, &qp, &qc);
    hmo_subtract(org[p] + t->y * org_stride[p] + t->x, org_stride[p], pred, N, resi, N, N, N);
    unsigned tmode = p ? HMO_REG_DCT : t->mode;
    uint32_t abs_sum = 0;
    if (ts) {
      hmo_transformNxN(resi, N, lvl, N, cfg->B, tmode, ts, 0, &qc, &abs_sum);
    } else {
      hmo_rdoq_cfg rc;
      rc.per = qp.per, rc.rem = qp.rem, rc.is_luma = p == 0, rc.is_intra = 1;
      rc.scan_idx = qc.scan_idx, rc.root_cbf = 0, rc.cbf_ctx = (t->flags >> 4) & 15, rc.sign_hide = cfg->sign_hide;
      rc.lambda = lambda[p ? 1 : 0];
      hmo_xT(tmode, resi, N, coef, N, cfg->B);
      hmo_xRateDistOptQuant(coef, lvl, N, cfg->B, &rc, &est[(p ? 4 : 0) + t->log2n - 2], &abs_sum);
    }
    if (abs_sum)
      hmo_invtransformNxN(0, tmode, resi, N, lvl, N, cfg->B, qp.per, qp.rem, ts);
    else
      memset(resi, 0, sizeof(resi));
    hmo_addClip(pred, N, resi, N, rec[p] + t->y * rec_stride[p] + t->x, rec_stride[p], N, N, cfg->B);
    for (int r = 0; r < N; r++) memcpy(level[p] + (t->y + r) * pw + t->x, lvl + r * N, sizeof(int32_t) * N);
  }
}

void hmo_intra_frame_decode(const hmo_frame_cfg *cfg, const hmo_tu *tus, int n_tu,
                            int16_t *const rec[3], const int rec_stride[3],
                            const int32_t *const level[3]) {
  int16_t pred[32 * 32], resi[32 * 32];
  int32_t lvl[32 * 32];
  for (int i = 0; i < n_tu; i++) {
    const hmo_tu *t = &tus[i];
    int p = t->plane, N = 1 << t->log2n, ts = t->flags & 1;
    int pw = p ? cfg->pic_w / 2 : cfg->pic_w;
    hmo_qp qp;
    hmo_quant_cfg qc;
    tu_predict(cfg, t, rec[p], rec_stride[p], pred);
    tu_quant_cfg(cfg, t, &qp, &qc);
    for (int r = 0; r < N; r++) memcpy(lvl + r * N, level[p] + (t->y + r) * pw + t->x, sizeof(int32_t) * N);
    hmo_invtransformNxN(0, p ? HMO_REG_DCT : t->mode, resi, N, lvl, N, cfg->B, qp.per, qp.rem, ts);
    hmo_addClip(pred, N, resi, N, rec[p] + t->y * rec_stride[p] + t->x, rec_stride[p], N, N, cfg->B);
  }
}

void hmo_mc_frame(const hmo_pu *pus, int n_pu, int B, const int16_t *const *ref_planes,
                  const int *ref_strides, int16_t *const dst[3], const int dst_stride[3]) {
  int16_t t0[64 * 64], t1[64 * 64];
  for (int i = 0; i < n_pu; i++) {
    const hmo_pu *u = &pus[i];
    int bi = (u->ref0 != 255 && u->ref1 != 255);
    for (int p = 0; p < 3; p++) {
      int c = p != 0, x = u->x >> c, y = u->y >> c, w = u->w >> c, h = u->h >> c;
      int16_t *d = dst[p] + y * dst_stride[p] + x;
      for (int l = 0; l < 2; l++) {
        int r = l ? u->ref1 : u->ref0;
        if (r == 255) continue;
        const int16_t *ref = ref_planes[r * 3 + p] + y * ref_strides[p] + x;
        int mvx = l ? u->mv1x : u->mv0x, mvy = l ? u->mv1y : u->mv0y;
        int16_t *o = bi ? (l ? t1 : t0) : d;
        int os = bi ? 64 : dst_stride[p];
        if (!c)
          hmo_predInterLumaBlk(ref, ref_strides[p], mvx, mvy, u->w, u->h, o, os, bi, B);
        else
          hmo_predInterChromaBlk(ref, ref_strides[p], mvx, mvy, u->w, u->h, o, os, bi, B);
      }
      if (bi) hmo_addAvg(t0, 64, t1, 64, d, dst_stride[p], w, h, B);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Planar YUV frames (VIO/TVideoIOYuv.cpp)
 * ---------------------------------------------------------------------------------------- */
static int16_t yuv_rescale(int16_t v, int shift, int bits) { /* scalePlane :62-127: shift > 0 multiplies */
  if (shift == 0) return v;
  if (shift > 0) return (int16_t)(v << shift);
  int16_t r = (int16_t)((v + (int16_t)(1 << (-shift - 1))) >> -shift);
  int maxv = (1 << bits) - 1;
  return (int16_t)(r < 0 ? 0 : (r > maxv ? maxv : r));
}
void hmo_yuv_unpack(const uint8_t *file, int file_bits, int internal_bits, int w_full, int h_full, int pad_x, int pad_y,
                    int16_t *const planes[3], const int strides[3]) {
  const int wide = file_bits > 8, shift = internal_bits - file_bits;
  for (int p = 0; p < 3; p++) {
    const int c = p ? 1 : 0, wf = w_full >> c, hf = h_full >> c, px = pad_x >> c, py = pad_y >> c, w = wf - px, h = hf - py;
    int16_t *d = planes[p];
    const int st = strides[p];
    for (int y = 0; y < hf; y++)
      for (int x = 0; x < wf; x++) {
        const int sx = x < w ? x : w - 1, sy = y < h ? y : h - 1; /* readPlane :226-275 */
        const uint8_t *s = file + ((size_t)sy * w + sx) * (wide ? 2 : 1);
        const int16_t v = wide ? (int16_t)((s[1] << 8) | s[0]) : (int16_t)s[0];
        d[y * st + x] = yuv_rescale(v, shift, internal_bits);
      }
    file += (size_t)w * h * (wide ? 2 : 1);
  }
}
void hmo_yuv_pack(const int16_t *const planes[3], const int strides[3], int w, int h, int crop_right, int crop_bottom,
                  int internal_bits, int file_bits, uint8_t *file) {
  const int wide = file_bits > 8, shift = file_bits - internal_bits; /* write() scales by -m_bitdepthShift :421-436 */
  int ww = w - crop_right, hh = h - crop_bottom;
  for (int p = 0; p < 3; p++) {
    if (p == 1) ww >>= 1, hh >>= 1;
    for (int y = 0; y < hh; y++)
      for (int x = 0; x < ww; x++) {
        const int16_t v = yuv_rescale(planes[p][y * strides[p] + x], shift, file_bits);
        if (wide) {
          *file++ = (uint8_t)(v & 0xff);
          *file++ = (uint8_t)((v >> 8) & 0xff);
        } else
          *file++ = (uint8_t)v;
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * Deblocking filter application (COM/TComLoopFilter.cpp:571-922)
 * ---------------------------------------------------------------------------------------- */
static const uint8_t dbk_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static const uint8_t dbk_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                                     16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};

/* one luma segment: 4 lines across an edge; s = sample Q0 of line 0, `across` = step over the edge, `along` = step to the next line */
static void dbk_luma_segment(int16_t *s, int across, int along, int bs, int qp, int B, int boff, int toff, int p_off, int q_off) {
  const int scale = 1 << (B - 8), maxv = (1 << B) - 1;
  const int tc = dbk_tc[clip3(0, 53, qp + 2 * (bs - 1) + (toff << 1))] * scale;
  const int beta = dbk_beta[clip3(0, 51, qp + (boff << 1))] * scale;
  const int side = (beta + (beta >> 1)) >> 3, cut = tc * 10;
#define PX(line, k) s[(line) * along + (k) * across] /* k = -4..3: P3..P0, Q0..Q3 */
  const int dp0 = abs(PX(0, -3) - 2 * PX(0, -2) + PX(0, -1)), dq0 = abs(PX(0, 0) - 2 * PX(0, 1) + PX(0, 2));
  const int dp3 = abs(PX(3, -3) - 2 * PX(3, -2) + PX(3, -1)), dq3 = abs(PX(3, 0) - 2 * PX(3, 1) + PX(3, 2));
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const int fp = dp < side, fq = dq < side;
  int strong = 1;
  for (int e = 0; e < 2; e++) { /* xUseStrongFiltering on lines 0 and 3 */
    const int line = e ? 3 : 0, dd = 2 * (e ? d3 : d0);
    const int ds = abs(PX(line, -4) - PX(line, -1)) + abs(PX(line, 3) - PX(line, 0));
    strong = strong && (ds < (beta >> 3)) && (dd < (beta >> 2)) && (abs(PX(line, -1) - PX(line, 0)) < ((tc * 5 + 1) >> 1));
  }
  for (int l = 0; l < 4; l++) {
    const int m0 = PX(l, -4), m1 = PX(l, -3), m2 = PX(l, -2), m3 = PX(l, -1), m4 = PX(l, 0), m5 = PX(l, 1), m6 = PX(l, 2), m7 = PX(l, 3);
    int n1 = m1, n2 = m2, n3 = m3, n4 = m4, n5 = m5, n6 = m6;
    if (strong) {
      n3 = clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
      n4 = clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
      n2 = clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
      n5 = clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
      n1 = clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
      n6 = clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    } else {
      int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
      if (abs(delta) < cut) {
        delta = clip3(-tc, tc, delta);
        n3 = clip3(0, maxv, m3 + delta);
        n4 = clip3(0, maxv, m4 - delta);
        const int tc2 = tc >> 1;
        if (fp) n2 = clip3(0, maxv, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
        if (fq) n5 = clip3(0, maxv, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
      }
    }
    if (!p_off) PX(l, -1) = (int16_t)n3, PX(l, -2) = (int16_t)n2, PX(l, -3) = (int16_t)n1;
    if (!q_off) PX(l, 0) = (int16_t)n4, PX(l, 1) = (int16_t)n5, PX(l, 2) = (int16_t)n6;
  }
#undef PX
}
static void dbk_chroma_line(int16_t *s, int across, int tc, int maxv, int p_off, int q_off) {
  const int m2 = s[-2 * across], m3 = s[-across], m4 = s[0], m5 = s[across];
  const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
  if (!p_off) s[-across] = (int16_t)clip3(0, maxv, m3 + delta);
  if (!q_off) s[0] = (int16_t)clip3(0, maxv, m4 - delta);
}
void hmo_deblock_picture(int16_t *const planes[3], const int strides[3], int pic_w, int pic_h, int B, const uint8_t *bs_ver,
                         const uint8_t *bs_hor, const int8_t *qp, const uint8_t *no_filter, int boff, int toff) {
  const int uw = pic_w / 4, uh = pic_h / 4, scale = 1 << (B - 8), maxv = (1 << B) - 1;
  for (int dir = 0; dir < 2; dir++) {
    const uint8_t *bs = dir ? bs_hor : bs_ver;
    for (int uy = 0; uy < uh; uy++)
      for (int ux = 0; ux < uw; ux++) {
        const int u = uy * uw + ux, b = bs[u];
        if (!b || ((dir ? uy : ux) & 1)) continue; /* the 8x8 luma grid */
        const int up = dir ? u - uw : u - 1;       /* the unit on the P side */
        const int pn = no_filter ? no_filter[up] : 0, qn = no_filter ? no_filter[u] : 0;
        const int q_avg = (qp[up] + qp[u] + 1) >> 1;
        int16_t *y0 = planes[0] + (4 * uy) * strides[0] + 4 * ux;
        dbk_luma_segment(y0, dir ? strides[0] : 1, dir ? 1 : strides[0], b, q_avg, B, boff, toff, pn, qn);
        if (b > 1 && !((dir ? uy : ux) & 3)) { /* chroma: its own 8x8 grid, strength 2 only (:709-712, :740) */
          const int qc = hmo_chroma_scale(clip3(0, 51, q_avg));
          const int tc = dbk_tc[clip3(0, 53, qc + 2 * (b - 1) + (toff << 1))] * scale;
          for (int p = 1; p < 3; p++)
            for (int k = 0; k < 2; k++) {
              int16_t *c0 = planes[p] + (2 * uy) * strides[p] + 2 * ux;
              dbk_chroma_line(c0 + k * (dir ? 1 : strides[p]), dir ? strides[p] : 1, tc, maxv, pn, qn);
            }
        }
      }
  }
}

static int dbk_mv_far(const int16_t *a, const int16_t *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }
static int dbk_strength(const hmo_dbk_unit *P, const hmo_dbk_unit *Pm /* motion of the P side */, const hmo_dbk_unit *Q, int tu_edge,
                        int is_b) {
  if (P->intra || Q->intra) return 2;
  if (tu_edge && (Q->cbf || P->cbf)) return 1;
  if (!is_b) return (Pm->ref[0] != Q->ref[0]) || dbk_mv_far(Pm->mv[0], Q->mv[0]);
  /* "no picture" compares equal to "no picture" (NULL == NULL in the reference) */
  const int p0 = Pm->ref[0] < 0 ? -1 : Pm->ref[0], p1 = Pm->ref[1] < 0 ? -1 : Pm->ref[1];
  const int q0 = Q->ref[0] < 0 ? -1 : Q->ref[0], q1 = Q->ref[1] < 0 ? -1 : Q->ref[1];
  if (!((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0))) return 1;
  if (p0 != p1) {
    if (p0 == q0) return dbk_mv_far(Pm->mv[0], Q->mv[0]) || dbk_mv_far(Pm->mv[1], Q->mv[1]);
    return dbk_mv_far(Pm->mv[0], Q->mv[1]) || dbk_mv_far(Pm->mv[1], Q->mv[0]);
  }
  return (dbk_mv_far(Pm->mv[0], Q->mv[1]) || dbk_mv_far(Pm->mv[1], Q->mv[0])) &&
         (dbk_mv_far(Pm->mv[0], Q->mv[0]) || dbk_mv_far(Pm->mv[1], Q->mv[1]));
}
void hmo_deblock_strengths(const hmo_dbk_unit *units, const uint8_t *edge_ver, const uint8_t *edge_hor, int pic_w, int pic_h,
                           int ctu, int is_b, uint8_t *bs_ver, uint8_t *bs_hor) {
  const int uw = pic_w / 4, uh = pic_h / 4;
  for (int dir = 0; dir < 2; dir++) {
    const uint8_t *edge = dir ? edge_hor : edge_ver;
    uint8_t *bs = dir ? bs_hor : bs_ver;
    for (int uy = 0; uy < uh; uy++)
      for (int ux = 0; ux < uw; ux++) {
        const int u = uy * uw + ux;
        bs[u] = 0;
        if (!(edge[u] & 1) || ((dir ? uy : ux) & 1) || (dir ? uy : ux) == 0) continue;
        const int up = dir ? u - uw : u - 1;
        int um = up;
        if (dir && (4 * uy) % ctu == 0) { /* the row above belongs to another CTU: compressed motion, [0 0 3 3] per 16 samples */
          const int g = ux & ~3, k = ux & 3;
          um = up - ux + g + (k < 2 ? 0 : 3);
        }
        bs[u] = (uint8_t)dbk_strength(&units[up], &units[um], &units[u], (edge[u] >> 1) & 1, is_b);
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * Sample adaptive offset (COM/TComSampleAdaptiveOffset.cpp:781-1240)
 * ---------------------------------------------------------------------------------------- */
static int sgn(int v) { return (v > 0) - (v < 0); }
void hmo_sao_picture(const int16_t *const in[3], int16_t *const out[3], const int strides[3], int pic_w, int pic_h, int B, int ctu,
                     const hmo_sao_lcu *const params[3]) {
  static const int eo_table[5] = {1, 2, 0, 3, 4}; /* m_auiEoTable :92-103: edge index -> offset slot, slot 0 = no offset */
  static const int dx[4] = {1, 0, 1, -1}, dy[4] = {0, 1, 1, 1}; /* b = c + d, a = c - d */
  const int cw = (pic_w + ctu - 1) / ctu, maxv = (1 << B) - 1, up = B - (B < 10 ? B : 10);
  for (int p = 0; p < 3; p++) {
    const int sh = p ? 1 : 0, w = pic_w >> sh, h = pic_h >> sh, cs = ctu >> sh, st = strides[p];
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const hmo_sao_lcu *q = &params[p][(y / cs) * cw + x / cs];
        const int c = in[p][y * st + x];
        int v = c;
        if (q->type >= 0 && q->type < 4) {
          const int ax = x - dx[q->type], ay = y - dy[q->type], bx = x + dx[q->type], by = y + dy[q->type];
          if (ax >= 0 && ax < w && ay >= 0 && bx >= 0 && bx < w && by < h) {
            const int slot = eo_table[sgn(c - in[p][ay * st + ax]) + sgn(c - in[p][by * st + bx]) + 2];
            if (slot) v = clip3(0, maxv, c + (q->offset[slot - 1] << up));
          }
        } else if (q->type == 4) {
          const int k = ((c >> (B - 5)) - q->band) & 31; /* band index relative to the first signalled band */
          if (k < 4) v = clip3(0, maxv, c + (q->offset[k] << up));
        }
        out[p][y * st + x] = (int16_t)v;
      }
  }
}

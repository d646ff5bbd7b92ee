// hmx_device.h -- device-side building blocks of libhmx (gfx950).
//
// Work decomposition used by every block kernel: a block (TU) of N x N samples is owned by a group
// of N consecutive lanes of one wavefront; lane r keeps one row (or one column) of the block in
// registers.  The two 1-D passes of a separable transform run entirely in registers, the N x N
// transposition between them goes through a padded LDS tile.  64/N blocks share a wavefront
// (16 4x4, 8 8x8, 4 16x16, 2 32x32), so all cross-lane traffic of a block stays inside a wave.
//
// Arithmetic contract (bit-exact with the reference, see include/hmx.h for file:line):
//   forward pass  y[k] = wrap16((sum_n M[k][n] x[n] + rnd) >> shift)      TComTrQuant.cpp:417-795
//   inverse pass  y[n] = clip16((sum_k M[k][n] c[k] + rnd) >> shift)
// The even/odd recursion below is an exact integer factorisation of those sums (no overflow:
// |sum| < 2^27), so the result equals the reference's partial butterflies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmx {

// ---------------------------------------------------------------------------------------------
// Tables
// ---------------------------------------------------------------------------------------------
// integer chosen by HEVC for cos(p*pi/64), p = 0..32 (first column of the 32-point matrix)
__host__ __device__ constexpr int cos64(int p) {
  constexpr int t[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                         61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
  return t[p];
}
// M_N[k][n]: fold the angle (2n+1)k*(32/N)*pi/64 into the first quadrant
__host__ __device__ constexpr int dct_coef(int N, int k, int n) {
  int p = ((2 * n + 1) * k * (32 / N)) & 127;
  int s = 1;
  if (p > 64) p = 128 - p;
  if (p > 32) {
    p = 64 - p;
    s = -1;
  }
  return s * cos64(p);
}
__host__ __device__ constexpr int dst_coef(int k, int n) {
  constexpr int t[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
  return t[k][n];
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
// Full-rate integer multiplies (v_mul_i32_i24 / v_mad_i32_i24 / v_mul_u32_u24; the plain 32-bit multiply
// v_mul_lo_u32 is a quarter-rate instruction).  Both operands must fit 24 bits; the result is the
// exact low 32 bits of the product.  Every use states why its operands fit.
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ unsigned umul24(unsigned a, unsigned b) { return __umul24(a, b); }
__device__ __forceinline__ int wrap16(int v) { return (int)(short)v; }

// ---------------------------------------------------------------------------------------------
// 1-D transforms on register arrays (raw sums, no rounding)
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void dct_fwd_raw(const int *x, int *y) {
  if constexpr (N == 2) {
    y[0] = 64 * (x[0] + x[1]);
    y[1] = 64 * (x[0] - x[1]);
  } else {
    int e[N / 2], o[N / 2], ye[N / 2];
#pragma unroll
    for (int n = 0; n < N / 2; n++) {
      e[n] = x[n] + x[N - 1 - n];
      o[n] = x[n] - x[N - 1 - n];
    }
    dct_fwd_raw<N / 2>(e, ye);
#pragma unroll
    for (int m = 0; m < N / 2; m++) {
      y[2 * m] = ye[m];
      int acc = 0;
#pragma unroll
      for (int n = 0; n < N / 2; n++) acc += mul24(dct_coef(N, 2 * m + 1, n), o[n]); // |o| < 2^21: sums of int16 inputs
      y[2 * m + 1] = acc;
    }
  }
}

template <int N>
__device__ __forceinline__ void dct_inv_raw(const int *c, int *out) {
  if constexpr (N == 2) {
    out[0] = 64 * (c[0] + c[1]);
    out[1] = 64 * (c[0] - c[1]);
  } else {
    int ce[N / 2], ee[N / 2];
#pragma unroll
    for (int m = 0; m < N / 2; m++) ce[m] = c[2 * m];
    dct_inv_raw<N / 2>(ce, ee);
#pragma unroll
    for (int n = 0; n < N / 2; n++) {
      int acc = 0;
#pragma unroll
      for (int m = 0; m < N / 2; m++) acc += mul24(dct_coef(N, 2 * m + 1, n), c[2 * m + 1]); // c: int16 inputs
      out[n] = ee[n] + acc;
      out[N - 1 - n] = ee[n] - acc;
    }
  }
}

// forward pass with the reference's rounding and implicit store-to-short wrap
template <int N>
__device__ __forceinline__ void fwd_pass(const int *x, int *y, int shift, bool use_dst) {
  int rnd = 1 << (shift - 1);
  if (N == 4 && use_dst) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int acc = 0;
#pragma unroll
      for (int n = 0; n < 4; n++) acc += mul24(dst_coef(k, n), x[n]); // x: int16-range residuals
      y[k] = wrap16((acc + rnd) >> shift);
    }
  } else {
    int raw[N];
    dct_fwd_raw<N>(x, raw);
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = wrap16((raw[k] + rnd) >> shift);
  }
}

// inverse pass with Clip3(-32768, 32767)
template <int N>
__device__ __forceinline__ void inv_pass(const int *c, int *y, int shift, bool use_dst) {
  int rnd = 1 << (shift - 1);
  if (N == 4 && use_dst) {
#pragma unroll
    for (int n = 0; n < 4; n++) {
      int acc = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) acc += mul24(dst_coef(k, n), c[k]); // c: int16 coefficients
      y[n] = clip3(-32768, 32767, (acc + rnd) >> shift);
    }
  } else {
    int raw[N];
    dct_inv_raw<N>(c, raw);
#pragma unroll
    for (int n = 0; n < N; n++) y[n] = clip3(-32768, 32767, (raw[n] + rnd) >> shift);
  }
}

// ---------------------------------------------------------------------------------------------
// Quantiser parameters (host-prepared, one set per plane type) and scan tables
// ---------------------------------------------------------------------------------------------
struct QuantDev {
  int q;          // g_quantScales[rem]
  int per_qbits;  // per used for iQBits (slice base QP)
  int iq_scale;   // g_invQuantScales[rem] << per
  int rnd_factor; // 171 (I) or 85
};

struct PicDev { // parameters shared by the block kernels
  int pic_w, pic_h; // luma
  int ctu;          // 64
  int bit_depth;
  int sign_hide;
  QuantDev qd[2];   // [0] luma, [1] chroma
};

// P.qd[luma ? 0 : 1] by value.  The parameters live in the kernel-argument segment; a select between two
// references makes the compiler select the ADDRESS and fetch every field with a per-lane load from that
// segment.  Here both sets arrive as scalars (v_readfirstlane pins them) and the lane selects values.
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *wave_uniform(T *p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
  return (T *)((unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ QuantDev pick_qd(const PicDev &P, bool luma) {
  const int a0 = wave_uniform(P.qd[0].q), a1 = wave_uniform(P.qd[0].per_qbits), a2 = wave_uniform(P.qd[0].iq_scale),
            a3 = wave_uniform(P.qd[0].rnd_factor);
  const int b0 = wave_uniform(P.qd[1].q), b1 = wave_uniform(P.qd[1].per_qbits), b2 = wave_uniform(P.qd[1].iq_scale),
            b3 = wave_uniform(P.qd[1].rnd_factor);
  return QuantDev{luma ? a0 : b0, luma ? a1 : b1, luma ? a2 : b2, luma ? a3 : b3};
}

// Scan tables g_auiSigLastScan (TComRom.cpp:564-698 with REMOVAL_8x2_2x8_CG): coefficient groups
// of 4x4, groups ordered like the samples inside a group.  Index 0 = diagonal (also used for the
// reference's "zigzag" index, TComTrQuant.cpp:1135), 1 = horizontal, 2 = vertical.  Entry = raster
// position inside the N x N block.  Built at compile time, kept in constant memory.
__host__ __device__ constexpr int diag_xy(int W, int i) { // i-th position of the up-right diagonal scan of a WxW grid -> y*W+x
  int c = 0;
  for (int d = 0; d <= 2 * W - 2; d++)
    for (int x = (d < W ? 0 : d - W + 1); x <= d && x < W; x++) {
      if (c == i) return (d - x) * W + x;
      c++;
    }
  return 0;
}
template <int N, typename T>
struct ScanTab {
  T t[3][N * N];
  constexpr ScanTab() : t{} {
    constexpr int G = N / 4;
    for (int sc = 0; sc < 3; sc++)
      for (int g = 0; g < G * G; g++)
        for (int i = 0; i < 16; i++) {
          int gy = 0, gx = 0, y = 0, x = 0;
          if (sc == 1) {
            gy = g / G, gx = g % G, y = i >> 2, x = i & 3;
          } else if (sc == 2) {
            gx = g / G, gy = g % G, x = i >> 2, y = i & 3;
          } else {
            int gp = diag_xy(G, g), ip = diag_xy(4, i);
            gy = gp / G, gx = gp % G, y = ip >> 2, x = ip & 3;
          }
          t[sc][g * 16 + i] = (T)((gy * 4 + y) * N + gx * 4 + x);
        }
  }
};
static __constant__ ScanTab<4, unsigned char> kScan4 = ScanTab<4, unsigned char>();
static __constant__ ScanTab<8, unsigned char> kScan8 = ScanTab<8, unsigned char>();
static __constant__ ScanTab<16, unsigned char> kScan16 = ScanTab<16, unsigned char>();
static __constant__ ScanTab<32, unsigned short> kScan32 = ScanTab<32, unsigned short>();

// the 16 raster positions of coefficient group g in scan order, kept PACKED as loaded (one or two
// 16-byte loads: 4 dwords of bytes, or 8 dwords of halfwords for 32x32) -- they stay live across the
// sign-hiding decision, so their register footprint matters
template <int N>
struct ScanPos {
  static constexpr int W = N == 32 ? 8 : 4;
  unsigned pk[W];
  __device__ __forceinline__ void load(int scan_idx, int g) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    if constexpr (N == 32) {
      const u4 *p = reinterpret_cast<const u4 *>(&kScan32.t[scan_idx][g * 16]);
      const u4 a = p[0], b = p[1];
#pragma unroll
      for (int q = 0; q < 4; q++) pk[q] = a[q], pk[4 + q] = b[q];
    } else {
      const unsigned char *base = N == 4 ? &kScan4.t[scan_idx][g * 16] : N == 8 ? &kScan8.t[scan_idx][g * 16] : &kScan16.t[scan_idx][g * 16];
      const u4 a = *reinterpret_cast<const u4 *>(base);
#pragma unroll
      for (int q = 0; q < 4; q++) pk[q] = a[q];
    }
  }
  __device__ __forceinline__ int at(int i) const { // i: compile-time constant after unrolling
    if constexpr (N == 32) return (pk[i >> 1] >> (16 * (i & 1))) & 0xffff;
    else return (pk[i >> 2] >> (8 * (i & 3))) & 255;
  }
  __device__ __forceinline__ int at_dyn(int i) const { // i: run-time index
    unsigned d = 0;
    if constexpr (N == 32) {
#pragma unroll
      for (int q = 0; q < 8; q++) d = (i >> 1) == q ? pk[q] : d;
      return (d >> (16 * (i & 1))) & 0xffff;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) d = (i >> 2) == q ? pk[q] : d;
      return (d >> (8 * (i & 3))) & 255;
    }
  }
};

// getCoefScanIdx (TComDataCU.cpp:4014-4063); 0 (zigzag) is used as diagonal by xQuant
__device__ __forceinline__ int coef_scan_idx(int N, bool luma, bool intra, int mode) {
  if (!intra) return 0;
  bool multi = luma ? (N == 4 || N == 8) : (N == 4);
  if (!multi) return 0;
  if (abs(mode - 26) < 5) return 1;
  if (abs(mode - 10) < 5) return 2;
  return 0;
}

// LDS scratch of one block.  tile: transposition buffer, then the quantiser's packed words
//   bits 0..15  level (signed)      bits 16..31  (deltaU << 1) | (unquantised coefficient < 0)
// i.e. everything sign-bit hiding reads about a coefficient in one word.
template <int N>
struct TuLds {
  int tile[N][N + 1];
  int me[3 * N + 2]; // extended main reference of the angular modes, me[N + k] = refMain[k], k = -N..2N
  int line[4 * N + 2];
  int fline[4 * N + 2];
  unsigned nzmask[2]; // bit g: coefficient group g (scan order) holds a non-zero level
};

// The scratch of an 8x8 block worked by FOUR lanes, sixteen blocks per wave (wave_chain_8x2): the extended main reference is
// spent when the prediction is formed and the tile is first written behind it, so they share memory -- 568 bytes per block, 9 KB
// per wave (sixteen TuLds<8> would be 10.5 KB, and sixteen waves per CU have 10 KB each).
struct TuLds8x2 {
  union {
    int tile[8][9];
    int me[3 * 8 + 2];
  };
  int line[4 * 8 + 2];
  int fline[4 * 8 + 2];
  unsigned nzmask[2];
};

__device__ __forceinline__ int level_of(int word) { return (int)(short)word; }

// Flat quantisation of one coefficient (TComTrQuant.cpp:1241-1259) -> packed word.  WIDE = false is
// for coefficients known to lie in [-32768, 32768] (anything that went through a forward pass or
// transform skip): |c| * q < 2^30 and every intermediate fits 32 bits, identical results.
template <bool WIDE>
__device__ __forceinline__ int quant_one(int c, int q, int qbits, int rnd_factor, int &abs_level) {
  int l, du;
  if (WIDE) {
    const long long t = (long long)abs(c) * q, add = (long long)rnd_factor << (qbits - 9);
    l = (int)((t + add) >> qbits);
    du = (int)((t - ((long long)l << qbits)) >> (qbits - 8));
  } else {
    const unsigned t = umul24((unsigned)abs(c), (unsigned)q), add = (unsigned)rnd_factor << (qbits - 9); // |c| <= 2^15, q < 2^15
    l = (int)((t + add) >> qbits);
    du = ((int)t - (l << qbits)) >> (qbits - 8);
  }
  abs_level = l;
  const int level = clip3(-32768, 32767, c < 0 ? -l : l);
  return (level & 0xffff) | ((du << 1 | (c < 0 ? 1 : 0)) << 16);
}

// w[idx] for a run-time idx: a four-level tree of 15 bit-field inserts (v_bfi_b32 with an all-ones /
// all-zeros mask per index bit).  Written with masks on purpose: a tree of ?: selects is turned into
// an indexed private array by the compiler, i.e. scratch memory.
__device__ __forceinline__ int blend(int m, int t, int f) { return (t & m) | (f & ~m); }
__device__ __forceinline__ int select16(const int *w, int idx) {
  int a[8], b[4], c[2];
  const int m0 = -(idx & 1), m1 = -((idx >> 1) & 1), m2 = -((idx >> 2) & 1), m3 = -((idx >> 3) & 1);
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = blend(m0, w[2 * k + 1], w[2 * k]);
#pragma unroll
  for (int k = 0; k < 4; k++) b[k] = blend(m1, a[2 * k + 1], a[2 * k]);
#pragma unroll
  for (int k = 0; k < 2; k++) c[k] = blend(m2, b[2 * k + 1], b[2 * k]);
  return blend(m3, c[1], c[0]);
}

// signBitHidingHDQ (TComTrQuant.cpp:977-1100) for ONE 16-coefficient group, given its 16 packed words
// w[] in scan order (level | neg << 16 | deltaU << 17).  Groups are independent except for the
// reference's lastCG flag: first_nz_group = this is the highest group in scan order that holds a
// non-zero level (its candidate loop starts at the last non-zero, not at 15).
// Returns the scan index whose level changes (or -1) and the new packed word.
//
// Branch-free restatement.  Lanes of a wave hold different groups, so an early exit saves nothing
// unless every lane takes it; instead
//  * one pass gathers three 16-bit masks (non-zero, negative, parity) -> first/last/sign by bit scans;
//  * the reference's candidate rules become an "invalid" mask built with a handful of mask ops:
//      zero level below the first non-zero whose sign differs from the hidden sign   (:1050-1064)
//      the first non-zero itself when |level| = 1 and deltaU <= 0                    (:1040-1043)
//      positions above the last non-zero in the last group                           (:1017, lastCG)
//  * the reference scans n = 15..0 keeping the strictly smaller cost, i.e. the minimum of
//    (cost, -n); cost = -|deltaU| for a non-zero level, -deltaU for a zero level (:1026-1073).
//    One integer key (cost + 512) << 5 | (15 - n), bit 30 set when invalid, and a min over 16 keys.
// deltaU lies in [-86, 255] (the rounding offset is at most 171/512 of a step), so the key is positive.
__device__ __forceinline__ int sbh_decide(const int *w, bool first_nz_group, int &new_word) {
  unsigned nzm = 0, ngm = 0, par = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    nzm |= ((w[i] & 0xffff) != 0 ? 1u : 0u) << i;
    ngm |= (((unsigned)w[i] >> 16) & 1u) << i;
    par ^= (unsigned)w[i];
  }
  const int first = __builtin_ctz(nzm | 0x10000u), last = 31 - __builtin_clz(nzm | 1u);
  const unsigned signbit = (ngm >> first) & 1u;
  const bool hide = nzm != 0 && last - first >= 4 && signbit != (par & 1u);
  const int wf = select16(w, first & 15);
  const int qf = level_of(wf);
  const bool first_unit_down = (qf == 1 || qf == -1) && (wf >> 17) <= 0;
  unsigned inv = ~nzm & ((1u << first) - 1u) & (ngm ^ (0u - signbit));
  inv |= first_unit_down ? (1u << first) : 0u;
  inv |= first_nz_group ? (0xfffffffeu << last) : 0u;
  unsigned best = 0xffffffffu;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int du = w[i] >> 17;
    const int cost = (w[i] & 0xffff) != 0 ? min(du, -du) : -du;
    unsigned key = (unsigned)((cost << 5) + ((512 << 5) | (15 - i)));
    key |= (inv << (30 - i)) & 0x40000000u;
    best = min(best, key);
  }
  if (!hide || (best & 0x40000000u)) return -1;
  const int best_i = 15 - (int)(best & 31u);
  const int wsel = select16(w, best_i);
  const int q = level_of(wsel), neg = (wsel >> 16) & 1;
  int chg = (q != 0 && (wsel >> 17) <= 0) ? -1 : 1;
  if (q == 32767 || q == -32768) chg = -1;
  const int nq = neg ? q - chg : q + chg;
  new_word = (wsel & 0xffff0000) | (nq & 0xffff);
  return best_i;
}

// ---------------------------------------------------------------------------------------------
// Intra reference samples and prediction
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned zorder4(unsigned cx, unsigned cy) { // 4-bit coordinates
  unsigned z = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) z |= ((cx >> b) & 1u) << (2 * b) | ((cy >> b) & 1u) << (2 * b + 1);
  return z;
}

// Availability mask of the 4n+1 neighbour units of a block at luma (x,y), luma size `size`
// (TComPattern.cpp:607-786 + TComDataCU.cpp:1221-1735; one slice, one tile, no constrained intra):
// bit u follows the bNeighborFlags order: below-left (bottom first), left, corner, above, above-right.
__host__ __device__ __forceinline__ unsigned long long intra_avail_mask(int x, int y, int size, const PicDev &P) {
  int n = size >> 2, U = P.ctu >> 2;
  int cx = (x & (P.ctu - 1)) >> 2, cy = (y & (P.ctu - 1)) >> 2;
  unsigned long long m = 0;
  if (x > 0 && y > 0) m |= 1ull << (2 * n);
  if (y > 0) m |= ((1ull << n) - 1) << (2 * n + 1);
  if (x > 0) m |= ((1ull << n) - 1) << n;
  int rx = cx + n - 1, by = cy + n - 1;
  int ctu_col = x / P.ctu, ctu_cols = (P.pic_w + P.ctu - 1) / P.ctu;
  for (int o = 1; o <= n; o++) {
    bool a;
    if (x + size - 4 + 4 * o >= P.pic_w)
      a = false;
    else if (rx + o < U)
      a = cy > 0 ? (zorder4(rx, cy) > zorder4(rx + o, cy - 1)) : (y > 0);
    else
      a = (cy == 0) && y > 0 && ctu_col < ctu_cols - 1;
    if (a) m |= 1ull << (3 * n + o);
    bool b;
    if (y + size - 4 + 4 * o >= P.pic_h)
      b = false;
    else if (by + o < U)
      b = cx > 0 ? (zorder4(cx, by) > zorder4(cx - 1, by + o)) : (x > 0);
    else
      b = false;
    if (b) m |= 1ull << (n - o);
  }
  return m;
}

// The same mask in closed form (no loop over the units; the plan kernels form it for every block of every picture).  The n
// above-right units of an n-aligned block lie in ONE n-aligned neighbour position (a + 1, b - 1) in units of n, so inside the
// CTU they share one answer: is that position earlier in Z-order?  The most significant differing Morton bit decides: the y bit
// at ctz(b) (set in b, clear in b - 1) against the x bit at cto(a) = trailing ones of a (clear in a, set in a + 1); y bits
// outrank x bits of the same position, so the neighbour precedes iff ctz(b) >= cto(a).  Below-left, (a - 1, b + 1): the y bit at
// cto(b) (clear in b) against the x bit at ctz(a) (set in a): the neighbour precedes iff cto(b) < ctz(a).  The picture's right and
// lower edges then cut the runs.  CTU size 64; held against intra_avail_mask at every position by tests/test_intra_dependencies.py.
__host__ __device__ __forceinline__ unsigned long long intra_avail_mask_fast(int x, int y, int size, const PicDev &P) {
  const int n = size >> 2, lgn = n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : 3;
  const int cx = (x & 63) >> 2, cy = (y & 63) >> 2;
  const unsigned a = (unsigned)cx >> lgn, b = (unsigned)cy >> lgn;
  const unsigned long long run = (1ull << n) - 1;
  unsigned long long m = 0;
  if (x > 0 && y > 0) m |= 1ull << (2 * n);
  if (y > 0) m |= run << (2 * n + 1);
  if (x > 0) m |= run << n;
#if defined(__HIP_DEVICE_COMPILE__)
  const int ctz_a = __ffs((int)(a | 16u)) - 1, ctz_b = __ffs((int)(b | 16u)) - 1, cto_a = __ffs((int)(~a)) - 1, cto_b = __ffs((int)(~b)) - 1;
#else
  const int ctz_a = __builtin_ctz(a | 16u), ctz_b = __builtin_ctz(b | 16u), cto_a = __builtin_ctz(~a), cto_b = __builtin_ctz(~b);
#endif
  bool ar;
  if (cx + n < 16) ar = cy > 0 ? ctz_b >= cto_a : y > 0;
  else ar = cy == 0 && y > 0 && (x >> 6) < ((P.pic_w + 63) >> 6) - 1;
  if (ar) {
    int k = (P.pic_w - x - size + 3) >> 2; // units of the run that start inside the picture
    k = k < 0 ? 0 : k > n ? n : k;
    m |= ((1ull << k) - 1) << (3 * n + 1);
  }
  const bool bl = cy + n < 16 && (cx > 0 ? cto_b < ctz_a : x > 0);
  if (bl) {
    int k = (P.pic_h - y - size + 3) >> 2;
    k = k < 0 ? 0 : k > n ? n : k;
    m |= ((1ull << k) - 1) << (n - k); // below-left counts upwards from the bottom: the k units nearest the block
  }
  return m;
}

// The same for a 64 x 64 LUMA block = a whole CTU (the prediction unit of a 64 x 64 coding unit, TEncSearch.cpp:2509-2540:
// initAdiPattern and the 35 modes run at the PU size; no transform of that size exists).  4n + 1 = 65 units of four samples
// do not fit the 64-bit mask, and need not: below-left of a CTU is never coded before it, left / above are all-or-nothing,
// and the picture's right edge cuts the above-right CTU at a multiple of the minimum CU size (8).  So the mask is kept in
// units of EIGHT samples (n = 8, 33 bits, same bit order); build_ref_line takes the unit size as an argument.
__host__ __device__ __forceinline__ unsigned long long intra_avail_mask_ctu(int x, int y, const PicDev &P) {
  unsigned long long m = 0;
  if (x > 0 && y > 0) m |= 1ull << 16;
  if (x > 0) m |= 0xffull << 8;
  if (y > 0) {
    m |= 0xffull << 17;
    for (int o = 0; o < 8; o++)
      if (x + 64 + 8 * o < P.pic_w) m |= 1ull << (25 + o);
  }
  return m;
}

// Reference line of a block: L[0..4N], L[0] = lowest below-left sample, L[2N] = corner,
// L[4N] = right-most above-right sample (fillReferenceSamples, TComPattern.cpp:368-552).
// Every sample is one independent load: an unavailable sample copies the nearest available
// sample before it (or the first available one for a leading run), which is what the reference's
// sequential padding loop produces.  fetch(dx, dy) returns the reconstructed sample at offset
// (dx, dy) from the block origin (plane or tiled addressing is the caller's business).
template <int N, int NL, typename Fetch>
__device__ __forceinline__ void build_ref_line(Fetch fetch, unsigned long long avail, int unit_log2, int bit_depth, int gl,
                                               int *L) {
  const int unit = 1 << unit_log2, n = N >> unit_log2;
  constexpr int IT = (4 * N + 1 + NL - 1) / NL;
  // two phases, fully unrolled: every load of this lane is issued before the first one is consumed,
  // so the gather costs ONE memory round trip instead of one per sample
  int v[IT];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int p = gl + it * NL;
    v[it] = 1 << (bit_depth - 1);
    if (p <= 4 * N && avail != 0) {
      int u = p < 2 * N ? (p >> unit_log2) : (p == 2 * N ? 2 * n : 2 * n + 1 + ((p - 2 * N - 1) >> unit_log2));
      int q = p;
      if (!((avail >> u) & 1)) {
        unsigned long long lower = avail & ((1ull << u) - 1);
        if (lower) {
          int u2 = 63 - __clzll((long long)lower); // last sample of the nearest available unit below
          q = u2 < 2 * n ? (u2 << unit_log2) + unit - 1 : (u2 == 2 * n ? 2 * N : 2 * N + ((u2 - 2 * n) << unit_log2));
        } else {
          int u2 = __ffsll((long long)avail) - 1; // first sample of the first available unit
          q = u2 < 2 * n ? (u2 << unit_log2) : (u2 == 2 * n ? 2 * N : 2 * N + 1 + ((u2 - 2 * n - 1) << unit_log2));
        }
      }
      // one address, one load (q always names an available sample)
      const int dx = q < 2 * N ? -1 : (q == 2 * N ? -1 : q - 2 * N - 1);
      const int dy = q < 2 * N ? 2 * N - 1 - q : -1;
      v[it] = fetch(dx, dy);
    }
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int p = gl + it * NL;
    if (p <= 4 * N) L[p] = v[it];
  }
}

template <int N, int NL>
__device__ __forceinline__ void smooth_ref_line(const int *L, int *F, int gl) { // [1 2 1], :265-306
  for (int p = gl; p <= 4 * N; p += NL)
    F[p] = (p == 0 || p == 4 * N) ? L[p] : (L[p - 1] + 2 * L[p] + L[p + 1] + 2) >> 2;
}

__device__ __forceinline__ bool use_filtered_refs(int mode, int log2n) { // TComPattern.cpp:49-56,577-605
  const int thr = log2n == 2 ? 10 : log2n == 3 ? 7 : log2n == 4 ? 1 : log2n == 5 ? 0 : 10; // m_aucIntraFilter: {10, 7, 1, 0, 10 (64x64)}
  if (mode == 1) return false;
  return min(abs(mode - 10), abs(mode - 26)) > thr;
}

// Angular parameters of a mode (TComPrediction.cpp:196-210)
struct AngParam {
  bool ver;
  int angle, inv_angle;
};
__device__ __forceinline__ AngParam ang_param(int mode) {
  AngParam a;
  a.ver = mode >= 18;
  const int idx = a.ver ? mode - 26 : 10 - mode, aidx = abs(idx);
  constexpr int ang_tab[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
  constexpr int inv_tab[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
  int angle = 0, inv = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
    if (i == aidx) {
      angle = ang_tab[i];
      inv = inv_tab[i];
    }
  a.angle = idx < 0 ? -angle : angle;
  a.inv_angle = inv;
  return a;
}

// Extended main reference of an angular mode (TComPrediction.cpp:227-258): ME[N + k] = refMain[k].
// k >= 0: the main reference (above for vertical modes, left for horizontal ones); k < 0 (negative
// angles only, down to (N*angle)>>5): the side reference projected with the inverse angle.
// NL lanes of the block fill it together.  R = reference line (top(k) = R[2N+k], left(k) = R[2N-k]).
template <int N, int NL>
__device__ __forceinline__ void build_main_ref(const int *R, int *ME, int mode, int gl) {
  if (mode < 2) return;
  const AngParam a = ang_param(mode);
  const int lim = a.angle < 0 ? (N * a.angle) >> 5 : 0, last = a.angle < 0 ? N : 2 * N;
  for (int e = gl; e <= 3 * N; e += NL) {
    const int k = e - N;
    if (k > last || (k < 0 && k <= lim)) continue;
    int v;
    if (k >= 0)
      v = a.ver ? R[2 * N + k] : R[2 * N - k];
    else {
      const int j = (128 + mul24(-k, a.inv_angle)) >> 8; // k >= -32, inv_angle <= 4096
      v = a.ver ? R[2 * N - j] : R[2 * N + j];
    }
    ME[e] = (short)v;
  }
}

// NS samples of the N x N prediction (TComPrediction.cpp:129-386, 689-730, 1010-1029):
// p[s] = pred(row(s), col(s)).  R = reference line (raw or smoothed), ME = extended main reference
// built from the same line (angular modes), dc_sum = sum of the N above and N left neighbours.
template <int N, int NS, typename RowFn, typename ColFn>
__device__ __forceinline__ void intra_pred_samples(const int *R, const int *ME, int mode, bool luma, int bit_depth, int dc_sum,
                                                   RowFn row, ColFn col, int *p) {
  constexpr int LOG2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int *top = R + 2 * N; // top[k]; left(k) = R[2N - k]
  if (mode == 0) { // planar, closed form of the accumulators
    const int tr = top[N + 1], bl = R[2 * N - (N + 1)];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int r = row(s), c = col(s), t = top[c + 1], left = R[2 * N - (r + 1)];
      int hor = (left << LOG2N) + N + mul24(c + 1, tr - left); // samples: at most 12 bits
      int ver = (t << LOG2N) + mul24(r + 1, bl - t);
      p[s] = (short)((hor + ver) >> (LOG2N + 1));
    }
    return;
  }
  if (mode == 1) { // DC (+ edge smoothing for luma, any size)
    const int dc = (dc_sum + N) >> (LOG2N + 1);
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int r = row(s), c = col(s);
      int v = dc;
      if (luma) {
        if (r == 0) v = c == 0 ? (top[1] + R[2 * N - 1] + 2 * dc + 2) >> 2 : (top[c + 1] + 3 * dc + 2) >> 2;
        else if (c == 0) v = (R[2 * N - (r + 1)] + 3 * dc + 2) >> 2;
      }
      p[s] = (short)v;
    }
    return;
  }
  const AngParam a = ang_param(mode);
  const int max_v = (1 << bit_depth) - 1;
  const int *M0 = ME + N; // M0[k] = refMain[k]
  if (a.angle == 0) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int r = row(s), c = col(s);
      // main-frame coordinates: k = distance from the main reference, l = position along it
      const int k = a.ver ? r : c, l = a.ver ? c : r;
      int v = M0[l + 1];
      if (luma && l == 0) { // edge filter on the first column of the main frame (:280-286)
        const int side_k1 = a.ver ? R[2 * N - (k + 1)] : top[k + 1];
        v = clip3(0, max_v, v + (((short)side_k1 - (short)top[0]) >> 1));
      }
      p[s] = v;
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int r = row(s), c = col(s);
    const int k = a.ver ? r : c, l = a.ver ? c : r;
    const int pos = mul24(k + 1, a.angle), di = pos >> 5, df = pos & 31; // k < 32, |angle| <= 32
    const int i = l + di + 1;
    const int m0 = M0[i];
    p[s] = df ? (short)((mul24(32 - df, m0) + mul24(df, M0[i + 1]) + 16) >> 5) : m0; // samples: at most 12 bits
  }
}

} // namespace hmx

// hmx_kernels.h -- block-level device routines built on hmx_device.h and the __global__ kernels.
#pragma once
#include "hmx_device.h"

#include "../../include/hmx.h"

namespace hmx {

template <int N>
struct Log2 {
  static constexpr int v = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
};
// lanes of a workgroup that take part for block size N: 256, except 32x32 whose LDS scratch
// (9.6 KB per block) is kept to four blocks per workgroup
template <int N>
struct Slots {
  static constexpr int v = N == 32 ? 4 : 256 / N;
};

struct PlanesDev { // one picture: three planes, element strides
  short *p[3];
  int s[3];
};
struct LevelsDev {
  int *p[3];
  int s[3];
};

__device__ __forceinline__ int group_sum(int v, int width) { // sum over `width` consecutive lanes
  for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, width);
  return v;
}

// ---------------------------------------------------------------------------------------------
// transformNxN core: residual row (lane gl holds row gl) -> final levels in L.tile[row][col].
// Called by all lanes of the workgroup (contains barriers); `active` masks idle block slots.
//   ts      transform skip (TComTrQuant.cpp:1622)      use_dst  4x4 luma intra (DST)
// Returns the reference's uiAbsSum (sum of |level| BEFORE sign-bit hiding, :1256).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int fwd_tq_block(TuLds<N> &L, int gl, bool active, const int *x, bool ts, bool use_dst,
                                            bool luma, int scan_idx, bool do_quant, const PicDev &P) {
  constexpr int LG = Log2<N>::v;
  const int B = P.bit_depth, tshift = 15 - B - LG;
  int coef[N];
  if (active && gl == 0) L.nzmask[0] = L.nzmask[1] = 0;
  if (ts) {
#pragma unroll
    for (int k = 0; k < N; k++) coef[k] = tshift >= 0 ? x[k] << tshift : (x[k] + (1 << (-tshift - 1))) >> (-tshift);
  } else {
    int y1[N];
    fwd_pass<N>(x, y1, LG - 1 + (B - 8), use_dst);
    if (active) {
#pragma unroll
      for (int k = 0; k < N; k++) L.tile[k][gl] = y1[k]; // transposed store, like dst[k*line + j]
    }
  }
  __syncthreads();
  if (!ts) {
    int z[N];
#pragma unroll
    for (int n = 0; n < N; n++) z[n] = L.tile[gl][n];
    fwd_pass<N>(z, coef, LG + 6, use_dst); // coef[k] is coefficient (row k, col gl)
  }
  __syncthreads();
  if (!do_quant) { // xT / xTransformSkip only: leave the Int coefficients in the tile
    if (active) {
#pragma unroll
      for (int k = 0; k < N; k++) {
        if (ts)
          L.tile[gl][k] = coef[k];
        else
          L.tile[k][gl] = coef[k];
      }
    }
    __syncthreads();
    return 0;
  }
  const QuantDev &qd = P.qd[luma ? 0 : 1];
  const int qbits = 14 + qd.per_qbits + tshift;
  const long long add = (long long)qd.rnd_factor << (qbits - 9);
  int sum = 0;
  if (active) { // idle lanes of a partly filled workgroup may alias another block's scratch
#pragma unroll
    for (int k = 0; k < N; k++) {
      int lvl, du, al;
      quant_one(coef[k], qd, qbits, add, lvl, du, al);
      sum += al;
      const int r = ts ? gl : k, c = ts ? k : gl;
      L.tile[r][c] = lvl;
      L.du[r][c] = (du << 1) | (coef[k] < 0 ? 1 : 0);
    }
  }
  sum = group_sum(active ? sum : 0, N);
  __syncthreads();
  constexpr int NG = (N / 4) * (N / 4);
  const bool hide = P.sign_hide && sum >= 2; // uniform over the block's lanes
  if (hide) {
    for (int g = gl; g < NG; g += N) {
      bool nz = false;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int p = scan_pos<N>(scan_idx, g, i);
        nz |= L.tile[p / N][p % N] != 0;
      }
      if (nz) atomicOr(&L.nzmask[g >> 5], 1u << (g & 31));
    }
  }
  __syncthreads();
  if (hide) {
    const unsigned long long mask = (unsigned long long)L.nzmask[0] | ((unsigned long long)L.nzmask[1] << 32);
    for (int g = gl; g < NG; g += N) {
      if (!((mask >> g) & 1)) continue;
      bool higher = g < 63 ? (mask >> (g + 1)) != 0 : false;
      sbh_group<N>(L, scan_idx, g, !higher);
    }
  }
  __syncthreads();
  return sum;
}

// ---------------------------------------------------------------------------------------------
// invtransformNxN core: levels in L.tile[row][col] -> residual row gl in out[N].
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void inv_tq_block(TuLds<N> &L, int gl, bool active, bool ts, bool use_dst, bool luma,
                                             bool do_dequant, const PicDev &P, int *out) {
  constexpr int LG = Log2<N>::v;
  const int B = P.bit_depth, tshift = 15 - B - LG;
  const QuantDev &qd = P.qd[luma ? 0 : 1];
  const int dshift = 6 - tshift, dadd = 1 << (dshift - 1);
  int c[N], t[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    int v = ts ? L.tile[gl][k] : L.tile[k][gl];
    if (do_dequant) { // xDeQuant flat path, 32-bit product like the reference (:1343-1354)
      int l = clip3(-32768, 32767, v);
      v = clip3(-32768, 32767, (int)((unsigned)l * (unsigned)qd.iq_scale + (unsigned)dadd) >> dshift);
    }
    c[k] = v;
  }
  if (ts) { // xITransformSkip (:1667-1704)
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = wrap16(tshift > 0 ? (c[k] + (1 << (tshift - 1))) >> tshift : c[k] << (-tshift));
  } else {
#pragma unroll
    for (int k = 0; k < N; k++) c[k] = wrap16(c[k]); // coeff[j] = (short)plCoef[j], :1602
    inv_pass<N>(c, t, 7, use_dst); // row gl of the intermediate block
  }
  __syncthreads(); // every lane has read its levels
  if (!ts && active) {
#pragma unroll
    for (int n = 0; n < N; n++) L.tile[gl][n] = t[n];
  }
  __syncthreads();
  if (!ts) {
    int u[N];
#pragma unroll
    for (int k = 0; k < N; k++) u[k] = L.tile[k][gl];
    inv_pass<N>(u, out, 12 - (B - 8), use_dst);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Intra prediction of one block: lane gl -> row gl of the prediction in p[N].
// rec_plane points at sample (0,0) of the plane the block lives in.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void intra_refs(TuLds<N> &L, int gl, bool active, const short *rec_plane, int stride, int x,
                                           int y, bool luma, const PicDev &P) {
  if (active) {
    const int c = luma ? 0 : 1;
    unsigned long long avail = intra_avail_mask(x << c, y << c, N << c, P);
    build_ref_line<N>(rec_plane + (size_t)y * stride + x, stride, avail, luma ? 2 : 1, P.bit_depth, gl, L.line);
  }
  __syncthreads();
  if (active && luma) smooth_ref_line<N>(L.line, L.fline, gl);
  __syncthreads();
}

template <int N>
__device__ __forceinline__ void intra_pred_block(TuLds<N> &L, int gl, int mode, bool luma, const PicDev &P, int *p) {
  const int *R = (luma && use_filtered_refs(mode, Log2<N>::v)) ? L.fline : L.line;
  intra_pred_row<N>(R, mode, luma, P.bit_depth, gl, p);
}

// vectorised row access helpers (rows of N int16 / int32, natural alignment not guaranteed
// for Pel rows of 4 samples inside a plane with odd strides, so go through 2-byte loads when needed)
template <int N>
__device__ __forceinline__ void load_row16(const short *src, int *x) {
#pragma unroll
  for (int k = 0; k < N; k++) x[k] = src[k];
}
template <int N>
__device__ __forceinline__ void store_row16(short *dst, const int *x) {
#pragma unroll
  for (int k = 0; k < N; k++) dst[k] = (short)x[k];
}
template <int N>
__device__ __forceinline__ void load_row32(const int *src, int *x) {
#pragma unroll
  for (int k = 0; k < N; k++) x[k] = src[k];
}
template <int N>
__device__ __forceinline__ void store_row32(int *dst, const int *x) {
#pragma unroll
  for (int k = 0; k < N; k++) dst[k] = x[k];
}

} // namespace hmx

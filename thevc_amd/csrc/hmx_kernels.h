// hmx_kernels.h -- block-level device routines built on hmx_device.h.
//
// Synchronisation model: a block never spans wavefronts (N <= 32 lanes per block, groups aligned to
// N; the 32x32 MFMA path uses exactly one wave), and LDS instructions of one wave execute in issue
// order.  So the only barrier these routines need is wave_sync(): a compiler-level fence that keeps
// LDS stores before / loads after it.  No s_barrier, no workgroup coupling: waves are autonomous.
#pragma once
#include "hmx_device.h"

#include "../../include/hmx.h"

namespace hmx {

template <int N>
struct Log2 {
  static constexpr int v = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
};

struct PlanesDev { // one picture: three planes, element strides
  short *p[3];
  int s[3];
};

// Working layout of the whole-picture path ("tiled"): the plane is cut into CTU blocks (C x C samples,
// raster order); a CTU block holds its 4x4 tiles in Z-order, a tile is row-major.  Every aligned
// N x N block is then one contiguous run of N*N samples, the right column / bottom row of a
// neighbour sit in one or two 64-byte sectors, and HBM sees whole sectors instead of 8-byte pieces
// of scattered rows.  The same addressing with 16 ints per 4x4 unit and row-major N x N blocks is the
// reference's own coefficient layout (TComDataCU::m_pcTrCoeffY + 16 * partition index).
struct TiledPlane {
  short *p;
  int ctu_w;        // CTU blocks per row
  int clog;         // log2 of the CTU size in this plane (6 luma, 5 chroma for CTU 64)
  unsigned qstride; // elements from one 8x8 quad (4 tiles = 64 samples = one 128-byte line) to the next
};
// The tiled offsets below ("o") address ONE picture: CTU blocks in raster order, tiles in Z-order.
// In memory consecutive quads of a picture are qstride elements apart: 64 when every picture has its
// own contiguous slot, 64 * P when P pictures are interleaved quad by quad -- the layout of the
// across-pictures schedule, where the 64/N lanes-groups of a wave hold the SAME block of consecutive
// pictures and so touch consecutive 128-byte lines.
__host__ __device__ __forceinline__ size_t tphys(unsigned qstride, unsigned o) { return (size_t)(o >> 6) * qstride + (o & 63); }
__host__ __device__ __forceinline__ unsigned spread4(unsigned t) { return (t & 1) | ((t & 2) << 1) | ((t & 4) << 2) | ((t & 8) << 3); }
// Z index of tile (tx, ty), both < 16: the two nibbles are spread together (7 operations)
__host__ __device__ __forceinline__ unsigned zorder_tile(unsigned tx, unsigned ty) {
  unsigned v = tx | (ty << 8);
  v = (v | (v << 2)) & 0x3333u;
  v = (v | (v << 1)) & 0x5555u;
  return (v | (v >> 7)) & 0xffu;
}
// offset o of the 4x4 tile that holds (x,y); ctu block base + Z index * 16.  A picture has fewer than
// 2^24 samples per plane set, so the offset is 32-bit arithmetic; only tphys() widens.
__host__ __device__ __forceinline__ unsigned tile_base(int ctu_w, int clog, int x, int y) {
  const unsigned m = (1u << clog) - 1;
  const unsigned tx = ((unsigned)x & m) >> 2, ty = ((unsigned)y & m) >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned cidx = (unsigned)__mul24(y >> clog, ctu_w) + (unsigned)(x >> clog); // CTU rows and columns: far below 2^23
#else
  const unsigned cidx = (unsigned)((y >> clog) * ctu_w + (x >> clog));
#endif
  return (cidx << (2 * clog)) + (zorder_tile(tx, ty) << 4);
}
// physical element index of sample (x,y)
__device__ __forceinline__ size_t taddr(const TiledPlane &T, int x, int y) {
  return tphys(T.qstride, tile_base(T.ctu_w, T.clog, x, y) + ((y & 3) << 2) + (x & 3));
}
// physical displacement of offset k RELATIVE to a quad-aligned block start (k < 1024: a block has at most
// 16 quads): 32-bit, full-rate multiply (quad index < 16, qstride < 2^24)
template <int N>
__device__ __forceinline__ unsigned trel(unsigned qstride, unsigned k) {
  if constexpr (N <= 8) return k; // the whole block is one quad
  return __umul24(k >> 6, qstride) + (k & 63);
}
// offset of tile (q, rr) (tile units) inside an aligned block whose origin tile has Z index z0
__host__ __device__ __forceinline__ unsigned tile_in_block(unsigned q, unsigned rr) { return (spread4(q) | (spread4(rr) << 1)) << 4; }

// Pointers that were loaded from a table in memory are generic to the compiler, and accesses through
// them become FLAT instructions; every buffer this library touches is global memory.
template <typename T>
__device__ __forceinline__ T *as_global(T *p) {
  return (T *)(__attribute__((address_space(1))) T *)p;
}

typedef short s4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Accesses to the working reconstruction.  In the packed schedule (k_intra_packed) blocks of one launch hand their
// reconstruction to later blocks of the SAME launch.  A CU's vector L1 is never refreshed by another CU's stores, so
// every LOAD of the reconstruction there bypasses the L1 (COH loads: sc1, served by the L2).  STORES come in two
// flavours:
//   plain          the producer and every consumer run on ONE XCD (the packed schedule binds a picture group to the XCD
//                  that claimed it, read from the hardware's XCC id): the L1 is write-through, a store whose vmcnt has
//                  drained is in that XCD's L2, and the L2 is the one cache all its CUs share -- nothing has to reach HBM;
//   write-through  (COH stores: sc1) for producers and consumers on different XCDs, whose L2s keep dirty lines to
//                  themselves (MI355X_MICROARCH.md, inter-workgroup visibility: "every store sc1 + drained, every load
//                  sc1").  The first cut of the packed schedule used this form; it is kept for A/B builds.
// In both the producer drains its stores (s_waitcnt vmcnt(0)) before it signals, and the consumer polls the signal with
// an sc1 load before its first load.  Relaxed agent-scope atomics are how HIP spells those instructions
// (global_load/store_dwordx2 ... sc1); 8 bytes = one tile row is the unit of every access.
// COH = false: the level-synchronous schedules, where a kernel boundary separates producer and consumer.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
template <bool COH>
__device__ __forceinline__ s4v ld_rec4(const short *p) { // 4 samples, 8-byte aligned
  if constexpr (COH) {
    const unsigned long long v = __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s4v r;
    __builtin_memcpy(&r, &v, 8);
    return r;
  } else {
    return *reinterpret_cast<const s4v *>(p);
  }
}
template <bool COH>
__device__ __forceinline__ void st_rec4(short *p, s4v v) {
  if constexpr (COH) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    __hip_atomic_store((gu64 *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *reinterpret_cast<s4v *>(p) = v;
  }
}

// Streams of the whole-picture chain that are touched once -- originals in, levels out -- can carry the non-temporal
// hint, so that they do not push the reconstruction lines the next dependency level gathers from out of the XCD's L2
// (2048 pictures of 2160p: 97.5 vs 93.4 Gpx/s; no difference at 256).  -DHMX_STREAM_PLAIN builds without the hint.
#ifndef HMX_STREAM_PLAIN
#define HMX_STREAM_NT 1
#endif
#ifdef HMX_STREAM_NT
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void stream_store(T *p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) { return *p; }
template <typename T>
__device__ __forceinline__ void stream_store(T *p, T v) { *p = v; }
#endif
// row r of the aligned N x N block (N >= 8) whose first sample has offset b0; pb = p + tphys(b0)
template <int N>
__device__ __forceinline__ void tload_row(const short *pb, unsigned qstride, int r, int *x) {
#pragma unroll
  for (int q = 0; q < N / 4; q++) {
    const s4v v = stream_load(reinterpret_cast<const s4v *>(pb + trel<N>(qstride, tile_in_block(q, r >> 2) + ((r & 3) << 2))));
    x[4 * q] = v[0];
    x[4 * q + 1] = v[1];
    x[4 * q + 2] = v[2];
    x[4 * q + 3] = v[3];
  }
}
template <int N, bool COH = false>
__device__ __forceinline__ void tstore_row(short *pb, unsigned qstride, int r, const int *x) {
#pragma unroll
  for (int q = 0; q < N / 4; q++) {
    s4v v = {(short)x[4 * q], (short)x[4 * q + 1], (short)x[4 * q + 2], (short)x[4 * q + 3]};
    st_rec4<COH>(pb + trel<N>(qstride, tile_in_block(q, r >> 2) + ((r & 3) << 2)), v);
  }
}
struct LevelsDev {
  int *p[3];
  int s[3];
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// make this wave's global stores visible to its own later loads issued by other lanes
__device__ __forceinline__ void wave_global_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int group_sum(int v, int width) { // sum over `width` consecutive lanes
  for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, width);
  return v;
}

// distortion of NS samples: sum of (org - rec)^2 >> 2 * (B - 8), TComRdCost::xGetSSE* with IBDI_DISTORTION 0 (TComRdCost.cpp:1313-1657)
template <int NS>
__device__ __forceinline__ unsigned sse_samples(const int *org, const int *rec, int B) {
  const unsigned sh = (unsigned)(B - 8) << 1;
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int d = org[k] - rec[k];
    s += (unsigned)mul24(d, d) >> sh; // |d| < 2^12
  }
  return s;
}

// ---------------------------------------------------------------------------------------------
// Quantise the coefficients a lane holds and run sign-bit hiding on the block (TComTrQuant.cpp
// :1130-1267, :977-1100).  coef[k] sits at (row, col) = pos(k); NL lanes own the block, NCOEF per lane.
// Leaves packed words (level in the low half) in L.tile[row][col]; returns uiAbsSum (before sign-bit
// hiding, :1256).  WIDE: see quant_one.
// ---------------------------------------------------------------------------------------------
template <int N, int NL, int NCOEF, bool WIDE, typename LT, typename RowFn, typename ColFn>
__device__ __forceinline__ int quant_sbh_block(LT &L, int gl, bool active, const int *coef, RowFn row_of, ColFn col_of,
                                               bool luma, int scan_idx, const PicDev &P, const int *qtab = nullptr) {
  constexpr int LG = Log2<N>::v;
  const int tshift = 15 - P.bit_depth - LG;
  const QuantDev qd = pick_qd(P, luma);
  const int qbits = 14 + qd.per_qbits + tshift;
  int sum = 0;
  if (active) { // idle lanes may alias another block's scratch: never let them write
    if (gl == 0) L.nzmask[0] = L.nzmask[1] = 0;
#pragma unroll
    for (int k = 0; k < NCOEF; k++) {
      int al;
      // qtab: getQuantCoeff of a scaling list, per position (TComTrQuant.cpp:1215, 1244); the chains pass none
      const int word = quant_one<WIDE>(coef[k], qtab ? qtab[row_of(k) * N + col_of(k)] : qd.q, qbits, qd.rnd_factor, al);
      sum += al;
      L.tile[row_of(k)][col_of(k)] = word;
    }
  }
  sum = group_sum(active ? sum : 0, NL);
  wave_sync();
  constexpr int NG = (N / 4) * (N / 4), PER = (NG + NL - 1) / NL;
#ifdef HMX_X_NO_SBH /* timing experiment (results wrong): the quantiser without sign-bit hiding */
  const bool hide = false;
#else
  const bool hide = P.sign_hide && sum >= 2; // uniform over the block's lanes
#endif
  int w[PER][16];
  ScanPos<N> pos[PER];
  if (hide) {
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int g = gl + q * NL;
      if (g < NG) {
        pos[q].load(scan_idx, g);
        bool nz = false;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int p = pos[q].at(i);
          w[q][i] = L.tile[p / N][p % N];
          nz |= (w[q][i] & 0xffff) != 0;
        }
        if (nz) atomicOr(&L.nzmask[g >> 5], 1u << (g & 31));
      }
    }
  }
  wave_sync();
  if (hide) {
    const unsigned long long mask = (unsigned long long)L.nzmask[0] | ((unsigned long long)L.nzmask[1] << 32);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int g = gl + q * NL;
      if (g < NG && ((mask >> g) & 1)) {
        const bool higher = g < 63 ? (mask >> (g + 1)) != 0 : false;
        int nw;
        const int bi = sbh_decide(w[q], !higher, nw);
        if (bi >= 0) {
          const int p = pos[q].at_dyn(bi);
          L.tile[p / N][p % N] = nw;
        }
      }
    }
  }
  wave_sync();
  return sum;
}

// ---------------------------------------------------------------------------------------------
// VALU path (N = 4, 8, 16; also 32 in the list kernels): lane gl owns row/column gl of the block.
// transformNxN core: residual row -> final levels in L.tile.  ts = transform skip (:1622),
// use_dst = 4x4 luma intra.  do_quant = false leaves the Int coefficients (xT / xTransformSkip).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int fwd_tq_block(TuLds<N> &L, int gl, bool active, const int *x, bool ts, bool use_dst,
                                            bool luma, int scan_idx, bool do_quant, const PicDev &P) {
  // everything quantised here went through a forward pass (int16) or transform skip (|x| << shift
  // with |x| < 2^(B+1)), so the 32-bit quantiser is exact
  constexpr int LG = Log2<N>::v;
  const int B = P.bit_depth, tshift = 15 - B - LG;
  int coef[N];
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
    wave_sync();
    int z[N];
#pragma unroll
    for (int n = 0; n < N; n++) z[n] = L.tile[gl][n];
    fwd_pass<N>(z, coef, LG + 6, use_dst); // coef[k] is coefficient (row k, col gl)
    wave_sync();
  }
  if (!do_quant) {
    if (active) {
#pragma unroll
      for (int k = 0; k < N; k++) {
        if (ts)
          L.tile[gl][k] = coef[k];
        else
          L.tile[k][gl] = coef[k];
      }
    }
    wave_sync();
    return 0;
  }
  return quant_sbh_block<N, N, N, false>(
      L, gl, active, coef, [&](int k) { return ts ? gl : k; }, [&](int k) { return ts ? k : gl; }, luma, scan_idx, P);
}

__device__ __forceinline__ int dequant_one(int v, int iq_scale, int dshift) { // :1343-1354, 32-bit product
  int l = clip3(-32768, 32767, v);
  // l is int16, iq_scale = g_invQuantScales[rem] << per <= 72 << 14 < 2^23: the low 32 bits of the 24-bit
  // multiply are the reference's wrapped 32-bit product
  return clip3(-32768, 32767, (int)((unsigned)mul24(l, iq_scale) + (1u << (dshift - 1))) >> dshift);
}

// invtransformNxN core: levels in L.tile[row][col] -> residual row gl in out[N].
template <int N>
__device__ __forceinline__ void inv_tq_block(TuLds<N> &L, int gl, bool active, bool ts, bool use_dst, bool luma,
                                             bool do_dequant, const PicDev &P, int *out) {
  constexpr int LG = Log2<N>::v;
  const int B = P.bit_depth, tshift = 15 - B - LG;
  const QuantDev qd = pick_qd(P, luma);
  const int dshift = 6 - tshift;
  int c[N], t[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    int v = ts ? L.tile[gl][k] : L.tile[k][gl];
    c[k] = do_dequant ? dequant_one(level_of(v), qd.iq_scale, dshift) : v;
  }
  if (ts) { // xITransformSkip (:1667-1704)
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = wrap16(tshift > 0 ? (c[k] + (1 << (tshift - 1))) >> tshift : c[k] << (-tshift));
    return;
  }
#pragma unroll
  for (int k = 0; k < N; k++) c[k] = wrap16(c[k]); // coeff[j] = (short)plCoef[j], :1602
  inv_pass<N>(c, t, 7, use_dst);                    // row gl of the intermediate block
  wave_sync();                                      // every lane has read its levels
  if (active) {
#pragma unroll
    for (int n = 0; n < N; n++) L.tile[gl][n] = t[n];
  }
  wave_sync();
  int u[N];
#pragma unroll
  for (int k = 0; k < N; k++) u[k] = L.tile[k][gl];
  inv_pass<N>(u, out, 12 - (B - 8), use_dst);
  wave_sync();
}

// ---------------------------------------------------------------------------------------------
// Intra references and prediction of one block owned by NL lanes.
// ---------------------------------------------------------------------------------------------
template <int N, int NL, typename Fetch>
__device__ __forceinline__ void intra_refs(TuLds<N> &L, int gl, bool active, Fetch fetch, bool luma,
                                           unsigned long long avail, const PicDev &P) {
  // N = 64 (a whole CTU, luma only): the mask counts units of eight samples (intra_avail_mask_ctu)
  if (active) build_ref_line<N, NL>(fetch, avail, N == 64 ? 3 : luma ? 2 : 1, P.bit_depth, gl, L.line);
  wave_sync();
  if (active && luma && N > 4) smooth_ref_line<N, NL>(L.line, L.fline, gl); // 4x4 never uses the smoothed line
  wave_sync();
}

// The same reference line for a block of the TILED working reconstruction, gathered tile row by tile row instead of
// sample by sample: the 4N+1 samples are the last sample of 2N tile rows to the left (left + below-left), the last
// sample of one tile row of the corner tile and N/2 whole tile rows above (above + above-right).  One 8-byte access
// each, 2N + 1 + N/2 of them shared by the block's NL lanes in ceil(.. / NL) rounds, every one issued before the first
// is consumed (the sample-by-sample gather made 4N+1 two-byte accesses, and HBM saw ~4x the bytes the block needs).
// Units outside the mask are not loaded (one address stands in for all of them); the reference's
// padding rule then runs on the raw samples in LDS: position p copies sample q(p), the nearest available sample
// before it (the first available one for a leading run), exactly as build_ref_line picks its load address.
// pb0 = element index of the block's first sample.  Leaves L.line (raw) and L.fline (smoothed, luma N > 4).
template <int N, int NL, bool COH, typename LT>
__device__ __forceinline__ void intra_refs_tiled(LT &L, int gl, bool active, const TiledPlane &R, int x, int y, size_t pb0,
                                                 bool luma, unsigned long long avail, const PicDev &P) {
  constexpr int T = 2 * N + 1 + N / 2, IT = (T + NL - 1) / NL;
  const int ul = luma ? 2 : 1, n = N >> ul;
  // loads of units outside the mask name ONE address in the whole wave (the first lane's block: a valid address in any picture's
  // pool), which the memory pipe serves as a single request
  const short *own = R.p + pb0;
  const short *idle = reinterpret_cast<const short *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((uintptr_t)own >> 32)) << 32) |
                                                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)own)));
  if (active) {
    s4v v[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int t = gl + it * NL;
      if (t < T) {
        int tx, ty, u;
        unsigned um = 1;
        if (t < 2 * N) { // left column, bottom to top: p = t, sample (x-1, y+2N-1-t)
          tx = x - 4, ty = y + 2 * N - 1 - t, u = t >> ul;
        } else if (t == 2 * N) { // corner
          tx = x - 4, ty = y - 1, u = 2 * n;
        } else { // tile j of the row above: p = 2N+1+4j .. +3 (one luma unit, two chroma units)
          const int j = t - 2 * N - 1;
          tx = x + 4 * j, ty = y - 1, u = 2 * n + 1 + ((4 * j) >> ul), um = luma ? 1u : 3u;
        }
        const bool on = ((avail >> u) & um) != 0;
        v[it] = ld_rec4<COH>(on ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, tx, ty) + ((unsigned)(ty & 3) << 2)) : idle);
      }
    }
#pragma unroll
    for (int it = 0; it < IT; it++) { // raw samples wait in L.fline (free until the smoothing pass)
      const int t = gl + it * NL;
      if (t <= 2 * N) {
        L.fline[t] = v[it][3];
      } else if (t < T) {
        const int p0 = 2 * N + 1 + 4 * (t - 2 * N - 1);
        L.fline[p0] = v[it][0], L.fline[p0 + 1] = v[it][1], L.fline[p0 + 2] = v[it][2], L.fline[p0 + 3] = v[it][3];
      }
    }
  }
  wave_sync();
  if (active) {
    const int unit = 1 << ul, dcv = 1 << (P.bit_depth - 1);
#pragma unroll
    for (int it = 0; it < (4 * N + 1 + NL - 1) / NL; it++) {
      const int p = gl + it * NL;
      if (p <= 4 * N) {
        int val = dcv;
        if (avail != 0) {
          const int u = p < 2 * N ? (p >> ul) : (p == 2 * N ? 2 * n : 2 * n + 1 + ((p - 2 * N - 1) >> ul));
          int q = p;
          if (!((avail >> u) & 1)) {
            const unsigned long long lower = avail & ((1ull << u) - 1);
            if (lower) {
              const int u2 = 63 - __clzll((long long)lower); // last sample of the nearest available unit below
              q = u2 < 2 * n ? (u2 << ul) + unit - 1 : (u2 == 2 * n ? 2 * N : 2 * N + ((u2 - 2 * n) << ul));
            } else {
              const int u2 = __ffsll((long long)avail) - 1; // first sample of the first available unit
              q = u2 < 2 * n ? (u2 << ul) : (u2 == 2 * n ? 2 * N : 2 * N + 1 + ((u2 - 2 * n - 1) << ul));
            }
          }
          val = L.fline[q];
        }
        L.line[p] = val;
      }
    }
  }
  wave_sync();
  if (active && luma && N > 4) smooth_ref_line<N, NL>(L.line, L.fline, gl); // 4x4 never uses the smoothed line
  wave_sync();
}

// sum of the N above + N left neighbours, shared by the block's NL lanes (DC mode)
template <int N, int NL, typename LT>
__device__ __forceinline__ int dc_sum_block(const LT &L, int gl) {
  int s = 0;
  for (int i = gl; i < 2 * N; i += NL) s += i < N ? L.line[2 * N + 1 + i] : L.line[2 * N - 1 - (i - N)];
  return group_sum(s, NL);
}

// luma: edge filters of the angular/DC modes (bFilter) and, unless raw_line, the smoothed reference line
// where getPredictorPtr picks it
template <int N>
__device__ __forceinline__ void intra_pred_block(TuLds<N> &L, int gl, int mode, bool luma, const PicDev &P, int *p, bool raw_line = false) {
  const int *R = (luma && !raw_line && use_filtered_refs(mode, Log2<N>::v)) ? L.fline : L.line;
  const int dcs = dc_sum_block<N, N>(L, gl); // shuffles: every lane of the wave executes this
  build_main_ref<N, N>(R, L.me, mode, gl);
  wave_sync();
  intra_pred_samples<N, N>(R, L.me, mode, luma, P.bit_depth, dcs, [&](int) { return gl; }, [](int s) { return s; }, p);
  wave_sync(); // L.me is rebuilt by the next call (mode fan-out)
}

// calcHAD of an N x N block (TComRdCost.cpp:404-450): Hadamard SATD over its 8x8 sub-blocks (4x4 for N = 4),
// each rounded on its own ((sum+2)>>2, (sum+1)>>1), summed.  Lane gl holds row gl of org - cur in d[]; all
// lanes of the block return the block's sum (before calcHAD's final >> bit increment).  The sum of magnitudes
// of a 2-D Hadamard transform does not depend on the order or signs of its rows: plain Walsh-Hadamard steps.
template <int S>
__device__ __forceinline__ void wht_regs(int *v) {
#pragma unroll
  for (int half = 1; half < S; half <<= 1)
#pragma unroll
    for (int i = 0; i < S; i += 2 * half)
#pragma unroll
      for (int j = i; j < i + half; j++) {
        const int a = v[j], b = v[j + half];
        v[j] = a + b;
        v[j + half] = a - b;
      }
}
template <int N>
__device__ __forceinline__ int satd_block(TuLds<N> &L, int gl, const int *d) {
  constexpr int S = N >= 8 ? 8 : 4, RND = S == 8 ? 2 : 1;
  int t[N];
#pragma unroll
  for (int k = 0; k < N; k++) t[k] = d[k];
#pragma unroll
  for (int c = 0; c < N; c += S) wht_regs<S>(t + c); // horizontal, per sub-block row
#pragma unroll
  for (int k = 0; k < N; k++) L.tile[gl][k] = t[k];
  wave_sync();
#pragma unroll
  for (int r = 0; r < N; r++) t[r] = L.tile[r][gl]; // column gl
  wave_sync();
  int total = 0;
#pragma unroll
  for (int c = 0; c < N; c += S) { // vertical, per sub-block column; then the sub-block's S columns are S lanes
    wht_regs<S>(t + c);
    int s = 0;
#pragma unroll
    for (int r = 0; r < S; r++) s += abs(t[c + r]);
    s = group_sum(s, S);
    total += (s + RND) >> RND;
  }
  return group_sum((gl % S) == 0 ? total : 0, N);
}

// row access helpers (a block row inside a plane is not guaranteed to be more than 2-byte aligned)
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
// Levels written by ONE lane as several 16-byte pieces (a row of an 8x8 / 16x16 block, the 64 bytes of a 4x4 block): with the
// non-temporal hint every piece goes out on its own, half a 32-byte sector at a time, and WRITE_SIZE counts 2.7x the bytes
// (tools/issue_probe.hip, k_write<4> against k_write<5>: the 60 GB by which the chain's writes exceed levels + reconstruction).
// Plain stores meet in the L2 first and write the bytes once -- and were measured 3 % SLOWER at 2048 pictures (104.3 against
// 107.5 Gpx/s, tools/ab.sh): the levels then evict the reconstruction lines the next dependency level gathers from, and HBM
// bandwidth is not what bounds the chain.  The hint stays; -DHMX_LEV_PIECES_PLAIN builds without it.
template <typename T>
__device__ __forceinline__ void piece_store(T *p, T v) {
#ifdef HMX_LEV_PIECES_PLAIN
  *p = v;
#else
  stream_store(p, v);
#endif
}
template <int N>
__device__ __forceinline__ void stream_store_row32(int *dst, const int *x) { // 16-byte aligned rows of levels (frame path)
#pragma unroll
  for (int k = 0; k < N; k += 4) {
    const i4v v = {x[k], x[k + 1], x[k + 2], x[k + 3]};
    piece_store(reinterpret_cast<i4v *>(dst + k), v);
  }
}

// ---------------------------------------------------------------------------------------------
// 32x32 on the matrix cores.  One wave owns one block; lane l = (r = l & 31, h = l >> 5).
// v_mfma_i32_32x32x32_i8 multiplies slot (h, s) of A's lane (row, h) with slot (h, s) of B's lane
// (col, h) and leaves C[row mrow(g,h)][col r] in accumulator g (layout verified on gfx950 with exact
// integer data, tools/mfma_i8_probe.hip).  Any consistent assignment of K indices to slots is valid,
// so slot (h, s) is bound to index mrow(s, h): a result tile can then be fed back as the next
// operand without any cross-lane movement:
//   forward  pass 1  C1 = X * M^T         data = A (lane = row j of X),        const B = TF
//            pass 2  C2 = M * C1          const A = TF,                         data = B (C1 as it lies)
//   inverse  pass 1  T1 = c^T * M         data = A (c as it lies: lane = col),  const B = TI
//            pass 2  R^T = M^T * T1       const A = TI,                         data = B (T1 as it lies)
// 16-bit data are split x = 256*hi + (lo + 128) with hi, lo in int8: two MFMAs per pass plus a
// constant 128 * (row or column sum of M), which is 2048 for DCT row 0 and 0 for every other row.
// Integer arithmetic, exact: bit-identical to the partial butterflies (TComTrQuant.cpp:678-795).
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int mrow(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

struct MfmaTables {
  signed char tf[32][2][16]; // tf[r][h][s] = M[r][mrow(s,h)]
  signed char ti[32][2][16]; // ti[r][h][s] = M[mrow(s,h)][r]
  int colsum[32];            // sum_k M[k][n]
  int csp[2][16];            // csp[h][g] = 128 * colsum[mrow(g,h)]
  constexpr MfmaTables() : tf{}, ti{}, colsum{}, csp{} {
    for (int r = 0; r < 32; r++) {
      int cs = 0;
      for (int k = 0; k < 32; k++) cs += dct_coef(32, k, r);
      colsum[r] = cs;
      for (int h = 0; h < 2; h++)
        for (int g = 0; g < 16; g++)
          if (mrow(g, h) == r) csp[h][g] = 128 * cs;
      for (int h = 0; h < 2; h++)
        for (int s = 0; s < 16; s++) {
          tf[r][h][s] = (signed char)dct_coef(32, r, mrow(s, h));
          ti[r][h][s] = (signed char)dct_coef(32, mrow(s, h), r);
        }
    }
  }
};
static __constant__ MfmaTables kMfma = MfmaTables();

// 16 int16-range values -> hi bytes / (lo bytes ^ 0x80) packed four per dword
__device__ __forceinline__ void split_hi_lo(const int *v, v4i &hi, v4i &lo) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    unsigned w01 = __builtin_amdgcn_perm((unsigned)v[4 * q + 1], (unsigned)v[4 * q], 0x05040100u);     // x1.lo16 : x0.lo16
    unsigned w23 = __builtin_amdgcn_perm((unsigned)v[4 * q + 3], (unsigned)v[4 * q + 2], 0x05040100u); // x3.lo16 : x2.lo16
    hi[q] = (int)__builtin_amdgcn_perm(w23, w01, 0x07050301u);
    lo[q] = (int)(__builtin_amdgcn_perm(w23, w01, 0x06040200u) ^ 0x80808080u);
  }
}

__device__ __forceinline__ v4i load_const_operand(const signed char (*tab)[2][16], int r, int h) {
  return *reinterpret_cast<const v4i *>(&tab[r][h][0]);
}

// data as the A operand: out[g] = sum_slot data(lane, slot) * konst(col r, slot) + 128 * fix_col (per lane).
// The high-byte product is scaled in place and handed to the low-byte MFMA as its accumulator, so one
// set of 16 accumulator registers carries the whole product.
__device__ __forceinline__ void mfma_data_a(const int *data, v4i konst, int fix, int *raw) {
  v4i hi, lo;
  split_hi_lo(data, hi, lo);
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, konst, c, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; g++) c[g] = 256 * c[g] + fix;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, konst, c, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; g++) raw[g] = c[g];
}
// data as the B operand; fix(g) = 128 * (row sum of the constant A for the row this accumulator holds)
template <typename FixFn>
__device__ __forceinline__ void mfma_data_b(const int *data, v4i konst, FixFn fix, int *raw) {
  v4i hi, lo;
  split_hi_lo(data, hi, lo);
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(konst, hi, c, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; g++) c[g] = 256 * c[g] + fix(g);
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(konst, lo, c, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; g++) raw[g] = c[g];
}

// forward 32x32 DCT: x[s] = residual (row r, col mrow(s,h)) -> coef[g] = coefficient (row mrow(g,h), col r)
__device__ __forceinline__ void fwd32_mfma(const int *x, int r, int h, int B, int *coef) {
  const v4i tf = load_const_operand(kMfma.tf, r, h);
  int raw[16], t[16];
  const int s1 = 4 + (B - 8), s2 = 11;
  mfma_data_a(x, tf, r == 0 ? 128 * 2048 : 0, raw);
#pragma unroll
  for (int g = 0; g < 16; g++) t[g] = wrap16((raw[g] + (1 << (s1 - 1))) >> s1);
  mfma_data_b(t, tf, [&](int g) { return (g == 0 && h == 0) ? 128 * 2048 : 0; }, raw);
#pragma unroll
  for (int g = 0; g < 16; g++) coef[g] = wrap16((raw[g] + (1 << (s2 - 1))) >> s2);
}

// inverse 32x32 DCT: c[g] = coefficient (row mrow(g,h), col r) -> out[g] = residual (row r, col mrow(g,h))
__device__ __forceinline__ void inv32_mfma(const int *c, int r, int h, int B, int *out) {
  const v4i ti = load_const_operand(kMfma.ti, r, h);
  int raw[16], t[16];
  const int s2 = 12 - (B - 8);
  mfma_data_a(c, ti, 128 * kMfma.colsum[r], raw);
#pragma unroll
  for (int g = 0; g < 16; g++) t[g] = clip3(-32768, 32767, (raw[g] + 64) >> 7);
  mfma_data_b(t, ti, [&](int g) { return kMfma.csp[h][g]; }, raw);
#pragma unroll
  for (int g = 0; g < 16; g++) out[g] = clip3(-32768, 32767, (raw[g] + (1 << (s2 - 1))) >> s2);
}

} // namespace hmx

// hmx_chain_dev.h -- device side of the whole-picture all-intra chain: the block chains (wave_chain_*), the
// level-synchronous kernels, the layout conversion and the packed schedule (tables' prep kernels, k_intra_packed).
// Included by hmx_chain.hip (defines HMX_CHAIN_MAIN: also gets the non-template kernels), hmx_chain_rdoq.hip (the RDOQ
// instantiations of k_intra_packed) and hmx_list.hip (the one-lane 4x4 transform halves).
#pragma once
#include "hmx_host.h"

#ifndef HMX_X_SKIP
#define HMX_X_SKIP 0 /* timing experiments only (results are wrong), 4x4 lane chain: 1 no level stores, 2 no reference loads, 4 no source loads, 8 no reconstruction stores; 16: no level stores of 8x8 and 16x16 blocks */
#endif
#define HMX_WAVE_SMEM 9088 /* max(16 * sizeof(TuLds8x2), 4 * sizeof(TuLds<16>), sizeof(Lane4Lds), ...) - checked below; 16 waves per CU have 10 KB each */
static_assert(4 * sizeof(TuLds<16>) <= HMX_WAVE_SMEM, "per-wave LDS scratch");
static_assert(16 * sizeof(TuLds<4>) <= HMX_WAVE_SMEM && 8 * sizeof(TuLds<8>) <= HMX_WAVE_SMEM &&
                  sizeof(TuLds<32>) <= HMX_WAVE_SMEM,
              "per-wave LDS scratch");

// What a block chain needs of the picture it works on, for the plane of its block.
struct PlaneView {
  const short *org; // tiled working copy of the original
  TiledPlane rec;   // tiled working reconstruction
  int *lev;
  int lev_stride;   // > 0: plane geometry; 0: the reference's Z-order coefficient layout
  uint32_t *sse = nullptr; // != NULL (encoder direction): xGetSSE(org, rec) of every block, at the index of its first 4x4 unit
};
// Two ways a wave finds its work.  "Own": every item of the wave is another block of ONE picture
// (descriptor i of a list).  "Across": every item is the SAME block of another picture -- pictures that
// follow one plan (same decisions) run in SIMD across pictures: the descriptor, its mode, position and
// availability are wave-uniform (scalar registers, no divergent mode branches), and a wave is full
// whenever the batch holds at least 64/N pictures.
// The lane index as a value the optimiser cannot see through.  Inside a persistent loop (k_intra_packed) everything
// derived from threadIdx.x is loop-invariant: the compiler hoists all of it -- slot, row, LDS addresses of four chains --
// out of the loop and, out of registers, parks it in scratch memory (176 bytes per lane, reloaded every iteration).
__device__ __forceinline__ int lane_id() {
  int l = threadIdx.x;
  asm volatile("" : "+v"(l));
  return l;
}
struct OwnPicture {
  static constexpr bool kCoherent = false; // producer and consumer are separated by a kernel boundary
  static constexpr bool kWriteThrough = false;
  static constexpr bool kSse = false; // distortion output: packed schedule only
  static constexpr bool kRdoq = false; // RDOQ as the chain's quantiser: packed schedule only
  const PicWork &W;
  const FTu *tus;
  __device__ __forceinline__ void wait() const {}
  __device__ __forceinline__ FTu desc(int i) const { return tus[i]; }
  __device__ __forceinline__ PlaneView view(int, int pl) const {
    TiledPlane r = W.rec[pl];
    r.p = as_global(r.p);
    return PlaneView{as_global(W.org[pl].p), r, as_global(W.lev[pl]), W.lev_stride[pl]};
  }
};
struct AcrossPictures {
  static constexpr bool kCoherent = false;
  static constexpr bool kWriteThrough = false;
  static constexpr bool kSse = false;
  static constexpr bool kRdoq = false;
  __device__ __forceinline__ void wait() const {}
  const PicWork *pics;
  const FTu *ft; // the one block this wave works on (wave-uniform address: scalar loads)
  int pic0, n_pics;
  const short *pool_org; // pictures interleaved quad by quad (TiledPlane::qstride = 64 * n_pics)
  short *pool_rec;
  uint32_t luma_elems, chroma_elems; // plane sizes of one picture (Y, Cb, Cr in this order)
  int ctu_w, clog_luma;
  __device__ __forceinline__ FTu desc(int) const { return *ft; }
  __device__ __forceinline__ PlaneView view(int i, int pl) const {
    const size_t o = (size_t)((pl > 0 ? luma_elems : 0u) + (pl > 1 ? chroma_elems : 0u)) * n_pics + (size_t)(pic0 + i) * 64;
    // the table row is read with computed addresses: an indexed member array would live in scratch
    const char *row = reinterpret_cast<const char *>(&pics[pic0 + i]);
    int *lv = *reinterpret_cast<int *const *>(row + offsetof(PicWork, lev) + pl * sizeof(int *));
    const int ls = *reinterpret_cast<const int *>(row + offsetof(PicWork, lev_stride) + pl * sizeof(int));
    return PlaneView{pool_org + o, TiledPlane{pool_rec + o, ctu_w, pl ? clog_luma - 1 : clog_luma, 64u * n_pics}, as_global(lv), ls};
  }
};

// element offset of row r of the N x N block at (x,y) in a level buffer
// (a level plane holds fewer than 2^32 elements; rows and strides are below 2^24: full-rate multiply)
template <int N>
__device__ __forceinline__ unsigned lev_row_off(const PlaneView &V, int x, int y, int r) {
  return V.lev_stride ? __umul24((unsigned)(y + r), (unsigned)V.lev_stride) + x : tile_base(V.rec.ctu_w, V.rec.clog, x, y) + (unsigned)r * N;
}

#ifdef HMX_MARKS /* tools/section_mix.py: section boundaries as comments in the ISA */
#define HMX_MARK(n, id) asm volatile("; HMXMARK %0 %1" ::"n"(n), "n"(id) : "memory")
#else
#define HMX_MARK(n, id)
#endif
template <int N, bool ENC, bool ONCE = false, typename SRC>
__device__ __forceinline__ void wave_chain_valu(char *smem, const SRC &src, const PicDev &P, int count) {
  constexpr int SL = 64 / N;
  const int lane = lane_id(), slot = lane / N, gl = lane % N;
  TuLds<N> &L = reinterpret_cast<TuLds<N> *>(smem)[slot];
  for (int base = 0; ONCE ? base < 1 : base < count; base += SL) { // ONCE: the level schedule hands a wave at most one pass
    HMX_MARK(N, 0);
    const int i = base + slot;
    const bool active = i < count;
    const FTu ft = src.desc(active ? i : 0);
    const hmx_tu t = ft.t;
    const int pl = t.plane, x = t.x, y = t.y;
    const bool luma = pl == 0, ts = t.flags & HMX_TU_TRANSFORM_SKIP;
    const int scan_idx = coef_scan_idx(N, luma, true, t.mode);
    const unsigned long long avail = (unsigned long long)ft.avail_lo | ((unsigned long long)ft.avail_hi << 32);
    const PlaneView V = src.view(active ? i : 0, pl);
    const TiledPlane &R = V.rec;
    const unsigned b0 = tile_base(R.ctu_w, R.clog, x, y); // same geometry for org and rec
    int pred[N], row[N];
    int *lev_row = V.lev + lev_row_off<N>(V, x, y, gl);
    const size_t pb0 = tphys(R.qstride, b0);
    if (ENC && active) tload_row<N>(V.org + pb0, R.qstride, gl, row); // independent of the references
    HMX_MARK(N, 1);
    src.wait(); // packed schedule: the blocks this one predicts from belong to earlier rows of the same launch
    HMX_MARK(N, 2);
    intra_refs_tiled<N, N, SRC::kCoherent>(L, gl, active, R, x, y, pb0, luma, avail, P);
    HMX_MARK(N, 3);
    intra_pred_block<N>(L, gl, t.mode, luma, P, pred);
    HMX_MARK(N, 4);
    if (ENC) {
#pragma unroll
      for (int k = 0; k < N; k++) row[k] = wrap16(row[k] - pred[k]);
      if constexpr (SRC::kRdoq) { // xRateDistOptQuant in the quantiser's place (transform-skip blocks keep the flat one)
        static_assert(N >= 8, "4x4 blocks with RDOQ run in the lane-per-block chain");
        wave_sync(); // the prediction has read the reference line
        if (gl == 0)
          L.line[0] = active && !ts, L.line[1] = src.picture(), L.line[2] = luma, L.line[3] = scan_idx, L.line[4] = src.cbf_ctx(), L.line[9] = src.group_slot() * 2 + (luma ? 0 : 1), L.line[10] = 0;
        fwd_tq_block<N>(L, gl, active, row, ts, luma, luma, scan_idx, ts, P);
        rdoq_wave_tiles<N, SL>(reinterpret_cast<TuLds<N> *>(smem), src.rdoq_lds(), src.rdoq(), P, lane);
      } else {
        fwd_tq_block<N>(L, gl, active, row, ts, luma, luma, scan_idx, true, P);
      }
      HMX_MARK(N, 5);
      if (active) {
        load_row32<N>(&L.tile[gl][0], row);
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = level_of(row[k]);
#ifdef HMX_STREAM_NT
        if (V.lev_stride == 0) {
          if (!(HMX_X_SKIP & 16) || row[0] == 0x7fffffff) stream_store_row32<N>(lev_row, row); // the reference's coefficient layout: 16-byte aligned rows
        } else
#endif
          store_row32<N>(lev_row, row);
      }
    } else {
      if (active) {
        load_row32<N>(lev_row, row);
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = clip3(-32768, 32767, row[k]) & 0xffff;
        store_row32<N>(&L.tile[gl][0], row);
      }
      wave_sync();
    }
    // inverse of all-zero levels is exactly zero, so the reference's "if (uiAbsSum)" needs no branch
    HMX_MARK(N, 6);
    inv_tq_block<N>(L, gl, active, ts, luma, luma, true, P, row);
    HMX_MARK(N, 7);
    if (active) {
      const int mx = (1 << P.bit_depth) - 1;
#pragma unroll
      for (int k = 0; k < N; k++) row[k] = clip3(0, mx, pred[k] + row[k]);
      tstore_row<N, SRC::kWriteThrough>(R.p + pb0, R.qstride, gl, row);
    }
    HMX_MARK(N, 8);
    if constexpr (ENC && SRC::kSse) {
      if (src.want_sse()) { // wave-uniform: getDistPart right behind the reconstruction (TEncSearch.cpp:1163), fused
        int o[N];
        unsigned d = 0;
        if (active) {
          tload_row<N>(V.org + pb0, R.qstride, gl, o); // the original row again (it went into the residual): an L2 hit
          d = sse_samples<N>(o, row, P.bit_depth);
        }
        d = (unsigned)group_sum((int)d, N);
        if (active && gl == 0) V.sse[b0 >> 4] = d;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 8x8 blocks on FOUR lanes, two rows (then two columns) per lane: sixteen blocks per wave-item.  The arithmetic per sample is
// wave_chain_valu<8>'s; what a block costs whatever its size -- descriptor and addresses, the reference gather and its padding,
// sign-bit hiding (four coefficient groups: one per lane, where eight lanes left four idle) -- is spread over twice the samples.
// PMC before: 3526 VALU instructions per 1024 samples of 8x8 blocks against 1972 / 2573 / 2158 for 4x4 / 16x16 / 32x32, and 37 % of
// the mixed workload's instructions.  Packed schedule, large batches (a wave-item's chain is longer: not for latency-bound ones).
// ---------------------------------------------------------------------------------------------
template <bool ENC, typename SRC>
__device__ __forceinline__ void wave_chain_8x2(char *smem, const SRC &src, const PicDev &P, int count) {
  constexpr int N = 8, NL = 4, LG = 3;
  HMX_MARK(82, 0);
  const int lane = lane_id(), slot = lane >> 2, gl = lane & 3, r0 = 2 * gl;
  TuLds8x2 &L = reinterpret_cast<TuLds8x2 *>(smem)[slot];
  const bool active = slot < count;
  const FTu ft = src.desc(active ? slot : 0);
  const hmx_tu t = ft.t;
  const int pl = t.plane, x = t.x, y = t.y;
  const bool luma = pl == 0, ts = t.flags & HMX_TU_TRANSFORM_SKIP;
  const int scan_idx = coef_scan_idx(N, luma, true, t.mode);
  const unsigned long long avail = (unsigned long long)ft.avail_lo | ((unsigned long long)ft.avail_hi << 32);
  const PlaneView V = src.view(active ? slot : 0, pl);
  const TiledPlane &R = V.rec;
  const unsigned b0 = tile_base(R.ctu_w, R.clog, x, y);
  const size_t pb0 = tphys(R.qstride, b0);
  const int B = P.bit_depth, tshift = 15 - B - LG, mx = (1 << B) - 1;
  int pred[16], v[16];
  if (ENC && active) {
    tload_row<N>(V.org + pb0, R.qstride, r0, v);
    tload_row<N>(V.org + pb0, R.qstride, r0 + 1, v + 8);
  }
  HMX_MARK(82, 1);
  src.wait();
  HMX_MARK(82, 2);
  intra_refs_tiled<N, NL, SRC::kCoherent>(L, gl, active, R, x, y, pb0, luma, avail, P);
  HMX_MARK(82, 3);
  {
    const int *RL = (luma && use_filtered_refs(t.mode, LG)) ? L.fline : L.line;
    const int dcs = dc_sum_block<N, NL>(L, gl);
    build_main_ref<N, NL>(RL, L.me, t.mode, gl);
    wave_sync();
    intra_pred_samples<N, 16>(RL, L.me, t.mode, luma, B, dcs, [&](int s) { return r0 + (s >> 3); }, [](int s) { return s & 7; }, pred);
    wave_sync(); // the main reference shares the tile's memory
  }
  HMX_MARK(82, 4);
  int *const lev_blk = V.lev + lev_row_off<N>(V, x, y, 0); // stride 0: the block's 64 levels are contiguous, row-major
  if (ENC) {
    int coef[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = wrap16(v[k] - pred[k]);
    if (ts) {
#pragma unroll
      for (int k = 0; k < 16; k++) coef[k] = tshift >= 0 ? v[k] << tshift : (v[k] + (1 << (-tshift - 1))) >> (-tshift);
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int y1[N];
        fwd_pass<N>(v + 8 * h, y1, LG - 1 + (B - 8), false);
        if (active) {
#pragma unroll
          for (int k = 0; k < N; k++) L.tile[k][r0 + h] = y1[k]; // transposed store
        }
      }
      wave_sync();
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int z[N];
#pragma unroll
        for (int n = 0; n < N; n++) z[n] = L.tile[r0 + h][n];
        fwd_pass<N>(z, coef + 8 * h, LG + 6, false); // coef[8h + k] = coefficient (row k, column r0 + h)
      }
      wave_sync();
    }
    HMX_MARK(82, 5);
    if (__any(ts)) { // (no transform-skip block is 8x8 in the reference's streams: the general routine keeps the case)
      quant_sbh_block<N, NL, 16, false>(
          L, gl, active, coef, [&](int k) { return ts ? r0 + (k >> 3) : (k & 7); }, [&](int k) { return ts ? (k & 7) : r0 + (k >> 3); }, luma, scan_idx, P);
    } else {
      // Quantiser and sign-bit hiding on a coefficient group held by ONE lane, in registers (what the lane-per-block 4x4 chain does):
      // the two lanes that hold a group's columns swap halves (a lane keeps rows 0-3 or 4-7 of four columns: group gx = gl >> 1,
      // gy = gl & 1), quantise their sixteen coefficients and decide for them.  quant_sbh_block gathers a group's words from LDS in
      // scan order, with a table load, an LDS atomic and two more passes over the tile: 813 -> ~580 instructions for the step.
      const bool odd = gl & 1;
      int rcv[8], cgc[16];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int snd = odd ? coef[8 * h + kk] : coef[8 * h + 4 + kk]; // the rows of the partner's group
          rcv[4 * h + kk] = __builtin_amdgcn_update_dpp(0, snd, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
        }
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int xx = 0; xx < 4; xx++) { // column xx of the group: own columns are 0,1 in the even lane, 2,3 in the odd one
          const int h = xx & 1, mine = odd ? coef[8 * h + 4 + kk] : coef[8 * h + kk];
          cgc[4 * kk + xx] = ((xx >= 2) == odd) ? mine : rcv[4 * h + kk];
        }
      const QuantDev qd = pick_qd(P, luma);
      const int qbits = 14 + qd.per_qbits + tshift;
      int w[16], sum = 0;
      unsigned nzb = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        int al;
        w[k] = quant_one<false>(cgc[k], qd.q, qbits, qd.rnd_factor, al);
        sum += al;
        nzb |= (unsigned)w[k];
      }
      const bool nz = (nzb & 0xffffu) != 0;
      sum = group_sum(active ? sum : 0, NL);
      const bool hide = P.sign_hide && sum >= 2; // uniform over the block's lanes
      // which groups of the block hold a level, in the order of the block's group scan (diagonal and vertical: gx * 2 + gy = gl;
      // horizontal: gy * 2 + gx)
      const unsigned nzq = (unsigned)(__ballot(nz) >> (lane & ~3)) & 15u;
      const bool hor = scan_idx == 1, ver = scan_idx == 2;
      const unsigned nzs = hor ? ((nzq & 9u) | ((nzq & 2u) << 1) | ((nzq & 4u) >> 1)) : nzq;
      const int sidx = hor ? (((gl & 1) << 1) | (gl >> 1)) : gl;
      if (hide && nz) {
        constexpr int dg[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15};
        int ws[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int d = w[dg[k]], hv = w[k], vv = w[((k & 3) << 2) | (k >> 2)];
          ws[k] = hor ? hv : (ver ? vv : d);
        }
        int nw;
        const int bi = sbh_decide(ws, (nzs >> (sidx + 1)) == 0, nw);
        if (bi >= 0) {
          const int bd = (int)((0xfbe7ad369c258140ull >> (4 * bi)) & 15); // dg[bi], one nibble per entry
          const int bp = hor ? bi : (ver ? (((bi & 3) << 2) | (bi >> 2)) : bd);
#pragma unroll
          for (int q = 0; q < 16; q++) w[q] = (q == bp) ? nw : w[q];
        }
      }
      if (active) {
        int *tb = &L.tile[4 * (gl & 1)][4 * (gl >> 1)];
#pragma unroll
        for (int k = 0; k < 16; k++) tb[(k >> 2) * 9 + (k & 3)] = w[k];
      }
      wave_sync();
    }
    HMX_MARK(82, 6);
    if (active) {
      if (V.lev_stride == 0) { // four lanes, 16 bytes each, in the order of the addresses: an instruction writes one whole 64-byte line
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * j + gl, r = p >> 1, c0 = 4 * (p & 1);
          const i4v o = {level_of(L.tile[r][c0]), level_of(L.tile[r][c0 + 1]), level_of(L.tile[r][c0 + 2]), level_of(L.tile[r][c0 + 3])};
          piece_store(reinterpret_cast<i4v *>(lev_blk + 4 * p), o);
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          int w[N];
          load_row32<N>(&L.tile[r0 + h][0], w);
#pragma unroll
          for (int k = 0; k < N; k++) w[k] = level_of(w[k]);
          store_row32<N>(V.lev + lev_row_off<N>(V, x, y, r0 + h), w);
        }
      }
    }
  } else {
    if (active) {
      if (V.lev_stride == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * j + gl, r = p >> 1, c0 = 4 * (p & 1);
          const i4v o = *reinterpret_cast<const i4v *>(lev_blk + 4 * p);
#pragma unroll
          for (int k = 0; k < 4; k++) L.tile[r][c0 + k] = clip3(-32768, 32767, o[k]) & 0xffff;
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          int w[N];
          load_row32<N>(V.lev + lev_row_off<N>(V, x, y, r0 + h), w);
#pragma unroll
          for (int k = 0; k < N; k++) L.tile[r0 + h][k] = clip3(-32768, 32767, w[k]) & 0xffff;
        }
      }
    }
    wave_sync();
  }
  HMX_MARK(82, 7);
  // inverse (inv_tq_block's steps for the two columns, then the two rows, of this lane)
  const QuantDev qd = pick_qd(P, luma);
  const int dshift = 6 - tshift;
  int out[16];
  if (ts) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = dequant_one(level_of(L.tile[r0 + (k >> 3)][k & 7]), qd.iq_scale, dshift);
      out[k] = wrap16(tshift > 0 ? (c + (1 << (tshift - 1))) >> tshift : c << (-tshift));
    }
  } else {
    int mid[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int c[N];
#pragma unroll
      for (int k = 0; k < N; k++) c[k] = wrap16(dequant_one(level_of(L.tile[k][r0 + h]), qd.iq_scale, dshift));
      inv_pass<N>(c, mid + 8 * h, 7, false);
    }
    wave_sync(); // every lane has read its levels
    if (active) {
#pragma unroll
      for (int k = 0; k < 16; k++) L.tile[r0 + (k >> 3)][k & 7] = mid[k];
    }
    wave_sync();
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int u[N];
#pragma unroll
      for (int k = 0; k < N; k++) u[k] = L.tile[k][r0 + h];
      inv_pass<N>(u, out + 8 * h, 12 - (B - 8), false);
    }
    wave_sync();
  }
  HMX_MARK(82, 8);
  if (active) {
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = clip3(0, mx, pred[k] + out[k]);
    tstore_row<N, SRC::kWriteThrough>(R.p + pb0, R.qstride, r0, out);
    tstore_row<N, SRC::kWriteThrough>(R.p + pb0, R.qstride, r0 + 1, out + 8);
  }
  HMX_MARK(82, 9);
  if constexpr (ENC && SRC::kSse) {
    if (src.want_sse()) {
      int o[16];
      unsigned d = 0;
      if (active) {
        tload_row<N>(V.org + pb0, R.qstride, r0, o);
        tload_row<N>(V.org + pb0, R.qstride, r0 + 1, o + 8);
        d = sse_samples<16>(o, out, B);
      }
      d = (unsigned)group_sum((int)d, NL);
      if (active && gl == 0) V.sse[b0 >> 4] = d;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 4x4 blocks, one LANE per block (64 blocks per wave).  Everything a 4x4 block needs fits one lane's
// registers: the two transposes of the separable transform are register renaming, sign-bit hiding
// runs in every lane instead of one lane in four, and no lane idles while its group waits.
// Only the reference line goes through LDS (the angular modes index it at run time).
// ---------------------------------------------------------------------------------------------
struct Lane4Lds {
  int line[64][17]; // line[lane][p], p = 0..16 as in build_ref_line (odd stride: conflict-free)
  int me[64][13];   // extended main reference per lane (3N+1 entries)
};
static_assert(sizeof(Lane4Lds) <= HMX_WAVE_SMEM, "per-wave LDS scratch");

__device__ __forceinline__ int scan4_pos(int scan_idx, int i) { // raster position of scan entry i of a 4x4 block
  constexpr unsigned char dg[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15};
  const int hor = i, ver = ((i & 3) << 2) | (i >> 2);
  return scan_idx == 1 ? hor : (scan_idx == 2 ? ver : dg[i]);
}

// The transform half of a 4x4 block held by ONE lane (used by the lane-per-block chain and by the inter list kernel).
// lane4_forward: residual (row-major) -> packed words (level | neg << 16 | deltaU << 17) after sign-bit hiding.
__device__ __forceinline__ void lane4_coef(const int *resid, bool use_dst, bool ts, const PicDev &P, int *coef) {
  const int B = P.bit_depth, tshift = 15 - B - 2;
  if (ts) {
#pragma unroll
    for (int k = 0; k < 16; k++) coef[k] = resid[k] << tshift; // tshift >= 1 for B <= 12
  } else {
    int t1[16];
#pragma unroll
    for (int r = 0; r < 4; r++) { // tmp[k][r] = pass1(row r)[k]
      int yk[4];
      fwd_pass<4>(resid + 4 * r, yk, 1 + (B - 8), use_dst);
#pragma unroll
      for (int k = 0; k < 4; k++) t1[4 * k + r] = yk[k];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { // coeff[k][r] = pass2(row r of tmp)[k]
      int yk[4];
      fwd_pass<4>(t1 + 4 * r, yk, 8, use_dst);
#pragma unroll
      for (int k = 0; k < 4; k++) coef[4 * k + r] = yk[k];
    }
  }
}
__device__ __forceinline__ void lane4_forward(const int *resid, bool use_dst, bool ts, bool luma, int scan_idx, const PicDev &P, int *w) {
  const int B = P.bit_depth, tshift = 15 - B - 2;
  int coef[16];
  lane4_coef(resid, use_dst, ts, P, coef);
  const QuantDev qd = pick_qd(P, luma);
  const int qbits = 14 + qd.per_qbits + tshift;
  int sum = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int al;
    w[k] = quant_one<false>(coef[k], qd.q, qbits, qd.rnd_factor, al);
    sum += al;
  }
  if (P.sign_hide && sum >= 2) { // one coefficient group = the whole block; it is "the last group"
    // the scan differs per lane, but there are only three of them: scan entry k of each is a
    // compile-time register, so the reorder is two selects per entry
    constexpr int dg[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15};
    const bool hor = scan_idx == 1, ver = scan_idx == 2;
    int ws[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int d = w[dg[k]], hv = w[k], vv = w[((k & 3) << 2) | (k >> 2)];
      ws[k] = hor ? hv : (ver ? vv : d);
    }
    int nw;
    const int bi = sbh_decide(ws, true, nw);
    if (bi >= 0) {
      const int bd = (int)((0xfbe7ad369c258140ull >> (4 * bi)) & 15); // dg[bi], one nibble per entry
      const int bp = hor ? bi : (ver ? (((bi & 3) << 2) | (bi >> 2)) : bd);
#pragma unroll
      for (int q = 0; q < 16; q++) w[q] = (q == bp) ? nw : w[q];
    }
  }
}
// lane4_inverse: levels (row-major) -> residual
__device__ __forceinline__ void lane4_inverse(const int *lv, bool use_dst, bool ts, bool luma, const PicDev &P, int *out) {
  const int B = P.bit_depth, tshift = 15 - B - 2;
  const QuantDev qd = pick_qd(P, luma);
  int c[16];
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = dequant_one(lv[k], qd.iq_scale, 6 - tshift);
  if (ts) {
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = wrap16((c[k] + (1 << (tshift - 1))) >> tshift);
    return;
  }
  int t1[16];
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = wrap16(c[k]);
#pragma unroll
  for (int j = 0; j < 4; j++) { // tmp[j][n] = sum_k M[k][n] * c[k][j]
    int col[4], yn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) col[k] = c[4 * k + j];
    inv_pass<4>(col, yn, 7, use_dst);
#pragma unroll
    for (int nn = 0; nn < 4; nn++) t1[4 * j + nn] = yn[nn];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { // block[j][n] = sum_k M[k][n] * tmp[k][j]
    int col[4], yn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) col[k] = t1[4 * k + j];
    inv_pass<4>(col, yn, 12 - (B - 8), use_dst);
#pragma unroll
    for (int nn = 0; nn < 4; nn++) out[4 * j + nn] = yn[nn];
  }
}

template <bool ENC, bool ONCE = false, typename SRC>
__device__ __forceinline__ void wave_chain_4_lane(char *smem, const SRC &src, const PicDev &P, int count) {
  Lane4Lds &LS = *reinterpret_cast<Lane4Lds *>(smem);
  const int lane = lane_id();
  const int B = P.bit_depth, mx = (1 << B) - 1;
  for (int base = 0; ONCE ? base < 1 : base < count; base += 64) {
    const int i = base + lane;
    const bool active = i < count;
    const FTu ft = src.desc(active ? i : 0);
    const hmx_tu t = ft.t;
    const int pl = t.plane, x = t.x, y = t.y, mode = t.mode;
    const bool luma = pl == 0, ts = t.flags & HMX_TU_TRANSFORM_SKIP;
    const unsigned avail = ft.avail_lo; // 4n+1 <= 9 units
    const PlaneView V = src.view(active ? i : 0, pl);
    const TiledPlane &R = V.rec;
    const unsigned b0 = tile_base(R.ctu_w, R.clog, x, y);
    const size_t pb0 = tphys(R.qstride, b0); // a tile never straddles quads
    // Packed schedule: the vector-memory pipe, not the ALU, bounds this chain (PMC, 2048 pictures of 4x4 blocks: texture addresser
    // busy 0.99 of the time, VALU 0.3; a lane's 16-byte access is a request of its own and a lane's four level stores are four partial
    // writes of one 64-byte line).  So the lanes move a block's bytes TOGETHER and hand them to the block's lane through LDS: two
    // lanes per 32-byte tile (source, reconstruction), four per 64 bytes of levels.  The lanes without a block stay for that (they
    // repeat item 0's arithmetic and store nothing).  Elsewhere a lane works alone from the wait on.
    constexpr bool COOP = ONCE && SRC::kCoherent && !SRC::kRdoq;
    unsigned long long *const co_addr = reinterpret_cast<unsigned long long *>(smem); // [64] a block's address, by lane
    i4v *const co_tile = reinterpret_cast<i4v *>(smem + 512);                          // [64][2] halves of the blocks' tiles
    int *const co_rows = reinterpret_cast<int *>(smem);                                // [64][20]: 16 levels, the block's address, padding to 80 bytes
    int *lev_ptr = V.lev;
    const bool zlev = V.lev_stride == 0;
    const unsigned l0 = zlev ? b0 : __umul24((unsigned)y, (unsigned)V.lev_stride) + x;
    const int lrow = zlev ? 4 : V.lev_stride;
    const bool co_lev = COOP && __all(zlev); // the reference's coefficient layout: a block's levels are 64 contiguous bytes (every plane of a call alike)
    int v[16], w[16];
    i4v o0, o1, lq[4];
    if ((HMX_X_SKIP & 4) && ENC) {
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = (lane * 7 + k * 13 + x) & 255;
    } else if (ENC && COOP) {
      co_addr[lane] = (unsigned long long)(uintptr_t)(V.org + pb0);
      wave_sync();
      const i4v *a0 = reinterpret_cast<const i4v *>((uintptr_t)co_addr[lane >> 1]), *a1 = reinterpret_cast<const i4v *>((uintptr_t)co_addr[32 + (lane >> 1)]);
      o0 = stream_load(a0 + (lane & 1)), o1 = stream_load(a1 + (lane & 1)); // blocks 0..31, 32..63: half (lane & 1) of block (lane >> 1)
    } else if (ENC && active) {
      o0 = stream_load(reinterpret_cast<const i4v *>(V.org + pb0)), o1 = stream_load(reinterpret_cast<const i4v *>(V.org + pb0 + 8));
    }
    if (!ENC && co_lev) { // decoder direction: the levels are there before the neighbours are
      wave_sync();
      *reinterpret_cast<unsigned long long *>(co_rows + lane * 20 + 16) = (unsigned long long)(uintptr_t)(lev_ptr + l0);
      wave_sync();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const i4v *from = reinterpret_cast<const i4v *>((uintptr_t) * reinterpret_cast<const unsigned long long *>(co_rows + (16 * k + (lane >> 2)) * 20 + 16));
        lq[k] = stream_load(from + (lane & 3));
      }
    }
    src.wait(); // the whole wave (packed schedule): the neighbours belong to earlier rows of the same launch
    if (ENC && !(HMX_X_SKIP & 4)) {
      if constexpr (COOP) {
        wave_sync();
        co_tile[lane] = o0, co_tile[64 + lane] = o1;
        wave_sync();
        o0 = co_tile[2 * lane], o1 = co_tile[2 * lane + 1];
        wave_sync(); // the reference lines take the scratch
      }
      if (COOP || active) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          v[2 * k] = (short)(o0[k] & 0xffff), v[2 * k + 1] = o0[k] >> 16;
          v[8 + 2 * k] = (short)(o1[k] & 0xffff), v[8 + 2 * k + 1] = o1[k] >> 16;
        }
      }
    }
    if (!ENC && co_lev) {
#pragma unroll
      for (int k = 0; k < 4; k++) *reinterpret_cast<i4v *>(co_rows + (16 * k + (lane >> 2)) * 20 + 4 * (lane & 3)) = lq[k];
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const i4v o = *reinterpret_cast<const i4v *>(co_rows + lane * 20 + 4 * r);
        w[4 * r] = o[0], w[4 * r + 1] = o[1], w[4 * r + 2] = o[2], w[4 * r + 3] = o[3];
      }
      wave_sync(); // the reference lines take the scratch
    }
    if (!COOP && !active) continue;
    // ---- reference line (fillReferenceSamples): sequential padding is natural inside one lane.
    int *line = LS.line[lane];
    {
      const int ul = luma ? 2 : 1, n = 4 >> ul; // unit = 4 (luma) / 2 (chroma) samples
      // The 17 reference samples lie in five neighbour tiles: column 3 of the below-left and left
      // tiles, sample (3,3) of the corner tile, row 3 of the above and above-right tiles.  One tile
      // address each.  `avail` is what the plan put into the descriptor: the units the block's mode READS among the available
      // ones, closed under the padding rule (hmx_plan.hip) -- padding and gathering work on it as on the availability.
      int raw[17];
      {
        const unsigned m_bl = (1u << n) - 1, m_lf = m_bl << n, m_c = 1u << (2 * n), m_a = m_bl << (2 * n + 1), m_ar = m_a << n;
        const bool has_bl = avail & m_bl, has_lf = avail & m_lf, has_c = avail & m_c, has_a = avail & m_a, has_ar = avail & m_ar;
        // A tile the block does not read is not fetched: its lanes all name ONE address (the first lane's own tile -- inside the pool
        // whatever picture that lane works on), which the memory pipe serves as a single request; the mask drops the values.
        const short *own = R.p + pb0;
        const short *idle = reinterpret_cast<const short *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((uintptr_t)own >> 32)) << 32) |
                                                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)own)));
        const short *t_bl = has_bl ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y + 4)) : idle;
        const short *t_lf = has_lf ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y)) : idle;
        const short *t_c = has_c ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y - 4)) : idle;
        const short *t_a = has_a ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, x, y - 4)) : idle;
        const short *t_ar = has_ar ? R.p + tphys(R.qstride, tile_base(R.ctu_w, R.clog, x + 4, y - 4)) : idle;
        // One request per tile.  A lane's load instruction is a request of its own to the L2 on the coherent path (sc1: the vector L1
        // is bypassed), and the vector-memory pipe, not the ALU, bounds this chain (PMC, 2048 pictures of 4x4 blocks: texture addresser
        // busy 0.99 of the time, 13 read requests per block, VALU 0.3): the right column of the two left tiles comes as the tile's two
        // 16-byte halves instead of four 8-byte rows.
        constexpr bool COH = SRC::kCoherent;
        s4v va, var, vc;
        i4v b0v, b1v, l0v, l1v; // rows 0-1 and rows 2-3 of the below-left and the left tile
        if constexpr ((HMX_X_SKIP & 2) != 0) {
          va = s4v{(short)x, (short)y, (short)lane, 3}, var = va, vc = va;
          b0v = i4v{x, y, lane, x ^ y}, b1v = b0v, l0v = b0v, l1v = b0v;
        } else if constexpr (COH) {
          asm volatile("global_load_dwordx2 %0, %7, off sc1\n\t"
                       "global_load_dwordx2 %1, %8, off sc1\n\t"
                       "global_load_dwordx2 %2, %9, off sc1\n\t"
                       "global_load_dwordx4 %3, %10, off sc1\n\t"
                       "global_load_dwordx4 %4, %10, off offset:16 sc1\n\t"
                       "global_load_dwordx4 %5, %11, off sc1\n\t"
                       "global_load_dwordx4 %6, %11, off offset:16 sc1\n\t"
                       "s_waitcnt vmcnt(0)"
                       : "=&v"(va), "=&v"(var), "=&v"(vc), "=&v"(b0v), "=&v"(b1v), "=&v"(l0v), "=&v"(l1v)
                       : "v"(t_a + 12), "v"(t_ar + 12), "v"(t_c + 12), "v"(t_bl), "v"(t_lf)
                       : "memory");
        } else {
          va = ld_rec4<false>(t_a + 12), var = ld_rec4<false>(t_ar + 12), vc = ld_rec4<false>(t_c + 12);
          b0v = *reinterpret_cast<const i4v *>(t_bl), b1v = *reinterpret_cast<const i4v *>(t_bl + 8);
          l0v = *reinterpret_cast<const i4v *>(t_lf), l1v = *reinterpret_cast<const i4v *>(t_lf + 8);
        }
        // p = 0..3: (x-1, y+7-p) = rows 3..0 of the below-left tile; p = 4..7: rows 3..0 of the left tile; column 3 = the odd words' high half
        raw[0] = b1v[3] >> 16, raw[1] = b1v[1] >> 16, raw[2] = b0v[3] >> 16, raw[3] = b0v[1] >> 16;
        raw[4] = l1v[3] >> 16, raw[5] = l1v[1] >> 16, raw[6] = l0v[3] >> 16, raw[7] = l0v[1] >> 16;
#pragma unroll
        for (int k = 0; k < 4; k++) raw[9 + k] = va[k], raw[13 + k] = var[k];
        raw[8] = vc[3];
      }
      const int dc = 1 << (B - 1);
      int carry = dc;
      bool have = false;
      int lead = dc; // value of a leading unavailable run = first sample of the first available unit
#pragma unroll
      for (int p = 16; p >= 0; p--) {
        const int u = p < 8 ? (p >> ul) : (p == 8 ? 2 * n : 2 * n + 1 + ((p - 9) >> ul));
        const bool first_of_unit = p < 8 ? (p & ((1 << ul) - 1)) == 0 : (p == 8 ? true : ((p - 9) & ((1 << ul) - 1)) == 0);
        if (((avail >> u) & 1) && first_of_unit) lead = raw[p];
      }
#pragma unroll
      for (int p = 0; p <= 16; p++) {
        const int u = p < 8 ? (p >> ul) : (p == 8 ? 2 * n : 2 * n + 1 + ((p - 9) >> ul));
        int val;
        if ((avail >> u) & 1) {
          val = raw[p];
          have = true;
        } else {
          val = have ? carry : lead; // an unavailable unit repeats the last sample before it
        }
        carry = val;
        line[p] = avail ? val : dc;
      }
    }
    // ---- prediction (4x4 never uses the smoothed line)
    int pred[16];
    {
      int dcs = 0;
#pragma unroll
      for (int k = 1; k <= 4; k++) dcs += line[8 + k] + line[8 - k];
      int *me = LS.me[lane];
      build_main_ref<4, 1>(line, me, mode, 0);
      intra_pred_samples<4, 16>(line, me, mode, luma, B, dcs, [](int s) { return s >> 2; }, [](int s) { return s & 3; }, pred);
    }
    if (ENC) {
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = wrap16(v[k] - pred[k]);
      bool flat = true;
      if constexpr (SRC::kRdoq) {
        if (!ts) { // transform-skip blocks keep the flat quantiser
          int coef[16];
          lane4_coef(v, luma, false, P, coef);
          // the lane's coefficients in scan order and its levels go through its LDS rows (the reference line and the main
          // reference are spent): no private array is indexed at run time
          const int scan_idx = coef_scan_idx(4, luma, true, mode);
          const bool hor = scan_idx == 1, ver = scan_idx == 2;
          constexpr int dg[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15}, inv_dg[16] = {0, 2, 5, 9, 1, 4, 8, 12, 3, 7, 11, 14, 6, 10, 13, 15};
          int *c16 = LS.line[lane];
          short *l16 = reinterpret_cast<short *>(LS.me[lane]);
#pragma unroll
          for (int k = 0; k < 16; k++) c16[k] = hor ? coef[k] : (ver ? coef[((k & 3) << 2) | (k >> 2)] : coef[dg[k]]);
          rdoq_lane_4x4(c16, l16, src.picture(), src.group_slot() * 2 + (luma ? 0 : 1), luma, scan_idx, src.cbf_ctx(), src.rdoq(), src.rdoq_lds(), P);
#pragma unroll
          for (int q = 0; q < 16; q++) w[q] = l16[hor ? q : (ver ? (((q & 3) << 2) | (q >> 2)) : inv_dg[q])];
          flat = false;
        }
      }
      if (flat) {
        lane4_forward(v, luma, ts, luma, coef_scan_idx(4, luma, true, mode), P, w);
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = level_of(w[k]);
      }
      // A lane's four 16-byte level stores cost more than the rest of the block's memory traffic together (timing experiment, 2048
      // pictures of 4x4 blocks: 254 ms with them, 155 ms without, against 14 ms for the reconstruction's two stores and 67 ms for the
      // arithmetic alone): lanes 4j..4j+3 write block j's line in ONE store instruction.
      if (co_lev) {
        wave_sync(); // the reference lines are spent in every lane
        int *mine = co_rows + lane * 20;
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<i4v *>(mine + 4 * r) = i4v{w[4 * r], w[4 * r + 1], w[4 * r + 2], w[4 * r + 3]};
        *reinterpret_cast<unsigned long long *>(mine + 16) = active ? (unsigned long long)(uintptr_t)(lev_ptr + l0) : 0ull;
        wave_sync();
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int *from = co_rows + (16 * k + (lane >> 2)) * 20;
          const i4v o = *reinterpret_cast<const i4v *>(from + 4 * (lane & 3));
          const unsigned long long to = *reinterpret_cast<const unsigned long long *>(from + 16);
          if (to && (!(HMX_X_SKIP & 1) || o[0] == 0x7fffffff)) piece_store(reinterpret_cast<i4v *>((int *)(uintptr_t)to) + (lane & 3), o);
        }
        wave_sync();
      } else if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          i4v o = {w[4 * r], w[4 * r + 1], w[4 * r + 2], w[4 * r + 3]};
          if (!(HMX_X_SKIP & 1) || o[0] == 0x7fffffff) piece_store(reinterpret_cast<i4v *>(lev_ptr + l0 + (unsigned)r * (unsigned)lrow), o);
        }
      }
    } else if (!co_lev) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const i4v o = *reinterpret_cast<const i4v *>(lev_ptr + l0 + (unsigned)r * (unsigned)lrow);
        w[4 * r] = o[0], w[4 * r + 1] = o[1], w[4 * r + 2] = o[2], w[4 * r + 3] = o[3];
      }
    }
    int out[16];
    lane4_inverse(w, luma, ts, luma, P, out);
    i4v r0, r1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      r0[k] = (clip3(0, mx, pred[2 * k] + out[2 * k]) & 0xffff) | (clip3(0, mx, pred[2 * k + 1] + out[2 * k + 1]) << 16);
      r1[k] = (clip3(0, mx, pred[8 + 2 * k] + out[8 + 2 * k]) & 0xffff) | (clip3(0, mx, pred[8 + 2 * k + 1] + out[8 + 2 * k + 1]) << 16);
    }
    if constexpr (ENC && SRC::kSse) {
      if (src.want_sse()) {
        const i4v o0 = *reinterpret_cast<const i4v *>(V.org + pb0), o1 = *reinterpret_cast<const i4v *>(V.org + pb0 + 8);
        int o[16], rc[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          o[2 * k] = (short)(o0[k] & 0xffff), o[2 * k + 1] = o0[k] >> 16;
          o[8 + 2 * k] = (short)(o1[k] & 0xffff), o[8 + 2 * k + 1] = o1[k] >> 16;
          rc[2 * k] = r0[k] & 0xffff, rc[2 * k + 1] = (int)((unsigned)r0[k] >> 16);
          rc[8 + 2 * k] = r1[k] & 0xffff, rc[8 + 2 * k + 1] = (int)((unsigned)r1[k] >> 16);
        }
        if (active) V.sse[b0 >> 4] = sse_samples<16>(o, rc, B);
      }
    }
    if constexpr (SRC::kWriteThrough) { // write-through, one tile row per store
#pragma unroll
      for (int k = 0; k < 2; k++) {
        s4v a, b;
        const int a2[2] = {r0[2 * k], r0[2 * k + 1]}, b2[2] = {r1[2 * k], r1[2 * k + 1]};
        __builtin_memcpy(&a, a2, 8);
        __builtin_memcpy(&b, b2, 8);
        st_rec4<true>(R.p + pb0 + 4 * k, a);
        st_rec4<true>(R.p + pb0 + 8 + 4 * k, b);
      }
    } else if constexpr (COOP) { // two lanes per tile: one 32-byte request instead of two of 16
      wave_sync();
      co_addr[lane] = active ? (unsigned long long)(uintptr_t)(R.p + pb0) : 0ull;
      co_tile[2 * lane] = r0, co_tile[2 * lane + 1] = r1;
      wave_sync();
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const unsigned long long to = co_addr[32 * j + (lane >> 1)];
        const i4v o = co_tile[64 * j + lane];
        if (to && (!(HMX_X_SKIP & 8) || o[0] == 0x7fffffff)) *(reinterpret_cast<i4v *>((uintptr_t)to) + (lane & 1)) = o;
      }
      wave_sync();
    } else if (active && (!(HMX_X_SKIP & 8) || r0[0] == 0x7fffffff)) {
      *reinterpret_cast<i4v *>(R.p + pb0) = r0;
      *reinterpret_cast<i4v *>(R.p + pb0 + 8) = r1;
    }
  }
}

template <bool ENC, bool ONCE = false, typename SRC>
__device__ __forceinline__ void wave_chain_32(char *smem, const SRC &src, const PicDev &P, int count) {
  const int lane = lane_id(), r = lane & 31, h = lane >> 5;
  TuLds<32> &L = *reinterpret_cast<TuLds<32> *>(smem);
  constexpr int LG = 5;
  for (int i = 0; ONCE ? i < 1 : i < count; i++) {
    const FTu ft = src.desc(i);
    const hmx_tu t = ft.t;
    const int pl = t.plane, x = t.x, y = t.y;
    const bool luma = pl == 0;
    const unsigned long long avail = (unsigned long long)ft.avail_lo | ((unsigned long long)ft.avail_hi << 32);
    const PlaneView V = src.view(i, pl);
    const TiledPlane &R = V.rec;
    const unsigned b0 = tile_base(R.ctu_w, R.clog, x, y);
    // this lane's samples of row r: columns mrow(s,h) = tile column 2*(s>>2)+h, all four samples of the tile row
    const size_t row_off = tphys(R.qstride, b0) + ((r & 3) << 2);
    int pred[16], v[16];
    s4v org4[4]; // kept packed until the residual is formed
    if (ENC) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        org4[q] = stream_load(reinterpret_cast<const s4v *>(V.org + row_off + trel<32>(R.qstride, tile_in_block(2 * q + h, r >> 2))));
    }
    src.wait();
    intra_refs_tiled<32, 64, SRC::kCoherent>(L, lane, true, R, x, y, tphys(R.qstride, b0), luma, avail, P);
    const int *RL = (luma && use_filtered_refs(t.mode, LG)) ? L.fline : L.line;
    const int dcs = dc_sum_block<32, 64>(L, lane);
    build_main_ref<32, 64>(RL, L.me, t.mode, lane);
    wave_sync();
    intra_pred_samples<32, 16>(RL, L.me, t.mode, luma, P.bit_depth, dcs, [&](int) { return r; }, [&](int s) { return mrow(s, h); }, pred);
    const bool zlev = V.lev_stride == 0;
    int *lev0 = V.lev + (zlev ? b0 + r : __umul24((unsigned)y, (unsigned)V.lev_stride) + x + r);
    const int lstep = zlev ? 32 : V.lev_stride;
    // the prediction is needed again only for the reconstruction: it waits as 8 packed registers
    unsigned pred2[8];
#pragma unroll
    for (int s = 0; s < 8; s++) pred2[s] = (unsigned)pred[2 * s] | ((unsigned)pred[2 * s + 1] << 16);
    if (ENC) {
      int coef[16];
#pragma unroll
      for (int s = 0; s < 16; s++) v[s] = wrap16((int)org4[s >> 2][s & 3] - pred[s]);
      fwd32_mfma(v, r, h, P.bit_depth, coef);
      if constexpr (SRC::kRdoq) {
        wave_sync();
        if (lane == 0) L.line[0] = 1, L.line[1] = src.picture(), L.line[2] = luma, L.line[3] = 0, L.line[4] = src.cbf_ctx(), L.line[9] = src.group_slot() * 2 + (luma ? 0 : 1), L.line[10] = 0;
#pragma unroll
        for (int g = 0; g < 16; g++) L.tile[mrow(g, h)][r] = coef[g];
        wave_sync();
        rdoq_wave_tiles<32, 1>(&L, src.rdoq_lds(), src.rdoq(), P, lane);
      } else {
        quant_sbh_block<32, 64, 16, false>(
            L, lane, true, coef, [&](int k) { return mrow(k, h); }, [&](int) { return r; }, luma, 0, P);
      }
#pragma unroll
      for (int g = 0; g < 16; g++) {
        v[g] = level_of(L.tile[mrow(g, h)][r]);
        stream_store(&lev0[__umul24((unsigned)mrow(g, h), (unsigned)lstep)], v[g]);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 16; g++) v[g] = lev0[__umul24((unsigned)mrow(g, h), (unsigned)lstep)];
    }
    const int tshift = 15 - P.bit_depth - LG;
    const QuantDev qd = pick_qd(P, luma);
    int out[16];
#pragma unroll
    for (int g = 0; g < 16; g++) v[g] = wrap16(dequant_one(v[g], qd.iq_scale, 6 - tshift));
    inv32_mfma(v, r, h, P.bit_depth, out);
    const int mx = (1 << P.bit_depth) - 1;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int p0 = pred2[2 * q] & 0xffff, p1 = pred2[2 * q] >> 16, p2 = pred2[2 * q + 1] & 0xffff, p3 = pred2[2 * q + 1] >> 16;
      s4v o = {(short)clip3(0, mx, p0 + out[4 * q]), (short)clip3(0, mx, p1 + out[4 * q + 1]),
               (short)clip3(0, mx, p2 + out[4 * q + 2]), (short)clip3(0, mx, p3 + out[4 * q + 3])};
      st_rec4<SRC::kWriteThrough>(R.p + row_off + trel<32>(R.qstride, tile_in_block(2 * q + h, r >> 2)), o);
    }
    if constexpr (ENC && SRC::kSse) {
      if (src.want_sse()) {
        int o[16], rc[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const s4v ov = *reinterpret_cast<const s4v *>(V.org + row_off + trel<32>(R.qstride, tile_in_block(2 * q + h, r >> 2)));
          const int p0 = pred2[2 * q] & 0xffff, p1 = pred2[2 * q] >> 16, p2 = pred2[2 * q + 1] & 0xffff, p3 = pred2[2 * q + 1] >> 16;
          const int pr4[4] = {p0, p1, p2, p3};
#pragma unroll
          for (int k = 0; k < 4; k++) o[4 * q + k] = ov[k], rc[4 * q + k] = clip3(0, mx, pr4[k] + out[4 * q + k]);
        }
        const unsigned d = (unsigned)group_sum((int)sse_samples<16>(o, rc, P.bit_depth), 64);
        if (lane == 0) V.sse[b0 >> 4] = d;
      }
    }
    wave_sync();
  }
}

// A 256-thread workgroup moves a 64 x 64 region, a thread one 4x4 tile: four 8-byte accesses on the plane side (16
// consecutive threads cover a 128-byte line of each row) and the tile's 32 contiguous bytes on the tiled side (the
// four threads of a quad complete its 128-byte line).  grid = (picture, region, plane): consecutive workgroups take
// the same region of consecutive pictures, which are consecutive lines of the interleaved pool.
// Rows [y0, y1) of the luma plane (and the chroma rows below them): a band of CTU rows, so that the conversion of one
// band can overlap the dependency chain working on the others.
template <bool TO_TILED>
__global__ __launch_bounds__(256) void k_convert_tiled(const ConvJob *jobs, int y0, int y1) {
  const ConvJob J = jobs[blockIdx.x * 3 + blockIdx.z];
  const int stride = J.stride, w = J.w, h = J.h;
  const TiledPlane T = J.T;
  const int c = blockIdx.z ? 1 : 0;
  const int spr = ((T.ctu_w << T.clog) + 63) >> 6; // regions per row of this plane
  const int sx = blockIdx.y % spr, sy = blockIdx.y / spr;
  const int x = (sx << 6) + ((threadIdx.x & 15) << 2), y = (y0 >> c) + (sy << 6) + ((threadIdx.x >> 4) << 2);
  const int yend = min(h, y1 >> c);
  if (x >= w || y >= yend) return;
  short *tp = T.p + tphys(T.qstride, tile_base(T.ctu_w, T.clog, x, y));
  short *pp = J.plane + (size_t)y * stride + x;
  const bool vec = (((reinterpret_cast<uintptr_t>(pp) | (uintptr_t)(2 * stride)) & 7) == 0) && x + 4 <= w && y + 4 <= yend;
  if (vec) { // the common case: whole tile inside the picture, 8-byte aligned plane rows
    if (TO_TILED) {
      s4v r[4];
#pragma unroll
      for (int k = 0; k < 4; k++) r[k] = *reinterpret_cast<const s4v *>(pp + (size_t)k * stride);
#pragma unroll
      for (int k = 0; k < 4; k++) *reinterpret_cast<s4v *>(tp + 4 * k) = r[k];
    } else {
      s4v r[4];
#pragma unroll
      for (int k = 0; k < 4; k++) r[k] = *reinterpret_cast<const s4v *>(tp + 4 * k);
#pragma unroll
      for (int k = 0; k < 4; k++) *reinterpret_cast<s4v *>(pp + (size_t)k * stride) = r[k];
    }
    return;
  }
  for (int j = 0; j < 4 && y + j < yend; j++)
    for (int k = 0; k < 4 && x + k < w; k++) {
      if (TO_TILED)
        tp[4 * j + k] = pp[(size_t)j * stride + k];
      else
        pp[(size_t)j * stride + k] = tp[4 * j + k];
    }
}

// Level-synchronous schedule: one launch per picture-wide dependency level.  Every block of a level
// is independent of every other, so the launch is a plain list kernel: blockIdx.y = picture,
// blockIdx.x = chunk of 64/N blocks (one 32x32 block) of that picture's level, all sizes in one grid.
struct LevelArgs {
  const PicWork *pics;
  int level;
  // pictures that share one plan: the level's row and block list travel as kernel arguments, so a
  // wave can fetch its block descriptors without first chasing the picture table (two dependent
  // memory hops less on the critical path of every launch)
  int shared;
  LevelRow row;
  const FTu *ltus;
  PicDev P;
};
template <bool ENC>
__global__ __launch_bounds__(64, 4) void k_intra_level(LevelArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[HMX_WAVE_SMEM];
  const PicWork &W = A.pics[blockIdx.y];
  if (!A.shared && A.level >= W.n_levels) return;
  const LevelRow row = A.shared ? A.row : W.ltab[A.level];
  const FTu *ltus = A.shared ? A.ltus : W.ltus;
  int c = blockIdx.x;
#pragma unroll
  for (int s = 3; s >= 0; s--) { // largest blocks first (see k_intra_level_across)
    const int per = s == 3 ? 1 : (16 >> (2 * s)) * 1; // blocks per wave: 16, 8(=64/8), 4, 1
    const int slots = s == 0 ? kSlots4Own : s == 1 ? 8 : s == 2 ? 4 : 1;
    (void)per;
    const int chunks = (int)(row.count[s] + slots - 1) / slots;
    if (c < chunks) {
      const FTu *tus = ltus + row.start[s] + (size_t)c * slots;
      const int n = min(slots, (int)row.count[s] - c * slots);
      const OwnPicture src{W, tus};
      if (s == 0) {
        if constexpr (kSlots4Own == 64) wave_chain_4_lane<ENC, true>(smem, src, A.P, n);
        else wave_chain_valu<4, ENC, true>(smem, src, A.P, n);
      } else if (s == 1) wave_chain_valu<8, ENC, true>(smem, src, A.P, n);
      else if (s == 2) wave_chain_valu<16, ENC, true>(smem, src, A.P, n);
      else wave_chain_32<ENC, true>(smem, src, A.P, n);
      return;
    }
    c -= chunks;
  }
}

// Level schedule for pictures that follow ONE plan: a wave takes one block of the level and works
// it for 64/N pictures at once (see AcrossPictures).  1-D grid: for each size class, count x cpb
// waves, cpb = picture chunks per block.
struct AcrossArgs {
  const PicWork *pics;
  const FTu *ltus;
  LevelRow row;
  int n_pics;
  uint32_t cpb[4];
  const short *pool_org;
  short *pool_rec;
  size_t pic_elems;
  uint32_t plane_off[3];
  int ctu_w, clog;
  PicDev P;
};
template <bool ENC>
__global__ __launch_bounds__(64, 4) void k_intra_level_across(AcrossArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[HMX_WAVE_SMEM];
  uint32_t c = blockIdx.x;
#pragma unroll
  // The size classes of a level in descending block size: waves are dispatched in workgroup order, the 32x32 waves
  // run longest, and a level lasts until its last wave ends (4x4 first: 84.4 Gpx/s, 32x32 first: 87.4, 1536 pictures).
  for (int s = 3; s >= 0; s--) {
    const int slots = s == 0 ? kSlots4 : s == 1 ? 8 : s == 2 ? 4 : 1;
    const uint32_t waves = A.row.count[s] * A.cpb[s];
    if (c < waves) {
      const uint32_t blk = c / A.cpb[s], chunk = c - blk * A.cpb[s];
      const int pic0 = (int)chunk * slots, n = min(slots, A.n_pics - pic0);
      const AcrossPictures src{A.pics, A.ltus + A.row.start[s] + blk, pic0, A.n_pics, A.pool_org, A.pool_rec,
                               A.plane_off[1], A.plane_off[2] - A.plane_off[1], A.ctu_w, A.clog};
      if (s == 0) {
        if constexpr (kSlots4 == 64) wave_chain_4_lane<ENC, true>(smem, src, A.P, n);
        else wave_chain_valu<4, ENC, true>(smem, src, A.P, n);
      } else if (s == 1) wave_chain_valu<8, ENC, true>(smem, src, A.P, n);
      else if (s == 2) wave_chain_valu<16, ENC, true>(smem, src, A.P, n);
      else wave_chain_32<ENC, true>(smem, src, A.P, n);
      return;
    }
    c -= waves;
  }
}

template <bool ENC>
__global__ __launch_bounds__(64, 4) void k_intra_wave(FrameArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[HMX_WAVE_SMEM];
  int w = blockIdx.x;
  const int plane = w % 3;
  w /= 3;
  const int ctu = A.wave_ctus[w % A.n_wave_ctus];
  const PicWork &W = A.pics[w / A.n_wave_ctus];
  const uint32_t sb = W.seg_range[(ctu * 3 + plane) * 2], se = W.seg_range[(ctu * 3 + plane) * 2 + 1];
  for (uint32_t s = sb; s < se; s++) {
    const Seg sg = W.segs[s];
    const FTu *tus = W.tus + sg.start;
    // a new dependency level gathers references from the reconstruction written by the previous one
    if (sg.new_level) wave_global_sync();
    wave_sync(); // the LDS scratch is re-interpreted per block size
    const OwnPicture src{W, tus};
    switch (sg.log2n) {
    case 2: wave_chain_4_lane<ENC>(smem, src, A.P, sg.count); break;
    case 3: wave_chain_valu<8, ENC>(smem, src, A.P, sg.count); break;
    case 4: wave_chain_valu<16, ENC>(smem, src, A.P, sg.count); break;
    default: wave_chain_32<ENC>(smem, src, A.P, sg.count); break;
    }
  }
}

// =============================================================================================
// Packed schedule: ONE persistent launch per whole-picture call.
//
// The level schedules pay one kernel launch per picture-wide dependency level (4844 at 2160p) and every launch lasts at
// least one block-chain latency however little work it carries.  Here the dependency order lives in memory instead:
//   * pictures form GROUPS of I <= 64 (the interleave domain of the working pool); a ROW = (dependency level, group)
//     holds every block of that level of the group's pictures -- each picture following ITS OWN plan -- bucketed by
//     transform size.  A WAVE-ITEM is one wave's worth of a bucket: 64/N blocks (one 32x32 block) taken from whichever
//     pictures have them, so waves are full whether the pictures share a plan or not (per item: picture + descriptor);
//   * groups are dealt to SHARDS (group mod n_shards, at most 8); the wave-items of a shard are numbered row after row,
//     level-major (tickets).  A persistent wave draws the next ticket of its shard with an atomic add, WAITS until the
//     previous row of the same group is complete (one counter per row, polled with an L1-bypassing load), runs the
//     block chain, drains its stores and adds 1 to its row's counter.
// A shard belongs to ONE XCD: the first wave that touches it claims it for the XCD it runs on (compare-and-swap on the
// shard's owner word with the hardware's XCC id; a wave starts at the shard with its XCD's number, moves on to shards its
// XCD already owns or that nobody owns when those are drained, and never works on another XCD's).  So every producer and
// every consumer of a group's reconstruction runs on the same XCD BY CONSTRUCTION -- read from the hardware, not assumed
// from the dispatch order -- and the hand-off stays inside that XCD's L2: plain stores (the vector L1 is write-through;
// a store whose vmcnt has drained is in the L2), loads that bypass the L1 (sc1), no write-through to HBM and no round
// trip to it on the dependency path.  An XCD is a 32-CU machine with its own L2; this schedule runs eight of them side
// by side on independent pictures.
// Forward progress: a wave waits only for wave-items with SMALLER tickets of the same shard, and a ticket is drawn by a
// wave that is already running, in ticket order.  So the unfinished wave-item with the smallest ticket of a shard is always
// held by a running wave whose own dependencies are complete: it finishes, and by induction all do, whatever the number of
// resident waves, the dispatch order or the placement (an XCD that gets no wave of the launch owns nothing: its shards
// are claimed by the waves of another XCD once those have drained their own).  There is no barrier between workgroups.
// (A spin that exceeds ~2^22 polls -- seconds -- raises the abort word and every wave leaves: a bug fails loudly.)
// Rows of different groups are independent, so while one group waits for its row's last wave-item the others compute.
// Latency hiding inside a wave: the ticket, the descriptor and the items of the NEXT wave-item are fetched while the current
// one runs (ticket drawn before the chain, descriptor loaded behind the dependency poll, items loaded behind the chain's
// stores), so that a wave-item starts with its block descriptors in registers.
// Reference for the dependency a row encodes: TLibCommon/TComPattern.cpp:389-425 (which neighbours a block reads).
// =============================================================================================
#ifdef HMX_CHAIN_MAIN
// prep 1: blocks per size class of every row (one wave per row, lane = picture of the group)
// (a wave covers 64 / I rows: lane = (row of the wave, picture of the group); the first cut spent one wave per row with
// I <= 4 lanes at work -- 9 M workgroups for 2048 pictures of 2160p, 115 ms of launch overhead for the fill alone)
__device__ __forceinline__ int seg_sum(int v, int I, int k) { // sum over the I lanes of a segment (k = lane in segment), any I <= 64
  const int lane = threadIdx.x, base = lane - k;
  int t = 0;
  for (int q = 0; q < I; q++) t += __shfl(v, base + q, 64);
  return t;
}
__global__ __launch_bounds__(64) void k_pack_count(const PackPic *pics, PackRow *rows, PackGeom G, int n_rows) {
  const int rpw = 64 / G.I, sub = threadIdx.x / G.I, k = threadIdx.x - sub * G.I;
  const int row = blockIdx.x * rpw + sub;
  const bool live = sub < rpw && row < n_rows;
  const int L = live ? row / G.n_groups : 0, g = live ? row - L * G.n_groups : 0;
  const int pic = g * G.I + k;
  uint32_t c[4] = {0, 0, 0, 0};
  if (live && pic < G.n_pics && L < pics[pic].n_levels) {
    const LevelRow r = pics[pic].ltab[L];
#pragma unroll
    for (int s = 0; s < 4; s++) c[s] = r.count[s];
  }
#pragma unroll
  for (int s = 0; s < 4; s++) c[s] = (uint32_t)seg_sum((int)c[s], G.I, k);
  if (live && k == 0) {
    PackRow R{};
    uint32_t nw = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      R.count[s] = c[s];
      const uint32_t sl = pack_slots(s, G.slots4, G.slots8);
      nw += (c[s] + sl - 1) / sl;
    }
    R.n_waves = nw;
    rows[row] = R;
  }
}
// position p of the ticket order (shard-major, then level, then group) -> row
__device__ __forceinline__ int pack_row_at(const PackGeom &G, int p, int &shard) {
  int sh = 0;
  for (;; sh++) {
    const int ng = (G.n_groups - sh + G.n_shards - 1) / G.n_shards, n = ng * G.max_levels;
    if (p < n || sh == G.n_shards - 1) {
      shard = sh;
      const int L = p / ng, gi = p - L * ng;
      return L * G.n_groups + sh + gi * G.n_shards;
    }
    p -= n;
  }
}
// prep 2: exclusive prefix of wave-items and items over the rows in ticket order.  Two launches of kPackScanWgs workgroups: the
// first leaves every workgroup's totals in `part`, the second turns the totals before it into its base and scans its rows (one
// workgroup for 2.25 M rows took 5.4 ms of every call with new plans).
constexpr int kPackScanWgs = 128;
template <bool APPLY>
__global__ __launch_bounds__(1024) void k_pack_scan(PackRow *rows, PackHdr *hdr, PackGeom G, uint32_t *part) {
  __shared__ uint32_t sw[1024], si[1024];
  __shared__ uint32_t base_w, base_i;
  const int n_rows = G.max_levels * G.n_groups, tid = threadIdx.x;
  const int per_wg = (n_rows + kPackScanWgs - 1) / kPackScanWgs, wg_lo = min((int)blockIdx.x * per_wg, n_rows), wg_hi = min(wg_lo + per_wg, n_rows);
  const int chunk = (wg_hi - wg_lo + 1023) / 1024, lo = min(wg_lo + tid * chunk, wg_hi), hi = min(lo + chunk, wg_hi);
  uint32_t w = 0, it = 0;
  for (int p = lo; p < hi; p++) {
    int sh;
    const PackRow &R = rows[pack_row_at(G, p, sh)];
    w += R.n_waves;
    it += R.count[0] + R.count[1] + R.count[2] + R.count[3];
  }
  sw[tid] = w, si[tid] = it;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) { // inclusive scan
    const uint32_t a = tid >= off ? sw[tid - off] : 0, b = tid >= off ? si[tid - off] : 0;
    __syncthreads();
    sw[tid] += a, si[tid] += b;
    __syncthreads();
  }
  if constexpr (!APPLY) {
    if (tid == 1023) part[2 * blockIdx.x] = sw[1023], part[2 * blockIdx.x + 1] = si[1023];
    return;
  } else {
    if (tid == 0) {
      uint32_t bw = 0, bi = 0;
      for (int q = 0; q < (int)blockIdx.x; q++) bw += part[2 * q], bi += part[2 * q + 1];
      base_w = bw, base_i = bi;
    }
    __syncthreads();
    uint32_t wb = base_w + sw[tid] - w, ib = base_i + si[tid] - it;
    int prev_shard = lo > 0 ? -2 : -1; // -2: find out
    if (lo > 0 && lo < n_rows) pack_row_at(G, lo - 1, prev_shard);
    for (int p = lo; p < hi; p++) {
      int sh;
      PackRow &R = rows[pack_row_at(G, p, sh)];
      if (sh != prev_shard)
        for (int q = prev_shard + 1; q <= sh; q++) hdr->shard_base[q] = wb; // empty shards in between do not occur, but stay safe
      prev_shard = sh;
      R.wave_base = wb;
      wb += R.n_waves;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        R.item_base[s] = ib;
        ib += R.count[s];
      }
    }
    if (blockIdx.x == kPackScanWgs - 1 && tid == 1023) {
      for (int q = G.n_shards; q <= 8; q++) hdr->shard_base[q] = base_w + sw[1023];
      hdr->total_items = base_i + si[1023];
    }
  }
}
// prep 3: the wave-item descriptors and the item array of one (row, size class) per wave.  Items of a bucket are ordered
// by rank inside their picture's bucket, then by picture: pictures that share a plan put the SAME block of consecutive
// pictures on consecutive lanes (consecutive lines of the interleaved pool, uniform control flow); pictures with their
// own plans put blocks of similar code path (the plan sorts a bucket by plane, transform skip, mode) next to each other.
// One workgroup per (dependency level, picture), one wave per size class, a LANE per item (rank r of the picture's bucket, 64 at a
// time): the item is read where the plan has it and written where its row wants it -- both ends coalesced over the wave.  (The
// first cut gave each picture of a row ONE lane that walked its bucket entry by entry: a chain of dependent loads per lane,
// 30 ms per 2048 pictures of 2160p when every step brings new plans.)  The wave of the group's first picture also writes the
// row's wave-item descriptors.
constexpr int kPackFillLevels = 8; // levels per workgroup: their table rows are fetched together, then the items move
__global__ __launch_bounds__(256) void k_pack_fill(const PackPic *pics, const PackRow *rows, PackDesc *descs, FTu *items, PackGeom G) {
  const int pic = blockIdx.y, s = 3 - (int)(threadIdx.x >> 6), lane = threadIdx.x & 63; // (largest blocks first in a row)
  const int g = pic / G.I, k = pic - g * G.I, L0 = blockIdx.x * kPackFillLevels;
  const bool have = pic < G.n_pics; // (a ragged last group)
  const int n_levels = have ? pics[pic].n_levels : 0;
  const FTu *ltus0 = have ? as_global(pics[pic].ltus) : nullptr;
  const LevelRow *ltab = have ? as_global(pics[pic].ltab) : nullptr;
  // lane q < I looks at picture q of the group: its level table, for the counts of this (level, size) over the group
  const bool peer = lane < G.I && g * G.I + lane < G.n_pics;
  const int peer_levels = peer ? pics[g * G.I + lane].n_levels : 0;
  const LevelRow *peer_tab = peer ? as_global(pics[g * G.I + lane].ltab) : nullptr;
  // phase 1: everything the levels of this workgroup need from the tables, all loads in flight together
  uint32_t wave_base[kPackFillLevels], item_base[kPackFillLevels], dep[kPackFillLevels], cnt[kPackFillLevels], start[kPackFillLevels], cq[kPackFillLevels];
  uint32_t rcount[kPackFillLevels][4];
#pragma unroll
  for (int q = 0; q < kPackFillLevels; q++) {
    const int L = L0 + q;
    const bool on = L < G.max_levels;
    const PackRow *R = rows + (size_t)(on ? L : 0) * G.n_groups + g;
    wave_base[q] = R->wave_base, item_base[q] = R->item_base[s];
#pragma unroll
    for (int t = 0; t < 4; t++) rcount[q][t] = on ? R->count[t] : 0;
    dep[q] = (on && L > 0) ? (R - G.n_groups)->n_waves : 0;
    const bool mine = on && L < n_levels;
    cnt[q] = mine ? ltab[L].count[s] : 0, start[q] = mine ? ltab[L].start[s] : 0;
    cq[q] = (on && L < peer_levels) ? peer_tab[L].count[s] : 0;
  }
  // phase 2: descriptors (the group's first picture) and items
#pragma unroll
  for (int q = 0; q < kPackFillLevels; q++) {
    const int L = L0 + q;
    if (L >= G.max_levels) break;
    if (k == 0) { // the row's wave-items of this size class: they follow those of the larger classes
      const uint32_t sl = pack_slots(s, G.slots4, G.slots8), total = rcount[q][s], nw = (total + sl - 1) / sl;
      uint32_t woff = 0;
      for (int t = 3; t > s; t--) woff += (rcount[q][t] + pack_slots(t, G.slots4, G.slots8) - 1) / pack_slots(t, G.slots4, G.slots8);
      const uint32_t row = (uint32_t)(L * G.n_groups + g);
      for (uint32_t c = (uint32_t)lane; c < nw; c += 64u)
        descs[wave_base[q] + woff + c] = PackDesc{item_base[q] + c * sl, min(sl, total - c * sl) | ((uint32_t)s << 28), row, dep[q]};
    }
    // an item (rank r, picture k) sits behind every item of a lower rank and the same-rank items of the pictures before k:
    //   sum over k' of min(count[k'], r)  +  #{k' < k : count[k'] > r}
    const FTu *ltus = ltus0 + start[q];
    FTu *dst = items + item_base[q];
    for (uint32_t r0 = 0; r0 < cnt[q]; r0 += 64u) { // (whole rounds: the shuffles below need every lane)
      const uint32_t r = r0 + (uint32_t)lane;
      uint32_t off = 0;
      for (int t = 0; t < G.I; t++) {
        const uint32_t c2 = (uint32_t)__shfl((int)cq[q], t, 64);
        off += min(c2, r) + ((t < k && c2 > r) ? 1u : 0u);
      }
      if (r < cnt[q]) {
        FTu f = ltus[r];
        f.t.plane = (uint8_t)(f.t.plane | (k << 2)); // picture of the group in the upper six bits
        dst[off] = f;
      }
    }
  }
}

#endif // HMX_CHAIN_MAIN

// The completion counters of different rows live on different 128-byte lines: the adds of a row's wave-items serialise
// on their word anyway (~12 ns each), but with neighbouring rows on one line every add and every poll of a whole level --
// all groups -- queued on ONE L2 channel (measured: the first cut ran 5x slower than the level schedule it replaces).
constexpr uint32_t kDoneStride = 32;
struct PackArgs {
  const PackPic *pics;
  const PackRow *rows;
  const PackDesc *descs;
  const FTu *items;
  uint32_t *done; // [rows][kDoneStride] completed wave-items, one 128-byte line per row (see kDoneStride)
  int sleep0, sleep1; // poll back-off in units of 64 clocks: previous row not started / in progress
  PackHdr *hdr;
  const short *pool_org;
  short *pool_rec;
  size_t pic_elems;      // one picture, three planes
  uint32_t plane_off[3]; // of one picture
  int ctu_w, clog;
  int n_groups, n_shards, I;
  // level buffers of the call laid out as ONE slab per plane (picture i at lev_base[p] + i * lev_pic_elems[p], one stride):
  // a wave-item then addresses its levels by arithmetic instead of a load from the picture table on its way to the wait
  int want_sse; // encoder direction: write xGetSSE(org, rec) of every block through PackPic::sse
  int lev_slab;
  int *lev_base[3];
  long long lev_pic_elems[3];
  int lev_stride[3];
  PicDev P;
  RdoqChain rq; // RDOQ variant of the kernel only
};
// what a wave-item of the packed schedule prefetches for its successor (see k_intra_packed)
struct PackNext {
  uint32_t ticket_raw; // lane 0: the ticket drawn for the next wave-item (the atomic's return value)
  uint32_t base, total;
  const PackDesc *descs;
  uint32_t t;          // the next ticket, wave-uniform (valid after the dependency wait)
  PackDesc d;          // its descriptor (in flight after the dependency wait)
};
template <bool SSE, bool RDOQ = false>
struct PackedSrc {
#ifdef HMX_PACKED_L1INV
  static constexpr bool kCoherent = false;     // A/B: the vector L1 invalidated behind the dependency wait, plain loads after it
#else
  static constexpr bool kCoherent = true;      // reconstruction loads bypass the vector L1 (sc1)
#endif
  static constexpr bool kWriteThrough = false; // producers and consumers share an XCD's L2: plain stores
  static constexpr bool kSse = SSE;            // a kernel variant of its own: the extra live registers would spill in the common one
  static constexpr bool kRdoq = RDOQ;          // likewise (doubles, and the 4x4 lane's private arrays)
  __device__ __forceinline__ bool want_sse() const { return SSE; }
  __device__ __forceinline__ int picture() const { return pic0 + (int)(ft.t.plane >> 2); }
  __device__ __forceinline__ int cbf_ctx() const { return (ft.t.flags >> 4) & 15; } // hmx_tu::flags bits 4..7
  __device__ __forceinline__ const RdoqChain &rdoq() const { return A->rq; }
  __device__ __forceinline__ RdoqWaveLds &rdoq_lds() const { return *rq_lds; }
  __device__ __forceinline__ int group_slot() const { return (int)(ft.t.plane >> 2); } // the picture's index in its group
  FTu ft;               // this lane's item, fetched during the previous wave-item
  const PackPic *gpics; // the group's pictures
  const short *org_g;   // the group's region of the pools
  short *rec_g;
  uint32_t off1, off2;  // plane offsets inside the group region
  uint32_t qstride;
  int ctu_w, clog_luma;
  const uint32_t *dep;  // counter of the previous row of the group (NULL: first level)
  uint32_t target;
  uint32_t *abort_word;
  int sleep0, sleep1;
  PackNext *nx;
  const PackArgs *A;
  int pic0; // first picture of the group
  RdoqWaveLds *rq_lds; // RDOQ variant: the wave's tables and buffers
#ifdef HMX_PACK_PROFILE
  unsigned long long *pt; // [0] wait entry, [1] wait exit, [2] polls
#endif
  __device__ __forceinline__ FTu desc(int) const { // the chains ask for item i = lane / (lanes per block): that is what was fetched
    FTu f = ft;
    f.t.plane &= 3;
    return f;
  }
  __device__ __forceinline__ PlaneView view(int, int pl) const {
    const unsigned k = ft.t.plane >> 2;
    const size_t o = (size_t)(pl == 0 ? 0u : pl == 1 ? off1 : off2) + (size_t)k * 64;
    int *lv;
    int ls;
    if (A->lev_slab) {
      int *const b = pl == 0 ? A->lev_base[0] : pl == 1 ? A->lev_base[1] : A->lev_base[2];
      const long long e = pl == 0 ? A->lev_pic_elems[0] : pl == 1 ? A->lev_pic_elems[1] : A->lev_pic_elems[2];
      lv = b + (long long)(pic0 + (int)k) * e;
      ls = pl == 0 ? A->lev_stride[0] : pl == 1 ? A->lev_stride[1] : A->lev_stride[2];
    } else {
      const char *row = reinterpret_cast<const char *>(&gpics[k]);
      lv = *reinterpret_cast<int *const *>(row + offsetof(PackPic, lev) + pl * sizeof(int *));
      ls = *reinterpret_cast<const int *>(row + offsetof(PackPic, lev_stride) + pl * sizeof(int));
    }
    uint32_t *sp = nullptr;
    if constexpr (SSE) sp = as_global(*reinterpret_cast<uint32_t *const *>(reinterpret_cast<const char *>(&gpics[k]) + offsetof(PackPic, sse) + pl * sizeof(uint32_t *)));
    return PlaneView{org_g + o, TiledPlane{rec_g + o, ctu_w, pl ? clog_luma - 1 : clog_luma, qstride}, as_global(lv), ls, sp};
  }
  // Wait until the previous row of the group is complete.  One L1-bypassing load per poll (the whole wave reads one
  // word: one request); everything the chain loads from the reconstruction afterwards is an sc1 load issued after this
  // loop has seen the count, and the producers' stores had reached the L2 (vmcnt drained) before they counted.
  __device__ __forceinline__ void wait() const {
#ifdef HMX_PACK_PROFILE
    pt[0] = wall_clock64();
#endif
    if (dep) {
      unsigned spins = 0;
      for (;;) {
#ifdef HMX_PACK_PROFILE
        pt[2]++;
#endif
        const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32 *)dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= target) break;
        // back off: a row that has not finished a single wave-item is at least one block chain away, one in progress
        // completes within a few hundred nanoseconds; every poll is a request to the L2 channel the producers add on
        for (int q = v == 0 ? sleep0 : sleep1; q > 0; q--) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
          if (spins >= (1u << 22)) __hip_atomic_store((gu32 *)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32 *)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
        }
      }
    }
#ifdef HMX_PACKED_L1INV
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // buffer_inv sc1
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler ordering: no reconstruction load moves above the poll
#endif
    // the next wave-item's ticket has long returned: fetch its descriptor behind the reference loads that follow
    nx->t = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx->ticket_raw);
    if (nx->t < nx->total) nx->d = nx->descs[nx->base + nx->t];
#ifdef HMX_PACK_PROFILE
    pt[1] = wall_clock64();
#endif
  }
};

// item index of a lane inside a wave-item of size class s: lane / (lanes per block)
template <int SL4, int SL8>
__device__ __forceinline__ int pack_lane_item(int lane, int s) {
  return s == 0 ? (SL4 == 64 ? lane : lane >> 2) : s == 1 ? (SL8 == 16 ? lane >> 2 : lane >> 3) : s == 2 ? lane >> 4 : 0;
}

template <bool ENC, int SL4, bool SSE = false, bool RDOQ = false, int SL8 = 8>
#ifndef HMX_PACKED_OCC
#define HMX_PACKED_OCC 4 /* waves per SIMD the register allocation is held to (A/B: -DHMX_PACKED_OCC=5 | 6 spill) */
#endif
__global__ __launch_bounds__(64, RDOQ ? 2 : HMX_PACKED_OCC) void k_intra_packed(PackArgs A) {
  static_assert(!RDOQ || (ENC && SL4 == 64 && SL8 == 8), "RDOQ: encoder direction, 4x4 blocks one per lane, 8x8 blocks on eight lanes (rdoq_wave_tiles)");
  static_assert(SL8 == 8 || (SL8 == 16 && 16 * sizeof(TuLds8x2) <= HMX_WAVE_SMEM), "8x8 blocks per wave-item");
  // RDOQ variant: every byte of LDS decides how many waves a CU holds (the walks are latency chains); the lane-per-block 4x4
  // chain, the largest user of the common scratch, borrows the round buffer its RDOQ does not need
  constexpr int kSmem = RDOQ ? (int)(4 * sizeof(TuLds<16>)) : HMX_WAVE_SMEM;
  static_assert(!RDOQ || (8 * sizeof(TuLds<8>) <= kSmem && sizeof(TuLds<32>) <= kSmem && sizeof(Lane4Lds) <= sizeof(RdoqWaveLds::u)), "RDOQ variant: LDS scratch");
  __shared__ __attribute__((aligned(16))) char smem[kSmem];
  __shared__ __attribute__((aligned(16))) char rq_raw[RDOQ ? sizeof(RdoqWaveLds) : 16];
  RdoqWaveLds *rq_lds = reinterpret_cast<RdoqWaveLds *>(rq_raw);
  char *const smem4 = RDOQ ? rq_lds->u.lane4 : smem;
  if constexpr (RDOQ) {
    if (lane_id() == 0) rq_lds->key = 0;
    wave_sync();
  }
  int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15;
  PackHdr *hdr = A.hdr;
#ifdef HMX_PACK_PROFILE
  unsigned long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt[3] = {0, 0, 0};
  const unsigned long long t_start = wall_clock64();
#define PROF_T(v) const unsigned long long v = wall_clock64()
#else
#define PROF_T(v)
#endif
  for (int si = 0; si < A.n_shards; si++) {
    const int s0 = xcc % A.n_shards, sh = s0 + si < A.n_shards ? s0 + si : s0 + si - A.n_shards;
    // whose shard?  mine if my XCD claimed it or nobody has yet (then it is mine from now on)
    uint32_t own = 0;
    if (lane_id() == 0) {
      own = __hip_atomic_load((gu32 *)&hdr->owner[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (own == 0) {
        uint32_t expect = 0;
        own = __hip_atomic_compare_exchange_strong((gu32 *)&hdr->owner[sh][0], &expect, (uint32_t)xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)
                  ? (uint32_t)xcc + 1u
                  : expect;
      }
    }
    own = (uint32_t)__builtin_amdgcn_readfirstlane((int)own);
    if (own != (uint32_t)xcc + 1u) continue;
    if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32 *)&hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
    PackNext nx;
    nx.base = hdr->shard_base[sh], nx.total = hdr->shard_base[sh + 1] - nx.base;
    nx.descs = A.descs;
    gu32 *ticket = (gu32 *)&hdr->ticket[sh][0];
    // prologue: the first wave-item's ticket, descriptor and items, unhidden
    PROF_T(p0);
    uint32_t t = 0;
    if (lane_id() == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= nx.total) continue;
    PackDesc d = A.descs[nx.base + t];
    FTu ft;
    {
      const int s = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)d.n_s) >> 28), n = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)d.n_s) & 0x0fffffffu);
      ft = A.items[(uint32_t)__builtin_amdgcn_readfirstlane((int)d.item_off) + (uint32_t)min(pack_lane_item<SL4, SL8>(lane_id(), s), n - 1)];
    }
    PROF_T(p1);
#ifdef HMX_PACK_PROFILE
    acc[0] += p1 - p0;
#endif
    for (;;) {
      PROF_T(p2);
      // draw the NEXT ticket now: its latency hides behind this wave-item
      nx.ticket_raw = 0;
      if (lane_id() == 0) nx.ticket_raw = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d.item_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.item_off), d.n_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.n_s);
      d.row = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.row), d.dep_target = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.dep_target);
      const int s = (int)(d.n_s >> 28), n = (int)(d.n_s & 0x0fffffffu);
      const int g = (int)(d.row % (uint32_t)A.n_groups);
      const size_t greg = (size_t)g * A.I * A.pic_elems;
      const PackedSrc<SSE, RDOQ> src{ft, A.pics + (size_t)g * A.I, A.pool_org + greg, A.pool_rec + greg,
                          A.plane_off[1] * (uint32_t)A.I, A.plane_off[2] * (uint32_t)A.I, 64u * (uint32_t)A.I, A.ctu_w, A.clog,
                          d.dep_target ? A.done + (size_t)(d.row - (uint32_t)A.n_groups) * kDoneStride : nullptr, d.dep_target, &hdr->abort,
                          A.sleep0, A.sleep1, &nx, &A, g * A.I, rq_lds
#ifdef HMX_PACK_PROFILE
                          , pt
#endif
      };
      wave_sync(); // the LDS scratch is re-interpreted per block size
      if constexpr (RDOQ) rdoq_stage_tables(*rq_lds, A.rq, g, s, A.I, lane_id());
      PROF_T(p3);
      if (s == 0) {
        if constexpr (SL4 == 64) wave_chain_4_lane<ENC, true>(smem4, src, A.P, n);
        else wave_chain_valu<4, ENC, true>(smem, src, A.P, n);
      } else if (s == 1) {
        if constexpr (SL8 == 16) wave_chain_8x2<ENC>(smem, src, A.P, n);
        else wave_chain_valu<8, ENC, true>(smem, src, A.P, n);
      } else if (s == 2) wave_chain_valu<16, ENC, true>(smem, src, A.P, n);
      else wave_chain_32<ENC, true>(smem, src, A.P, n);
      // the next wave-item's items, behind this one's stores (its descriptor was fetched behind the dependency poll)
      const bool more = nx.t < nx.total;
      FTu ftn = ft;
      if (more) {
        const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx.d.n_s);
        ftn = A.items[(uint32_t)__builtin_amdgcn_readfirstlane((int)nx.d.item_off) +
                      (uint32_t)min(pack_lane_item<SL4, SL8>(lane_id(), (int)(ns >> 28)), (int)(ns & 0x0fffffffu) - 1)];
      }
      // publish: every store of this wave has reached the L2 before the row's count moves
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      PROF_T(p4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PROF_T(p5);
      if (lane_id() == 0) __hip_atomic_fetch_add((gu32 *)(A.done + (size_t)d.row * kDoneStride), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef HMX_PACK_PROFILE
      const unsigned long long p6 = wall_clock64();
      acc[1] += p3 - p2, acc[2] += pt[0] - p3, acc[3] += pt[1] - pt[0], acc[4] += p4 - pt[1], acc[5] += p5 - p4, acc[6] += p6 - p5, acc[7] += 1;
#endif
      if (!more) break;
      d = nx.d, ft = ftn;
    }
  }
#ifdef HMX_PACK_PROFILE
  acc[8] = pt[2], acc[9] = wall_clock64() - t_start;
  if (lane_id() == 0)
    for (int q = 0; q < 10; q++) atomicAdd(&hdr->prof[q], acc[q]);
#endif
}


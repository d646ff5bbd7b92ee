// hmx_lib.hip -- libhmx: kernels + C-ABI (include/hmx.h) for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hmx_kernels.h"
#include "hmx_rdoq.h"

using namespace hmx;

// =============================================================================================
// Kernels
// =============================================================================================
struct DTu { // device descriptor of the list kernels: hmx_tu + index in the caller's order
  hmx_tu t;
  uint32_t idx;
};
__device__ __forceinline__ DTu load_dtu(const DTu *p) { // one 12-byte load instead of one per field read
  typedef __attribute__((address_space(1))) const int gint;
  int w[3];
  __builtin_memcpy(w, (gint *)p, 12);
  DTu d;
  __builtin_memcpy(&d, w, 12);
  return d;
}

// list kernels: 256-thread workgroups = four autonomous waves; blocks per workgroup
template <int N>
struct Slots {
  static constexpr int v = N == 64 ? 1 : N == 32 ? 4 : 256 / N; // 32x32 scratch (9.5 KB) is kept to four blocks; 64x64 (prediction only, 19.5 KB): one
};
#define HMX_SMEM_BYTES (16 * (int)sizeof(TuLds<16>)) /* largest of Slots<N> * sizeof(TuLds<N>) */
static_assert(64 * sizeof(TuLds<4>) <= HMX_SMEM_BYTES && 32 * sizeof(TuLds<8>) <= HMX_SMEM_BYTES &&
                  4 * sizeof(TuLds<32>) <= HMX_SMEM_BYTES && sizeof(TuLds<64>) <= HMX_SMEM_BYTES,
              "LDS scratch");

enum ListOp { OP_TRANSFORM_NXN, OP_INVTRANSFORM_NXN, OP_XT, OP_XIT, OP_XQUANT, OP_XDEQUANT, OP_PRED, OP_TRANSFORM_RECON };

struct ListPic { // planes of one picture of a multi-picture list call
  PlanesDev a, b;
  LevelsDev lev;
  PlanesDev rec; // OP_TRANSFORM_RECON: reconstruction out
};
struct ListArgs {
  const DTu *tus;
  int n;
  const ListPic *pics; // != NULL: grid.y pictures, planes from this table instead of a / b / lev
  int n_pics, abs_stride;
  PlanesDev a;   // residual in (transform) / prediction in (inverse with recon) / recon (pred)
  PlanesDev b;   // output planes
  LevelsDev lev; // levels / coefficients (Int)
  LevelsDev lev2;
  uint32_t *abs_sum;
  uint32_t *sse; // OP_TRANSFORM_RECON: xGetSSE(org, rec) per block, indexed like abs_sum (NULL = none)
  int have_pred;
  PlanesDev org;        // OP_PRED with cost: the original the predictions are costed against
  uint32_t *cost;       // OP_PRED: calcHAD of every (block, mode), [block idx][n_modes]; NULL = none
  const uint8_t *modes; // OP_PRED fan-out
  int n_modes;
  size_t mode_elems[3];
  PicDev P;
};

template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int i) { // a[i] without a run-time index (see k_list)
  T r = a[0];
  r = i == 1 ? a[1] : r;
  return i == 2 ? a[2] : r;
}
// The same for a table entry that is uniform over the wave: the three values are pinned as wave-uniform (scalar
// loads), or the compiler turns the select of loads back into one per-lane load from a selected address.
template <typename T>
__device__ __forceinline__ T uniform3v(T a0, T a1, T a2, int i) {
  const T v0 = wave_uniform(a0), v1 = wave_uniform(a1), v2 = wave_uniform(a2);
  T r = v0;
  r = i == 1 ? v1 : r;
  return i == 2 ? v2 : r;
}
#define uniform3(arr, i) uniform3v((arr)[0], (arr)[1], (arr)[2], (i))

// One kernel per (operation, block size): every block of the launch has size N.
template <int N, int OP>
__global__ __launch_bounds__(256) void k_list(ListArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[HMX_SMEM_BYTES];
  constexpr int SL = Slots<N>::v;
  const int tid = threadIdx.x, slot = tid / N, gl = tid % N;
  const bool lane_on = slot < SL;
  TuLds<N> &L = reinterpret_cast<TuLds<N> *>(smem)[lane_on ? slot : 0];
  const int i = blockIdx.x * SL + slot;
  const bool active = lane_on && i < A.n;
  const DTu d = load_dtu(A.tus + (active ? i : 0));
  const hmx_tu t = d.t;
  const int pl = t.plane, x = t.x, y = t.y;
  // blockIdx.y = picture of a multi-picture call (planes from the table); single calls carry theirs inline
  const ListPic *Q = A.pics ? A.pics + blockIdx.y : nullptr;
  // (members of the by-value argument struct are picked with constant indices: a run-time index into it makes
  // the compiler copy all of ListArgs to scratch memory, 22 stores per wave before the first useful load)
  // The picture table entry is uniform over the launch's x dimension: its fields come in by scalar loads and the
  // lane's plane selects among them, instead of one vector load per field and lane.
  short *a_p = Q ? uniform3(Q->a.p, pl) : pick3(A.a.p, pl), *b_p = Q ? uniform3(Q->b.p, pl) : pick3(A.b.p, pl);
  const int a_s = Q ? uniform3(Q->a.s, pl) : pick3(A.a.s, pl), b_s = Q ? uniform3(Q->b.s, pl) : pick3(A.b.s, pl);
  int *lev_p = Q ? uniform3(Q->lev.p, pl) : pick3(A.lev.p, pl), *lev2_p = pick3(A.lev2.p, pl);
  const int lev_s = Q ? uniform3(Q->lev.s, pl) : pick3(A.lev.s, pl), lev2_s = pick3(A.lev2.s, pl);
  uint32_t *abs_sum = A.abs_sum ? A.abs_sum + (size_t)blockIdx.y * A.abs_stride : nullptr;
  const bool luma = pl == 0, inter = t.flags & HMX_TU_INTER, ts = t.flags & HMX_TU_TRANSFORM_SKIP;
  const bool use_dst = luma && !inter; // uiMode != REG_DCT, only consulted for N == 4
  const int scan_idx = coef_scan_idx(N, luma, !inter, t.mode);
  int row[N];

  if constexpr (OP == OP_TRANSFORM_NXN || OP == OP_XT || OP == OP_TRANSFORM_RECON) {
    int pr[N];
    if (active) {
      load_row16<N>(a_p + (size_t)(y + gl) * a_s + x, row);
      if (A.have_pred) { // residual = original - prediction (TComYuv::subtract, TComYuv.cpp:461) fused in
        load_row16<N>(b_p + (size_t)(y + gl) * b_s + x, pr);
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = wrap16(row[k] - pr[k]);
      }
    }
    int sum = fwd_tq_block<N>(L, gl, active, row, ts, use_dst, luma, scan_idx, OP != OP_XT, A.P);
    if (active) {
      load_row32<N>(&L.tile[gl][0], row);
      if (OP != OP_XT) {
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = level_of(row[k]);
      }
      store_row32<N>(lev_p + (size_t)(y + gl) * lev_s + x, row);
      if (OP != OP_XT && gl == 0 && abs_sum) abs_sum[d.idx] = (uint32_t)sum;
    }
    if constexpr (OP == OP_TRANSFORM_RECON) { // the packed words are still in the tile: IQ, IT, Clip(pred + resi) in the same pass
      inv_tq_block<N>(L, gl, active, ts, use_dst, luma, true, A.P, row);
      if (active) {
        const int mx = (1 << A.P.bit_depth) - 1;
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = clip3(0, mx, pr[k] + row[k]);
        store_row16<N>(uniform3(Q->rec.p, pl) + (size_t)(y + gl) * uniform3(Q->rec.s, pl) + x, row);
      }
      if (A.sse) { // getDistPart(rec, org, DF_SSE) behind the reconstruction (TEncSearch.cpp:4990), in the same pass
        unsigned dsum = 0;
        if (active) {
          int o[N];
          load_row16<N>(a_p + (size_t)(y + gl) * a_s + x, o); // the original row again: it went into the residual
          dsum = sse_samples<N>(o, row, A.P.bit_depth);
        }
        dsum = (unsigned)group_sum((int)dsum, N);
        if (active && gl == 0) A.sse[(size_t)blockIdx.y * A.abs_stride + d.idx] = dsum;
      }
    }
  } else if constexpr (OP == OP_XQUANT) {
    // Int coefficients in lev -> levels in lev2 (the quantiser half of transformNxN on its own)
    if (active) load_row32<N>(lev_p + (size_t)(y + gl) * lev_s + x, row);
    int sum = quant_sbh_block<N, N, N, true>(
        L, gl, active, row, [&](int) { return gl; }, [&](int k) { return k; }, luma, scan_idx, A.P);
    if (active) {
      load_row32<N>(&L.tile[gl][0], row);
#pragma unroll
      for (int k = 0; k < N; k++) row[k] = level_of(row[k]);
      store_row32<N>(lev2_p + (size_t)(y + gl) * lev2_s + x, row);
      if (gl == 0 && abs_sum) abs_sum[d.idx] = (uint32_t)sum;
    }
  } else if constexpr (OP == OP_INVTRANSFORM_NXN || OP == OP_XIT) {
    if (active) {
      load_row32<N>(lev_p + (size_t)(y + gl) * lev_s + x, row);
      if (OP == OP_INVTRANSFORM_NXN) { // the tile holds packed words: xDeQuant's input clip happens here
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = clip3(-32768, 32767, row[k]) & 0xffff;
      }
      store_row32<N>(&L.tile[gl][0], row);
    }
    wave_sync();
    inv_tq_block<N>(L, gl, active, ts, use_dst, luma, OP == OP_INVTRANSFORM_NXN, A.P, row);
    if (active) {
      if (A.have_pred) {
        int pr[N];
        load_row16<N>(a_p + (size_t)(y + gl) * a_s + x, pr);
        const int mx = (1 << A.P.bit_depth) - 1;
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = clip3(0, mx, pr[k] + row[k]);
      }
      store_row16<N>(b_p + (size_t)(y + gl) * b_s + x, row);
    }
  } else if constexpr (OP == OP_XDEQUANT) {
    constexpr int LG = Log2<N>::v;
    const int tshift = 15 - A.P.bit_depth - LG, dshift = 6 - tshift, dadd = 1 << (dshift - 1);
    const QuantDev qd = pick_qd(A.P, luma);
    if (active) {
      load_row32<N>(lev_p + (size_t)(y + gl) * lev_s + x, row);
#pragma unroll
      for (int k = 0; k < N; k++) {
        int l = clip3(-32768, 32767, row[k]);
        row[k] = clip3(-32768, 32767, (int)((unsigned)l * (unsigned)qd.iq_scale + (unsigned)dadd) >> dshift);
      }
      store_row32<N>(lev2_p + (size_t)(y + gl) * lev2_s + x, row);
    }
  } else { // OP_PRED
    const int sh = luma ? 0 : 1;
    unsigned long long avail = 0;
    if constexpr (N == 64) avail = active ? intra_avail_mask_ctu(x, y, A.P) : 0; // a whole CTU, luma (hmx_tu_list_create checks)
    else avail = active ? intra_avail_mask(x << sh, y << sh, N << sh, A.P) : 0;
    const short *rec0 = a_p + (size_t)y * a_s + x;
    const int rst = a_s;
    intra_refs<N, N>(L, gl, active, [&](int dx, int dy) { return (int)rec0[(ptrdiff_t)dy * rst + dx]; }, luma, avail, A.P);
    if (active) {
      int org_row[N];
      if (A.cost) load_row16<N>(pick3(A.org.p, pl) + (size_t)(y + gl) * pick3(A.org.s, pl) + x, org_row);
      const int nm = A.n_modes <= 0 ? 1 : A.n_modes;
      for (int m = 0; m < nm; m++) {
        intra_pred_block<N>(L, gl, A.n_modes <= 0 ? (int)t.mode : (int)A.modes[m], luma, A.P, row);
        if (b_p) store_row16<N>(b_p + (A.n_modes <= 0 ? 0 : m * pick3(A.mode_elems, pl)) + (size_t)(y + gl) * b_s + x, row);
        if (A.cost) { // the prediction never leaves the registers: estIntraPredQT's calcHAD(org, pred) fused in
#pragma unroll
          for (int k = 0; k < N; k++) row[k] = org_row[k] - row[k];
          const int satd = satd_block<N>(L, gl, row);
          if (gl == 0) A.cost[(size_t)d.idx * nm + m] = (uint32_t)satd >> (A.P.bit_depth - 8);
        }
      }
    }
  }
}

// ---- whole-picture all-intra reconstruction: one launch per CTU diagonal ----
// Work item = (picture, CTU of the diagonal, plane), owned by ONE autonomous wave (64-thread
// workgroup): no workgroup barrier anywhere.  The host plan lists the item's blocks as segments of
// equal size and equal dependency level; a wave walks its segments, 64/N blocks at a time on the
// VALU path (N <= 16), one 32x32 block at a time on the matrix cores.
struct Seg { // a run of same-size blocks of one dependency level of one (CTU, plane)
  uint32_t start;
  uint16_t count;
  uint8_t log2n;
  uint8_t new_level; // 1: first segment of a dependency level (needs the previous level's recon)
};
struct FTu { // block descriptor of the frame path: geometry + precomputed neighbour availability
  hmx_tu t;
  uint32_t avail_lo, avail_hi;
};
struct LevelRow { // blocks of one picture-wide dependency level, bucketed by size (log2n - 2)
  uint32_t start[4];
  uint32_t count[4];
};
struct PicWork { // per picture: working planes + the plan it follows
  TiledPlane org[3], rec[3]; // tiled working copies (see TiledPlane)
  int *lev[3];
  int lev_stride[3];         // > 0: plane geometry; 0: the reference's Z-order coefficient layout
  const FTu *tus;
  const Seg *segs;
  const uint32_t *seg_range; // [(ctu*3+plane)*2 + {0,1}] -> begin,end in segs
  const FTu *ltus;           // level schedule: blocks sorted by (level, size)
  const LevelRow *ltab;      // [n_levels]
  int n_levels;
};
struct FrameArgs {
  const PicWork *pics;
  const uint32_t *wave_ctus; // CTU ids of this diagonal
  int n_wave_ctus;
  PicDev P;
};

#define HMX_WAVE_SMEM 7680 /* max(4 * sizeof(TuLds<16>), sizeof(Lane4Lds), ...) - checked below */
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
// 4x4 blocks per wave in the across-pictures level schedule: 16 = four lanes per block (one row each, through LDS like the 8x8 and
// 16x16 blocks), 64 = one lane per block (wave_chain_4_lane).  Measured at 1024 pictures of the 2160p mix: four lanes
// +3 % encoder direction, +16 % decoder direction (more, shorter waves); the one-lane form stays for A/B builds.
#ifndef HMX_SLOTS4
#define HMX_SLOTS4 16
#endif
constexpr int kSlots4 = HMX_SLOTS4; // across pictures
constexpr int kSlots4Own = 64;       // per-picture level kernel: one lane per block (four lanes: 51.5 vs 54.5 Gpx/s at 1024 pictures)
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

template <int N, bool ENC, bool ONCE = false, typename SRC>
__device__ __forceinline__ void wave_chain_valu(char *smem, const SRC &src, const PicDev &P, int count) {
  constexpr int SL = 64 / N;
  const int lane = lane_id(), slot = lane / N, gl = lane % N;
  TuLds<N> &L = reinterpret_cast<TuLds<N> *>(smem)[slot];
  for (int base = 0; ONCE ? base < 1 : base < count; base += SL) { // ONCE: the level schedule hands a wave at most one pass
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
    src.wait(); // packed schedule: the blocks this one predicts from belong to earlier rows of the same launch
    intra_refs_tiled<N, N, SRC::kCoherent>(L, gl, active, R, x, y, pb0, luma, avail, P);
    intra_pred_block<N>(L, gl, t.mode, luma, P, pred);
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
      if (active) {
        load_row32<N>(&L.tile[gl][0], row);
#pragma unroll
        for (int k = 0; k < N; k++) row[k] = level_of(row[k]);
#ifdef HMX_STREAM_NT
        if (V.lev_stride == 0) stream_store_row32<N>(lev_row, row); // the reference's coefficient layout: 16-byte aligned rows
        else
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
    inv_tq_block<N>(L, gl, active, ts, luma, luma, true, P, row);
    if (active) {
      const int mx = (1 << P.bit_depth) - 1;
#pragma unroll
      for (int k = 0; k < N; k++) row[k] = clip3(0, mx, pred[k] + row[k]);
      tstore_row<N, SRC::kWriteThrough>(R.p + pb0, R.qstride, gl, row);
    }
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
    int v[16];
    if (ENC && active) {
      const i4v o0 = stream_load(reinterpret_cast<const i4v *>(V.org + pb0)), o1 = stream_load(reinterpret_cast<const i4v *>(V.org + pb0 + 8));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v[2 * k] = (short)(o0[k] & 0xffff), v[2 * k + 1] = o0[k] >> 16;
        v[8 + 2 * k] = (short)(o1[k] & 0xffff), v[8 + 2 * k + 1] = o1[k] >> 16;
      }
    }
    src.wait(); // the whole wave (packed schedule): the neighbours belong to earlier rows of the same launch
    if (!active) continue; // from here a lane works alone: nothing below needs the other lanes
    // ---- reference line (fillReferenceSamples): sequential padding is natural inside one lane.
    int *line = LS.line[lane];
    {
      const int ul = luma ? 2 : 1, n = 4 >> ul; // unit = 4 (luma) / 2 (chroma) samples
      // The 17 reference samples lie in five neighbour tiles: column 3 of the below-left and left
      // tiles, sample (3,3) of the corner tile, row 3 of the above and above-right tiles.  One tile
      // address each; a tile none of whose units is available is replaced by the block's own tile
      // (a valid address; the availability mask drops the values).  11 loads, one round trip.
      int raw[17];
      {
        const unsigned m_bl = (1u << n) - 1, m_lf = m_bl << n, m_c = 1u << (2 * n), m_a = m_bl << (2 * n + 1), m_ar = m_a << n;
        const bool has_bl = avail & m_bl, has_lf = avail & m_lf, has_c = avail & m_c, has_a = avail & m_a, has_ar = avail & m_ar;
        const short *t_bl = R.p + (has_bl ? tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y + 4)) : pb0);
        const short *t_lf = R.p + (has_lf ? tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y)) : pb0);
        const short *t_c = R.p + (has_c ? tphys(R.qstride, tile_base(R.ctu_w, R.clog, x - 4, y - 4)) : pb0);
        const short *t_a = R.p + (has_a ? tphys(R.qstride, tile_base(R.ctu_w, R.clog, x, y - 4)) : pb0);
        const short *t_ar = R.p + (has_ar ? tphys(R.qstride, tile_base(R.ctu_w, R.clog, x + 4, y - 4)) : pb0);
        // whole tile rows (8 bytes): the access width of the coherent path, and no narrower request reaches the L2
        constexpr bool COH = SRC::kCoherent;
        const s4v va = ld_rec4<COH>(t_a + 12), var = ld_rec4<COH>(t_ar + 12), vc = ld_rec4<COH>(t_c + 12);
        s4v vb[4], vl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) vb[k] = ld_rec4<COH>(t_bl + 4 * (3 - k)), vl[k] = ld_rec4<COH>(t_lf + 4 * (3 - k));
#pragma unroll
        for (int k = 0; k < 4; k++) {
          raw[k] = vb[k][3];     // p = 0..3: (x-1, y+7-p)
          raw[4 + k] = vl[k][3]; // p = 4..7: (x-1, y+7-p)
          raw[9 + k] = va[k];
          raw[13 + k] = var[k];
        }
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
    int *lev_ptr = V.lev;
    const bool zlev = V.lev_stride == 0;
    const unsigned l0 = zlev ? b0 : __umul24((unsigned)y, (unsigned)V.lev_stride) + x;
    const int lrow = zlev ? 4 : V.lev_stride;
    int w[16];
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
#pragma unroll
      for (int r = 0; r < 4; r++) {
        i4v o = {w[4 * r], w[4 * r + 1], w[4 * r + 2], w[4 * r + 3]};
        stream_store(reinterpret_cast<i4v *>(lev_ptr + l0 + (unsigned)r * (unsigned)lrow), o);
      }
    } else {
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
        V.sse[b0 >> 4] = sse_samples<16>(o, rc, B);
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
    } else {
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

// plane <-> tiled conversion: one thread per tile row (4 samples), a 256-thread workgroup = 64 tiles
// = a 32x32 region in Z-order: tiled side fully coalesced, plane side whole 64-byte sectors.
struct ConvJob { // one plane of one picture
  short *plane;
  int stride, w, h;
  TiledPlane T;
};
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
struct PackRow { // one (level, group)
  uint32_t wave_base;    // index of its first wave-item in the call's descriptor array
  uint32_t n_waves;
  uint32_t item_base[4]; // first entry of each size class in the call's item array
  uint32_t count[4];     // blocks per size class
  uint32_t pad[2];
};
struct PackDesc { // one wave-item
  uint32_t item_off;   // first item
  uint32_t n_s;        // items | size class << 28
  uint32_t row;
  uint32_t dep_target; // wave-items of the previous row of the group (0: nothing to wait for)
};
struct PackHdr {
  uint32_t shard_base[9]; // wave-items of shard s: [shard_base[s], shard_base[s+1])
  uint32_t total_items;
  uint32_t pad0[22];
  uint32_t abort;         // set by a wave whose wait timed out
  uint32_t pad1[31];
  uint32_t ticket[8][32]; // one 128-byte line per shard
  uint32_t owner[8][32];  // 0: unclaimed, x + 1: claimed by XCD x
  // -DHMX_PACK_PROFILE builds only: phases of a wave-item in ticks of the 100 MHz wall clock, summed over all wave-items
  // [0] draw a ticket [1] descriptor [2] item loads issued up to the wait [3] wait for the previous row [4] references +
  // arithmetic + stores issued [5] drain of the stores [6] count; [7] wave-items; [8] polls; [9] waves' lifetimes
  unsigned long long prof[16];
};
struct PackPic { // per picture: where its levels go, and the plan it follows
  int *lev[3];
  int lev_stride[3];
  int n_levels;
  const LevelRow *ltab;
  const FTu *ltus;
  uint32_t *sse[3]; // distortion output per plane (hmx_set_sse_output), NULL = none
};
struct PackGeom {
  int n_pics, I, n_groups, n_shards, max_levels, slots4;
};
__host__ __device__ __forceinline__ uint32_t pack_slots(int s, int slots4) { return s == 0 ? (uint32_t)slots4 : s == 1 ? 8u : s == 2 ? 4u : 1u; }

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
      const uint32_t sl = pack_slots(s, G.slots4);
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
// prep 2: exclusive prefix of wave-items and items over the rows in ticket order (one workgroup)
__global__ __launch_bounds__(1024) void k_pack_scan(PackRow *rows, PackHdr *hdr, PackGeom G) {
  __shared__ uint32_t sw[1024], si[1024];
  const int n_rows = G.max_levels * G.n_groups, tid = threadIdx.x;
  const int chunk = (n_rows + 1023) / 1024, lo = min(tid * chunk, n_rows), hi = min(lo + chunk, n_rows);
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
  uint32_t wb = sw[tid] - w, ib = si[tid] - it;
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
  if (tid == 1023) {
    for (int q = G.n_shards; q <= 8; q++) hdr->shard_base[q] = sw[1023];
    hdr->total_items = si[1023];
  }
}
// prep 3: the wave-item descriptors and the item array of one (row, size class) per wave.  Items of a bucket are ordered
// by rank inside their picture's bucket, then by picture: pictures that share a plan put the SAME block of consecutive
// pictures on consecutive lanes (consecutive lines of the interleaved pool, uniform control flow); pictures with their
// own plans put blocks of similar code path (the plan sorts a bucket by plane, transform skip, mode) next to each other.
__global__ __launch_bounds__(64) void k_pack_fill(const PackPic *pics, const PackRow *rows, PackDesc *descs, FTu *items, PackGeom G, int n_rows) {
  __shared__ uint32_t cnts[64];
  const int rpw = 64 / G.I, sub = threadIdx.x / G.I, k = threadIdx.x - sub * G.I, seg0 = threadIdx.x - k;
  const int row = blockIdx.x * rpw + sub;
  const bool live = sub < rpw && row < n_rows;
  const int L = live ? row / G.n_groups : 0, g = live ? row - L * G.n_groups : 0;
  const PackRow R = rows[live ? row : 0];
  const int pic = g * G.I + k;
  LevelRow lr{};
  const FTu *ltus = nullptr;
  if (live && pic < G.n_pics && L < pics[pic].n_levels) {
    lr = pics[pic].ltab[L];
    ltus = as_global(pics[pic].ltus);
  }
  const uint32_t dep = (live && L > 0) ? rows[row - G.n_groups].n_waves : 0;
  uint32_t woff = 0;
#pragma unroll
  for (int s = 3; s >= 0; s--) { // largest blocks first
    const uint32_t sl = pack_slots(s, G.slots4), total = live ? R.count[s] : 0, nw = (total + sl - 1) / sl;
    for (uint32_t c = (uint32_t)k; c < nw; c += (uint32_t)G.I)
      descs[R.wave_base + woff + c] = PackDesc{R.item_base[s] + c * sl, min(sl, total - c * sl) | ((uint32_t)s << 28), (uint32_t)row, dep};
    woff += nw;
    // item (rank r, picture k) of the bucket sits behind every item of a lower rank and the same-rank items of the pictures
    // before k:  sum over k' of min(count[k'], r)  +  #{k' < k : count[k'] > r}
    const uint32_t cnt = ltus ? lr.count[s] : 0, start = lr.start[s];
    wave_sync();
    cnts[threadIdx.x] = cnt; // the counts of the row's pictures, for lanes that loop longer than their neighbours
    wave_sync();
    for (uint32_t r = 0; r < cnt; r++) {
      uint32_t off = 0;
      for (int q = 0; q < G.I; q++) {
        const uint32_t cq = cnts[seg0 + q];
        off += min(cq, r) + ((q < k && cq > r) ? 1u : 0u);
      }
      FTu f = ltus[start + r];
      f.t.plane = (uint8_t)(f.t.plane | (k << 2)); // picture of the group in the upper six bits
      items[R.item_base[s] + off] = f;
    }
  }
}

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
  static constexpr bool kCoherent = true;      // reconstruction loads bypass the vector L1 (sc1)
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler ordering: no reconstruction load moves above the poll
    // the next wave-item's ticket has long returned: fetch its descriptor behind the reference loads that follow
    nx->t = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx->ticket_raw);
    if (nx->t < nx->total) nx->d = nx->descs[nx->base + nx->t];
#ifdef HMX_PACK_PROFILE
    pt[1] = wall_clock64();
#endif
  }
};

// item index of a lane inside a wave-item of size class s: lane / (lanes per block)
template <int SL4>
__device__ __forceinline__ int pack_lane_item(int lane, int s) {
  return s == 0 ? (SL4 == 64 ? lane : lane >> 2) : s == 1 ? lane >> 3 : s == 2 ? lane >> 4 : 0;
}

template <bool ENC, int SL4, bool SSE = false, bool RDOQ = false>
__global__ __launch_bounds__(64, RDOQ ? 2 : 4) void k_intra_packed(PackArgs A) {
  static_assert(!RDOQ || (ENC && SL4 == 64), "RDOQ: encoder direction, 4x4 blocks one per lane");
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
      ft = A.items[(uint32_t)__builtin_amdgcn_readfirstlane((int)d.item_off) + (uint32_t)min(pack_lane_item<SL4>(lane_id(), s), n - 1)];
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
      } else if (s == 1) wave_chain_valu<8, ENC, true>(smem, src, A.P, n);
      else if (s == 2) wave_chain_valu<16, ENC, true>(smem, src, A.P, n);
      else wave_chain_32<ENC, true>(smem, src, A.P, n);
      // the next wave-item's items, behind this one's stores (its descriptor was fetched behind the dependency poll)
      const bool more = nx.t < nx.total;
      FTu ftn = ft;
      if (more) {
        const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx.d.n_s);
        ftn = A.items[(uint32_t)__builtin_amdgcn_readfirstlane((int)nx.d.item_off) +
                      (uint32_t)min(pack_lane_item<SL4>(lane_id(), (int)(ns >> 28)), (int)(ns & 0x0fffffffu) - 1)];
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

// The inter block chain for 32x32 blocks on the matrix cores, ONE WAVE PER BLOCK (k_list<32> spends 32 lanes on a
// block and multiplies on the VALU): the data layout and the MFMA passes of wave_chain_32 with the prediction read
// from its plane.  grid = (blocks, pictures of a multi-picture call).
__global__ __launch_bounds__(64) void k_inter32(ListArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[sizeof(TuLds<32>)];
  TuLds<32> &L = *reinterpret_cast<TuLds<32> *>(smem);
  typedef __attribute__((address_space(1))) const short gpel;
  typedef __attribute__((address_space(1))) short gpel_w;
  typedef __attribute__((address_space(1))) int gint_w;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const DTu d = load_dtu(A.tus + blockIdx.x);
  const hmx_tu t = d.t;
  const int pl = wave_uniform((int)t.plane), x = wave_uniform((int)t.x), y = wave_uniform((int)t.y);
  const ListPic &Q = A.pics[blockIdx.y];
  const bool luma = pl == 0;
  const int a_s = uniform3(Q.a.s, pl), b_s = uniform3(Q.b.s, pl), l_s = uniform3(Q.lev.s, pl), r_s = uniform3(Q.rec.s, pl);
  // this lane: row r, the four 4-sample pieces at columns 8q + 4h (mrow), as the MFMA passes want them
  gpel *org = (gpel *)uniform3(Q.a.p, pl) + (size_t)(y + r) * a_s + x + 4 * h;
  gpel *prd = (gpel *)uniform3(Q.b.p, pl) + (size_t)(y + r) * b_s + x + 4 * h;
  int v[16], coef[16];
  unsigned pred2[8]; // the prediction waits packed for the reconstruction
#pragma unroll
  for (int q = 0; q < 4; q++) {
    short o4[4], p4[4];
    __builtin_memcpy(o4, org + 8 * q, 8);
    __builtin_memcpy(p4, prd + 8 * q, 8);
#pragma unroll
    for (int k = 0; k < 4; k++) v[4 * q + k] = wrap16(o4[k] - p4[k]);
    pred2[2 * q] = (unsigned)(unsigned short)p4[0] | ((unsigned)(unsigned short)p4[1] << 16);
    pred2[2 * q + 1] = (unsigned)(unsigned short)p4[2] | ((unsigned)(unsigned short)p4[3] << 16);
  }
  fwd32_mfma(v, r, h, A.P.bit_depth, coef);
  const int sum = quant_sbh_block<32, 64, 16, false>(
      L, lane, true, coef, [&](int k) { return mrow(k, h); }, [&](int) { return r; }, luma, 0, A.P);
  gint_w *lev = (gint_w *)uniform3(Q.lev.p, pl) + (size_t)y * l_s + x + r;
#pragma unroll
  for (int g = 0; g < 16; g++) {
    v[g] = level_of(L.tile[mrow(g, h)][r]);
    lev[(size_t)mrow(g, h) * l_s] = v[g]; // 32 lanes = one 128-byte row of levels
  }
  if (lane == 0 && A.abs_sum) A.abs_sum[(size_t)blockIdx.y * A.abs_stride + d.idx] = (uint32_t)sum;
  const int tshift = 15 - A.P.bit_depth - 5;
  const QuantDev qd = pick_qd(A.P, luma);
  int out[16];
#pragma unroll
  for (int g = 0; g < 16; g++) v[g] = wrap16(dequant_one(v[g], qd.iq_scale, 6 - tshift));
  inv32_mfma(v, r, h, A.P.bit_depth, out);
  const int mx = (1 << A.P.bit_depth) - 1;
  gpel_w *rec = (gpel_w *)uniform3(Q.rec.p, pl) + (size_t)(y + r) * r_s + x + 4 * h;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int p0 = pred2[2 * q] & 0xffff, p1 = pred2[2 * q] >> 16, p2 = pred2[2 * q + 1] & 0xffff, p3 = pred2[2 * q + 1] >> 16;
    short r4[4] = {(short)clip3(0, mx, p0 + out[4 * q]), (short)clip3(0, mx, p1 + out[4 * q + 1]),
                   (short)clip3(0, mx, p2 + out[4 * q + 2]), (short)clip3(0, mx, p3 + out[4 * q + 3])};
    __builtin_memcpy(rec + 8 * q, r4, 8);
    if (A.sse) { // v[] becomes org - rec for the distortion
      short o4[4];
      __builtin_memcpy(o4, org + 8 * q, 8);
#pragma unroll
      for (int k = 0; k < 4; k++) v[4 * q + k] = o4[k] - r4[k];
    }
  }
  if (A.sse) {
    const unsigned sh = (unsigned)(A.P.bit_depth - 8) << 1;
    unsigned dsum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) dsum += (unsigned)mul24(v[k], v[k]) >> sh;
    dsum = (unsigned)group_sum((int)dsum, 64);
    if (lane == 0) A.sse[(size_t)blockIdx.y * A.abs_stride + d.idx] = dsum;
  }
}

// The inter block chain for 4x4 blocks, ONE LANE PER BLOCK (k_list spends four lanes on a 4x4 block and runs its
// sign-bit hiding in one of them): residual org - pred, T, Q + sign hiding, levels out, IQ, IT, Clip(pred + resi) out.
// grid.y = picture of a multi-picture call (ListArgs::pics).
__global__ __launch_bounds__(256) void k_inter4(ListArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n) return;
  const DTu d = load_dtu(A.tus + i);
  const hmx_tu t = d.t;
  const int pl = t.plane, x = t.x, y = t.y;
  const ListPic &Q = A.pics[blockIdx.y];
  const bool luma = pl == 0, inter = t.flags & HMX_TU_INTER, ts = t.flags & HMX_TU_TRANSFORM_SKIP;
  const bool use_dst = luma && !inter;
  // table fields by scalar loads + per-lane plane select; rows as single 8 / 16-byte accesses (dword-aligned planes)
  typedef __attribute__((address_space(1))) const short gpel;
  typedef __attribute__((address_space(1))) short gpel_w;
  typedef __attribute__((address_space(1))) int gint_w;
  const int a_s = uniform3(Q.a.s, pl), b_s = uniform3(Q.b.s, pl), l_s = uniform3(Q.lev.s, pl), r_s = uniform3(Q.rec.s, pl);
  gpel *org = (gpel *)uniform3(Q.a.p, pl) + (size_t)y * a_s + x, *prd = (gpel *)uniform3(Q.b.p, pl) + (size_t)y * b_s + x;
  int pred[16], v[16], w[16];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    short o4[4], p4[4];
    __builtin_memcpy(o4, org + (size_t)r * a_s, 8);
    __builtin_memcpy(p4, prd + (size_t)r * b_s, 8);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      pred[4 * r + k] = p4[k];
      v[4 * r + k] = wrap16(o4[k] - p4[k]);
    }
  }
  lane4_forward(v, use_dst, ts, luma, coef_scan_idx(4, luma, !inter, t.mode), A.P, w);
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = level_of(w[k]);
  gint_w *lev = (gint_w *)uniform3(Q.lev.p, pl) + (size_t)y * l_s + x;
#pragma unroll
  for (int r = 0; r < 4; r++) __builtin_memcpy(lev + (size_t)r * l_s, w + 4 * r, 16);
  int out[16];
  lane4_inverse(w, use_dst, ts, luma, A.P, out);
  const int mx = (1 << A.P.bit_depth) - 1;
  gpel_w *rec = (gpel_w *)uniform3(Q.rec.p, pl) + (size_t)y * r_s + x;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    short r4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r4[k] = (short)clip3(0, mx, pred[4 * r + k] + out[4 * r + k]);
    __builtin_memcpy(rec + (size_t)r * r_s, r4, 8);
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * r + k] = v[4 * r + k] + pred[4 * r + k] - r4[k]; // org - rec (org = residual + prediction)
  }
  if (A.sse) {
    const unsigned sh = (unsigned)(A.P.bit_depth - 8) << 1;
    unsigned dsum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) dsum += (unsigned)mul24(out[k], out[k]) >> sh;
    A.sse[(size_t)blockIdx.y * A.abs_stride + d.idx] = dsum;
  }
}

// =============================================================================================
// Host side
// =============================================================================================
struct hmx_ctx {
  hmx_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // scratch for the scalar drop-ins (one block): device staging
  char *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  // working pictures of the whole-picture path in tiled layout (grow-only pool, one slot per picture)
  // ONE allocation per direction, picture i at element offset i * tiled_pic_elems, its planes at
  // tiled_plane_off[]: a wave that works across pictures reaches any picture with a multiply-add
  short *pool_org = nullptr, *pool_rec = nullptr; // pools of the call being issued: the context's own or the caller's (hmx_tpool)
  short *own_pool_org = nullptr, *own_pool_rec = nullptr; // allocations behind the plane-geometry entry points
  int pool_pics = 0;
  size_t tiled_pic_elems = 0;
  uint32_t tiled_plane_off[3] = {0, 0, 0};
  int tiled_cw = 0, tiled_ch = 0; // CTU grid of the call being issued
  int own_cw = 0, own_ch = 0;     // CTU grid the context's own pools were sized for
  bool resident_call = false;     // the call works on the caller's resident pools: no layout conversion
  ConvJob *d_jobs = nullptr;      // [2][n_pics*3]: to-tiled jobs, then from-tiled jobs
  int jobs_cap = 0;
  // whole-picture calls recorded as HIP graphs (see frame_intra)
  struct GraphEntry {
    uint64_t key;
    int n_pics;
    hipGraphExec_t exec;
    PicWork *d_work;
    uint64_t stamp;
  };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int last_schedule = 0, last_groups = 1; // of the last whole-picture call (hmx_last_call_shape)
  bool onto_call = false;                 // this call reconstructs onto what the reconstruction planes already hold
  bool pipeline_conv = false;  // this call converts CTU row by CTU row, overlapped with the chain
  bool across_call = false;    // the call being issued uses the across-pictures schedule (interleaved pool)
  int level_mode_min_pics = 1; // measured: the level schedule is at least as fast as the wave schedule at every batch size
  // optional timing of the last whole-picture call: events around the layout conversions and the chain
  bool timing = false;
  hipEvent_t tev[4] = {};
  bool tev_valid = false;
  // level schedule: picture groups run on side streams so that launches of different groups overlap
  static const int kMaxSide = 8;
  hipStream_t side[kMaxSide] = {};
  // layout conversions pipelined with the chain (across schedule): one stream for the conversions, one event per CTU row
  // and direction, one per (group, CTU row) for "this row is final"
  hipStream_t conv_stream = nullptr;
  std::vector<hipEvent_t> ev_rows; // [ch] converted in, [ch] converted out marker unused, then [groups][ch] row final
  hipEvent_t ev_conv_join = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {};
  int n_side = 0;
  // Argument arena: small per-call tables (picture planes, job lists) travel through a pinned host ring
  // and a device ring by asynchronous copies; the stream is synchronised only when the ring wraps.
  double *rdoq_wd = nullptr; // RDOQ per-lane records (hmx_rdoq.h), sized for rdoq_T lanes
  int *rdoq_wi = nullptr;
  RdoqBlock *rdoq_blocks = nullptr;
  EstBitsDev *rdoq_est = nullptr;
  int rdoq_T = 0, rdoq_est_cap = 0, rdoq_blocks_cap = 0;
  uint64_t rdoq_key = 0;      // of the block list and tables resident on the device
  bool rdoq_resident = false;
  size_t rdoq_class_n[4] = {0, 0, 0, 0}; // blocks of 32, 16, 8, 4 in the resident list
  uint64_t rdoq_in_key = 0;              // of the caller's arguments that produced the resident list (hmx_batch_xRateDistOptQuant)
  bool rdoq_join[2] = {false, false}; // side streams of the current RDOQ call still to be joined
  double rdoq_consts_h[4] = {};
  bool rdoq_consts_valid = false;
  double *rdoq_consts = nullptr; // k_rdoq_tiles: lambda [luma, chroma], then the Int64 factors of sign hiding
  int *d_mcmap = nullptr; // cell -> PU maps of the last motion-compensation call
  size_t mcmap_cap = 0;
  char *arena_h = nullptr, *arena_d = nullptr;
  size_t arena_cap = 0, arena_head = 0;
  // packed schedule (k_intra_packed): tables of the last call; rebuilt on the device when the pictures / plans change
  struct Packed {
    PackPic *d_pics = nullptr;
    PackRow *d_rows = nullptr;
    PackDesc *d_descs = nullptr;
    FTu *d_items = nullptr;
    uint32_t *d_done = nullptr;
    PackHdr *d_hdr = nullptr;
    size_t cap_pics = 0, cap_rows = 0, cap_descs = 0, cap_items = 0, cap_done = 0;
    uint64_t key = 0;
    bool valid = false;
    PackGeom G{};
    int n_wg = 0;
    uint64_t waves_bound = 0;
  } pk;
  int max_resident_waves = 0; // of k_intra_packed on this device
  int pack_I = 1;             // pictures per group (interleave domain of the pool) of the call being issued
  const hmx_levels *call_lev = nullptr; // the call's level planes (host array, valid while the call is issued)
  bool pk_pending = false;    // a packed launch was issued since the last check of its abort word
  // hmx_set_rdoq: xRateDistOptQuant as the quantiser of the next whole-picture encode calls
  struct ChainRdoq {
    int n = 0;                    // pictures described (1: one set for all pictures of a call); 0 = off
    EstBitsDev *d_est = nullptr;  // [n][2][4]
    size_t cap_est = 0;
    std::vector<double> lambda;   // [n][2]
    double *d_lambda = nullptr;   // [n][2], then the Int64 factors [n][2]
    size_t cap_lambda = 0;
    int max_waves = 0;            // resident waves of the RDOQ kernel variant
    uint64_t serial = 0;          // counts hmx_set_rdoq calls (part of the schedule key)
  } crq;
  std::vector<hmx_sse> sse_out; // hmx_set_sse_output: per-picture distortion arrays of the next whole-picture encode calls
  uint64_t table_key = 0;     // of the picture table resident in d_jobs (whole-picture calls)
  bool table_valid = false;
  // knobs, read once from the environment in hmx_create (A/B runs and the cross-checks of the tests)
  struct Knobs {
    int schedule = -1;     // HMX_INTRA_SCHEDULE: wave / level / packed (default: packed)
    int across = -1;       // HMX_INTRA_ACROSS: 0 keeps shared-plan batches of the level schedule per picture
    int streams = 0;       // HMX_INTRA_STREAMS: picture groups of the across schedule
    bool pipeline_conv = false, graph = false;
    int slots4 = 0;        // HMX_PACK_SLOTS4: 16 or 64 4x4 blocks per wave-item (0: by batch size)
    int pack_group = 0;    // HMX_PACK_GROUP: pictures per group, 1..64 (0: by batch size, see pack_group_size)
    int pack_waves = 0;    // HMX_PACK_WAVES: persistent waves (0: by batch size)
    int pack_sleep0 = -1, pack_sleep1 = -1; // HMX_PACK_SLEEP0 / 1: poll back-off, units of 64 clocks (-1: default)
    bool rdoq_lane_only = false; // HMX_RDOQ_LANE: every block through the one-lane-per-block kernel (round 1's, A/B and cross-check)
  } knob;
};

struct hmx_intra_plan {
  FTu *d_tus = nullptr;
  Seg *d_segs = nullptr;
  uint32_t *d_seg_range = nullptr;
  uint32_t *d_wave_ctus = nullptr;
  std::vector<std::pair<uint32_t, uint32_t>> waves; // offset,count into d_wave_ctus
  FTu *d_ltus = nullptr;       // level schedule
  LevelRow *d_ltab = nullptr;
  std::vector<uint32_t> level_chunks; // waves needed per level
  std::vector<LevelRow> h_ltab;       // host copy of the level table
  uint64_t size_total[4] = {0, 0, 0, 0}; // blocks per transform size
  uint64_t serial = 0;                    // unique per plan: a freed plan's address may be handed out again
  // per CTU row: the first and the last dependency level that touches it (the layout conversions are pipelined by CTU
  // row: a row is converted in before its first level and out after its last one)
  std::vector<int> row_first_level, row_last_level;
  PicDev P;
  int n_tu = 0;
  int qp = 0, chroma_qp_offset = 0, slice_type = 0;
};

static int fail(hmx_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}
#define HIPCHK(ctx, call)                                                   \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) return fail(ctx, HMX_ERR_DEVICE, #call, e_);      \
  } while (0)

static inline int ilog2i(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return l;
}
static const int kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564}; // TComRom.cpp:293
static const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};                // TComRom.cpp:298
static int chroma_scale(int idx) { // g_aucChromaScale[58], TComRom.cpp:380
  static const unsigned char mid[13] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37};
  return idx < 30 ? idx : (idx >= 43 ? idx - 6 : mid[idx - 30]);
}

extern "C" hmx_qp hmx_setQPforQuant(int qpy, int text_type, int qp_bd_offset, int chroma_qp_offset) {
  int q;
  if (text_type == HMX_TEXT_LUMA)
    q = qpy + qp_bd_offset;
  else {
    q = std::min(57, std::max(-qp_bd_offset, qpy + chroma_qp_offset));
    q = q < 0 ? q + qp_bd_offset : chroma_scale(q) + qp_bd_offset;
  }
  hmx_qp r = {q, q / 6, q % 6, 15 + q / 6};
  return r;
}

static QuantDev make_qd(const hmx_qp &qp, int per_base, int slice_type) {
  QuantDev d;
  d.q = kQuantScales[qp.rem];
  d.per_qbits = per_base >= 0 ? per_base : qp.per;
  d.iq_scale = kInvQuantScales[qp.rem] << qp.per;
  d.rnd_factor = slice_type == HMX_I_SLICE ? 171 : 85;
  return d;
}

static PicDev make_picdev(const hmx_ctx *c, const hmx_pic_param *pp) {
  PicDev P;
  P.pic_w = pp->pic_w;
  P.pic_h = pp->pic_h;
  P.ctu = c->cfg.ctu_size;
  P.bit_depth = c->cfg.bit_depth;
  P.sign_hide = pp->sign_hide;
  const int bd = 6 * (c->cfg.bit_depth - 8);
  P.qd[0] = make_qd(hmx_setQPforQuant(pp->qp, HMX_TEXT_LUMA, bd, 0), -1, pp->slice_type);
  P.qd[1] = make_qd(hmx_setQPforQuant(pp->qp, HMX_TEXT_CHROMA, bd, pp->chroma_qp_offset), -1, pp->slice_type);
  return P;
}

// Tuning knobs: read from the environment ONCE, in hmx_create; hmx_set_option changes one afterwards (A/B runs, and the
// parity tests that hold the schedules against each other).  value == NULL restores the default.
static const char *const kKnobNames[] = {"HMX_INTRA_SCHEDULE", "HMX_INTRA_ACROSS", "HMX_INTRA_STREAMS", "HMX_PIPELINE_CONV", "HMX_GRAPH",
                                         "HMX_PACK_SLOTS4",    "HMX_PACK_GROUP",      "HMX_PACK_WAVES",    "HMX_PACK_SLEEP0",   "HMX_PACK_SLEEP1",
                                         "HMX_RDOQ_LANE"};
static bool apply_knob(hmx_ctx *c, const char *name, const char *v) {
  auto &k = c->knob;
  const std::string n(name);
  if (n == "HMX_INTRA_SCHEDULE") k.schedule = !v ? -1 : v[0] == 'w' ? 0 : v[0] == 'l' ? 1 : v[0] == 'p' ? 3 : -1;
  else if (n == "HMX_INTRA_ACROSS") k.across = !v ? -1 : v[0] != '0';
  else if (n == "HMX_INTRA_STREAMS") k.streams = v ? atoi(v) : 0;
  else if (n == "HMX_PIPELINE_CONV") k.pipeline_conv = v && v[0] != '0';
  else if (n == "HMX_GRAPH") k.graph = v != nullptr;
  else if (n == "HMX_PACK_SLOTS4") k.slots4 = !v ? 0 : atoi(v) == 16 ? 16 : 64;
  else if (n == "HMX_PACK_GROUP") k.pack_group = v ? std::min(64, std::max(1, atoi(v))) : 0;
  else if (n == "HMX_PACK_WAVES") k.pack_waves = v ? std::max(1, atoi(v)) : 0;
  else if (n == "HMX_PACK_SLEEP0") k.pack_sleep0 = v ? std::max(0, atoi(v)) : -1;
  else if (n == "HMX_PACK_SLEEP1") k.pack_sleep1 = v ? std::max(0, atoi(v)) : -1;
  else if (n == "HMX_RDOQ_LANE") k.rdoq_lane_only = v && v[0] != '0';
  else return false;
  return true;
}
extern "C" int hmx_set_option(hmx_ctx *c, const char *name, const char *value) {
  if (!c || !name) return HMX_ERR_ARG;
  return apply_knob(c, name, value) ? HMX_OK : fail(c, HMX_ERR_ARG, "hmx_set_option: unknown option");
}

extern "C" int hmx_create(const hmx_config *cfg, hmx_ctx **out) {
  if (!cfg || !out) return HMX_ERR_ARG;
  if (cfg->bit_depth < 8 || cfg->bit_depth > 12 || (cfg->ctu_size != 64 && cfg->ctu_size != 32 && cfg->ctu_size != 16))
    return HMX_ERR_ARG;
  hmx_ctx *c = new hmx_ctx;
  c->cfg = *cfg;
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) {
    delete c;
    return HMX_ERR_DEVICE;
  }
  if (cfg->stream)
    c->stream = (hipStream_t)cfg->stream;
  else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return HMX_ERR_DEVICE;
    }
    c->own_stream = true;
  }
  c->scratch_bytes = 1 << 20;
  e = hipMalloc((void **)&c->d_scratch, c->scratch_bytes);
  if (e != hipSuccess) {
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return HMX_ERR_NOMEM;
  }
  for (const char *name : kKnobNames)
    if (const char *v = getenv(name)) apply_knob(c, name, v);
  *out = c;
  return HMX_OK;
}

extern "C" void hmx_destroy(hmx_ctx *c) {
  if (!c) return;
  hipStreamSynchronize(c->stream);
  hipFree(c->d_scratch);
  for (auto &e : c->graphs) {
    hipGraphExecDestroy(e.exec);
    hipFree(e.d_work);
  }
  hipFree(c->own_pool_org);
  hipFree(c->own_pool_rec);
  hipFree(c->d_jobs);
  if (c->arena_h) hipHostFree(c->arena_h);
  hipFree(c->arena_d);
  hipFree(c->d_mcmap);
  hipFree(c->rdoq_wd);
  hipFree(c->rdoq_wi);
  hipFree(c->rdoq_blocks);
  hipFree(c->rdoq_est);
  hipFree(c->rdoq_consts);
  hipFree(c->crq.d_est);
  hipFree(c->crq.d_lambda);
  hipFree(c->pk.d_pics);
  hipFree(c->pk.d_rows);
  hipFree(c->pk.d_descs);
  hipFree(c->pk.d_items);
  hipFree(c->pk.d_done);
  hipFree(c->pk.d_hdr);
  for (int g = 0; g < c->n_side; g++) {
    hipStreamDestroy(c->side[g]);
    hipEventDestroy(c->ev_join[g]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->conv_stream) hipStreamDestroy(c->conv_stream);
  for (auto e : c->ev_rows) hipEventDestroy(e);
  if (c->ev_conv_join) hipEventDestroy(c->ev_conv_join);
  for (int i = 0; i < 4; i++)
    if (c->tev[i]) hipEventDestroy(c->tev[i]);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}
extern "C" const char *hmx_last_error(const hmx_ctx *c) { return c ? c->err.c_str() : "null context"; }
// after a synchronisation: did a wave of the packed schedule give up waiting (its bounded spin ran out)?
static int check_packed_abort(hmx_ctx *c) {
  if (!c->pk_pending || !c->pk.d_hdr) return HMX_OK;
  c->pk_pending = false;
#ifdef HMX_PACK_PROFILE
  {
    unsigned long long pr[16];
    HIPCHK(c, hipMemcpy(pr, c->pk.d_hdr->prof, sizeof(pr), hipMemcpyDeviceToHost));
    const double n = pr[7] ? (double)pr[7] : 1.0;
    fprintf(stderr, "[pack profile] last call: %llu wave-items, per item (us): ticket %.2f desc %.2f pre-wait %.2f wait %.2f (%.1f polls) chain %.2f drain %.2f count %.2f; "
                    "wave lifetime %.1f us avg over %d waves\n", pr[7], pr[0] / n / 100, pr[1] / n / 100, pr[2] / n / 100, pr[3] / n / 100, pr[8] / n, pr[4] / n / 100,
            pr[5] / n / 100, pr[6] / n / 100, c->pk.n_wg ? pr[9] / 100.0 / c->pk.n_wg : 0.0, c->pk.n_wg);
    unsigned long long rq[40], zero[40] = {};
    HIPCHK(c, hipMemcpyFromSymbol(rq, HIP_SYMBOL(g_rdoq_prof), sizeof(rq)));
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_rdoq_prof), zero, sizeof(zero)));
    if (rq[9]) fprintf(stderr, "[rdoq profile] 4x4 in a lane: %llu blocks, %.2f us each\n", rq[9], rq[0] / (double)rq[9] / 100);
    for (int g = 1; g < 4; g++)
      if (rq[g * 10 + 9]) {
        const double m = (double)rq[g * 10 + 9] * 100;
        fprintf(stderr, "[rdoq profile] %dx%d: %llu wave calls, us per call: prep %.2f walk8 %.2f resolve %.2f walk %.2f lastpos %.2f levels %.2f signhide %.2f store %.2f\n",
                4 << g, 4 << g, rq[g * 10 + 9], rq[g * 10] / m, rq[g * 10 + 1] / m, rq[g * 10 + 2] / m, rq[g * 10 + 3] / m, rq[g * 10 + 4] / m, rq[g * 10 + 5] / m,
                rq[g * 10 + 6] / m, rq[g * 10 + 7] / m);
      }
  }
#endif
  uint32_t ab = 0;
  HIPCHK(c, hipMemcpy(&ab, &c->pk.d_hdr->abort, sizeof(ab), hipMemcpyDeviceToHost));
  if (!ab) return HMX_OK;
  HIPCHK(c, hipMemset(&c->pk.d_hdr->abort, 0, sizeof(ab))); // read and reported: the next call starts clean
  return fail(c, HMX_ERR_DEVICE, "packed schedule: a dependency wait timed out, the outputs of every call since the last hmx_sync are invalid");
}
extern "C" int hmx_sync(hmx_ctx *c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_packed_abort(c);
}
extern "C" int hmx_malloc(hmx_ctx *c, size_t bytes, void **dptr) {
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc", e);
  return HMX_OK;
}
extern "C" int hmx_free(hmx_ctx *c, void *dptr) {
  HIPCHK(c, hipFree(dptr));
  return HMX_OK;
}
extern "C" int hmx_upload(hmx_ctx *c, void *dst, const void *src, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HMX_OK;
}
extern "C" int hmx_download(hmx_ctx *c, void *dst, const void *src, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_packed_abort(c);
}
extern "C" int hmx_memset(hmx_ctx *c, void *dst, int value, size_t bytes) {
  HIPCHK(c, hipMemsetAsync(dst, value, bytes, c->stream));
  return HMX_OK;
}
extern "C" int hmx_event_create(hmx_ctx *c, void **ev) {
  hipEvent_t e;
  HIPCHK(c, hipEventCreate(&e));
  *ev = (void *)e;
  return HMX_OK;
}
extern "C" int hmx_event_record(hmx_ctx *c, void *ev) {
  HIPCHK(c, hipEventRecord((hipEvent_t)ev, c->stream));
  return HMX_OK;
}
extern "C" int hmx_event_elapsed_ms(hmx_ctx *c, void *a, void *b, float *ms) {
  HIPCHK(c, hipEventSynchronize((hipEvent_t)b));
  HIPCHK(c, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return HMX_OK;
}
extern "C" int hmx_event_destroy(hmx_ctx *c, void *ev) {
  HIPCHK(c, hipEventDestroy((hipEvent_t)ev));
  return HMX_OK;
}

// ---- list launches ----
template <int OP>
static int launch_list(hmx_ctx *c, int log2n, const ListArgs &A) {
  if (A.n <= 0) return HMX_OK;
  dim3 blk(256);
  const unsigned ny = A.pics ? (unsigned)A.n_pics : 1u;
  switch (log2n) {
  case 2: hipLaunchKernelGGL((k_list<4, OP>), dim3((A.n + Slots<4>::v - 1) / Slots<4>::v, ny), blk, 0, c->stream, A); break;
  case 3: hipLaunchKernelGGL((k_list<8, OP>), dim3((A.n + Slots<8>::v - 1) / Slots<8>::v, ny), blk, 0, c->stream, A); break;
  case 4: hipLaunchKernelGGL((k_list<16, OP>), dim3((A.n + Slots<16>::v - 1) / Slots<16>::v, ny), blk, 0, c->stream, A); break;
  case 5: hipLaunchKernelGGL((k_list<32, OP>), dim3((A.n + Slots<32>::v - 1) / Slots<32>::v, ny), blk, 0, c->stream, A); break;
  case 6: // 64 x 64: the luma prediction unit of a 64 x 64 coding unit (TEncSearch.cpp:2509-2540); no transform of that size exists
    if constexpr (OP == OP_PRED) {
      hipLaunchKernelGGL((k_list<64, OP>), dim3((unsigned)A.n, ny), blk, 0, c->stream, A);
      break;
    } else {
      return fail(c, HMX_ERR_ARG, "64x64 blocks: intra prediction only (the largest transform is 32x32)");
    }
  default: return fail(c, HMX_ERR_ARG, "unsupported block size");
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

static int launch_op(hmx_ctx *c, int op, int log2n, const ListArgs &A) {
  switch (op) {
  case OP_TRANSFORM_NXN: return launch_list<OP_TRANSFORM_NXN>(c, log2n, A);
  case OP_INVTRANSFORM_NXN: return launch_list<OP_INVTRANSFORM_NXN>(c, log2n, A);
  case OP_XT: return launch_list<OP_XT>(c, log2n, A);
  case OP_XIT: return launch_list<OP_XIT>(c, log2n, A);
  case OP_XQUANT: return launch_list<OP_XQUANT>(c, log2n, A);
  case OP_XDEQUANT: return launch_list<OP_XDEQUANT>(c, log2n, A);
  case OP_TRANSFORM_RECON: return launch_list<OP_TRANSFORM_RECON>(c, log2n, A);
  default: return launch_list<OP_PRED>(c, log2n, A);
  }
}

// A block list resident on the device, bucketed by block size.
struct hmx_tu_list {
  DTu *d = nullptr;
  int off[5] = {0, 0, 0, 0, 0}, cnt[5] = {0, 0, 0, 0, 0}; // size classes 4 .. 64 (64: luma prediction units, hmx_batch_predIntra[_cost] only)
  int n = 0;
};

extern "C" int hmx_tu_list_create(hmx_ctx *c, const hmx_tu *tus, int n, hmx_tu_list **out) {
  if (!c || !out || (n > 0 && !tus)) return fail(c, HMX_ERR_ARG, "hmx_tu_list_create: null argument");
  hmx_tu_list *l = new hmx_tu_list;
  std::vector<DTu> v;
  v.reserve(n);
  for (int s = 2; s <= 6; s++) {
    l->off[s - 2] = (int)v.size();
    for (int i = 0; i < n; i++)
      if (tus[i].log2n == s && (s < 6 || (tus[i].plane == 0 && tus[i].x % 64 == 0 && tus[i].y % 64 == 0 && c->cfg.ctu_size == 64)))
        v.push_back(DTu{tus[i], (uint32_t)i});
    l->cnt[s - 2] = (int)v.size() - l->off[s - 2];
    // The blocks of a list call are independent, so the order inside a size class is ours: raster order per plane
    // puts horizontally adjacent blocks on adjacent lanes, whose row accesses then share cache lines (coding order
    // only ever pairs them).  Results that are per block (abs sums, costs) go by DTu::idx, the caller's index.
    std::stable_sort(v.begin() + l->off[s - 2], v.end(), [](const DTu &a, const DTu &b) {
      if (a.t.plane != b.t.plane) return a.t.plane < b.t.plane;
      if (a.t.y != b.t.y) return a.t.y < b.t.y;
      return a.t.x < b.t.x;
    });
  }
  if ((int)v.size() != n) {
    delete l;
    return fail(c, HMX_ERR_ARG, "hmx_tu_list_create: block size outside 4..32 (64: luma, CTU-aligned, CTU size 64 only)");
  }
  l->n = n;
  if (n) {
    if (hipMalloc((void **)&l->d, sizeof(DTu) * n) != hipSuccess) {
      delete l;
      return fail(c, HMX_ERR_NOMEM, "hipMalloc tu list");
    }
    int r = hmx_upload(c, l->d, v.data(), sizeof(DTu) * n);
    if (r) {
      hipFree(l->d);
      delete l;
      return r;
    }
  }
  *out = l;
  return HMX_OK;
}
extern "C" void hmx_tu_list_destroy(hmx_ctx *c, hmx_tu_list *l) {
  (void)c;
  if (!l) return;
  hipFree(l->d);
  delete l;
}

static PlanesDev to_dev(const hmx_pic *p) {
  PlanesDev d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = p ? p->plane[i] : nullptr;
    d.s[i] = p ? p->stride[i] : 0;
  }
  return d;
}
static LevelsDev to_dev(const hmx_levels *p) {
  LevelsDev d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = p ? p->plane[i] : nullptr;
    d.s[i] = p ? p->stride[i] : 0;
  }
  return d;
}

// device copy of a small host table, valid for the kernels issued after it on the context's stream
static void *arena_push(hmx_ctx *c, const void *src, size_t bytes) {
  const size_t kCap = 8u << 20;
  if (!c->arena_h) {
    if (hipHostMalloc((void **)&c->arena_h, kCap, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (hipMalloc((void **)&c->arena_d, kCap) != hipSuccess) return nullptr;
    c->arena_cap = kCap;
  }
  const size_t raw = bytes;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > c->arena_cap) return nullptr;
  if (c->arena_head + bytes > c->arena_cap) { // wrap: everything issued so far has consumed its tables after this
    if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
    c->arena_head = 0;
  }
  char *h = c->arena_h + c->arena_head, *d = c->arena_d + c->arena_head;
  memcpy(h, src, raw);
  if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
  c->arena_head += bytes;
  return d;
}

static int run_list(hmx_ctx *c, int op, const hmx_tu_list *l, ListArgs A) {
  if (l->cnt[4] && op != OP_PRED) return fail(c, HMX_ERR_ARG, "the list holds 64x64 blocks: intra prediction only (the largest transform is 32x32)");
  for (int s = 0; s < 5; s++) {
    if (!l->cnt[s]) continue;
    A.tus = l->d + l->off[s];
    A.n = l->cnt[s];
    if (op == OP_TRANSFORM_RECON && s == 0 && A.pics && !A.abs_sum) { // 4x4 blocks: one lane per block
      hipLaunchKernelGGL(k_inter4, dim3((unsigned)((A.n + 255) / 256), (unsigned)A.n_pics), dim3(256), 0, c->stream, A);
      HIPCHK(c, hipGetLastError());
      continue;
    }
    if (op == OP_TRANSFORM_RECON && s == 3 && A.pics) { // 32x32 blocks: one wave per block on the matrix cores
      hipLaunchKernelGGL(k_inter32, dim3((unsigned)A.n, (unsigned)A.n_pics), dim3(64), 0, c->stream, A);
      HIPCHK(c, hipGetLastError());
      continue;
    }
    int r = launch_op(c, op, s + 2, A);
    if (r) return r;
  }
  return HMX_OK;
}

extern "C" int hmx_batch_transformNxN(hmx_ctx *c, const hmx_tu_list *l, const hmx_pic *resi, const hmx_levels *lev,
                                      uint32_t *d_abs_sum, const hmx_pic_param *pp) {
  if (!c || !l || !resi || !lev || !pp) return fail(c, HMX_ERR_ARG, "hmx_batch_transformNxN: null argument");
  ListArgs A{};
  A.a = to_dev(resi);
  A.lev = to_dev(lev);
  A.abs_sum = d_abs_sum;
  A.P = make_picdev(c, pp);
  return run_list(c, OP_TRANSFORM_NXN, l, A);
}

extern "C" int hmx_batch_residual_transformNxN(hmx_ctx *c, const hmx_tu_list *l, const hmx_pic *org, const hmx_pic *pred,
                                               const hmx_levels *lev, uint32_t *d_abs_sum, const hmx_pic_param *pp) {
  if (!c || !l || !org || !pred || !lev || !pp) return fail(c, HMX_ERR_ARG, "hmx_batch_residual_transformNxN: null argument");
  ListArgs A{};
  A.a = to_dev(org);
  A.b = to_dev(pred);
  A.have_pred = 1;
  A.lev = to_dev(lev);
  A.abs_sum = d_abs_sum;
  A.P = make_picdev(c, pp);
  return run_list(c, OP_TRANSFORM_NXN, l, A);
}

extern "C" int hmx_batch_invtransformNxN(hmx_ctx *c, const hmx_tu_list *l, const hmx_levels *lev, const hmx_pic *pred,
                                         const hmx_pic *out, const hmx_pic_param *pp) {
  if (!c || !l || !out || !lev || !pp) return fail(c, HMX_ERR_ARG, "hmx_batch_invtransformNxN: null argument");
  ListArgs A{};
  A.a = to_dev(pred);
  A.have_pred = pred != nullptr;
  A.b = to_dev(out);
  A.lev = to_dev(lev);
  A.P = make_picdev(c, pp);
  return run_list(c, OP_INVTRANSFORM_NXN, l, A);
}

static int run_list_multi(hmx_ctx *c, int op, const hmx_tu_list *l, int n_pics, const hmx_pic *a, const hmx_pic *b,
                          const hmx_levels *lev, uint32_t *d_abs_sum, const hmx_pic_param *pp, bool have_pred,
                          const hmx_pic *rec = nullptr, uint32_t *d_sse = nullptr) {
  std::vector<ListPic> t(n_pics);
  for (int i = 0; i < n_pics; i++) {
    t[i].a = to_dev(a ? &a[i] : nullptr);
    t[i].b = to_dev(b ? &b[i] : nullptr);
    t[i].lev = to_dev(&lev[i]);
    t[i].rec = to_dev(rec ? &rec[i] : nullptr);
  }
  ListArgs A{};
  A.pics = static_cast<const ListPic *>(arena_push(c, t.data(), sizeof(ListPic) * n_pics));
  if (!A.pics) return fail(c, HMX_ERR_NOMEM, "argument arena");
  A.n_pics = n_pics;
  A.abs_sum = d_abs_sum;
  A.sse = d_sse;
  A.abs_stride = l->n;
  A.have_pred = have_pred;
  A.P = make_picdev(c, pp);
  return run_list(c, op, l, A);
}

extern "C" int hmx_batch_residual_transformNxN_multi(hmx_ctx *c, const hmx_tu_list *l, int n_pics, const hmx_pic *org,
                                                     const hmx_pic *pred, const hmx_levels *lev, uint32_t *d_abs_sum,
                                                     const hmx_pic_param *pp) {
  if (!c || !l || !org || !pred || !lev || !pp || n_pics <= 0 || n_pics > 65535)
    return fail(c, HMX_ERR_ARG, "hmx_batch_residual_transformNxN_multi: bad argument");
  return run_list_multi(c, OP_TRANSFORM_NXN, l, n_pics, org, pred, lev, d_abs_sum, pp, true);
}

extern "C" int hmx_batch_residual_transform_recon_multi(hmx_ctx *c, const hmx_tu_list *l, int n_pics, const hmx_pic *org,
                                                       const hmx_pic *pred, const hmx_levels *lev, const hmx_pic *rec,
                                                       uint32_t *d_abs_sum, const hmx_pic_param *pp) {
  if (!c || !l || !org || !pred || !lev || !rec || !pp || n_pics <= 0 || n_pics > 65535)
    return fail(c, HMX_ERR_ARG, "hmx_batch_residual_transform_recon_multi: bad argument");
  return run_list_multi(c, OP_TRANSFORM_RECON, l, n_pics, org, pred, lev, d_abs_sum, pp, true, rec);
}

extern "C" int hmx_batch_residual_transform_recon_sse_multi(hmx_ctx *c, const hmx_tu_list *l, int n_pics, const hmx_pic *org,
                                                           const hmx_pic *pred, const hmx_levels *lev, const hmx_pic *rec,
                                                           uint32_t *d_abs_sum, uint32_t *d_sse, const hmx_pic_param *pp) {
  if (!c || !l || !org || !pred || !lev || !rec || !pp || n_pics <= 0 || n_pics > 65535)
    return fail(c, HMX_ERR_ARG, "hmx_batch_residual_transform_recon_sse_multi: bad argument");
  return run_list_multi(c, OP_TRANSFORM_RECON, l, n_pics, org, pred, lev, d_abs_sum, pp, true, rec, d_sse);
}

extern "C" int hmx_batch_invtransformNxN_multi(hmx_ctx *c, const hmx_tu_list *l, int n_pics, const hmx_levels *lev,
                                               const hmx_pic *pred, const hmx_pic *out, const hmx_pic_param *pp) {
  if (!c || !l || !out || !lev || !pp || n_pics <= 0 || n_pics > 65535)
    return fail(c, HMX_ERR_ARG, "hmx_batch_invtransformNxN_multi: bad argument");
  return run_list_multi(c, OP_INVTRANSFORM_NXN, l, n_pics, pred, out, lev, nullptr, pp, pred != nullptr);
}

extern "C" int hmx_batch_predIntra(hmx_ctx *c, const hmx_tu_list *l, const hmx_pic *rec, const hmx_pic *pred,
                                   const hmx_pic_param *pp, const uint8_t *d_modes, int n_modes,
                                   const size_t mode_plane_elems[3]) {
  if (!c || !l || !rec || !pred || !pp) return fail(c, HMX_ERR_ARG, "hmx_batch_predIntra: null argument");
  ListArgs A{};
  A.a = to_dev(rec);
  A.b = to_dev(pred);
  A.P = make_picdev(c, pp);
  A.modes = d_modes;
  A.n_modes = d_modes ? n_modes : 0;
  for (int i = 0; i < 3; i++) A.mode_elems[i] = mode_plane_elems ? mode_plane_elems[i] : 0;
  return run_list(c, OP_PRED, l, A);
}

extern "C" int hmx_batch_predIntra_cost(hmx_ctx *c, const hmx_tu_list *l, const hmx_pic *rec, const hmx_pic *org,
                                        const hmx_pic_param *pp, const uint8_t *d_modes, int n_modes, uint32_t *d_satd) {
  if (!c || !l || !rec || !org || !pp || !d_satd || (d_modes && (n_modes <= 0 || n_modes > 35)))
    return fail(c, HMX_ERR_ARG, "hmx_batch_predIntra_cost: bad argument");
  ListArgs A{};
  A.a = to_dev(rec);
  A.org = to_dev(org);
  A.cost = d_satd;
  A.P = make_picdev(c, pp);
  A.modes = d_modes;
  A.n_modes = d_modes ? n_modes : 0;
  return run_list(c, OP_PRED, l, A);
}

// ---- intra frame plan: dependency schedule ----
// Which neighbour units can the prediction of a block actually DEPEND on?  The availability mask says which neighbours exist;
// a mode reads only part of the reference line (a horizontal mode never looks above-right, DC and the negative angles stay
// inside left + above), and the order of the blocks only has to respect what is read.  The kernels still gather the whole
// line -- a unit nobody depends on may hold a stale reconstruction, which then sits in line positions the prediction does
// not touch.  Exactly as the prediction indexes its references (TComPrediction.cpp:179-290 xPredIntraAng, :689-730 planar,
// :129-167 DC, :1010-1029 DC filter), widened by one sample either side where the smoothed line is used (TComPattern.cpp:
// 265-306), and closed under the padding rule: an unavailable unit that is read takes its value from the nearest available
// unit before it (the first available one for a leading run, TComPattern.cpp:368-552).
// n_s = block size in samples, avail = intra_avail_mask's bits (units of 4 luma / 2 chroma samples).  Returns unit bits.
static unsigned long long intra_needed_units(int n_s, bool luma, int mode) {
  static const int ang_tab[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32}, inv_tab[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
  const int N = n_s, U = luma ? 4 : 2, n = N / U, lg = ilog2i(N);
  bool need[4 * 32 + 1] = {};
  // line position of above[k] (k = -1: corner) and left[k]
  auto above = [&](int k) { need[2 * N + 1 + k] = true; };
  auto left = [&](int k) { need[2 * N - 1 - k] = true; };
  if (mode == 0) { // planar
    for (int k = 0; k <= N; k++) above(k), left(k);
  } else if (mode == 1) { // DC (and its edge filter): left and above, N each
    for (int k = 0; k < N; k++) above(k), left(k);
  } else {
    const bool ver = mode >= 18;
    const int idx = ver ? mode - 26 : -(mode - 10);
    const int angle = (idx < 0 ? -1 : 1) * ang_tab[abs(idx)], inv_angle = inv_tab[abs(idx)];
    auto mainr = [&](int j) { // refMain[j], j >= 0; 0 = corner
      if (j == 0) need[2 * N] = true;
      else if (ver) above(j - 1);
      else left(j - 1);
    };
    auto side = [&](int j) {
      if (j == 0) need[2 * N] = true;
      else if (ver) left(j - 1);
      else above(j - 1);
    };
    if (angle == 0) {
      for (int l = 0; l < N; l++) mainr(l + 1);
      if (luma)
        for (int k = 0; k <= N; k++) side(k); // edge filter: refSide[k + 1] - refSide[0]
    } else {
      int acc = 128;
      const int lim = (N * angle) >> 5;
      int side_of[33]; // refMain[-j] = refSide[side_of[j]], j = 1 .. -lim - 1
      for (int k = -1; k > lim; k--) {
        acc += inv_angle;
        side_of[-k] = acc >> 8;
      }
      int pos = 0;
      for (int k = 0; k < N; k++) {
        pos += angle;
        const int di = pos >> 5, df = pos & 31;
        for (int l = 0; l < N; l++)
          for (int i = l + di + 1; i <= l + di + 1 + (df ? 1 : 0); i++) {
            if (i >= 0) mainr(i);
            else side(side_of[-i]);
          }
      }
    }
  }
  if (luma && mode != 1) { // the smoothed line: a sample of it is (raw[p - 1] + 2 raw[p] + raw[p + 1] + 2) >> 2
    const int dh = abs(mode - 10), dv = abs(mode - 26);
    static const int thr[4] = {10, 7, 1, 0};
    if ((dh < dv ? dh : dv) > thr[lg - 2]) {
      bool wide[4 * 32 + 1];
      for (int p = 0; p <= 4 * N; p++) wide[p] = need[p] || (p > 0 && need[p - 1]) || (p < 4 * N && need[p + 1]);
      for (int p = 0; p <= 4 * N; p++) need[p] = wide[p];
    }
  }
  unsigned long long units = 0;
  for (int p = 0; p <= 4 * N; p++)
    if (need[p]) units |= 1ull << (p < 2 * N ? p / U : p == 2 * N ? 2 * n : 2 * n + 1 + (p - 2 * N - 1) / U);
  return units;
}
static unsigned long long intra_dependency_mask(int n_s, bool luma, int mode, unsigned long long avail) {
  if ((n_s != 4 && n_s != 8 && n_s != 16 && n_s != 32) || mode < 0 || mode > 34) return avail; // not a mode this function knows: every neighbour
  struct Table { // what a mode reads depends on (size, texture type, mode) only: 280 masks, formed once
    unsigned long long u[4][2][35];
    Table() {
      for (int lg = 2; lg <= 5; lg++)
        for (int l = 0; l < 2; l++)
          for (int m = 0; m < 35; m++) u[lg - 2][l][m] = intra_needed_units(1 << lg, l != 0, m);
    }
  };
  static const Table T;
  const int n = n_s / (luma ? 4 : 2);
  const unsigned long long units = T.u[ilog2i(n_s) - 2][luma ? 1 : 0][mode];
  unsigned long long dep = units & avail;
  if (!(units & ~avail)) return dep;
  for (int u = 0; u <= 4 * n; u++) // padding: the value of an unavailable unit that is read
    if (((units >> u) & 1) && !((avail >> u) & 1) && avail) {
      const unsigned long long below = avail & ((1ull << u) - 1ull);
      dep |= below ? 1ull << (63 - __builtin_clzll(below)) : avail & (0 - avail);
    }
  return dep;
}
extern "C" unsigned long long hmx_intra_dependency_mask(int n_samples, int is_luma, int mode, unsigned long long avail) {
  return intra_dependency_mask(n_samples, is_luma != 0, mode, avail);
}

// The host half of a plan: the dependency analysis of one picture's decisions.  Touches nothing of the context but its
// configuration, so the pictures of a batch are analysed on as many host threads as there are (hmx_intra_plan_create_multi):
// 45 ms per 2160p picture on one core is 500x the picture's share of a whole-picture call.
struct PlanHost {
  std::vector<FTu> stus, ltus;
  std::vector<Seg> segs;
  std::vector<uint32_t> seg_range, level_chunks, wave_ctus;
  std::vector<LevelRow> ltab;
  std::vector<int> row_first, row_last;
  std::vector<std::pair<uint32_t, uint32_t>> waves;
  PicDev P;
  int n_tu = 0;
};
static const char *plan_build_host(const hmx_ctx *c, const hmx_tu *tus, int n_tu, const hmx_pic_param *pp, PlanHost &H) {
  const int ctu = c->cfg.ctu_size, U = ctu / 4;
  const int cw = (pp->pic_w + ctu - 1) / ctu, ch = (pp->pic_h + ctu - 1) / ctu, n_ctu = cw * ch;
  PicDev P = make_picdev(c, pp);
  // bucket blocks per (CTU, plane), keeping coding order
  std::vector<std::vector<int>> bucket((size_t)n_ctu * 3);
  for (int i = 0; i < n_tu; i++) {
    const hmx_tu &t = tus[i];
    if (t.plane > 2 || t.log2n < 2 || t.log2n > 5) return "hmx_intra_plan_create: bad block";
    const int sh = t.plane ? 1 : 0, lx = t.x << sh, ly = t.y << sh, ls = (1 << t.log2n) << sh;
    if ((lx % ctu) + ls > ctu || (ly % ctu) + ls > ctu) return "hmx_intra_plan_create: block crosses a CTU";
    // the CTU grid is padded, the caller's planes are not: a block in the padding would be written past their end
    if (lx + ls > pp->pic_w || ly + ls > pp->pic_h) return "hmx_intra_plan_create: block outside the picture";
    bucket[((size_t)(ly / ctu) * cw + lx / ctu) * 3 + t.plane].push_back(i);
  }
  std::vector<FTu> stus;
  stus.reserve(n_tu);
  std::vector<unsigned long long> masks(n_tu), deps(n_tu);
  std::vector<Seg> segs;
  std::vector<uint32_t> seg_range((size_t)n_ctu * 3 * 2);
  std::vector<int> level(n_tu);
  std::vector<int> grid((size_t)U * U);
  for (int b = 0; b < n_ctu * 3; b++) {
    std::fill(grid.begin(), grid.end(), 0);
    auto &ids = bucket[b];
    for (int id : ids) {
      const hmx_tu &t = tus[id];
      const int sh = t.plane ? 1 : 0, lx = t.x << sh, ly = t.y << sh, ls = (1 << t.log2n) << sh;
      const int n = ls / 4, cx = (lx % ctu) / 4, cy = (ly % ctu) / 4;
      unsigned long long m = intra_avail_mask(lx, ly, ls, P);
      masks[id] = m;
      m = intra_dependency_mask(1 << t.log2n, t.plane == 0, t.mode, m); // the order follows what the mode reads
      deps[id] = m;
      int lv = 0;
      auto dep = [&](int ux, int uy) { // unit coordinates relative to the CTU
        if (ux >= 0 && uy >= 0 && ux < U && uy < U) lv = std::max(lv, grid[uy * U + ux]);
      };
      for (int u = 0; u < 4 * n + 1; u++) {
        if (!((m >> u) & 1)) continue;
        if (u < 2 * n)
          dep(cx - 1, cy + 2 * n - 1 - u);
        else if (u == 2 * n)
          dep(cx - 1, cy - 1);
        else
          dep(cx + (u - 2 * n - 1), cy - 1);
      }
      level[id] = lv + 1;
      for (int j = 0; j < n; j++)
        for (int i2 = 0; i2 < n; i2++) grid[(cy + j) * U + cx + i2] = lv + 1;
    }
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b2) {
      if (level[a] != level[b2]) return level[a] < level[b2];
      return tus[a].log2n < tus[b2].log2n;
    });
    seg_range[(size_t)b * 2] = (uint32_t)segs.size();
    for (size_t k = 0; k < ids.size();) {
      size_t e = k;
      while (e < ids.size() && level[ids[e]] == level[ids[k]] && tus[ids[e]].log2n == tus[ids[k]].log2n &&
             e - k < 65535)
        e++;
      Seg s;
      s.start = (uint32_t)stus.size();
      s.count = (uint16_t)(e - k);
      s.log2n = tus[ids[k]].log2n;
      s.new_level = (k == 0 || level[ids[k]] != level[ids[k - 1]]) ? 1 : 0;
      segs.push_back(s);
      for (size_t q = k; q < e; q++)
        stus.push_back(FTu{tus[ids[q]], (uint32_t)masks[ids[q]], (uint32_t)(masks[ids[q]] >> 32)});
      k = e;
    }
    seg_range[(size_t)b * 2 + 1] = (uint32_t)segs.size();
  }
  // Picture-wide dependency levels (level schedule): level = 1 + max level of the blocks that
  // cover the available neighbour units, over the whole plane, blocks visited in coding order.
  std::vector<FTu> ltus(n_tu);
  std::vector<LevelRow> ltab;
  std::vector<uint32_t> level_chunks;
  std::vector<int> row_first, row_last;
  {
    const int uw = cw * U, uh = ch * U;
    std::vector<int> g3((size_t)uw * uh * 3, 0);
    std::vector<int> glevel(n_tu);
    int max_level = 0;
    for (int i = 0; i < n_tu; i++) {
      const hmx_tu &t = tus[i];
      const int sh = t.plane ? 1 : 0, lx = t.x << sh, ly = t.y << sh, ls = (1 << t.log2n) << sh;
      const int n = ls / 4, ux = lx / 4, uy = ly / 4;
      int *g = g3.data() + (size_t)t.plane * uw * uh;
      const unsigned long long m = deps[i];
      int lv = 0;
      for (int u = 0; u < 4 * n + 1; u++) {
        if (!((m >> u) & 1)) continue;
        int qx, qy;
        if (u < 2 * n) qx = ux - 1, qy = uy + 2 * n - 1 - u;
        else if (u == 2 * n) qx = ux - 1, qy = uy - 1;
        else qx = ux + (u - 2 * n - 1), qy = uy - 1;
        lv = std::max(lv, g[(size_t)qy * uw + qx]); // available => inside the picture
      }
      glevel[i] = lv; // zero-based level
      max_level = std::max(max_level, lv);
      for (int j = 0; j < n; j++)
        for (int i2 = 0; i2 < n; i2++) g[(size_t)(uy + j) * uw + ux + i2] = lv + 1;
    }
    row_first.assign(ch, 0x7fffffff);
    row_last.assign(ch, -1);
    for (int i = 0; i < n_tu; i++) {
      const int sh2 = tus[i].plane ? 1 : 0, r0 = (tus[i].y << sh2) / ctu, r1 = (((tus[i].y + (1 << tus[i].log2n)) << sh2) - 1) / ctu;
      for (int r = r0; r <= r1 && r < ch; r++) {
        row_first[r] = std::min(row_first[r], glevel[i]);
        row_last[r] = std::max(row_last[r], glevel[i]);
      }
    }
    ltab.assign((size_t)max_level + 1, LevelRow{{0, 0, 0, 0}, {0, 0, 0, 0}});
    for (int i = 0; i < n_tu; i++) ltab[glevel[i]].count[tus[i].log2n - 2]++;
    uint32_t off = 0;
    level_chunks.resize(ltab.size());
    for (size_t l = 0; l < ltab.size(); l++) {
      uint32_t chunks = 0;
      for (int sidx = 0; sidx < 4; sidx++) {
        ltab[l].start[sidx] = off;
        off += ltab[l].count[sidx];
        const uint32_t slots = sidx == 0 ? kSlots4Own : sidx == 1 ? 8 : sidx == 2 ? 4 : 1;
        chunks += (ltab[l].count[sidx] + slots - 1) / slots;
      }
      level_chunks[l] = chunks;
    }
    // Blocks of one (level, size) bucket are independent: order them so that the 64/N blocks that
    // share a wave take the same code paths (plane class = DST vs DCT and chroma rules, transform
    // skip, prediction mode class, then mode) instead of diverging.
    auto mode_class = [](int m) { return m == 0 ? 0 : m == 1 ? 1 : (m == 10 || m == 26) ? 2 : (m > 10 && m < 26) ? 3 : 4; };
    auto path_key = [&](const hmx_tu &t) {
      return (uint32_t)((t.plane ? 1u : 0u) << 24 | (uint32_t)(t.flags & 1u) << 20 | (uint32_t)mode_class(t.mode) << 16 |
                        (uint32_t)t.mode << 8 | t.plane);
    };
    // one 64-bit key per block (level | size | path | coding order): a plain sort of integers, no comparator that chases indices
    std::vector<uint64_t> order(n_tu);
    for (int i = 0; i < n_tu; i++)
      order[i] = ((uint64_t)(uint32_t)glevel[i] << 48) | ((uint64_t)(tus[i].log2n - 2) << 46) | ((uint64_t)(path_key(tus[i]) & 0x3ffffffu) << 20) |
                 (uint64_t)(uint32_t)i;
    static_assert(sizeof(int) == 4, "block index in the low 20 bits needs n_tu < 2^20");
    if (n_tu >= (1 << 20) || max_level >= (1 << 16)) return "hmx_intra_plan_create: picture too large for one plan";
    std::sort(order.begin(), order.end());
    for (int k = 0; k < n_tu; k++) {
      const int i = (int)(order[k] & 0xfffffu);
      ltus[k] = FTu{tus[i], (uint32_t)masks[i], (uint32_t)(masks[i] >> 32)};
    }
  }
  // CTU diagonals d = X + 2Y: (X,Y) needs (X-1,Y), (X-1,Y-1), (X,Y-1), (X+1,Y-1)
  for (int d = 0; d <= (cw - 1) + 2 * (ch - 1); d++) {
    uint32_t off = (uint32_t)H.wave_ctus.size();
    for (int Y = 0; Y < ch; Y++) {
      int X = d - 2 * Y;
      if (X >= 0 && X < cw) H.wave_ctus.push_back((uint32_t)(Y * cw + X));
    }
    H.waves.push_back({off, (uint32_t)H.wave_ctus.size() - off});
  }
  H.stus.swap(stus), H.ltus.swap(ltus), H.segs.swap(segs), H.seg_range.swap(seg_range), H.level_chunks.swap(level_chunks);
  H.ltab.swap(ltab), H.row_first.swap(row_first), H.row_last.swap(row_last);
  H.P = P, H.n_tu = n_tu;
  return nullptr;
}
// the device half: the tables go up in ONE allocation and one copy
static int plan_upload(hmx_ctx *c, PlanHost &H, const hmx_pic_param *pp, hmx_intra_plan **out) {
  hmx_intra_plan *pl = new hmx_intra_plan;
  static std::atomic<uint64_t> plan_serial{0}; // plans are created from several host threads / contexts: the serial is part of a cache key
  pl->serial = ++plan_serial;
  pl->level_chunks = H.level_chunks;
  for (const LevelRow &lr : H.ltab)
    for (int sidx = 0; sidx < 4; sidx++) pl->size_total[sidx] += lr.count[sidx];
  pl->h_ltab = H.ltab;
  pl->row_first_level = H.row_first;
  pl->row_last_level = H.row_last;
  pl->P = H.P;
  pl->n_tu = H.n_tu;
  pl->qp = pp->qp;
  pl->chroma_qp_offset = pp->chroma_qp_offset;
  pl->slice_type = pp->slice_type;
  for (auto &w : H.waves) pl->waves.push_back({w.first, w.second});
  auto up = [&](void **dp, const void *src, size_t bytes) -> int {
    if (hipMalloc(dp, bytes ? bytes : 4) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc plan");
    if (bytes) HIPCHK(c, hipMemcpyAsync(*dp, src, bytes, hipMemcpyHostToDevice, c->stream));
    return HMX_OK;
  };
  int r = up((void **)&pl->d_tus, H.stus.data(), H.stus.size() * sizeof(FTu));
  if (!r) r = up((void **)&pl->d_segs, H.segs.data(), H.segs.size() * sizeof(Seg));
  if (!r) r = up((void **)&pl->d_seg_range, H.seg_range.data(), H.seg_range.size() * sizeof(uint32_t));
  if (!r) r = up((void **)&pl->d_wave_ctus, H.wave_ctus.data(), H.wave_ctus.size() * sizeof(uint32_t));
  if (!r) r = up((void **)&pl->d_ltus, H.ltus.data(), H.ltus.size() * sizeof(FTu));
  if (!r) r = up((void **)&pl->d_ltab, H.ltab.data(), H.ltab.size() * sizeof(LevelRow));
  if (!r && hipStreamSynchronize(c->stream) != hipSuccess) r = fail(c, HMX_ERR_DEVICE, "plan upload"); // pageable sources
  if (r) {
    hmx_intra_plan_destroy(c, pl);
    return r;
  }
  *out = pl;
  return HMX_OK;
}
extern "C" int hmx_intra_plan_create(hmx_ctx *c, const hmx_tu *tus, int n_tu, const hmx_pic_param *pp, hmx_intra_plan **out) {
  if (!c || !tus || !pp || !out || n_tu <= 0) return fail(c, HMX_ERR_ARG, "hmx_intra_plan_create: bad argument");
  PlanHost H;
  if (const char *e = plan_build_host(c, tus, n_tu, pp, H)) return fail(c, HMX_ERR_ARG, e);
  return plan_upload(c, H, pp, out);
}
extern "C" int hmx_intra_plan_create_multi(hmx_ctx *c, const hmx_tu *const *tus, const int *n_tu, int n_pics, const hmx_pic_param *pp,
                                           hmx_intra_plan **out) {
  if (!c || !tus || !n_tu || !pp || !out || n_pics <= 0) return fail(c, HMX_ERR_ARG, "hmx_intra_plan_create_multi: bad argument");
  for (int i = 0; i < n_pics; i++) {
    out[i] = nullptr;
    if (!tus[i] || n_tu[i] <= 0) return fail(c, HMX_ERR_ARG, "hmx_intra_plan_create_multi: bad argument");
  }
  const int T = (int)std::max(1u, std::min({std::thread::hardware_concurrency(), (unsigned)n_pics, 32u}));
  int r = HMX_OK;
  for (int base = 0; base < n_pics && !r; base += 2 * T) { // chunks: a 2160p picture's host tables are ~15 MB
    const int n = std::min(2 * T, n_pics - base);
    // nothing thrown inside may leave a C entry point: allocation failures and thread-creation errors become HMX_ERR_NOMEM,
    // after every thread that did start has been joined
    std::vector<std::thread> th;
    bool oom = false;
    try {
      std::vector<PlanHost> H(n);
      std::vector<const char *> err(n, nullptr);
      static const char *const kOom = "hmx_intra_plan_create_multi: out of host memory";
      auto work = [&](int t) {
        for (int i = t; i < n; i += T) {
          try {
            err[i] = plan_build_host(c, tus[base + i], n_tu[base + i], pp, H[i]);
          } catch (...) {
            err[i] = kOom;
          }
        }
      };
      try {
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
      } catch (...) { // std::system_error: fewer threads than planned; their shares are picked up below
      }
      const int started = (int)th.size() + 1;
      work(0);
      for (auto &x : th) x.join();
      th.clear();
      for (int t = started; t < T; t++) work(t); // shares of the threads that could not be started
      for (int i = 0; i < n && !r; i++)
        r = err[i] == kOom ? fail(c, HMX_ERR_NOMEM, err[i]) : err[i] ? fail(c, HMX_ERR_ARG, err[i]) : plan_upload(c, H[i], pp, &out[base + i]);
    } catch (...) {
      oom = true;
    }
    for (auto &x : th)
      if (x.joinable()) x.join();
    if (oom) r = fail(c, HMX_ERR_NOMEM, "hmx_intra_plan_create_multi: out of host memory");
  }
  if (r)
    for (int i = 0; i < n_pics; i++)
      if (out[i]) hmx_intra_plan_destroy(c, out[i]), out[i] = nullptr;
  return r;
}

extern "C" int hmx_set_timing(hmx_ctx *c, int enable) {
  if (!c) return HMX_ERR_ARG;
  if (enable && !c->tev[0])
    for (int i = 0; i < 4; i++) HIPCHK(c, hipEventCreate(&c->tev[i]));
  c->timing = enable != 0;
  c->tev_valid = false;
  return HMX_OK;
}
extern "C" int hmx_last_call_timing(hmx_ctx *c, float *to_tiled_ms, float *chain_ms, float *from_tiled_ms) {
  if (!c || !c->tev_valid) return fail(c, HMX_ERR_ARG, "hmx_last_call_timing: no timed call");
  HIPCHK(c, hipEventSynchronize(c->tev[3]));
  float a = 0, b = 0, d = 0;
  HIPCHK(c, hipEventElapsedTime(&a, c->tev[0], c->tev[1]));
  HIPCHK(c, hipEventElapsedTime(&b, c->tev[1], c->tev[2]));
  HIPCHK(c, hipEventElapsedTime(&d, c->tev[2], c->tev[3]));
  if (to_tiled_ms) *to_tiled_ms = a;
  if (chain_ms) *chain_ms = b;
  if (from_tiled_ms) *from_tiled_ms = d;
  return HMX_OK;
}
extern "C" int hmx_intra_plan_info(const hmx_intra_plan *pl, int *n_blocks, int *n_levels, int *n_diagonals) {
  if (!pl) return HMX_ERR_ARG;
  if (n_blocks) *n_blocks = pl->n_tu;
  if (n_levels) *n_levels = (int)pl->level_chunks.size();
  if (n_diagonals) *n_diagonals = (int)pl->waves.size();
  return HMX_OK;
}
extern "C" int hmx_last_call_shape(const hmx_ctx *c, int *schedule, int *stream_groups) {
  if (!c) return HMX_ERR_ARG;
  if (schedule) *schedule = c->last_schedule;
  if (stream_groups) *stream_groups = c->last_groups;
  return HMX_OK;
}
extern "C" int hmx_intra_plan_level(const hmx_intra_plan *pl, int level, uint32_t counts[4], uint32_t *n_waves) {
  if (!pl || level < 0 || level >= (int)pl->h_ltab.size()) return HMX_ERR_ARG;
  if (counts)
    for (int s = 0; s < 4; s++) counts[s] = pl->h_ltab[level].count[s];
  if (n_waves) *n_waves = pl->level_chunks[level];
  return HMX_OK;
}
extern "C" int hmx_intra_schedule_for(const hmx_ctx *c, int n_pics) { // 3 = packed, 1 = level, 0 = wave
  (void)n_pics;
  return c->knob.schedule >= 0 ? c->knob.schedule : 3;
}

extern "C" void hmx_intra_plan_destroy(hmx_ctx *c, hmx_intra_plan *pl) {
  if (!pl) return;
  if (c) { // recorded graphs may refer to this plan (and its addresses may be re-used): drop them
    hipStreamSynchronize(c->stream);
    for (auto &e : c->graphs) {
      hipGraphExecDestroy(e.exec);
      hipFree(e.d_work);
    }
    c->graphs.clear();
  }
  hipFree(pl->d_tus);
  hipFree(pl->d_segs);
  hipFree(pl->d_seg_range);
  hipFree(pl->d_wave_ctus);
  hipFree(pl->d_ltus);
  hipFree(pl->d_ltab);
  delete pl;
}

// Issue the launches of one whole-picture call on `main` (and the side streams).  Also used under
// stream capture to record the call as a HIP graph.
static int issue_chain_launches(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                const PicWork *d_work, bool enc, bool use_level, int groups, hipStream_t main);
static int issue_packed(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics, bool enc, hipStream_t st);

// ---- packed schedule, host side ----
static int grow_dev(hmx_ctx *c, void **p, size_t *cap, size_t need) {
  if (need <= *cap) return HMX_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 16 + 256;
  if (hipMalloc(p, want) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc packed schedule tables");
  *cap = want;
  return HMX_OK;
}
static int issue_packed(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics, bool enc, hipStream_t st) {
  auto &pk = c->pk;
  const hmx_intra_plan *p0 = plans[0];
  PackGeom G{};
  G.n_pics = n_pics;
  G.I = c->pack_I;
  G.n_groups = (n_pics + G.I - 1) / G.I;
  G.n_shards = std::min(8, G.n_groups);
  uint64_t items = 0, sz[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_pics; i++) {
    const hmx_intra_plan *pl = plans[i * plan_stride];
    G.max_levels = std::max(G.max_levels, (int)pl->level_chunks.size());
    items += (uint64_t)pl->n_tu;
    for (int s = 0; s < 4; s++) sz[s] += pl->size_total[s];
  }
  // one lane per 4x4 block is the throughput shape, four lanes per block make more, shorter waves (small batches)
  // (measured with the mode-aware dependency order, 2160p mix: 64 lanes/wave-item ahead at 8..128 and from 384 pictures,
  // 16 ahead at 192 and 256)
  G.slots4 = c->knob.slots4 ? c->knob.slots4 : ((n_pics >= 160 && n_pics < 320) ? 16 : 64);
  const bool rdoq = enc && c->crq.n > 0;
  if (rdoq) {
    if (c->crq.n != 1 && c->crq.n != n_pics) return fail(c, HMX_ERR_ARG, "frame_intra: hmx_set_rdoq described another number of pictures");
    if (G.I > kRdoqMaxGroup) return fail(c, HMX_ERR_ARG, "frame_intra: RDOQ keeps the bit-estimate tables of a packing group in LDS: at most 2 pictures per group (HMX_PACK_GROUP)");
    G.slots4 = 64; // a 4x4 block's RDOQ runs inside one lane
  }
  const uint64_t n_rows = (uint64_t)G.max_levels * G.n_groups;
  uint64_t waves_bound = 4 * n_rows + 4;
  for (int s = 0; s < 4; s++) waves_bound += sz[s] / pack_slots(s, G.slots4);
  if (items >= 0xffffffffull || waves_bound >= 0x0fffffffull || n_rows >= 0x7fffffffull / 4)
    return fail(c, HMX_ERR_ARG, "frame_intra: batch too large for one packed call (split it)");
  const bool same = pk.valid && pk.key == c->table_key && pk.G.n_pics == G.n_pics && pk.G.I == G.I && pk.G.slots4 == G.slots4 &&
                    pk.G.max_levels == G.max_levels;
  if (!same) {
    pk.valid = false;
    int r = grow_dev(c, (void **)&pk.d_pics, &pk.cap_pics, sizeof(PackPic) * n_pics);
    if (!r) r = grow_dev(c, (void **)&pk.d_descs, &pk.cap_descs, sizeof(PackDesc) * waves_bound);
    if (!r) r = grow_dev(c, (void **)&pk.d_items, &pk.cap_items, sizeof(FTu) * items);
    if (!r) r = grow_dev(c, (void **)&pk.d_rows, &pk.cap_rows, sizeof(PackRow) * n_rows);
    if (!r) r = grow_dev(c, (void **)&pk.d_done, &pk.cap_done, sizeof(uint32_t) * kDoneStride * n_rows);
    if (!r && !pk.d_hdr) {
      if (hipMalloc((void **)&pk.d_hdr, sizeof(PackHdr)) != hipSuccess) r = fail(c, HMX_ERR_NOMEM, "hipMalloc packed header");
      else if (hipMemsetAsync(pk.d_hdr, 0, sizeof(PackHdr), st) != hipSuccess) r = fail(c, HMX_ERR_DEVICE, "hipMemsetAsync packed header");
    }
    if (r) return r;
    std::vector<PackPic> hp(n_pics);
    for (int i = 0; i < n_pics; i++) {
      const hmx_intra_plan *pl = plans[i * plan_stride];
      for (int p = 0; p < 3; p++) hp[i].lev[p] = c->call_lev[i].plane[p], hp[i].lev_stride[p] = c->call_lev[i].stride[p];
      hp[i].n_levels = (int)pl->level_chunks.size();
      hp[i].ltab = pl->d_ltab;
      hp[i].ltus = pl->d_ltus;
      for (int p = 0; p < 3; p++) hp[i].sse[p] = (enc && (int)c->sse_out.size() >= n_pics) ? c->sse_out[i].plane[p] : nullptr;
    }
    HIPCHK(c, hipMemcpyAsync(pk.d_pics, hp.data(), sizeof(PackPic) * n_pics, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st)); // hp goes out of scope
    HIPCHK(c, hipMemsetAsync(pk.d_hdr, 0, offsetof(PackHdr, abort), st)); // everything but the sticky abort word (below)
    const unsigned prep_waves = (unsigned)((n_rows + (uint64_t)(64 / G.I) - 1) / (uint64_t)(64 / G.I));
    hipLaunchKernelGGL(k_pack_count, dim3(prep_waves), dim3(64), 0, st, pk.d_pics, pk.d_rows, G, (int)n_rows);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, pk.d_rows, pk.d_hdr, G);
    hipLaunchKernelGGL(k_pack_fill, dim3(prep_waves), dim3(64), 0, st, pk.d_pics, pk.d_rows, pk.d_descs, pk.d_items, G, (int)n_rows);
    HIPCHK(c, hipGetLastError());
    pk.key = c->table_key;
    pk.G = G;
    pk.waves_bound = waves_bound;
    pk.valid = true;
  }
  // counters and ticket words start from zero every call
  HIPCHK(c, hipMemsetAsync(pk.d_done, 0, sizeof(uint32_t) * kDoneStride * n_rows, st));
  // The abort word is STICKY: it is cleared only by check_packed_abort after the host has read it (hmx_sync / hmx_download).
  // Calls queued behind a call whose dependency wait timed out see it set, leave at once and the next hmx_sync reports it --
  // a per-call clear would let call k+1 erase the failure of call k.
  HIPCHK(c, hipMemsetAsync(pk.d_hdr->ticket, 0, sizeof(PackHdr) - offsetof(PackHdr, ticket), st));
  const uint64_t wpl = waves_bound / (uint64_t)std::max(1, G.max_levels); // wave-items per dependency level, all groups
  if (!c->max_resident_waves) {
    int nb = 0;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->cfg.device));
    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_intra_packed<true, 64>, 64, 0));
    c->max_resident_waves = std::max(64, nb * prop.multiProcessorCount);
  }
  // Enough persistent waves to hold about two levels' worth of wave-items (the waves of the next row load their
  // descriptors and originals while the current row finishes), never more than the device keeps resident.
  int resident = c->max_resident_waves;
  if (rdoq) {
    if (!c->crq.max_waves) {
      int nb = 0;
      hipDeviceProp_t prop;
      HIPCHK(c, hipGetDeviceProperties(&prop, c->cfg.device));
      HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_intra_packed<true, 64, true, true>, 64, 0));
      c->crq.max_waves = std::max(64, nb * prop.multiProcessorCount);
    }
    resident = c->crq.max_waves;
  }
  const uint64_t want = std::max<uint64_t>(256, 2 * wpl);
  pk.n_wg = c->knob.pack_waves ? std::min(c->knob.pack_waves, resident) : (int)std::min<uint64_t>((uint64_t)resident, want);
  PackArgs A{};
  A.pics = pk.d_pics;
  A.rows = pk.d_rows;
  A.descs = pk.d_descs;
  A.items = pk.d_items;
  A.done = pk.d_done;
  A.hdr = pk.d_hdr;
  A.pool_org = c->pool_org;
  A.pool_rec = c->pool_rec;
  A.pic_elems = c->tiled_pic_elems;
  for (int p = 0; p < 3; p++) A.plane_off[p] = c->tiled_plane_off[p];
  A.ctu_w = c->tiled_cw;
  A.clog = ilog2i(p0->P.ctu);
  A.n_groups = G.n_groups;
  A.n_shards = G.n_shards;
  A.I = G.I;
  A.want_sse = enc && (int)c->sse_out.size() >= n_pics;
  {
    const hmx_levels *lv = c->call_lev;
    bool slab = true;
    for (int p = 0; p < 3 && slab; p++) {
      const ptrdiff_t d = n_pics > 1 ? (const char *)lv[1].plane[p] - (const char *)lv[0].plane[p] : 0;
      slab = d >= 0 && d % (ptrdiff_t)sizeof(int) == 0;
      for (int i = 0; i < n_pics && slab; i++)
        slab = (const char *)lv[i].plane[p] == (const char *)lv[0].plane[p] + (ptrdiff_t)i * d && lv[i].stride[p] == lv[0].stride[p];
      A.lev_base[p] = lv[0].plane[p];
      A.lev_pic_elems[p] = d / (ptrdiff_t)sizeof(int);
      A.lev_stride[p] = lv[0].stride[p];
    }
    A.lev_slab = slab ? 1 : 0;
  }
  A.sleep0 = c->knob.pack_sleep0 >= 0 ? c->knob.pack_sleep0 : 16;
  A.sleep1 = c->knob.pack_sleep1 >= 0 ? c->knob.pack_sleep1 : 2;
  A.P = p0->P;
  if (rdoq) {
#pragma clang fp contract(off)
    auto &q = c->crq;
    // lambda and the factor of the sign-hiding cost per picture, the error scale per size: the quotients are formed here, in
    // the reference's operation order (setErrScaleCoeff TComTrQuant.cpp:2794-2818, :2205)
    const int B = A.P.bit_depth, inc = B - 8;
    std::vector<double> up((size_t)q.n * 4);
    for (int t = 0; t < 2; t++) {
      const int qs = A.P.qd[t].q, per = A.P.qd[t].per_qbits, iq = A.P.qd[t].iq_scale >> per;
      for (int lg = 2; lg <= 5; lg++) {
        const int tshift = 15 - B - lg;
        double e = (double)(1 << 15);
        e = e * ldexp(1.0, -2 * tshift);
        e = e / (double)qs / (double)qs / (double)(1 << (2 * inc));
        A.rq.err_scale[t][lg - 2] = e;
      }
      for (int i = 0; i < q.n; i++) {
        const double lam = q.lambda[(size_t)i * 2 + t];
        up[(size_t)i * 2 + t] = lam;
        const long long f = (long long)((double)iq * (double)iq * (double)(1 << (2 * per)) / lam / 16 / (double)(1 << (2 * inc)) + 0.5);
        memcpy(&up[(size_t)q.n * 2 + (size_t)i * 2 + t], &f, sizeof(f));
      }
    }
    HIPCHK(c, hipMemcpyAsync(q.d_lambda, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st)); // `up` goes out of scope
    A.rq.est = q.d_est;
    A.rq.lambda = q.d_lambda;
    A.rq.rd_factor = reinterpret_cast<const long long *>(q.d_lambda + (size_t)q.n * 2);
    A.rq.pic_mul = q.n == 1 ? 0 : 1;
    A.rq.n_pics = n_pics;
  }
  const dim3 grid((unsigned)pk.n_wg), blk(64);
  if (rdoq) {
    if (A.want_sse) hipLaunchKernelGGL((k_intra_packed<true, 64, true, true>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((k_intra_packed<true, 64, false, true>), grid, blk, 0, st, A);
  } else if (A.want_sse) {
    if (G.slots4 == 64) hipLaunchKernelGGL((k_intra_packed<true, 64, true>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((k_intra_packed<true, 16, true>), grid, blk, 0, st, A);
  } else if (G.slots4 == 64) {
    if (enc) hipLaunchKernelGGL((k_intra_packed<true, 64>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((k_intra_packed<false, 64>), grid, blk, 0, st, A);
  } else {
    if (enc) hipLaunchKernelGGL((k_intra_packed<true, 16>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((k_intra_packed<false, 16>), grid, blk, 0, st, A);
  }
  HIPCHK(c, hipGetLastError());
  c->pk_pending = true;
  return HMX_OK;
}

// The across schedule with the layout conversions pipelined by CTU row.  The chain is latency-bound and leaves the
// memory system idle; the conversions are pure traffic.  CTU row r is converted in (stream `conv`) before the first
// dependency level that touches it and converted out after the last one, so both conversions hide behind the chain:
//   conv:   in(0) in(1) ... in(R-1)            wait(final 0) out(0)  wait(final 1) out(1) ...
//   group:  wait(in 0) level 0 ... wait(in r) level first[r] ... level last[r] record(final r) ...
static int issue_across_pipelined(hmx_ctx *c, const hmx_intra_plan *p0, int n_pics, const PicWork *d_work, const ConvJob *d_jobs,
                                  bool enc, int groups, hipStream_t main) {
  const int ctu = p0->P.ctu, cw = (p0->P.pic_w + ctu - 1) / ctu, ch = (p0->P.pic_h + ctu - 1) / ctu;
  int prio_lo = 0, prio_hi = 0; // the conversions are background traffic: lowest priority, the chain highest
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (!c->conv_stream) HIPCHK(c, hipStreamCreateWithPriority(&c->conv_stream, hipStreamNonBlocking, prio_lo));
  if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  if (!c->ev_conv_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_conv_join, hipEventDisableTiming));
  for (int g = c->n_side; g < groups; g++) {
    HIPCHK(c, hipStreamCreateWithPriority(&c->side[g], hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming));
    c->n_side = g + 1;
  }
  const size_t need = (size_t)ch * (1 + groups);
  while (c->ev_rows.size() < need) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_rows.push_back(e);
  }
  hipEvent_t *ev_in = c->ev_rows.data(), *ev_final = c->ev_rows.data() + ch; // ev_final[g * ch + r]
  hipStream_t conv = c->conv_stream;
  const unsigned spr = (unsigned)(cw * ctu + 63) / 64, strip_rows = (unsigned)(ctu + 63) / 64;
  const dim3 cgrid((unsigned)n_pics, spr * strip_rows, 3);
  // fork: everything starts after what is already on main
  HIPCHK(c, hipEventRecord(c->ev_fork, main));
  HIPCHK(c, hipStreamWaitEvent(conv, c->ev_fork, 0));
  for (int g = 0; g < groups; g++) HIPCHK(c, hipStreamWaitEvent(c->side[g], c->ev_fork, 0));
  if (c->timing) HIPCHK(c, hipEventRecord(c->tev[1], main)); // conversion-in is not a separate phase any more
  if (enc)
    for (int r = 0; r < ch; r++) {
      hipLaunchKernelGGL(k_convert_tiled<true>, cgrid, dim3(256), 0, conv, d_jobs, r * ctu, (r + 1) * ctu);
      HIPCHK(c, hipEventRecord(ev_in[r], conv));
    }
  AcrossArgs AA{};
  AA.ltus = p0->d_ltus;
  AA.pic_elems = c->tiled_pic_elems;
  for (int p = 0; p < 3; p++) AA.plane_off[p] = c->tiled_plane_off[p];
  AA.ctu_w = c->tiled_cw;
  AA.clog = 0;
  while ((1 << AA.clog) < ctu) AA.clog++;
  AA.P = p0->P;
  std::vector<int> first(groups + 1);
  for (int g = 0; g <= groups; g++) first[g] = (int)((long long)n_pics * g / groups);
  // rows in the order their first level comes up / their last level passes
  std::vector<int> by_first(ch), by_last(ch);
  for (int r = 0; r < ch; r++) by_first[r] = by_last[r] = r;
  std::stable_sort(by_first.begin(), by_first.end(), [&](int a, int b) { return p0->row_first_level[a] < p0->row_first_level[b]; });
  std::stable_sort(by_last.begin(), by_last.end(), [&](int a, int b) { return p0->row_last_level[a] < p0->row_last_level[b]; });
  int nf = 0, nl = 0;
  const size_t n_levels = p0->h_ltab.size();
  for (size_t l = 0; l < n_levels; l++) {
    if (enc)
      for (; nf < ch && p0->row_first_level[by_first[nf]] <= (int)l; nf++)
        for (int g = 0; g < groups; g++) HIPCHK(c, hipStreamWaitEvent(c->side[g], ev_in[by_first[nf]], 0));
    AA.row = p0->h_ltab[l];
    for (int g = 0; g < groups; g++) {
      const int np = first[g + 1] - first[g];
      if (np <= 0) continue;
      AA.pics = d_work + first[g];
      AA.n_pics = np;
      AA.pool_org = c->pool_org + (size_t)first[g] * c->tiled_pic_elems;
      AA.pool_rec = c->pool_rec + (size_t)first[g] * c->tiled_pic_elems;
      uint64_t waves = 0;
      for (int s2 = 0; s2 < 4; s2++) {
        const int slots = s2 == 0 ? kSlots4 : s2 == 1 ? 8 : s2 == 2 ? 4 : 1;
        AA.cpb[s2] = (uint32_t)((np + slots - 1) / slots);
        waves += (uint64_t)AA.row.count[s2] * AA.cpb[s2];
      }
      if (!waves) continue;
      if (waves > 0x7fffffffull) return fail(c, HMX_ERR_ARG, "frame_intra: level too large for one launch");
      if (enc)
        hipLaunchKernelGGL(k_intra_level_across<true>, dim3((unsigned)waves), dim3(64), 0, c->side[g], AA);
      else
        hipLaunchKernelGGL(k_intra_level_across<false>, dim3((unsigned)waves), dim3(64), 0, c->side[g], AA);
    }
    for (; nl < ch && p0->row_last_level[by_last[nl]] <= (int)l; nl++) { // these rows are final: convert them out
      const int r = by_last[nl];
      for (int g = 0; g < groups; g++) {
        HIPCHK(c, hipEventRecord(ev_final[g * ch + r], c->side[g]));
        HIPCHK(c, hipStreamWaitEvent(conv, ev_final[g * ch + r], 0));
      }
      hipLaunchKernelGGL(k_convert_tiled<false>, cgrid, dim3(256), 0, conv, d_jobs + (size_t)n_pics * 3, r * ctu, (r + 1) * ctu);
    }
  }
  // join
  for (int g = 0; g < groups; g++) {
    HIPCHK(c, hipEventRecord(c->ev_join[g], c->side[g]));
    HIPCHK(c, hipStreamWaitEvent(main, c->ev_join[g], 0));
  }
  if (c->timing) HIPCHK(c, hipEventRecord(c->tev[2], main)); // the chain is done
  HIPCHK(c, hipEventRecord(c->ev_conv_join, conv));
  HIPCHK(c, hipStreamWaitEvent(main, c->ev_conv_join, 0));
  return HMX_OK;
}

static int issue_intra_launches(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                const PicWork *d_work, const ConvJob *d_jobs, bool enc, bool use_level, int groups,
                                hipStream_t main) {
  // original planes -> tiled working copies (encode), chain, tiled reconstruction -> caller's planes
  const hmx_intra_plan *p0 = plans[0];
  const int cw = (p0->P.pic_w + p0->P.ctu - 1) / p0->P.ctu, ch = (p0->P.pic_h + p0->P.ctu - 1) / p0->P.ctu;
  // 64 x 64 regions of the padded luma plane (the chroma planes need a quarter of them; the rest exit)
  const unsigned spr = (unsigned)(cw * p0->P.ctu + 63) / 64, rows = (unsigned)(ch * p0->P.ctu + 63) / 64;
  dim3 cgrid((unsigned)n_pics, spr * rows, 3);
  const bool tm = c->timing;
  if (tm) HIPCHK(c, hipEventRecord(c->tev[0], main));
  if (c->across_call && c->pipeline_conv) {
    int r = issue_across_pipelined(c, p0, n_pics, d_work, d_jobs, enc, groups, main);
    if (r) return r;
    if (tm) {
      HIPCHK(c, hipEventRecord(c->tev[3], main));
      c->tev_valid = true;
    }
    HIPCHK(c, hipGetLastError());
    return HMX_OK;
  }
  const bool conv = !c->resident_call;
  if (conv && enc) hipLaunchKernelGGL(k_convert_tiled<true>, cgrid, dim3(256), 0, main, d_jobs, 0, 1 << 30);
  if (conv && c->onto_call) // the pool starts from the caller's reconstruction (the inter-coded parts of the picture)
    hipLaunchKernelGGL(k_convert_tiled<true>, cgrid, dim3(256), 0, main, d_jobs + (size_t)n_pics * 3, 0, 1 << 30);
  if (tm) HIPCHK(c, hipEventRecord(c->tev[1], main));
  int r = c->last_schedule == 3 ? issue_packed(c, plans, plan_stride, n_pics, enc, main)
                                : issue_chain_launches(c, plans, plan_stride, n_pics, d_work, enc, use_level, groups, main);
  if (r) return r;
  if (tm) HIPCHK(c, hipEventRecord(c->tev[2], main));
  if (conv) hipLaunchKernelGGL(k_convert_tiled<false>, cgrid, dim3(256), 0, main, d_jobs + (size_t)n_pics * 3, 0, 1 << 30);
  if (tm) {
    HIPCHK(c, hipEventRecord(c->tev[3], main));
    c->tev_valid = true;
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

static int issue_chain_launches(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                const PicWork *d_work, bool enc, bool use_level, int groups, hipStream_t main) {
  const hmx_intra_plan *p0 = plans[0];
  if (use_level) {
    // Pictures are split into groups; each group walks its levels on its own stream.  A launch
    // of one group fills only part of the chip (its duration is one block-chain latency), so
    // launches of different groups overlap.  Fork/join through events on the main stream.
    if (groups > 1) HIPCHK(c, hipEventRecord(c->ev_fork, main));
    std::vector<int> first(groups + 1);
    std::vector<size_t> glevels(groups, 0);
    for (int g = 0; g <= groups; g++) first[g] = (int)((long long)n_pics * g / groups);
    for (int g = 0; g < groups; g++) {
      if (groups > 1) HIPCHK(c, hipStreamWaitEvent(c->side[g], c->ev_fork, 0));
      for (int i = first[g]; i < first[g + 1]; i++) glevels[g] = std::max(glevels[g], plans[i * plan_stride]->level_chunks.size());
    }
    size_t n_levels = 0;
    for (int g = 0; g < groups; g++) n_levels = std::max(n_levels, glevels[g]);
    // one plan for every picture: SIMD across pictures (k_intra_level_across); every group of pictures
    // is its own interleave domain of the pool (see frame_intra) and walks the levels on its own stream
    if (c->across_call) {
      AcrossArgs AA{};
      AA.ltus = p0->d_ltus;
      AA.pic_elems = c->tiled_pic_elems;
      for (int p = 0; p < 3; p++) AA.plane_off[p] = c->tiled_plane_off[p];
      AA.ctu_w = c->tiled_cw;
      AA.clog = 0;
      while ((1 << AA.clog) < p0->P.ctu) AA.clog++;
      AA.P = p0->P;
      for (size_t l = 0; l < n_levels; l++) {
        AA.row = p0->h_ltab[l];
        for (int g = 0; g < groups; g++) {
          const int np = first[g + 1] - first[g];
          if (np <= 0) continue;
          AA.pics = d_work + first[g];
          AA.n_pics = np;
          AA.pool_org = c->pool_org + (size_t)first[g] * c->tiled_pic_elems;
          AA.pool_rec = c->pool_rec + (size_t)first[g] * c->tiled_pic_elems;
          uint64_t waves = 0;
          for (int s2 = 0; s2 < 4; s2++) {
            const int slots = s2 == 0 ? kSlots4 : s2 == 1 ? 8 : s2 == 2 ? 4 : 1;
            AA.cpb[s2] = (uint32_t)((np + slots - 1) / slots);
            waves += (uint64_t)AA.row.count[s2] * AA.cpb[s2];
          }
          if (!waves) continue;
          if (waves > 0x7fffffffull) return fail(c, HMX_ERR_ARG, "frame_intra: level too large for one launch");
          hipStream_t st = groups > 1 ? c->side[g] : main;
          if (enc)
            hipLaunchKernelGGL(k_intra_level_across<true>, dim3((unsigned)waves), dim3(64), 0, st, AA);
          else
            hipLaunchKernelGGL(k_intra_level_across<false>, dim3((unsigned)waves), dim3(64), 0, st, AA);
        }
      }
      if (groups > 1)
        for (int g = 0; g < groups; g++) {
          HIPCHK(c, hipEventRecord(c->ev_join[g], c->side[g]));
          HIPCHK(c, hipStreamWaitEvent(main, c->ev_join[g], 0));
        }
      HIPCHK(c, hipGetLastError());
      return HMX_OK;
    }
    LevelArgs LA{};
    LA.P = p0->P;
    for (size_t l = 0; l < n_levels; l++)
      for (int g = 0; g < groups; g++) {
        if (l >= glevels[g]) continue;
        uint32_t chunks = 0;
        if (plan_stride == 0)
          chunks = p0->level_chunks[l];
        else
          for (int i = first[g]; i < first[g + 1]; i++) {
            const auto &lc = plans[i]->level_chunks;
            if (l < lc.size()) chunks = std::max(chunks, lc[l]);
          }
        if (!chunks) continue;
        LA.pics = d_work + first[g];
        LA.level = (int)l;
        LA.shared = plan_stride == 0;
        if (LA.shared) {
          LA.row = p0->h_ltab[l];
          LA.ltus = p0->d_ltus;
        }
        dim3 grid(chunks, (unsigned)(first[g + 1] - first[g]));
        hipStream_t st = groups > 1 ? c->side[g] : main;
        if (enc)
          hipLaunchKernelGGL(k_intra_level<true>, grid, dim3(64), 0, st, LA);
        else
          hipLaunchKernelGGL(k_intra_level<false>, grid, dim3(64), 0, st, LA);
      }
    if (groups > 1)
      for (int g = 0; g < groups; g++) {
        HIPCHK(c, hipEventRecord(c->ev_join[g], c->side[g]));
        HIPCHK(c, hipStreamWaitEvent(main, c->ev_join[g], 0));
      }
    HIPCHK(c, hipGetLastError());
    return HMX_OK;
  }
  FrameArgs A;
  A.pics = d_work;
  A.P = p0->P;
  for (auto &w : p0->waves) {
    if (!w.second) continue;
    A.wave_ctus = p0->d_wave_ctus + w.first;
    A.n_wave_ctus = (int)w.second;
    dim3 grid((unsigned)(w.second * n_pics * 3));
    if (enc)
      hipLaunchKernelGGL(k_intra_wave<true>, grid, dim3(64), 0, main, A);
    else
      hipLaunchKernelGGL(k_intra_wave<false>, grid, dim3(64), 0, main, A);
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

// Pictures per group of the packed schedule.  A group is an interleave domain of the pool and the unit that advances level
// by level: its pictures move in lockstep, and a row is complete only when its slowest wave-item is.  Small groups keep
// that coupling small (measured at 2048 pictures of 2160p, 64 distinct plans: 70 / 79 / 86 / 93 / 94 / 93 Gpx/s with groups
// of 64 / 32 / 16 / 8 / 4 / 2); what is left of "packing across pictures" at 4 is enough to fill the waves of the large
// batches, and small batches are latency-bound whatever the packing (256 pictures: 29 / 32 Gpx/s with 4 / 2; 64
// pictures: 9.5 / 9.9 with 2 / 1).  Groups are dealt to the 8 XCDs round-robin.
static int pack_group_size(const hmx_ctx *c, int n_pics) {
  if (c && c->knob.pack_group > 0) return std::min(c->knob.pack_group, n_pics);
  if (c && c->crq.n > 0) return n_pics >= 512 ? kRdoqMaxGroup : 1; // RDOQ on (hmx_set_rdoq): the tables of a group wait in LDS
  return n_pics >= 1536 ? 4 : n_pics >= 512 ? 2 : 1;
}
// ---- pictures resident in the working layout (include/hmx.h: hmx_tpool) ----
struct hmx_tpool {
  short *base = nullptr;
  int n_pics = 0, I = 1;   // groups of I pictures are interleaved quad by quad
  int cw = 0, ch = 0, ctu = 64, pic_w = 0, pic_h = 0;
  size_t pic_elems = 0;
  uint32_t plane_off[3] = {0, 0, 0};
};
static TiledPlane tpool_plane(const hmx_tpool *t, int i, int p) {
  const int g0 = i / t->I * t->I, clog = ilog2i(t->ctu);
  const size_t base = (size_t)g0 * t->pic_elems + (size_t)t->plane_off[p] * t->I + (size_t)(i - g0) * 64;
  return TiledPlane{t->base + base, t->cw, p ? clog - 1 : clog, 64u * (unsigned)t->I};
}
extern "C" int hmx_tpool_create(hmx_ctx *c, int pic_w, int pic_h, int n_pics, hmx_tpool **out) {
  if (!c || !out || pic_w <= 0 || pic_h <= 0 || n_pics <= 0) return fail(c, HMX_ERR_ARG, "hmx_tpool_create: bad argument");
  hmx_tpool *t = new hmx_tpool;
  t->ctu = c->cfg.ctu_size;
  t->pic_w = pic_w, t->pic_h = pic_h;
  t->cw = (pic_w + t->ctu - 1) / t->ctu, t->ch = (pic_h + t->ctu - 1) / t->ctu;
  t->n_pics = n_pics;
  t->I = pack_group_size(c, n_pics);
  size_t off = 0;
  for (int p = 0; p < 3; p++) {
    t->plane_off[p] = (uint32_t)off;
    off += (size_t)t->cw * t->ch * ((size_t)t->ctu * t->ctu >> (p ? 2 : 0));
  }
  t->pic_elems = off;
  const size_t slots = (size_t)(n_pics + t->I - 1) / t->I * t->I;
  if (hipMalloc((void **)&t->base, off * 2 * slots) != hipSuccess) {
    delete t;
    return fail(c, HMX_ERR_NOMEM, "hipMalloc resident pictures");
  }
  *out = t;
  return HMX_OK;
}
extern "C" void hmx_tpool_destroy(hmx_ctx *c, hmx_tpool *t) {
  if (!t) return;
  if (c) {
    hipStreamSynchronize(c->stream);
    c->table_valid = false; // a later pool may get the same address
    c->pk.valid = false;
  }
  hipFree(t->base);
  delete t;
}
static int tpool_convert(hmx_ctx *c, const hmx_tpool *t, int first, int n, const hmx_pic *planes, bool to_tiled) {
  if (!c || !t || !planes || first < 0 || n <= 0 || first + n > t->n_pics) return fail(c, HMX_ERR_ARG, "hmx_tpool import/export: bad argument");
  std::vector<ConvJob> jobs((size_t)n * 3);
  for (int i = 0; i < n; i++)
    for (int p = 0; p < 3; p++)
      jobs[(size_t)i * 3 + p] = ConvJob{planes[i].plane[p], planes[i].stride[p], t->pic_w >> (p ? 1 : 0), t->pic_h >> (p ? 1 : 0), tpool_plane(t, first + i, p)};
  for (size_t done = 0; done < jobs.size();) { // through the argument arena, a few thousand jobs at a time
    const size_t part = std::min(jobs.size() - done, (size_t)3 * 8192);
    const ConvJob *d = static_cast<const ConvJob *>(arena_push(c, jobs.data() + done, sizeof(ConvJob) * part));
    if (!d) return fail(c, HMX_ERR_NOMEM, "argument arena");
    const unsigned spr = (unsigned)(t->cw * t->ctu + 63) / 64, rows = (unsigned)(t->ch * t->ctu + 63) / 64;
    const dim3 grid((unsigned)(part / 3), spr * rows, 3);
    if (to_tiled) hipLaunchKernelGGL(k_convert_tiled<true>, grid, dim3(256), 0, c->stream, d, 0, 1 << 30);
    else hipLaunchKernelGGL(k_convert_tiled<false>, grid, dim3(256), 0, c->stream, d, 0, 1 << 30);
    done += part;
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}
extern "C" int hmx_tpool_import(hmx_ctx *c, hmx_tpool *t, int first, int n, const hmx_pic *src) { return tpool_convert(c, t, first, n, src, true); }
extern "C" int hmx_tpool_export(hmx_ctx *c, const hmx_tpool *t, int first, int n, const hmx_pic *dst) { return tpool_convert(c, t, first, n, dst, false); }

// org / rec: pictures in plane geometry (converted into / out of the context's own working pools around the chain), or
// NULL with torg / trec: pictures resident in the working layout (no conversion; packed schedule only)
static int frame_intra(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics, const hmx_pic *org,
                       const hmx_pic *rec, const hmx_levels *lev, bool enc, const hmx_tpool *torg = nullptr,
                       const hmx_tpool *trec = nullptr) {
  const bool resident = trec != nullptr;
  if (!c || !plans || !plans[0] || n_pics <= 0 || !lev || (!resident && (!rec || (enc && !org))) || (resident && enc && !torg))
    return fail(c, HMX_ERR_ARG, "frame_intra: null argument");
  const hmx_intra_plan *p0 = plans[0];
  const int ctu = p0->P.ctu, cw = (p0->P.pic_w + ctu - 1) / ctu, ch = (p0->P.pic_h + ctu - 1) / ctu;
  const int clog = ilog2i(ctu);
  // Schedules.  "packed" (default): ONE persistent launch, blocks of equal size and dependency level
  // packed into waves across pictures, each picture following its own plan, the dependency order kept by counters in
  // memory.  The level-synchronous schedules stay as cross-checks and for A/B runs (HMX_INTRA_SCHEDULE=level|wave):
  // "level" = one launch per picture-wide dependency level (pictures that share ONE plan run it across pictures on a
  // pool interleaved per stream group), "wave" = one launch per CTU diagonal with autonomous waves.
  const int sched_base = resident ? 3 : c->knob.schedule >= 0 ? c->knob.schedule : 3;
  const bool packed = sched_base == 3, use_level = sched_base == 1;
  const bool across = use_level && plan_stride == 0 && c->knob.across != 0;
  // Picture groups of the across schedule on separate streams: measured at 1024 pictures 64.8 / 73.6 / 76.1 / 51.8 Gpx/s
  // with 1 / 2 / 3 / 4 groups.
  int groups = !across ? 1 : n_pics >= 640 ? 3 : n_pics >= 384 ? 2 : 1;
  if (use_level && c->knob.streams > 0) groups = std::min(std::max(c->knob.streams, 1), std::min(n_pics, (int)hmx_ctx::kMaxSide));
  // packed: groups of I pictures are the interleave domains of the pool and the lanes of the tables' prep kernels
  const int I = resident ? trec->I : packed ? pack_group_size(c, n_pics) : 1, pool_need = packed ? (n_pics + I - 1) / I * I : n_pics;
  {
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
      c->tiled_plane_off[p] = (uint32_t)off;
      off += (size_t)cw * ch * ((size_t)ctu * ctu >> (p ? 2 : 0));
    }
    c->tiled_pic_elems = off;
  }
  if (resident) { // the caller's pools: same geometry as the plans, at least n_pics pictures, one interleave
    for (const hmx_tpool *t : {trec, enc ? torg : trec})
      if (t->cw != cw || t->ch != ch || t->ctu != ctu || t->pic_w != p0->P.pic_w || t->pic_h != p0->P.pic_h || t->n_pics < n_pics || t->I != I)
        return fail(c, HMX_ERR_ARG, "frame_intra: resident pool does not match the call (picture size, CTU size, pictures, group size)");
    c->pool_org = enc ? torg->base : nullptr;
    c->pool_rec = trec->base;
    c->tiled_cw = cw, c->tiled_ch = ch;
  } else {
    // the context's own working pools: one slot per picture, planes padded to whole CTUs
    if (c->own_cw != cw || c->own_ch != ch || c->pool_pics < pool_need) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipFree(c->own_pool_org);
      hipFree(c->own_pool_rec);
      c->own_pool_org = c->own_pool_rec = nullptr;
      c->pool_pics = 0;
      c->own_cw = cw, c->own_ch = ch;
      c->table_valid = false;
      c->pk.valid = false;
      if (hipMalloc((void **)&c->own_pool_org, c->tiled_pic_elems * 2 * pool_need) != hipSuccess ||
          hipMalloc((void **)&c->own_pool_rec, c->tiled_pic_elems * 2 * pool_need) != hipSuccess) {
        hipFree(c->own_pool_org);
        c->own_pool_org = nullptr;
        return fail(c, HMX_ERR_NOMEM, "hipMalloc tiled working pictures");
      }
      c->pool_pics = pool_need;
    }
    c->pool_org = c->own_pool_org, c->pool_rec = c->own_pool_rec;
    c->tiled_cw = cw, c->tiled_ch = ch;
  }
  c->resident_call = resident;
  c->across_call = across;
  c->pack_I = I;
  c->call_lev = lev;
  // Conversions pipelined with the chain, CTU row by CTU row (issue_across_pipelined): opt-in, across schedule only.
  c->pipeline_conv = c->knob.pipeline_conv && across && !c->knob.graph && !c->onto_call;
  c->last_schedule = packed ? 3 : !use_level ? 0 : (across ? 2 : 1);
  c->last_groups = groups;
  std::vector<PicWork> hw(n_pics);
  std::vector<ConvJob> jobs((size_t)n_pics * 6);
  for (int i = 0; i < n_pics; i++) {
    const hmx_intra_plan *pl = plans[i * plan_stride];
    if (!pl || pl->P.pic_w != p0->P.pic_w || pl->P.pic_h != p0->P.pic_h || pl->qp != p0->qp ||
        pl->chroma_qp_offset != p0->chroma_qp_offset || pl->slice_type != p0->slice_type ||
        pl->P.sign_hide != p0->P.sign_hide)
      return fail(c, HMX_ERR_ARG, "frame_intra: plans of one call must share picture size and quantiser settings");
    memset(&hw[i], 0, sizeof(PicWork));
    for (int p = 0; p < 3; p++) {
      const int pclog = p ? clog - 1 : clog, pw = p0->P.pic_w >> (p ? 1 : 0), ph = p0->P.pic_h >> (p ? 1 : 0);
      // interleave domain [g0, g1) of picture i: a stream group (across), a group of I pictures (packed), itself
      int g0 = i, g1 = i + 1;
      if (packed) {
        g0 = i / I * I, g1 = g0 + I;
      } else if (across) {
        const int g = (int)(((long long)(i + 1) * groups - 1) / n_pics); // the g with first[g] <= i < first[g+1]
        g0 = (int)((long long)n_pics * g / groups), g1 = (int)((long long)n_pics * (g + 1) / groups);
      }
      const size_t base = (size_t)g0 * c->tiled_pic_elems + (size_t)c->tiled_plane_off[p] * (g1 - g0) + (size_t)(i - g0) * 64;
      const unsigned qstride = 64u * (unsigned)(g1 - g0);
      hw[i].org[p] = TiledPlane{c->pool_org ? c->pool_org + base : nullptr, cw, pclog, qstride};
      hw[i].rec[p] = TiledPlane{c->pool_rec + base, cw, pclog, qstride};
      hw[i].lev[p] = lev[i].plane[p];
      hw[i].lev_stride[p] = lev[i].stride[p];
      if (!resident) {
        if (enc) jobs[(size_t)i * 3 + p] = ConvJob{org[i].plane[p], org[i].stride[p], pw, ph, hw[i].org[p]};
        jobs[(size_t)(n_pics + i) * 3 + p] = ConvJob{rec[i].plane[p], rec[i].stride[p], pw, ph, hw[i].rec[p]};
      }
    }
    hw[i].tus = pl->d_tus;
    hw[i].segs = pl->d_segs;
    hw[i].seg_range = pl->d_seg_range;
    hw[i].ltus = pl->d_ltus;
    hw[i].ltab = pl->d_ltab;
    hw[i].n_levels = (int)pl->level_chunks.size();
  }
  if (use_level && groups > 1 && c->n_side < groups) {
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int g = c->n_side; g < groups; g++) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->side[g], hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming));
    }
    c->n_side = groups;
  }
  // Key of the call = the picture table itself (planes, level buffers, plans, schedule).  A steady-state pipeline
  // re-uses its picture pools: the device copy of the table (and the packed schedule's tables) is then kept as it is,
  // and the call is queued behind the previous one without any synchronisation.
  uint64_t key = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) key = (key ^ b[i]) * 1099511628211ull;
  };
  mix(hw.data(), sizeof(PicWork) * n_pics);
  mix(jobs.data(), sizeof(ConvJob) * jobs.size());
  const int flags[6] = {enc, c->last_schedule, groups, n_pics, across, I};
  mix(flags, sizeof(flags));
  if (enc && c->crq.n > 0) {
    if (!packed) return fail(c, HMX_ERR_ARG, "frame_intra: RDOQ as the quantiser (hmx_set_rdoq) needs the packed schedule");
    mix(&c->crq.serial, sizeof(c->crq.serial)); // 4x4 blocks then always go one per lane: another table
  }
  if (enc && (int)c->sse_out.size() >= n_pics) {
    if (!packed) return fail(c, HMX_ERR_ARG, "frame_intra: the distortion output (hmx_set_sse_output) needs the packed schedule");
    mix(c->sse_out.data(), sizeof(hmx_sse) * n_pics);
  }
  for (int i = 0; i < n_pics; i++) {
    const hmx_intra_plan *pp = plans[i * plan_stride];
    mix(&pp, sizeof(pp));
    mix(&pp->serial, sizeof(pp->serial)); // a destroyed plan's address may come back
  }
  const bool use_graph = c->knob.graph && !packed; // measured: replay is not faster than eager launches here
  hmx_ctx::GraphEntry *hit = nullptr;
  for (auto &e : c->graphs)
    if (e.key == key && e.n_pics == n_pics) hit = &e;
  if (use_graph && hit) {
    hit->stamp = ++c->graph_clock;
    HIPCHK(c, hipGraphLaunch(hit->exec, c->stream));
    return HMX_OK;
  }
  PicWork *d_work = nullptr;
  const size_t table_bytes = sizeof(PicWork) * n_pics + sizeof(ConvJob) * jobs.size();
  if (!use_graph) { // eager path: one grow-only table in the context
    if ((int)table_bytes > c->jobs_cap) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipFree(c->d_jobs);
      c->jobs_cap = 0;
      c->table_valid = false;
      if (hipMalloc((void **)&c->d_jobs, table_bytes) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc picture table");
      c->jobs_cap = (int)table_bytes;
    }
    char *base = reinterpret_cast<char *>(c->d_jobs);
    if (!c->table_valid || c->table_key != key) {
      // earlier calls may still read the table: the copies are ordered behind them on the stream; the host vectors go
      // out of scope, hence the synchronisation -- on this path only
      HIPCHK(c, hipMemcpyAsync(base, hw.data(), sizeof(PicWork) * n_pics, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(base + sizeof(PicWork) * n_pics, jobs.data(), sizeof(ConvJob) * jobs.size(), hipMemcpyHostToDevice,
                               c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->table_key = key;
      c->table_valid = true;
    }
    return issue_intra_launches(c, plans, plan_stride, n_pics, reinterpret_cast<PicWork *>(base),
                                reinterpret_cast<ConvJob *>(base + sizeof(PicWork) * n_pics), enc, use_level, groups, c->stream);
  }
  if (hipMalloc((void **)&d_work, table_bytes) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc picture table");
  HIPCHK(c, hipMemcpyAsync(d_work, hw.data(), sizeof(PicWork) * n_pics, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(reinterpret_cast<char *>(d_work) + sizeof(PicWork) * n_pics, jobs.data(),
                           sizeof(ConvJob) * jobs.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream)); // hw / jobs go out of scope
  const ConvJob *d_jobs = reinterpret_cast<const ConvJob *>(reinterpret_cast<char *>(d_work) + sizeof(PicWork) * n_pics);
  hipGraph_t graph = nullptr;
  HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int r = issue_intra_launches(c, plans, plan_stride, n_pics, d_work, d_jobs, enc, use_level, groups, c->stream);
  hipError_t ce = hipStreamEndCapture(c->stream, &graph);
  if (r != HMX_OK || ce != hipSuccess) {
    if (graph) hipGraphDestroy(graph);
    hipFree(d_work);
    return r != HMX_OK ? r : fail(c, HMX_ERR_DEVICE, "hipStreamEndCapture", ce);
  }
  hipGraphExec_t exec = nullptr;
  ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (ce != hipSuccess) {
    hipFree(d_work);
    return fail(c, HMX_ERR_DEVICE, "hipGraphInstantiate", ce);
  }
  if (c->graphs.size() >= 6) { // evict the least recently used entry
    size_t v = 0;
    for (size_t i = 1; i < c->graphs.size(); i++)
      if (c->graphs[i].stamp < c->graphs[v].stamp) v = i;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipGraphExecDestroy(c->graphs[v].exec);
    hipFree(c->graphs[v].d_work);
    c->graphs.erase(c->graphs.begin() + v);
  }
  c->graphs.push_back(hmx_ctx::GraphEntry{key, n_pics, exec, d_work, ++c->graph_clock});
  HIPCHK(c, hipGraphLaunch(exec, c->stream));
  return HMX_OK;
}

extern "C" int hmx_frame_intra_encode(hmx_ctx *c, const hmx_intra_plan *pl, int n_pics, const hmx_pic *org,
                                      const hmx_pic *rec, const hmx_levels *lev) {
  return frame_intra(c, &pl, 0, n_pics, org, rec, lev, true);
}
extern "C" int hmx_frame_intra_decode(hmx_ctx *c, const hmx_intra_plan *pl, int n_pics, const hmx_pic *rec,
                                      const hmx_levels *lev) {
  return frame_intra(c, &pl, 0, n_pics, nullptr, rec, lev, false);
}
extern "C" int hmx_frame_intra_decode_onto(hmx_ctx *c, const hmx_intra_plan *pl, int n_pics, const hmx_pic *rec,
                                           const hmx_levels *lev) {
  if (!c) return HMX_ERR_ARG;
  if (c->knob.graph) return fail(c, HMX_ERR_ARG, "hmx_frame_intra_decode_onto: not available with HMX_GRAPH");
  c->onto_call = true;
  const int r = frame_intra(c, &pl, 0, n_pics, nullptr, rec, lev, false);
  c->onto_call = false;
  return r;
}
extern "C" int hmx_frame_intra_encode_onto(hmx_ctx *c, const hmx_intra_plan *pl, int n_pics, const hmx_pic *org, const hmx_pic *rec,
                                           const hmx_levels *lev) {
  if (!c) return HMX_ERR_ARG;
  if (c->knob.graph) return fail(c, HMX_ERR_ARG, "hmx_frame_intra_encode_onto: not available with HMX_GRAPH");
  c->onto_call = true;
  const int r = frame_intra(c, &pl, 0, n_pics, org, rec, lev, true);
  c->onto_call = false;
  return r;
}
extern "C" int hmx_frame_intra_encode_multi(hmx_ctx *c, const hmx_intra_plan *const *plans, int n_pics, const hmx_pic *org,
                                            const hmx_pic *rec, const hmx_levels *lev) {
  return frame_intra(c, plans, 1, n_pics, org, rec, lev, true);
}
extern "C" int hmx_frame_intra_decode_multi(hmx_ctx *c, const hmx_intra_plan *const *plans, int n_pics, const hmx_pic *rec,
                                            const hmx_levels *lev) {
  return frame_intra(c, plans, 1, n_pics, nullptr, rec, lev, false);
}
extern "C" int hmx_set_rdoq(hmx_ctx *c, const hmx_rdoq_pic *pics, int n_pics) {
  if (!c || (pics && n_pics <= 0)) return HMX_ERR_ARG;
  auto &q = c->crq;
  // every input is checked BEFORE the context's state moves: a rejected call leaves the previous setting as it was
  if (pics)
    for (int i = 0; i < n_pics; i++)
      if (!(pics[i].lambda_luma > 0) || !(pics[i].lambda_chroma > 0)) return fail(c, HMX_ERR_ARG, "hmx_set_rdoq: lambda must be positive");
  q.serial++;
  if (!pics) {
    q.n = 0;
    return HMX_OK;
  }
  static_assert(sizeof(hmx_rdoq_pic) == 8 * sizeof(EstBitsDev) + 2 * sizeof(double), "hmx_rdoq_pic: eight tables and two multipliers");
  q.n = 0; // from here on a failure (device memory, copy) leaves RDOQ OFF, never a half-written table set
  HIPCHK(c, hipStreamSynchronize(c->stream)); // a queued call may still read the previous tables
  int r = grow_dev(c, (void **)&q.d_est, &q.cap_est, sizeof(EstBitsDev) * 8 * (size_t)n_pics);
  if (!r) r = grow_dev(c, (void **)&q.d_lambda, &q.cap_lambda, sizeof(double) * 4 * (size_t)n_pics);
  if (r) return r;
  q.lambda.resize((size_t)n_pics * 2);
  for (int i = 0; i < n_pics; i++) {
    q.lambda[(size_t)i * 2] = pics[i].lambda_luma, q.lambda[(size_t)i * 2 + 1] = pics[i].lambda_chroma;
    HIPCHK(c, hipMemcpyAsync(q.d_est + (size_t)i * 8, pics[i].est, sizeof(EstBitsDev) * 8, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  q.n = n_pics;
  return HMX_OK;
}
extern "C" int hmx_set_sse_output(hmx_ctx *c, const hmx_sse *sse, int n_pics) {
  if (!c || (sse && n_pics <= 0)) return HMX_ERR_ARG;
  c->sse_out.clear();
  if (sse) c->sse_out.assign(sse, sse + n_pics);
  return HMX_OK;
}
extern "C" int hmx_frame_intra_encode_resident(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                               const hmx_tpool *org, hmx_tpool *rec, const hmx_levels *lev) {
  if (!org || !rec || (plan_stride != 0 && plan_stride != 1)) return fail(c, HMX_ERR_ARG, "hmx_frame_intra_encode_resident: bad argument");
  return frame_intra(c, plans, plan_stride, n_pics, nullptr, nullptr, lev, true, org, rec);
}
extern "C" int hmx_frame_intra_decode_resident(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                               hmx_tpool *rec, const hmx_levels *lev) {
  if (!rec || (plan_stride != 0 && plan_stride != 1)) return fail(c, HMX_ERR_ARG, "hmx_frame_intra_decode_resident: bad argument");
  return frame_intra(c, plans, plan_stride, n_pics, nullptr, nullptr, lev, false, nullptr, rec);
}

// =============================================================================================
// Scalar drop-ins: host pointers, one block, same kernels (batch of one)
// =============================================================================================
namespace {
struct Scratch { // carve the context's device scratch
  hmx_ctx *c;
  size_t off = 0;
  template <typename T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = reinterpret_cast<T *>(c->d_scratch + off);
    off += n * sizeof(T);
    return p;
  }
};

int up2d(hmx_ctx *c, void *dst, const void *src, size_t elem, int w, int h, size_t src_stride_elems) {
  HIPCHK(c, hipMemcpy2DAsync(dst, w * elem, src, src_stride_elems * elem, w * elem, h, hipMemcpyHostToDevice, c->stream));
  return HMX_OK;
}
int down2d(hmx_ctx *c, void *dst, size_t dst_stride_elems, const void *src, size_t elem, int w, int h) {
  HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride_elems * elem, src, w * elem, w * elem, h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HMX_OK;
}

struct One { // a one-block launch: dense N x N buffers at plane origin
  ListArgs A{};
  DTu *d_tu;
};
int one_block(hmx_ctx *c, Scratch &s, One &o, int n, int plane, unsigned mode, unsigned flags, const PicDev &P) {
  DTu h{};
  h.t.x = h.t.y = 0;
  h.t.log2n = (uint8_t)ilog2i(n);
  h.t.plane = (uint8_t)plane;
  h.t.mode = (uint8_t)(mode > 255 ? 255 : mode);
  h.t.flags = (uint8_t)flags;
  h.idx = 0;
  o.d_tu = s.take<DTu>(1);
  HIPCHK(c, hipMemcpyAsync(o.d_tu, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream)); // h is a stack object
  o.A.tus = o.d_tu;
  o.A.n = 1;
  o.A.P = P;
  return HMX_OK;
}
bool size_ok(int w, int h) { return w == h && (w == 4 || w == 8 || w == 16 || w == 32); }
// intra prediction also runs at 64 x 64, the prediction unit of a 64 x 64 coding unit (TComPrediction.cpp:343-345 asserts 4..128)
bool size_ok_intra(int w, int h) { return size_ok(w, h) || (w == 64 && h == 64); }

PicDev scalar_picdev(hmx_ctx *c, const hmx_qp *qp, int per_base, int slice_type, int sign_hide) {
  PicDev P{};
  P.pic_w = P.pic_h = 1 << 14;
  P.ctu = c->cfg.ctu_size;
  P.bit_depth = c->cfg.bit_depth;
  P.sign_hide = sign_hide;
  hmx_qp q = qp ? *qp : hmx_qp{0, 0, 0, 15};
  P.qd[0] = P.qd[1] = make_qd(q, per_base, slice_type);
  return P;
}
} // namespace

// uiMode -> flags: the list kernels derive DST/scan from (plane, INTER flag, mode)
static unsigned mode_flags(unsigned mode) { return mode == HMX_REG_DCT ? HMX_TU_INTER : 0; }

extern "C" int hmx_xT(hmx_ctx *c, unsigned mode, const hmx_pel *resi, unsigned stride, int32_t *coef, int w, int h) {
  if (!c || !resi || !coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xT: unsupported size or null");
  Scratch s{c};
  short *d_in = s.take<short>(w * h);
  int *d_out = s.take<int>(w * h);
  int r = up2d(c, d_in, resi, 2, w, h, stride);
  if (r) return r;
  One o;
  if ((r = one_block(c, s, o, w, 0, mode, mode_flags(mode), scalar_picdev(c, nullptr, -1, HMX_I_SLICE, 0)))) return r;
  o.A.a.p[0] = d_in;
  o.A.a.s[0] = w;
  o.A.lev.p[0] = d_out;
  o.A.lev.s[0] = w;
  if ((r = launch_op(c, OP_XT, ilog2i(w), o.A))) return r;
  return down2d(c, coef, w, d_out, 4, w, h);
}

extern "C" int hmx_xIT(hmx_ctx *c, unsigned mode, const int32_t *coef, hmx_pel *resi, unsigned stride, int w, int h) {
  if (!c || !resi || !coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xIT: unsupported size or null");
  Scratch s{c};
  int *d_in = s.take<int>(w * h);
  short *d_out = s.take<short>(w * h);
  int r = up2d(c, d_in, coef, 4, w, h, w);
  if (r) return r;
  One o;
  if ((r = one_block(c, s, o, w, 0, mode, mode_flags(mode), scalar_picdev(c, nullptr, -1, HMX_I_SLICE, 0)))) return r;
  o.A.lev.p[0] = d_in;
  o.A.lev.s[0] = w;
  o.A.b.p[0] = d_out;
  o.A.b.s[0] = w;
  if ((r = launch_op(c, OP_XIT, ilog2i(w), o.A))) return r;
  return down2d(c, resi, stride, d_out, 2, w, h);
}

extern "C" int hmx_xTransformSkip(hmx_ctx *c, const hmx_pel *resi, unsigned stride, int32_t *coef, int w, int h) {
  if (!c || !resi || !coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xTransformSkip: unsupported size or null");
  Scratch s{c};
  short *d_in = s.take<short>(w * h);
  int *d_out = s.take<int>(w * h);
  int r = up2d(c, d_in, resi, 2, w, h, stride);
  if (r) return r;
  One o;
  if ((r = one_block(c, s, o, w, 0, 0, HMX_TU_TRANSFORM_SKIP, scalar_picdev(c, nullptr, -1, HMX_I_SLICE, 0)))) return r;
  o.A.a.p[0] = d_in;
  o.A.a.s[0] = w;
  o.A.lev.p[0] = d_out;
  o.A.lev.s[0] = w;
  if ((r = launch_op(c, OP_XT, ilog2i(w), o.A))) return r;
  return down2d(c, coef, w, d_out, 4, w, h);
}

extern "C" int hmx_xITransformSkip(hmx_ctx *c, const int32_t *coef, hmx_pel *resi, unsigned stride, int w, int h) {
  if (!c || !resi || !coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xITransformSkip: unsupported size or null");
  Scratch s{c};
  int *d_in = s.take<int>(w * h);
  short *d_out = s.take<short>(w * h);
  int r = up2d(c, d_in, coef, 4, w, h, w);
  if (r) return r;
  One o;
  if ((r = one_block(c, s, o, w, 0, 0, HMX_TU_TRANSFORM_SKIP, scalar_picdev(c, nullptr, -1, HMX_I_SLICE, 0)))) return r;
  o.A.lev.p[0] = d_in;
  o.A.lev.s[0] = w;
  o.A.b.p[0] = d_out;
  o.A.b.s[0] = w;
  if ((r = launch_op(c, OP_XIT, ilog2i(w), o.A))) return r;
  return down2d(c, resi, stride, d_out, 2, w, h);
}

static int plane_of(int text_type) { return text_type == HMX_TEXT_LUMA ? 0 : (text_type == HMX_TEXT_CHROMA_V ? 2 : 1); }

extern "C" int hmx_xQuant(hmx_ctx *c, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *ac_sum, int text_type,
                          const hmx_quant_param *qp) {
  if (!c || !src || !dst || !qp || !ac_sum || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xQuant: unsupported size or null");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_out = s.take<int>(w * h);
  uint32_t *d_sum = s.take<uint32_t>(1);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (r) return r;
  One o;
  unsigned flags = qp->is_intra ? 0 : HMX_TU_INTER;
  if ((r = one_block(c, s, o, w, plane_of(text_type), qp->dir_mode, flags,
                     scalar_picdev(c, &qp->qp, qp->per_base, qp->slice_type, qp->sign_hide))))
    return r;
  o.A.lev.p[plane_of(text_type)] = d_in;
  o.A.lev.s[plane_of(text_type)] = w;
  o.A.lev2.p[plane_of(text_type)] = d_out;
  o.A.lev2.s[plane_of(text_type)] = w;
  o.A.abs_sum = d_sum;
  if ((r = launch_op(c, OP_XQUANT, ilog2i(w), o.A))) return r;
  uint32_t hs = 0;
  HIPCHK(c, hipMemcpyAsync(&hs, d_sum, 4, hipMemcpyDeviceToHost, c->stream));
  r = down2d(c, dst, w, d_out, 4, w, h);
  *ac_sum += hs; // uiAcSum is accumulated by reference (:1256)
  return r;
}

// ---- rate-distortion optimised quantisation (hmx_rdoq.h) ----
static_assert(sizeof(hmx_est_bits) == sizeof(EstBitsDev), "hmx_est_bits mirrors estBitsSbacStruct");
static const int kRdoqChunk = 16384; // lanes per launch: 41 KB of records each

static int rdoq_scan_index(int n, bool luma, bool intra, int mode) { // getCoefScanIdx (TComDataCU.cpp:4014): 0 diag, 1 hor, 2 ver
  if (!intra) return 0;
  const bool multi = luma ? (n == 4 || n == 8) : (n == 4);
  if (!multi) return 0;
  if (abs(mode - 26) < 5) return 1;
  if (abs(mode - 10) < 5) return 2;
  return 0;
}

// the per-call constants; the two quotients are formed here, in the reference's operation order
static void rdoq_constants(RdoqArgs &A, int B, const hmx_qp qp[2], const double lambda[2]) {
#pragma clang fp contract(off)
  const int inc = B - 8;
  for (int t = 0; t < 2; t++) {
    A.per[t] = qp[t].per;
    A.rem[t] = qp[t].rem;
    A.q[t] = kQuantScales[qp[t].rem];
    A.lambda[t] = lambda[t];
    for (int lg = 2; lg <= 5; lg++) { // setErrScaleCoeff, TComTrQuant.cpp:2794-2818 (flat quantiser coefficients)
      const int tshift = 15 - B - lg;
      double e = (double)(1 << 15);
      e = e * ldexp(1.0, -2 * tshift);
      e = e / (double)A.q[t] / (double)A.q[t] / (double)(1 << (2 * inc));
      A.err_scale[t][lg - 2] = e;
    }
    const int iq = kInvQuantScales[qp[t].rem];
    A.rd_factor[t] = (long long)((double)iq * (double)iq * (double)(1 << (2 * qp[t].per)) / lambda[t] / 16 / (double)(1 << (2 * inc)) + 0.5); // :2205
  }
  A.bit_depth = B;
}

static int rdoq_issue(hmx_ctx *c, RdoqArgs A);
static uint64_t hash_words(uint64_t h, const void *p, size_t bytes) {
  const uint64_t *w = static_cast<const uint64_t *>(p);
  for (size_t i = 0; i < bytes / 8; i++) h = (h ^ w[i]) * 0x9e3779b97f4a7c15ull, h ^= h >> 29;
  const unsigned char *t = static_cast<const unsigned char *>(p) + (bytes & ~(size_t)7);
  for (size_t i = 0; i < (bytes & 7); i++) h = (h ^ t[i]) * 0x100000001b3ull;
  return h;
}
static int rdoq_launch(hmx_ctx *c, RdoqArgs A, const std::vector<RdoqBlock> &blocks, const hmx_est_bits *est, int n_est) {
  if (n_est > c->rdoq_est_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(c->rdoq_est);
    c->rdoq_est = nullptr;
    c->rdoq_est_cap = 0;
    if (hipMalloc((void **)&c->rdoq_est, sizeof(EstBitsDev) * n_est) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc bit-estimate tables");
    c->rdoq_est_cap = n_est;
    c->rdoq_resident = false;
  }
  if (!c->rdoq_wd) {
    const size_t T = kRdoqChunk;
    if (hipMalloc((void **)&c->rdoq_wd, sizeof(double) * (3 * 1024 + 64) * T) != hipSuccess ||
        hipMalloc((void **)&c->rdoq_wi, sizeof(int) * 4 * 1024 * T) != hipSuccess)
      return fail(c, HMX_ERR_NOMEM, "hipMalloc RDOQ workspace");
    c->rdoq_T = (int)T;
  }
  if ((int)blocks.size() > c->rdoq_blocks_cap) { // the whole list goes up once; the launches below follow without a synchronisation
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(c->rdoq_blocks);
    c->rdoq_blocks = nullptr;
    c->rdoq_blocks_cap = 0;
    const size_t cap = blocks.size() + blocks.size() / 8 + 1024;
    if (hipMalloc((void **)&c->rdoq_blocks, sizeof(RdoqBlock) * cap) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc RDOQ block list");
    c->rdoq_blocks_cap = (int)cap;
    c->rdoq_resident = false;
  }
  // a pipeline quantises the same block structure picture after picture: when the list and the tables are the ones already
  // resident (64-bit hash over both), nothing is uploaded and nothing synchronises
  uint64_t key = hash_words(0x243f6a8885a308d3ull ^ blocks.size(), blocks.data(), sizeof(RdoqBlock) * blocks.size());
  key = hash_words(key ^ (uint64_t)n_est, est, sizeof(EstBitsDev) * n_est);
  if (!c->rdoq_resident || c->rdoq_key != key) {
    HIPCHK(c, hipMemcpyAsync(c->rdoq_est, est, sizeof(EstBitsDev) * n_est, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->rdoq_blocks, blocks.data(), sizeof(RdoqBlock) * blocks.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // pageable sources
    c->rdoq_key = key;
    c->rdoq_resident = true;
    c->rdoq_in_key = 0; // whoever built this list names its arguments afterwards
  }
  for (int k = 0; k < 4; k++) c->rdoq_class_n[k] = 0; // blocks per size, the list is sorted largest first
  for (const RdoqBlock &b : blocks) c->rdoq_class_n[5 - b.log2n]++;
  return rdoq_issue(c, A);
}
// the launches over the block list resident on the device
static int rdoq_issue(hmx_ctx *c, RdoqArgs A) {
  A.est = c->rdoq_est;
  A.wd = c->rdoq_wd;
  A.wi = c->rdoq_wi;
  A.T = c->rdoq_T;
  // blocks arrive sorted by size, largest first: 8x8 and larger go through the wave-cooperative routine (the decomposition
  // of hmx_rdoq_core.h), 4x4 blocks -- a single coefficient group, nothing to decompose -- one LANE per block (k_rdoq)
  const size_t n_wave = c->rdoq_class_n[0] + c->rdoq_class_n[1] + c->rdoq_class_n[2], n_all = n_wave + c->rdoq_class_n[3];
  const bool lane_only = c->knob.rdoq_lane_only;
  if (n_wave && !lane_only) {
    // 8x8 and larger: the wave-cooperative routine of the whole-picture chain (rdoq_wave_tiles), a wave per 8 / 4 / 1 blocks
    if (!c->rdoq_consts && hipMalloc((void **)&c->rdoq_consts, 4 * sizeof(double)) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc RDOQ constants");
    double up[4];
    up[0] = A.lambda[0], up[1] = A.lambda[1];
    memcpy(&up[2], A.rd_factor, 2 * sizeof(long long));
    if (!c->rdoq_consts_valid || memcmp(up, c->rdoq_consts_h, sizeof(up))) { // a pipeline calls with the same multipliers
      HIPCHK(c, hipStreamSynchronize(c->stream)); // an earlier call may still read them
      memcpy(c->rdoq_consts_h, up, sizeof(up));
      HIPCHK(c, hipMemcpyAsync(c->rdoq_consts, c->rdoq_consts_h, sizeof(up), hipMemcpyHostToDevice, c->stream));
      c->rdoq_consts_valid = true;
    }
    RdoqChain RC{};
    RC.est = nullptr, RC.lambda = c->rdoq_consts, RC.rd_factor = reinterpret_cast<const long long *>(c->rdoq_consts + 2);
    RC.pic_mul = 0, RC.n_pics = 1;
    memcpy(RC.err_scale, A.err_scale, sizeof(RC.err_scale));
    PicDev P{};
    P.bit_depth = A.bit_depth, P.sign_hide = A.sign_hide;
    for (int t = 0; t < 2; t++) P.qd[t].q = A.q[t], P.qd[t].per_qbits = A.per[t];
    // the size classes are independent, and a picture's worth of one class does not fill the chip (a launch lasts about as
    // long as one block): they run side by side, 16x16 and 8x8 on side streams that fork from and join the caller's
    if (c->n_side < 2) {
      if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      for (int g = c->n_side; g < 2; g++) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->side[g], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming));
      }
      c->n_side = 2;
    }
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    size_t o = 0;
    bool used[2] = {false, false};
    for (int lg = 5; lg >= 3; lg--) {
      const size_t e = o + c->rdoq_class_n[5 - lg];
      if (e > o) {
        A.blocks = c->rdoq_blocks + o;
        A.n = (int)(e - o);
        hipStream_t st = c->stream;
        if (lg < 5) {
          st = c->side[4 - lg];
          used[4 - lg] = true;
          HIPCHK(c, hipStreamWaitEvent(st, c->ev_fork, 0));
        }
        if (lg == 5) hipLaunchKernelGGL((k_rdoq_tiles<32, 1>), dim3((unsigned)A.n), dim3(64), 0, st, A, RC, P);
        else if (lg == 4) hipLaunchKernelGGL((k_rdoq_tiles<16, 4>), dim3((unsigned)((A.n + 3) / 4)), dim3(64), 0, st, A, RC, P);
        else hipLaunchKernelGGL((k_rdoq_tiles<8, 8>), dim3((unsigned)((A.n + 7) / 8)), dim3(64), 0, st, A, RC, P);
        HIPCHK(c, hipGetLastError());
      }
      o = e;
    }
    for (int g = 0; g < 2; g++) // the 4x4 launches that follow on the caller's stream overlap too; the call ends joined
      if (used[g]) {
        HIPCHK(c, hipEventRecord(c->ev_join[g], c->side[g]));
      }
    c->rdoq_join[0] = used[0], c->rdoq_join[1] = used[1];
  }
  // 4x4 blocks -- a single coefficient group, nothing to decompose -- one LANE per block (k_rdoq); HMX_RDOQ_LANE: every block
  for (size_t o = lane_only ? 0 : n_wave; o < n_all;) { // chunks share the workspace: launches of one stream run one after the other
    const size_t n = std::min(n_all - o, (size_t)kRdoqChunk);
    A.blocks = c->rdoq_blocks + o;
    A.n = (int)n;
    hipLaunchKernelGGL(k_rdoq, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, A);
    HIPCHK(c, hipGetLastError());
    o += n;
  }
  for (int g = 0; g < 2; g++)
    if (c->rdoq_join[g]) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[g], 0));
      c->rdoq_join[g] = false;
    }
  return HMX_OK;
}

extern "C" int hmx_xRateDistOptQuant(hmx_ctx *c, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *abs_sum, int text_type,
                                     const hmx_rdoq_param *rp, const hmx_est_bits *est) {
  if (!c || !src || !dst || !rp || !est || !abs_sum || !size_ok(w, h) || !(rp->lambda > 0))
    return fail(c, HMX_ERR_ARG, "hmx_xRateDistOptQuant: unsupported size, null or non-positive lambda");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_out = s.take<int>(w * h);
  uint32_t *d_sum = s.take<uint32_t>(1);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (r) return r;
  RdoqArgs A{};
  const hmx_qp qps[2] = {rp->qp, rp->qp};
  const double lam[2] = {rp->lambda, rp->lambda};
  rdoq_constants(A, c->cfg.bit_depth, qps, lam);
  A.sign_hide = rp->sign_hide;
  const bool luma = text_type == HMX_TEXT_LUMA;
  std::vector<RdoqBlock> b(1);
  b[0] = RdoqBlock{d_in, d_out, w, w, d_sum, (unsigned char)ilog2i(w), (unsigned char)luma,
                   (unsigned char)rdoq_scan_index(w, luma, rp->is_intra != 0, rp->dir_mode), (unsigned char)(rp->root_cbf != 0),
                   (unsigned char)rp->cbf_ctx, 0, 0};
  if ((r = rdoq_launch(c, A, b, est, 1))) return r;
  uint32_t hs = 0;
  HIPCHK(c, hipMemcpyAsync(&hs, d_sum, 4, hipMemcpyDeviceToHost, c->stream));
  r = down2d(c, dst, w, d_out, 4, w, h);
  *abs_sum += hs; // uiAbsSum accumulates (:2187)
  return r;
}

extern "C" int hmx_batch_xRateDistOptQuant(hmx_ctx *c, const hmx_tu *tus, const hmx_rdoq_side *side, int n, const hmx_levels *coef,
                                           const hmx_levels *lev, uint32_t *d_abs_sum, const hmx_pic_param *pp,
                                           const hmx_est_bits *est, int n_est, double lambda_luma, double lambda_chroma) {
  if (!c || !tus || !side || n <= 0 || !coef || !lev || !pp || !est || n_est <= 0 || !(lambda_luma > 0) || !(lambda_chroma > 0))
    return fail(c, HMX_ERR_ARG, "hmx_batch_xRateDistOptQuant: bad argument");
  RdoqArgs A{};
  const int bd = 6 * (c->cfg.bit_depth - 8);
  const hmx_qp qps[2] = {hmx_setQPforQuant(pp->qp, HMX_TEXT_LUMA, bd, 0), hmx_setQPforQuant(pp->qp, HMX_TEXT_CHROMA, bd, pp->chroma_qp_offset)};
  const double lam[2] = {lambda_luma, lambda_chroma};
  rdoq_constants(A, c->cfg.bit_depth, qps, lam);
  A.sign_hide = pp->sign_hide;
  // a pipeline quantises the same block structure picture after picture: when the arguments are the ones that produced the
  // list resident on the device (64-bit hash), the list is neither rebuilt nor sorted nor uploaded
  uint64_t in_key = hash_words(0x13198a2e03707344ull ^ (uint64_t)n, tus, sizeof(hmx_tu) * (size_t)n);
  in_key = hash_words(in_key, side, sizeof(hmx_rdoq_side) * (size_t)n);
  in_key = hash_words(in_key, coef, sizeof(*coef));
  in_key = hash_words(in_key, lev, sizeof(*lev));
  in_key = hash_words(in_key ^ (uint64_t)(uintptr_t)d_abs_sum ^ (uint64_t)n_est, est, sizeof(hmx_est_bits) * (size_t)n_est);
  if (c->rdoq_resident && c->rdoq_in_key == in_key && in_key) {
    // the multipliers and the QP may differ from call to call: they travel with the launch, not with the list
    const int r = rdoq_issue(c, A);
    return r;
  }
  std::vector<RdoqBlock> b(n);
  for (int i = 0; i < n; i++) {
    const hmx_tu &t = tus[i];
    if (t.plane > 2 || t.log2n < 2 || t.log2n > 5 || side[i].est_idx >= n_est || side[i].cbf_ctx >= 15)
      return fail(c, HMX_ERR_ARG, "hmx_batch_xRateDistOptQuant: bad block");
    const int p = t.plane, N = 1 << t.log2n;
    const bool luma = p == 0, intra = !(t.flags & HMX_TU_INTER);
    b[i] = RdoqBlock{coef->plane[p] + (size_t)t.y * coef->stride[p] + t.x,
                     lev->plane[p] + (size_t)t.y * lev->stride[p] + t.x,
                     coef->stride[p], lev->stride[p], d_abs_sum ? d_abs_sum + i : nullptr, t.log2n, (unsigned char)luma,
                     (unsigned char)rdoq_scan_index(N, luma, intra, t.mode), side[i].root_cbf, side[i].cbf_ctx,
                     (unsigned char)(luma ? 0 : 1), side[i].est_idx};
  }
  // a block is one lane and its cost grows with its size: lanes of a wave should hold blocks of one size,
  // the long ones first
  std::stable_sort(b.begin(), b.end(), [](const RdoqBlock &x, const RdoqBlock &y) { return x.log2n > y.log2n; });
  const int r = rdoq_launch(c, A, b, est, n_est);
  c->rdoq_in_key = r ? 0 : in_key;
  return r;
}

extern "C" int hmx_xDeQuant(hmx_ctx *c, const hmx_coeff *src, int32_t *dst, int w, int h, const hmx_qp *qp) {
  if (!c || !src || !dst || !qp || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xDeQuant: unsupported size or null");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_out = s.take<int>(w * h);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (r) return r;
  One o;
  if ((r = one_block(c, s, o, w, 0, 0, 0, scalar_picdev(c, qp, -1, HMX_I_SLICE, 0)))) return r;
  o.A.lev.p[0] = d_in;
  o.A.lev.s[0] = w;
  o.A.lev2.p[0] = d_out;
  o.A.lev2.s[0] = w;
  if ((r = launch_op(c, OP_XDEQUANT, ilog2i(w), o.A))) return r;
  return down2d(c, dst, w, d_out, 4, w, h);
}

extern "C" int hmx_transformNxN(hmx_ctx *c, const hmx_pel *resi, unsigned stride, hmx_coeff *level, unsigned w, unsigned h,
                                uint32_t *abs_sum, int text_type, const hmx_quant_param *qp, int use_ts, int bypass) {
  if (!c || !resi || !level || !qp || !abs_sum || !size_ok((int)w, (int)h))
    return fail(c, HMX_ERR_ARG, "hmx_transformNxN: unsupported size or null");
  if (bypass) { // TComTrQuant.cpp:1388-1399: a copy, no arithmetic to offload
    *abs_sum = 0;
    for (unsigned k = 0; k < h; k++)
      for (unsigned j = 0; j < w; j++) {
        level[k * w + j] = resi[k * stride + j];
        *abs_sum += (uint32_t)abs((int)resi[k * stride + j]);
      }
    return HMX_OK;
  }
  Scratch s{c};
  const int pl = plane_of(text_type);
  short *d_in = s.take<short>(w * h);
  int *d_out = s.take<int>(w * h);
  uint32_t *d_sum = s.take<uint32_t>(1);
  int r = up2d(c, d_in, resi, 2, (int)w, (int)h, stride);
  if (r) return r;
  One o;
  unsigned flags = (qp->is_intra ? 0 : HMX_TU_INTER) | (use_ts ? HMX_TU_TRANSFORM_SKIP : 0);
  if ((r = one_block(c, s, o, (int)w, pl, qp->dir_mode, flags,
                     scalar_picdev(c, &qp->qp, qp->per_base, qp->slice_type, qp->sign_hide))))
    return r;
  o.A.a.p[pl] = d_in;
  o.A.a.s[pl] = (int)w;
  o.A.lev.p[pl] = d_out;
  o.A.lev.s[pl] = (int)w;
  o.A.abs_sum = d_sum;
  if ((r = launch_op(c, OP_TRANSFORM_NXN, ilog2i((int)w), o.A))) return r;
  HIPCHK(c, hipMemcpyAsync(abs_sum, d_sum, 4, hipMemcpyDeviceToHost, c->stream));
  return down2d(c, level, w, d_out, 4, (int)w, (int)h);
}

extern "C" int hmx_invtransformNxN(hmx_ctx *c, int bypass, int text_type, unsigned mode, hmx_pel *resi, unsigned stride,
                                   const hmx_coeff *level, unsigned w, unsigned h, const hmx_qp *qp, int use_ts) {
  if (!c || !resi || !level || !qp || !size_ok((int)w, (int)h))
    return fail(c, HMX_ERR_ARG, "hmx_invtransformNxN: unsupported size or null");
  if (bypass) { // :1430-1440
    for (unsigned k = 0; k < h; k++)
      for (unsigned j = 0; j < w; j++) resi[k * stride + j] = (hmx_pel)level[k * w + j];
    return HMX_OK;
  }
  Scratch s{c};
  const int pl = plane_of(text_type);
  int *d_in = s.take<int>(w * h);
  short *d_out = s.take<short>(w * h);
  int r = up2d(c, d_in, level, 4, (int)w, (int)h, w);
  if (r) return r;
  One o;
  // the caller passes uiMode explicitly here (REG_DCT for chroma / inter), like the reference
  unsigned flags = (mode == HMX_REG_DCT ? HMX_TU_INTER : 0) | (use_ts ? HMX_TU_TRANSFORM_SKIP : 0);
  // DST is selected by (luma plane && !INTER); a chroma call with a luma mode must still be DCT
  if ((r = one_block(c, s, o, (int)w, mode == HMX_REG_DCT ? pl : 0, mode, flags, scalar_picdev(c, qp, -1, HMX_I_SLICE, 0))))
    return r;
  const int kp = mode == HMX_REG_DCT ? pl : 0;
  o.A.lev.p[kp] = d_in;
  o.A.lev.s[kp] = (int)w;
  o.A.b.p[kp] = d_out;
  o.A.b.s[kp] = (int)w;
  if ((r = launch_op(c, OP_INVTRANSFORM_NXN, ilog2i((int)w), o.A))) return r;
  return down2d(c, resi, stride, d_out, 2, (int)w, (int)h);
}

// =============================================================================================
// Intra scalar drop-ins: initAdiPattern, predIntraLumaAng / predIntraChromaAng
// =============================================================================================
template <int N>
__global__ __launch_bounds__(64) void k_adi(const short *win, int stride, int bx, int by, int x, int y, int chroma,
                                            PicDev P, int *adi) {
  __shared__ TuLds<N> L;
  const int gl = threadIdx.x;
  const bool on = gl < N;
  constexpr int W = 2 * N + 1;
  if (on) {
    unsigned long long avail;
    if constexpr (N == 64) avail = intra_avail_mask_ctu(x, y, P); // a whole CTU, luma: units of eight samples
    else avail = intra_avail_mask(x << chroma, y << chroma, N << chroma, P);
    const short *rec0 = win + (size_t)by * stride + bx;
    build_ref_line<N, N>([&](int dx, int dy) { return (int)rec0[(ptrdiff_t)dy * stride + dx]; }, avail, N == 64 ? 3 : chroma ? 1 : 2,
                         P.bit_depth, gl, L.line);
  }
  __syncthreads();
  if (on && !chroma) smooth_ref_line<N, N>(L.line, L.fline, gl);
  __syncthreads();
  // reference layout: row 0 = corner + 2N above, column 0 = 2N left; second buffer = smoothed (luma)
  for (int i = threadIdx.x; i < 2 * W * W; i += blockDim.x) adi[i] = 0;
  __syncthreads();
  if (on) {
    for (int p = gl; p <= 4 * N; p += N) {
      int cell = p >= 2 * N ? p - 2 * N : (2 * N - p) * W;
      adi[cell] = L.line[p];
      if (!chroma) adi[W * W + cell] = L.fline[p];
    }
  }
}

// predIntraGetPredValDC (TComPrediction.cpp:129-167) on a border buffer; fill >= 0: write it to n*n samples
__global__ void k_dcval(const int *adi, int n, int above, int left, int *out, short *fill) {
  const int W = 2 * n + 1;
  const int *src = adi + W + 1;
  int sum = 0;
  if (above)
    for (int i = 0; i < n; i++) sum += src[i - W];
  if (left)
    for (int i = 0; i < n; i++) sum += src[i * W - 1];
  int dc;
  if (above && left)
    dc = (sum + n) / (2 * n);
  else if (above || left)
    dc = (sum + n / 2) / n;
  else
    dc = src[-1];
  dc = (short)dc;
  if (out) *out = dc;
  if (fill)
    for (int i = 0; i < n * n; i++) fill[i] = (short)dc;
}

template <int N>
__global__ __launch_bounds__(64) void k_pred_adi(const int *adi, int mode, int luma, PicDev P, short *pred, int raw_line = 0) {
  __shared__ TuLds<N> L;
  const int gl = threadIdx.x;
  constexpr int W = 2 * N + 1;
  if (gl < N) {
    for (int p = gl; p <= 4 * N; p += N) {
      int cell = p >= 2 * N ? p - 2 * N : (2 * N - p) * W;
      L.line[p] = adi[cell];
      L.fline[p] = luma ? adi[W * W + cell] : 0;
    }
  }
  __syncthreads();
  if (gl < N) {
    int row[N];
    intra_pred_block<N>(L, gl, mode, luma != 0, P, row, raw_line != 0);
    store_row16<N>(pred + gl * N, row);
  }
}

extern "C" int hmx_initAdiPattern(hmx_ctx *c, const hmx_pel *rec, int stride, int x, int y, int n, int is_chroma,
                                  int pic_w, int pic_h, int32_t *adi) {
  if (!c || !rec || !adi || !size_ok_intra(n, n)) return fail(c, HMX_ERR_ARG, "hmx_initAdiPattern: unsupported size or null");
  if (n == 64 && (is_chroma || c->cfg.ctu_size != 64 || x % 64 || y % 64))
    return fail(c, HMX_ERR_ARG, "hmx_initAdiPattern: a 64x64 block is the luma prediction unit of a whole CTU (CTU size 64, aligned)");
  const int pw = is_chroma ? pic_w / 2 : pic_w, ph = is_chroma ? pic_h / 2 : pic_h;
  if (x < 0 || y < 0 || x + n > pw || y + n > ph) return fail(c, HMX_ERR_ARG, "hmx_initAdiPattern: block outside picture");
  const int x0 = std::max(x - 1, 0), y0 = std::max(y - 1, 0), x1 = std::min(x + 2 * n, pw), y1 = std::min(y + 2 * n, ph);
  const int ww = x1 - x0, wh = y1 - y0, W = 2 * n + 1;
  Scratch s{c};
  short *d_win = s.take<short>((size_t)ww * wh);
  int *d_adi = s.take<int>((size_t)2 * W * W);
  int r = up2d(c, d_win, rec + (size_t)y0 * stride + x0, 2, ww, wh, stride);
  if (r) return r;
  hmx_pic_param pp{pic_w, pic_h, 0, 0, HMX_I_SLICE, 0};
  PicDev P = make_picdev(c, &pp);
  const int bx = x - x0, by = y - y0;
  switch (n) {
  case 4: hipLaunchKernelGGL(k_adi<4>, dim3(1), dim3(64), 0, c->stream, d_win, ww, bx, by, x, y, is_chroma, P, d_adi); break;
  case 8: hipLaunchKernelGGL(k_adi<8>, dim3(1), dim3(64), 0, c->stream, d_win, ww, bx, by, x, y, is_chroma, P, d_adi); break;
  case 16: hipLaunchKernelGGL(k_adi<16>, dim3(1), dim3(64), 0, c->stream, d_win, ww, bx, by, x, y, is_chroma, P, d_adi); break;
  case 32: hipLaunchKernelGGL(k_adi<32>, dim3(1), dim3(64), 0, c->stream, d_win, ww, bx, by, x, y, is_chroma, P, d_adi); break;
  default: hipLaunchKernelGGL(k_adi<64>, dim3(1), dim3(64), 0, c->stream, d_win, ww, bx, by, x, y, is_chroma, P, d_adi); break;
  }
  HIPCHK(c, hipGetLastError());
  return hmx_download(c, adi, d_adi, sizeof(int) * 2 * W * W);
}

static int pred_from_adi(hmx_ctx *c, const int32_t *adi, unsigned mode, hmx_pel *pred, unsigned stride, int w, int h,
                         int luma, int raw_line = 0) {
  if (!c || !adi || !pred || !size_ok_intra(w, h) || mode > 34) return fail(c, HMX_ERR_ARG, "predIntra: unsupported size/mode or null");
  const int W = 2 * w + 1;
  Scratch s{c};
  int *d_adi = s.take<int>((size_t)2 * W * W);
  short *d_pred = s.take<short>((size_t)w * h);
  int r = hmx_upload(c, d_adi, adi, sizeof(int) * (luma ? 2 : 1) * W * W);
  if (r) return r;
  hmx_pic_param pp{1 << 14, 1 << 14, 0, 0, HMX_I_SLICE, 0};
  PicDev P = make_picdev(c, &pp);
  switch (w) {
  case 4: hipLaunchKernelGGL(k_pred_adi<4>, dim3(1), dim3(64), 0, c->stream, d_adi, (int)mode, luma, P, d_pred, raw_line); break;
  case 8: hipLaunchKernelGGL(k_pred_adi<8>, dim3(1), dim3(64), 0, c->stream, d_adi, (int)mode, luma, P, d_pred, raw_line); break;
  case 16: hipLaunchKernelGGL(k_pred_adi<16>, dim3(1), dim3(64), 0, c->stream, d_adi, (int)mode, luma, P, d_pred, raw_line); break;
  case 32: hipLaunchKernelGGL(k_pred_adi<32>, dim3(1), dim3(64), 0, c->stream, d_adi, (int)mode, luma, P, d_pred, raw_line); break;
  default: hipLaunchKernelGGL(k_pred_adi<64>, dim3(1), dim3(64), 0, c->stream, d_adi, (int)mode, luma, P, d_pred, raw_line); break;
  }
  HIPCHK(c, hipGetLastError());
  return down2d(c, pred, stride, d_pred, 2, w, h);
}
extern "C" int hmx_predIntraLumaAng(hmx_ctx *c, const int32_t *adi, unsigned mode, hmx_pel *pred, unsigned stride, int w,
                                    int h) {
  return pred_from_adi(c, adi, mode, pred, stride, w, h, 1);
}
extern "C" int hmx_predIntraChromaAng(hmx_ctx *c, const int32_t *adi, unsigned mode, hmx_pel *pred, unsigned stride, int w,
                                      int h) {
  return pred_from_adi(c, adi, mode, pred, stride, w, h, 0);
}

// The protected building blocks of the two wrappers above, named by the north star.  `adi` is ONE
// (2w+1) x (2w+1) border buffer (the caller chose raw or smoothed, as the reference's callers do by
// passing a pointer); the reference's pSrc is its cell (1,1).
extern "C" int hmx_predIntraGetPredValDC(hmx_ctx *c, const int32_t *adi, int w, int h, int above, int left, hmx_pel *dc) {
  if (!c || !adi || !dc || !size_ok_intra(w, h)) return fail(c, HMX_ERR_ARG, "hmx_predIntraGetPredValDC: unsupported size or null");
  const int W = 2 * w + 1;
  Scratch s{c};
  int *d_adi = s.take<int>((size_t)W * W), *d_out = s.take<int>(1);
  int r = hmx_upload(c, d_adi, adi, sizeof(int) * W * W);
  if (r) return r;
  hipLaunchKernelGGL(k_dcval, dim3(1), dim3(1), 0, c->stream, d_adi, w, above, left, d_out, (short *)nullptr);
  HIPCHK(c, hipGetLastError());
  int v = 0;
  r = hmx_download(c, &v, d_out, sizeof(int));
  *dc = (hmx_pel)v;
  return r;
}
extern "C" int hmx_xPredIntraPlanar(hmx_ctx *c, const int32_t *adi, hmx_pel *pred, unsigned stride, int w, int h) {
  return pred_from_adi(c, adi, 0, pred, stride, w, h, 0); // planar has no luma-only step: the chroma path on the given buffer
}
extern "C" int hmx_xPredIntraAng(hmx_ctx *c, const int32_t *adi, hmx_pel *pred, unsigned stride, int w, int h, unsigned dir_mode,
                                 int above, int left, int filter) {
  if (!c || !adi || !pred || !size_ok_intra(w, h) || dir_mode < 1 || dir_mode > 34)
    return fail(c, HMX_ERR_ARG, "hmx_xPredIntraAng: unsupported size, null or mode outside 1..34");
  if (dir_mode == 1) { // DC from the sides flagged available; no edge smoothing here (xDCPredFiltering is the wrapper's)
    const int W = 2 * w + 1;
    Scratch s{c};
    int *d_adi = s.take<int>((size_t)W * W);
    short *d_pred = s.take<short>((size_t)w * h);
    int r = hmx_upload(c, d_adi, adi, sizeof(int) * W * W);
    if (r) return r;
    hipLaunchKernelGGL(k_dcval, dim3(1), dim3(1), 0, c->stream, d_adi, w, above, left, (int *)nullptr, d_pred);
    HIPCHK(c, hipGetLastError());
    return down2d(c, pred, stride, d_pred, 2, w, h);
  }
  if (!filter) return pred_from_adi(c, adi, dir_mode, pred, stride, w, h, 0);
  // bFilter: the luma edge filter of the pure vertical / horizontal modes, on the buffer as given.  The luma
  // kernel expects the smoothed copy behind the raw one; it is told not to select it.
  const int W = 2 * w + 1;
  std::vector<int32_t> two((size_t)2 * W * W);
  memcpy(two.data(), adi, sizeof(int32_t) * W * W);
  memcpy(two.data() + (size_t)W * W, adi, sizeof(int32_t) * W * W);
  return pred_from_adi(c, two.data(), dir_mode, pred, stride, w, h, 1, 1);
}

// ---- distortion drop-ins (TComRdCost.cpp): calcHAD :404-450, getDistPart(DF_SSE) -> xGetSSE* :1313-1657 ----
// one thread per 8x8 / 4x4 sub-block (HAD) or per row (SSE); partial sums by atomicAdd
__global__ void k_dist(const short *org, int so, const short *cur, int sc, int w, int h, int inc, int hads, unsigned *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (hads) {
    const int n = (w % 8 == 0 && h % 8 == 0) ? 8 : 4, bw = w / n, nb = bw * (h / n);
    if (i >= nb) return;
    const short *o = org + (size_t)(i / bw) * n * so + (i % bw) * n, *c = cur + (size_t)(i / bw) * n * sc + (i % bw) * n;
    int d[64];
    for (int r = 0; r < n; r++)
      for (int k = 0; k < n; k++) d[r * 8 + k] = o[r * so + k] - c[r * sc + k];
    int sum = 0;
    if (n == 8) {
      for (int r = 0; r < 8; r++) wht_regs<8>(d + r * 8);
      for (int k = 0; k < 8; k++) {
        int col[8];
        for (int r = 0; r < 8; r++) col[r] = d[r * 8 + k];
        wht_regs<8>(col);
        for (int r = 0; r < 8; r++) sum += abs(col[r]);
      }
      sum = (sum + 2) >> 2;
    } else {
      for (int r = 0; r < 4; r++) wht_regs<4>(d + r * 8);
      for (int k = 0; k < 4; k++) {
        int col[4];
        for (int r = 0; r < 4; r++) col[r] = d[r * 8 + k];
        wht_regs<4>(col);
        for (int r = 0; r < 4; r++) sum += abs(col[r]);
      }
      sum = (sum + 1) >> 1;
    }
    atomicAdd(out, (unsigned)sum);
  } else {
    if (i >= h) return;
    unsigned sum = 0;
    for (int k = 0; k < w; k++) {
      const int t = org[(size_t)i * so + k] - cur[(size_t)i * sc + k];
      sum += (unsigned)((t * t) >> (2 * inc));
    }
    atomicAdd(out, sum);
  }
}
static int dist_scalar(hmx_ctx *c, const hmx_pel *org, int so, const hmx_pel *cur, int sc, int w, int h, int hads, uint32_t *out) {
  if (!c || !org || !cur || !out || w <= 0 || h <= 0 || w > 64 || h > 64 || (hads && ((w | h) & 3)))
    return fail(c, HMX_ERR_ARG, "distortion: unsupported size or null");
  Scratch s{c};
  short *d_o = s.take<short>((size_t)w * h), *d_c = s.take<short>((size_t)w * h);
  unsigned *d_out = s.take<unsigned>(1);
  int r = up2d(c, d_o, org, 2, w, h, so);
  if (!r) r = up2d(c, d_c, cur, 2, w, h, sc);
  if (r) return r;
  HIPCHK(c, hipMemsetAsync(d_out, 0, 4, c->stream));
  const int items = hads ? (w / 4) * (h / 4) : h;
  hipLaunchKernelGGL(k_dist, dim3((items + 63) / 64), dim3(64), 0, c->stream, d_o, w, d_c, w, w, h, c->cfg.bit_depth - 8, hads, d_out);
  HIPCHK(c, hipGetLastError());
  unsigned v = 0;
  r = hmx_download(c, &v, d_out, 4);
  *out = hads ? v >> (c->cfg.bit_depth - 8) : v; // calcHAD returns uiSum >> g_uiBitIncrement (:449)
  return r;
}
extern "C" int hmx_calcHAD(hmx_ctx *c, const hmx_pel *pi0, int stride0, const hmx_pel *pi1, int stride1, int w, int h, uint32_t *satd) {
  return dist_scalar(c, pi0, stride0, pi1, stride1, w, h, 1, satd);
}
extern "C" int hmx_getSSE(hmx_ctx *c, const hmx_pel *cur, int cur_stride, const hmx_pel *org, int org_stride, int w, int h, uint32_t *sse) {
  return dist_scalar(c, org, org_stride, cur, cur_stride, w, h, 0, sse);
}

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

// ---- deblocking filter, application part (TLibCommon/TComLoopFilter.cpp:571-922) ----
// One launch per direction over the whole picture (loopFilterPic :153-201 filters every vertical edge of the
// picture before the first horizontal one).  Work item = one 4x4 luma unit whose left (top) side is an edge of
// the 8x8 grid with a non-zero strength: the thread filters the unit's four luma lines and, on the chroma grid
// with strength 2, two lines of Cb and Cr.  Edges are 8 samples apart and a filter reads 4 and writes 3 samples
// per side, so the work items of one launch touch disjoint samples.
__constant__ unsigned char kDbkTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                         2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ unsigned char kDbkBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                                           16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ unsigned char kChromaScale[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
                                               20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35,
                                               36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};
struct DbkArgs {
  PlanesDev rec;
  const unsigned char *bs; // of this direction
  const signed char *qp;
  const unsigned char *no_filter;
  int uw, uh, dir, B, boff, toff;
};
__global__ __launch_bounds__(256) void k_deblock(DbkArgs A) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= A.uw * A.uh) return;
  const int ux = u % A.uw, uy = u / A.uw, dir = A.dir;
  const int b = A.bs[u];
  if (!b || ((dir ? uy : ux) & 1)) return;
  const int up = dir ? u - A.uw : u - 1;
  const bool pn = A.no_filter && A.no_filter[up], qn = A.no_filter && A.no_filter[u];
  const int q_avg = ((int)A.qp[up] + (int)A.qp[u] + 1) >> 1;
  const int B = A.B, scale = 1 << (B - 8), maxv = (1 << B) - 1;
  {
    const int tc = kDbkTc[clip3(0, 53, q_avg + 2 * (b - 1) + (A.toff << 1))] * scale;
    const int beta = kDbkBeta[clip3(0, 51, q_avg + (A.boff << 1))] * scale;
    const int side = (beta + (beta >> 1)) >> 3, cut = tc * 10;
    const int st = A.rec.s[0], across = dir ? st : 1, along = dir ? 1 : st;
    short *s = A.rec.p[0] + (size_t)(4 * uy) * st + 4 * ux;
    int m[4][8];
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
      for (int k = 0; k < 8; k++) m[l][k] = s[(ptrdiff_t)l * along + (ptrdiff_t)(k - 4) * across];
    const int dp0 = abs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = abs(m[0][4] - 2 * m[0][5] + m[0][6]);
    const int dp3 = abs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = abs(m[3][4] - 2 * m[3][5] + m[3][6]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
    if (d < beta) {
      const bool fp = dp < side, fq = dq < side;
      const bool s0 = (abs(m[0][0] - m[0][3]) + abs(m[0][7] - m[0][4]) < (beta >> 3)) && (2 * d0 < (beta >> 2)) &&
                      (abs(m[0][3] - m[0][4]) < ((tc * 5 + 1) >> 1));
      const bool s3 = (abs(m[3][0] - m[3][3]) + abs(m[3][7] - m[3][4]) < (beta >> 3)) && (2 * d3 < (beta >> 2)) &&
                      (abs(m[3][3] - m[3][4]) < ((tc * 5 + 1) >> 1));
      const bool strong = s0 && s3;
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const int m0 = m[l][0], m1 = m[l][1], m2 = m[l][2], m3 = m[l][3], m4 = m[l][4], m5 = m[l][5], m6 = m[l][6], m7 = m[l][7];
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
        short *pl = s + (ptrdiff_t)l * along;
        if (!pn) pl[-across] = (short)n3, pl[-2 * across] = (short)n2, pl[-3 * across] = (short)n1;
        if (!qn) pl[0] = (short)n4, pl[across] = (short)n5, pl[2 * across] = (short)n6;
      }
    }
  }
  if (b > 1 && !((dir ? uy : ux) & 3)) { // chroma: its own 8x8 grid, strength 2 only (:709-712, :740)
    const int qc = kChromaScale[clip3(0, 51, q_avg)];
    const int tc = kDbkTc[clip3(0, 53, qc + 2 * (b - 1) + (A.toff << 1))] * scale;
#pragma unroll
    for (int p = 1; p < 3; p++) {
      const int st = A.rec.s[p], across = dir ? st : 1, along = dir ? 1 : st;
      short *c0 = A.rec.p[p] + (size_t)(2 * uy) * st + 2 * ux;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        short *s = c0 + (ptrdiff_t)k * along;
        const int m2 = s[-2 * across], m3 = s[-across], m4 = s[0], m5 = s[across];
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        if (!pn) s[-across] = (short)clip3(0, maxv, m3 + delta);
        if (!qn) s[0] = (short)clip3(0, maxv, m4 - delta);
      }
    }
  }
}
// boundary strengths (xGetBoundaryStrengthSingle :444-569): one thread per 4x4 unit, both directions
__device__ __forceinline__ bool dbk_mv_far(const short *a, const short *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }
__device__ __forceinline__ int dbk_strength(const hmx_dbk_unit &P, const hmx_dbk_unit &Pm, const hmx_dbk_unit &Q, bool tu_edge, bool is_b) {
  if (P.intra || Q.intra) return 2;
  if (tu_edge && (Q.cbf || P.cbf)) return 1;
  if (!is_b) return (Pm.ref[0] != Q.ref[0]) || dbk_mv_far(Pm.mv[0], Q.mv[0]);
  const int p0 = Pm.ref[0] < 0 ? -1 : Pm.ref[0], p1 = Pm.ref[1] < 0 ? -1 : Pm.ref[1];
  const int q0 = Q.ref[0] < 0 ? -1 : Q.ref[0], q1 = Q.ref[1] < 0 ? -1 : Q.ref[1];
  if (!((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0))) return 1;
  if (p0 != p1) {
    if (p0 == q0) return dbk_mv_far(Pm.mv[0], Q.mv[0]) || dbk_mv_far(Pm.mv[1], Q.mv[1]);
    return dbk_mv_far(Pm.mv[0], Q.mv[1]) || dbk_mv_far(Pm.mv[1], Q.mv[0]);
  }
  return (dbk_mv_far(Pm.mv[0], Q.mv[1]) || dbk_mv_far(Pm.mv[1], Q.mv[0])) && (dbk_mv_far(Pm.mv[0], Q.mv[0]) || dbk_mv_far(Pm.mv[1], Q.mv[1]));
}
__global__ __launch_bounds__(256) void k_dbk_strengths(const hmx_dbk_unit *units, const unsigned char *edge_ver, const unsigned char *edge_hor,
                                                       int uw, int uh, int ctu, int is_b, unsigned char *bs_ver, unsigned char *bs_hor) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= uw * uh) return;
  const int ux = u % uw, uy = u / uw;
  const hmx_dbk_unit Q = units[u];
  int bv = 0, bh = 0;
  if ((edge_ver[u] & 1) && !(ux & 1) && ux) {
    const hmx_dbk_unit P = units[u - 1];
    bv = dbk_strength(P, P, Q, (edge_ver[u] >> 1) & 1, is_b);
  }
  if ((edge_hor[u] & 1) && !(uy & 1) && uy) {
    const int up = u - uw;
    int um = up;
    if ((4 * uy) % ctu == 0) um = up - ux + (ux & ~3) + ((ux & 3) < 2 ? 0 : 3); // compressed motion of the CTU row above: [0 0 3 3]
    bh = dbk_strength(units[up], units[um], Q, (edge_hor[u] >> 1) & 1, is_b);
  }
  bs_ver[u] = (unsigned char)bv;
  bs_hor[u] = (unsigned char)bh;
}
extern "C" int hmx_deblock_strengths(hmx_ctx *c, const hmx_dbk_unit *d_units, const uint8_t *d_edge_ver, const uint8_t *d_edge_hor, int pic_w,
                                     int pic_h, int is_b_slice, uint8_t *d_bs_ver, uint8_t *d_bs_hor) {
  if (!c || !d_units || !d_edge_ver || !d_edge_hor || !d_bs_ver || !d_bs_hor || pic_w <= 0 || pic_h <= 0 || (pic_w & 7) || (pic_h & 7))
    return fail(c, HMX_ERR_ARG, "hmx_deblock_strengths: null argument or picture size not a multiple of 8");
  const int uw = pic_w / 4, uh = pic_h / 4;
  hipLaunchKernelGGL(k_dbk_strengths, dim3((unsigned)(((size_t)uw * uh + 255) / 256)), dim3(256), 0, c->stream, d_units, d_edge_ver, d_edge_hor,
                     uw, uh, c->cfg.ctu_size, is_b_slice, d_bs_ver, d_bs_hor);
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

extern "C" int hmx_deblock_picture(hmx_ctx *c, const hmx_pic *rec, int pic_w, int pic_h, const uint8_t *d_bs_ver, const uint8_t *d_bs_hor,
                                   const int8_t *d_qp, const uint8_t *d_no_filter, int beta_offset_div2, int tc_offset_div2) {
  if (!c || !rec || !d_bs_ver || !d_bs_hor || !d_qp || pic_w <= 0 || pic_h <= 0 || (pic_w & 7) || (pic_h & 7))
    return fail(c, HMX_ERR_ARG, "hmx_deblock_picture: null argument or picture size not a multiple of 8");
  DbkArgs A{to_dev(rec), nullptr, d_qp, d_no_filter, pic_w / 4, pic_h / 4, 0, c->cfg.bit_depth, beta_offset_div2, tc_offset_div2};
  const unsigned blocks = (unsigned)(((size_t)A.uw * A.uh + 255) / 256);
  for (int dir = 0; dir < 2; dir++) {
    A.dir = dir;
    A.bs = dir ? d_bs_hor : d_bs_ver;
    hipLaunchKernelGGL(k_deblock, dim3(blocks), dim3(256), 0, c->stream, A);
  }
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

// ---- sample adaptive offset, application (TLibCommon/TComSampleAdaptiveOffset.cpp:781-1240) ----
// The reference filters in place, CTU by CTU, with line buffers that keep the unfiltered neighbours: the same as one
// pass from `in` to `out`, a thread per sample.
// A thread filters 8 consecutive samples of a row (a CTU is a multiple of 8 wide in both planes, so they share their
// parameters): three 16-byte loads (the row, the rows above and below) and the six samples just outside, one 16-byte store.
typedef short s8v __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_sao(PlanesDev in, PlanesDev out, int pic_w, int pic_h, int B, int ctu, const hmx_sao_lcu *prm, int n_lcu) {
  const int p = blockIdx.y, sh = p ? 1 : 0, w = pic_w >> sh, h = pic_h >> sh, cs = ctu >> sh, w8 = (w + 7) >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w8 * h) return;
  const int x0 = (i % w8) << 3, y = i / w8, cw = (pic_w + ctu - 1) / ctu;
  const hmx_sao_lcu q = prm[(size_t)p * n_lcu + (y / cs) * cw + x0 / cs];
  // the four offsets in one register, picked by shifts (an indexed copy of the struct would live in scratch)
  const unsigned offs = (unsigned)(unsigned char)q.offset[0] | (unsigned)(unsigned char)q.offset[1] << 8 | (unsigned)(unsigned char)q.offset[2] << 16 |
                        (unsigned)(unsigned char)q.offset[3] << 24;
  const short *s = in.p[p];
  const int st = in.s[p], maxv = (1 << B) - 1, up = B - min(B, 10), n = min(8, w - x0);
  short *d = out.p[p] + (size_t)y * out.s[p] + x0;
  // rows y-1, y, y+1 at x0-1 .. x0+8 (clamped addresses; out-of-picture neighbours are excluded by the tests below)
  int r[3][10];
  const bool vec = n == 8 && (((uintptr_t)(s + (size_t)y * st + x0) | (uintptr_t)(2 * st)) & 15) == 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int yy = min(max(y + j - 1, 0), h - 1);
    const short *row = s + (size_t)yy * st;
    if (vec) {
      const s8v v = *reinterpret_cast<const s8v *>(row + x0);
#pragma unroll
      for (int k = 0; k < 8; k++) r[j][k + 1] = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) r[j][k + 1] = row[min(x0 + k, w - 1)];
    }
    r[j][0] = row[max(x0 - 1, 0)];
    r[j][9] = row[min(x0 + 8, w - 1)];
  }
  int v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int x = x0 + k, c = r[1][k + 1];
    int o = c;
    if (q.type >= 0 && q.type < 4) {
      const int dx = q.type == 1 ? 0 : (q.type == 3 ? -1 : 1), dy = q.type == 0 ? 0 : 1; // b = c + d, a = c - d
      if (x - dx >= 0 && x - dx < w && y - dy >= 0 && x + dx >= 0 && x + dx < w && y + dy < h) {
        // select the neighbours from the register rows (dx, dy are uniform over the thread's samples)
        const int a = dy ? (dx == 0 ? r[0][k + 1] : (dx > 0 ? r[0][k] : r[0][k + 2])) : r[1][k];
        const int bb = dy ? (dx == 0 ? r[2][k + 1] : (dx > 0 ? r[2][k + 2] : r[2][k])) : r[1][k + 2];
        const int e = ((c > a) - (c < a)) + ((c > bb) - (c < bb)) + 2; // 0..4; m_auiEoTable {1, 2, 0, 3, 4} picks the offset
        const int slot = e == 2 ? 0 : (e < 2 ? e + 1 : e);
        if (slot) o = clip3(0, maxv, c + ((int)(signed char)(offs >> (8 * (slot - 1))) << up));
      }
    } else if (q.type == 4) {
      const int kk = ((c >> (B - 5)) - q.band) & 31;
      if (kk < 4) o = clip3(0, maxv, c + ((int)(signed char)(offs >> (8 * kk)) << up));
    }
    v[k] = o;
  }
  if (n == 8 && (((uintptr_t)d) & 15) == 0) {
    s8v ov;
#pragma unroll
    for (int k = 0; k < 8; k++) ov[k] = (short)v[k];
    *reinterpret_cast<s8v *>(d) = ov;
  } else {
    for (int k = 0; k < n; k++) d[k] = (short)v[k];
  }
}
extern "C" int hmx_sao_picture(hmx_ctx *c, const hmx_pic *in, const hmx_pic *out, int pic_w, int pic_h, const hmx_sao_lcu *d_params, int n_lcu) {
  const int ctu = c ? c->cfg.ctu_size : 64;
  if (!c || !in || !out || !d_params || pic_w <= 0 || pic_h <= 0 || (pic_w & 1) || (pic_h & 1) ||
      n_lcu != ((pic_w + ctu - 1) / ctu) * ((pic_h + ctu - 1) / ctu))
    return fail(c, HMX_ERR_ARG, "hmx_sao_picture: bad argument (n_lcu must be the CTU count of the picture)");
  for (int p = 0; p < 3; p++)
    if (in->plane[p] == out->plane[p]) return fail(c, HMX_ERR_ARG, "hmx_sao_picture: in and out must be different pictures");
  hipLaunchKernelGGL(k_sao, dim3((unsigned)(((size_t)((pic_w + 7) / 8) * pic_h + 255) / 256), 3), dim3(256), 0, c->stream, to_dev(in), to_dev(out), pic_w, pic_h,
                     c->cfg.bit_depth, ctu, d_params, n_lcu);
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

// ---- planar 4:2:0 YUV frames (TLibVideoIO/TVideoIOYuv.cpp:226-480) ----
// A frame travels as the bytes of the file (1 or 2 bytes per sample, Y then Cb then Cr): half or a quarter of
// the PCIe traffic of int16 planes; widening, bit-depth scaling and the right/bottom padding happen in HBM.
__device__ __forceinline__ short yuv_rescale(short v, int shift, int bits) { // scalePlane :62-127
  if (shift == 0) return v;
  if (shift > 0) return (short)(v << shift);
  const short r = (short)((v + (short)(1 << (-shift - 1))) >> -shift);
  return (short)min(max((int)r, 0), (1 << bits) - 1);
}
// 8 consecutive samples of a row per thread (16-byte plane accesses when aligned)
struct TiledPic { // the three planes of one resident picture
  TiledPlane T[3];
};
template <bool TILED>
__global__ __launch_bounds__(256) void k_yuv_unpack(const unsigned char *file, int wide, int shift, int bits, int w_full, int h_full,
                                                    int pad_x, int pad_y, PlanesDev D, TiledPic TP) {
  const int p = blockIdx.y, c = p ? 1 : 0;
  const int wf = w_full >> c, hf = h_full >> c, w = wf - (pad_x >> c), h = hf - (pad_y >> c), w8 = (wf + 7) >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w8 * hf) return;
  const int x0 = (i % w8) << 3, y = i / w8, sy = min(y, h - 1), n = min(8, wf - x0); // readPlane :226-275: replicate right, then down
  const size_t luma = (size_t)(w_full - pad_x) * (h_full - pad_y), chroma = (size_t)w * h;
  const size_t plane_off = (p == 0 ? 0 : luma + (p == 2 ? chroma : 0)) * (wide ? 2 : 1);
  const unsigned char *row = file + plane_off + (size_t)sy * w * (wide ? 2 : 1);
  short v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int sx = min(x0 + k, w - 1);
    const short t = wide ? (short)((row[2 * sx + 1] << 8) | row[2 * sx]) : (short)row[sx];
    v[k] = yuv_rescale(t, shift, bits);
  }
  if constexpr (TILED) { // eight samples of a row = one row of two neighbouring tiles (widths are even: n is 2, 4, 6 or 8)
    const TiledPlane T = p == 0 ? TP.T[0] : p == 1 ? TP.T[1] : TP.T[2];
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      if (k + 4 <= n) {
        s4v o = {v[k], v[k + 1], v[k + 2], v[k + 3]};
        *reinterpret_cast<s4v *>(T.p + taddr(T, x0 + k, y)) = o;
      } else {
        for (int q = k; q < n; q++) T.p[taddr(T, x0 + q, y)] = v[q];
      }
    }
    return;
  }
  short *d = D.p[p] + (size_t)y * D.s[p] + x0;
  if (n == 8 && (((uintptr_t)d) & 15) == 0) {
    s8v ov;
#pragma unroll
    for (int k = 0; k < 8; k++) ov[k] = v[k];
    *reinterpret_cast<s8v *>(d) = ov;
  } else {
    for (int k = 0; k < n; k++) d[k] = v[k];
  }
}
template <bool TILED>
__global__ __launch_bounds__(256) void k_yuv_pack(PlanesDev S, TiledPic TP, int wide, int shift, int bits, int ww, int hh, unsigned char *file) {
  const int p = blockIdx.y, c = p ? 1 : 0, w = ww >> c, h = hh >> c, w8 = (w + 7) >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w8 * h) return;
  const int x0 = (i % w8) << 3, y = i / w8, n = min(8, w - x0);
  const size_t luma = (size_t)ww * hh, chroma = (size_t)w * h;
  unsigned char *d = file + ((p == 0 ? 0 : luma + (p == 2 ? chroma : 0)) + (size_t)y * w + x0) * (wide ? 2 : 1);
  const short *s = TILED ? nullptr : S.p[p] + (size_t)y * S.s[p] + x0;
  short v[8];
  if constexpr (TILED) {
    const TiledPlane T = p == 0 ? TP.T[0] : p == 1 ? TP.T[1] : TP.T[2];
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      if (k + 4 <= n) {
        const s4v iv = *reinterpret_cast<const s4v *>(T.p + taddr(T, x0 + k, y));
        v[k] = iv[0], v[k + 1] = iv[1], v[k + 2] = iv[2], v[k + 3] = iv[3];
      } else {
        for (int q = k; q < k + 4; q++) v[q] = T.p[taddr(T, x0 + min(q, n - 1), y)];
      }
    }
  } else if (n == 8 && (((uintptr_t)s) & 15) == 0) {
    const s8v iv = *reinterpret_cast<const s8v *>(s);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = iv[k];
  } else {
    for (int k = 0; k < 8; k++) v[k] = s[min(k, n - 1)];
  }
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = yuv_rescale(v[k], shift, bits);
  if (wide) {
    if (n == 8 && (((uintptr_t)d) & 15) == 0) {
      s8v ov;
#pragma unroll
      for (int k = 0; k < 8; k++) ov[k] = v[k]; // little-endian 16-bit samples are the register layout
      *reinterpret_cast<s8v *>(d) = ov;
    } else {
      for (int k = 0; k < n; k++) d[2 * k] = (unsigned char)(v[k] & 0xff), d[2 * k + 1] = (unsigned char)((v[k] >> 8) & 0xff);
    }
  } else {
    if (n == 8 && (((uintptr_t)d) & 7) == 0) {
      unsigned long long o = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) o |= (unsigned long long)(unsigned char)v[k] << (8 * k);
      *reinterpret_cast<unsigned long long *>(d) = o;
    } else {
      for (int k = 0; k < n; k++) d[k] = (unsigned char)v[k];
    }
  }
}
extern "C" size_t hmx_yuv_frame_bytes(int w, int h, int file_bits) { return (size_t)w * h * 3 / 2 * (file_bits > 8 ? 2 : 1); }
extern "C" int hmx_yuv_unpack(hmx_ctx *c, const void *d_file, int file_bits, const hmx_pic *dst, int w_full, int h_full, int pad_x,
                              int pad_y) {
  if (!c || !d_file || !dst || file_bits < 8 || file_bits > 16 || w_full <= 0 || h_full <= 0 || (w_full & 1) || (h_full & 1) ||
      pad_x < 0 || pad_y < 0 || (pad_x & 1) || (pad_y & 1) || pad_x >= w_full || pad_y >= h_full)
    return fail(c, HMX_ERR_ARG, "hmx_yuv_unpack: bad argument");
  hipLaunchKernelGGL(k_yuv_unpack<false>, dim3((unsigned)(((size_t)((w_full + 7) / 8) * h_full + 255) / 256), 3), dim3(256), 0, c->stream,
                     static_cast<const unsigned char *>(d_file), file_bits > 8 ? 1 : 0, c->cfg.bit_depth - file_bits, c->cfg.bit_depth,
                     w_full, h_full, pad_x, pad_y, to_dev(dst), TiledPic{});
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}
extern "C" int hmx_yuv_pack(hmx_ctx *c, const hmx_pic *src, int w, int h, int crop_right, int crop_bottom, int file_bits, void *d_file) {
  if (!c || !d_file || !src || file_bits < 8 || file_bits > 16 || crop_right < 0 || crop_bottom < 0 || crop_right >= w ||
      crop_bottom >= h || ((w - crop_right) & 1) || ((h - crop_bottom) & 1))
    return fail(c, HMX_ERR_ARG, "hmx_yuv_pack: bad argument");
  const int ww = w - crop_right, hh = h - crop_bottom;
  hipLaunchKernelGGL(k_yuv_pack<false>, dim3((unsigned)(((size_t)((ww + 7) / 8) * hh + 255) / 256), 3), dim3(256), 0, c->stream, to_dev(src),
                     TiledPic{}, file_bits > 8 ? 1 : 0, file_bits - c->cfg.bit_depth, file_bits, ww, hh, static_cast<unsigned char *>(d_file));
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}
// The same straight into / out of a resident picture: the frame crosses PCIe as file bytes and is widened, scaled, padded and
// laid out for the block kernels in ONE pass over it; no plane-geometry copy exists on the device.
extern "C" int hmx_yuv_unpack_resident(hmx_ctx *c, const void *d_file, int file_bits, hmx_tpool *t, int index, int pad_x, int pad_y) {
  if (!c || !d_file || !t || index < 0 || index >= t->n_pics || file_bits < 8 || file_bits > 16 || (t->pic_w & 1) || (t->pic_h & 1) || pad_x < 0 ||
      pad_y < 0 || (pad_x & 1) || (pad_y & 1) || pad_x >= t->pic_w || pad_y >= t->pic_h)
    return fail(c, HMX_ERR_ARG, "hmx_yuv_unpack_resident: bad argument");
  TiledPic TP;
  for (int p = 0; p < 3; p++) TP.T[p] = tpool_plane(t, index, p);
  hipLaunchKernelGGL(k_yuv_unpack<true>, dim3((unsigned)(((size_t)((t->pic_w + 7) / 8) * t->pic_h + 255) / 256), 3), dim3(256), 0, c->stream,
                     static_cast<const unsigned char *>(d_file), file_bits > 8 ? 1 : 0, c->cfg.bit_depth - file_bits, c->cfg.bit_depth,
                     t->pic_w, t->pic_h, pad_x, pad_y, PlanesDev{}, TP);
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}
extern "C" int hmx_yuv_pack_resident(hmx_ctx *c, const hmx_tpool *t, int index, int crop_right, int crop_bottom, int file_bits, void *d_file) {
  if (!c || !d_file || !t || index < 0 || index >= t->n_pics || file_bits < 8 || file_bits > 16 || crop_right < 0 || crop_bottom < 0 ||
      crop_right >= t->pic_w || crop_bottom >= t->pic_h || ((t->pic_w - crop_right) & 1) || ((t->pic_h - crop_bottom) & 1))
    return fail(c, HMX_ERR_ARG, "hmx_yuv_pack_resident: bad argument");
  const int ww = t->pic_w - crop_right, hh = t->pic_h - crop_bottom;
  TiledPic TP;
  for (int p = 0; p < 3; p++) TP.T[p] = tpool_plane(t, index, p);
  hipLaunchKernelGGL(k_yuv_pack<true>, dim3((unsigned)(((size_t)((ww + 7) / 8) * hh + 255) / 256), 3), dim3(256), 0, c->stream, PlanesDev{}, TP,
                     file_bits > 8 ? 1 : 0, file_bits - c->cfg.bit_depth, file_bits, ww, hh, static_cast<unsigned char *>(d_file));
  HIPCHK(c, hipGetLastError());
  return HMX_OK;
}

extern "C" void hmx_clipMv(int *mvx, int *mvy, int cu_x, int cu_y, int pic_w, int pic_h, int ctu) {
  const int hmax = (pic_w + 8 - cu_x - 1) << 2, hmin = (-ctu - 8 - cu_x + 1) * 4; // TComDataCU.cpp:3505-3517
  const int vmax = (pic_h + 8 - cu_y - 1) << 2, vmin = (-ctu - 8 - cu_y + 1) * 4;
  *mvx = std::min(hmax, std::max(hmin, *mvx));
  *mvy = std::min(vmax, std::max(vmin, *mvy));
}

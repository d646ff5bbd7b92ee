// hmx_list.hip: block-list kernels and the hmx_batch_* entry points over lists -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#include "hmx_chain_dev.h"

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

int launch_op(hmx_ctx *c, int op, int log2n, const ListArgs &A) {
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


// hmx_scalar.hip: scalar drop-ins (host pointers, one block, the batch kernels with a batch of one), RDOQ block list -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#define HMX_RDOQ_KERNELS 1
#include "hmx_host.h"

// =============================================================================================
// Scalar drop-ins: host pointers, one block, same kernels (batch of one)
// =============================================================================================

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

static int rdoq_scan_index(int n, bool luma, bool intra, int mode);
// ---- xQuant's flat branch with a scaling list: getQuantCoeff per position (TComTrQuant.cpp:1215, 1244-1255), then signBitHidingHDQ ----
// One block, one wave: lane gl holds row gl (quant_sbh_block, the quantiser of the list kernels, with the table).
template <int N>
__global__ __launch_bounds__(64) void k_quant_scaled(const int *src, const int *qtab, int *dst, uint32_t *sum_out, PicDev P, int luma, int scan_idx) {
  __shared__ TuLds<N> L;
  const int gl = threadIdx.x;
  const bool active = gl < N;
  int row[N];
#pragma unroll
  for (int k = 0; k < N; k++) row[k] = active ? src[gl * N + k] : 0;
  const int sum = quant_sbh_block<N, N, N, true>(
      L, gl, active, row, [&](int) { return gl; }, [&](int k) { return k; }, luma != 0, scan_idx, P, qtab);
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k++) dst[gl * N + k] = level_of(L.tile[gl][k]);
    if (gl == 0) *sum_out = (uint32_t)sum;
  }
}
extern "C" int hmx_xQuant_scaled(hmx_ctx *c, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *ac_sum, int text_type,
                                 const hmx_quant_param *qp, const int32_t *quant_coef) {
  if (!c || !src || !dst || !qp || !ac_sum || !quant_coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xQuant_scaled: unsupported size or null");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_tab = s.take<int>(w * h), *d_out = s.take<int>(w * h);
  uint32_t *d_sum = s.take<uint32_t>(1);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (!r) r = up2d(c, d_tab, quant_coef, 4, w, h, w);
  if (r) return r;
  const PicDev P = scalar_picdev(c, &qp->qp, qp->per_base, qp->slice_type, qp->sign_hide);
  const bool luma = text_type == HMX_TEXT_LUMA;
  const int scan = rdoq_scan_index(w, luma, qp->is_intra != 0, qp->dir_mode); // getCoefScanIdx: 0 diagonal, 1 horizontal, 2 vertical
  switch (w) {
  case 4: hipLaunchKernelGGL(k_quant_scaled<4>, dim3(1), dim3(64), 0, c->stream, d_in, d_tab, d_out, d_sum, P, (int)luma, scan); break;
  case 8: hipLaunchKernelGGL(k_quant_scaled<8>, dim3(1), dim3(64), 0, c->stream, d_in, d_tab, d_out, d_sum, P, (int)luma, scan); break;
  case 16: hipLaunchKernelGGL(k_quant_scaled<16>, dim3(1), dim3(64), 0, c->stream, d_in, d_tab, d_out, d_sum, P, (int)luma, scan); break;
  default: hipLaunchKernelGGL(k_quant_scaled<32>, dim3(1), dim3(64), 0, c->stream, d_in, d_tab, d_out, d_sum, P, (int)luma, scan); break;
  }
  HIPCHK(c, hipGetLastError());
  uint32_t hs = 0;
  HIPCHK(c, hipMemcpyAsync(&hs, d_sum, 4, hipMemcpyDeviceToHost, c->stream));
  r = down2d(c, dst, w, d_out, 4, w, h);
  *ac_sum += hs;
  return r;
}

// ---- the pArlDes output of the quantiser (ADAPTIVE_QP_SELECTION) ----
// One thread per coefficient: the coefficient scaled like a level with ARL_C_PRECISION = 7 more fractional bits (TComTrQuant.cpp
// :1229-1249 flat branch, iQBits from the slice's base QP; :1757-1765, 1886-1891 inside xRateDistOptQuant, iQBits from m_cQP and the
// product limited first).  Independent of every quantiser decision, so it is a pass of its own beside hmx_xQuant / hmx_xRateDistOptQuant.
__global__ __launch_bounds__(256) void k_arl(const int *src, const int *qtab, int *arl, int n, int q, int qbits, int rdoq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int qbits_c = qbits - 7;
  const long long t = (long long)abs(src[i]) * (qtab ? qtab[i] : q);
  if (rdoq) {
    const long long lim = 2147483647ll - (1ll << (qbits - 1));
    const int ld = (int)(t < lim ? t : lim);
    arl[i] = (ld + (1 << (qbits_c - 1))) >> qbits_c;
  } else {
    arl[i] = (int)((t + (1ll << (qbits_c - 1))) >> qbits_c);
  }
}
extern "C" int hmx_arlCoeff(hmx_ctx *c, const int32_t *src, int32_t *arl, int w, int h, int text_type, const hmx_quant_param *qp, int rdoq_form,
                            const int32_t *quant_coef) {
  (void)text_type;
  if (!c || !src || !arl || !qp || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_arlCoeff: unsupported size or null");
  if (qp->qp.rem < 0 || qp->qp.rem > 5 || qp->qp.per < 0) return fail(c, HMX_ERR_ARG, "hmx_arlCoeff: bad QP");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_out = s.take<int>(w * h), *d_tab = quant_coef ? s.take<int>(w * h) : nullptr;
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (!r && quant_coef) r = up2d(c, d_tab, quant_coef, 4, w, h, w);
  if (r) return r;
  const int per = rdoq_form ? qp->qp.per : (qp->per_base >= 0 ? qp->per_base : qp->qp.per);
  const int qbits = 14 + per + (15 - c->cfg.bit_depth - ilog2i(w));
  hipLaunchKernelGGL(k_arl, dim3((unsigned)((w * h + 255) / 256)), dim3(256), 0, c->stream, d_in, d_tab, d_out, w * h, kQuantScales[qp->qp.rem], qbits, rdoq_form != 0);
  HIPCHK(c, hipGetLastError());
  return down2d(c, arl, w, d_out, 4, w, h);
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
  const bool lane_only = c->knob.rdoq_lane_only || A.qtab; // per-position tables: the sequential kernel reads them
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

// xRateDistOptQuant under a scaling list: getQuantCoeff and getErrScaleCoeff per position (TComTrQuant.cpp:1759-1762, 1882-1883).  The
// block goes through the sequential lane kernel (k_rdoq), which reads the two tables where it reads the flat values otherwise.
extern "C" int hmx_xRateDistOptQuant_scaled(hmx_ctx *c, const int32_t *src, hmx_coeff *dst, int w, int h, uint32_t *abs_sum, int text_type,
                                            const hmx_rdoq_param *rp, const hmx_est_bits *est, const int32_t *quant_coef, const double *err_scale) {
  if (!c || !src || !dst || !rp || !est || !abs_sum || !quant_coef || !err_scale || !size_ok(w, h) || !(rp->lambda > 0))
    return fail(c, HMX_ERR_ARG, "hmx_xRateDistOptQuant_scaled: unsupported size, null or non-positive lambda");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_out = s.take<int>(w * h), *d_q = s.take<int>(w * h);
  double *d_e = s.take<double>(w * h);
  uint32_t *d_sum = s.take<uint32_t>(1);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (!r) r = up2d(c, d_q, quant_coef, 4, w, h, w);
  if (!r) r = up2d(c, d_e, err_scale, 8, w, h, w);
  if (r) return r;
  RdoqArgs A{};
  const hmx_qp qps[2] = {rp->qp, rp->qp};
  const double lam[2] = {rp->lambda, rp->lambda};
  rdoq_constants(A, c->cfg.bit_depth, qps, lam);
  A.sign_hide = rp->sign_hide;
  A.qtab = d_q, A.estab = d_e;
  const bool luma = text_type == HMX_TEXT_LUMA;
  std::vector<RdoqBlock> b(1);
  b[0] = RdoqBlock{d_in, d_out, w, w, d_sum, (unsigned char)ilog2i(w), (unsigned char)luma,
                   (unsigned char)rdoq_scan_index(w, luma, rp->is_intra != 0, rp->dir_mode), (unsigned char)(rp->root_cbf != 0),
                   (unsigned char)rp->cbf_ctx, 0, 0};
  if ((r = rdoq_launch(c, A, b, est, 1))) return r;
  uint32_t hs = 0;
  HIPCHK(c, hipMemcpyAsync(&hs, d_sum, 4, hipMemcpyDeviceToHost, c->stream));
  r = down2d(c, dst, w, d_out, 4, w, h);
  *abs_sum += hs;
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

// xDeQuant with a scaling list (TComTrQuant.cpp:1311-1342): per position the table entry HM's setScalingListDec left for (list type,
// QP remainder, size); one thread per coefficient, 32-bit products as the reference forms them.
__global__ __launch_bounds__(256) void k_dequant_scaled(const int *src, const int *coef, int *dst, int n, int shift, int per, int limit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int v;
  if (shift > per) {
    const int l = clip3(-32768, 32767, src[i]);
    v = (int)((unsigned)l * (unsigned)coef[i] + (1u << (shift - per - 1))) >> (shift - per);
  } else {
    const int l = clip3(-limit, limit - 1, src[i]);
    v = (int)(((unsigned)l * (unsigned)coef[i]) << (per - shift));
  }
  dst[i] = clip3(-32768, 32767, v);
}
extern "C" int hmx_xDeQuant_scaled(hmx_ctx *c, const hmx_coeff *src, int32_t *dst, int w, int h, const hmx_qp *qp, const int32_t *dequant_coef) {
  if (!c || !src || !dst || !qp || !dequant_coef || !size_ok(w, h)) return fail(c, HMX_ERR_ARG, "hmx_xDeQuant_scaled: unsupported size or null");
  if (qp->per < 0) return fail(c, HMX_ERR_ARG, "hmx_xDeQuant_scaled: bad QP");
  Scratch s{c};
  int *d_in = s.take<int>(w * h), *d_tab = s.take<int>(w * h), *d_out = s.take<int>(w * h);
  int r = up2d(c, d_in, src, 4, w, h, w);
  if (!r) r = up2d(c, d_tab, dequant_coef, 4, w, h, w);
  if (r) return r;
  const int lg = ilog2i(w), shift = 20 - 14 - (15 - c->cfg.bit_depth - lg) + 4;
  const int bit_range = std::min(15, 12 + lg + c->cfg.bit_depth - qp->per);
  hipLaunchKernelGGL(k_dequant_scaled, dim3((unsigned)((w * h + 255) / 256)), dim3(256), 0, c->stream, d_in, d_tab, d_out, w * h, shift, qp->per, 1 << bit_range);
  HIPCHK(c, hipGetLastError());
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


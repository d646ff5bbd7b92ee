// hmx_loop.hip: deblocking and SAO (application), YUV file formats -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#include "hmx_host.h"

// ---- deblocking filter, application part (TLibCommon/TComLoopFilter.cpp:571-922) ----
// One launch per direction over the whole picture (loopFilterPic :153-201 filters every vertical edge of the
// picture before the first horizontal one).  Work item = one 4x4 luma unit whose left (top) side is an edge of
// the 8x8 grid with a non-zero strength: the thread filters the unit's four luma lines and, on the chroma grid
// with strength 2, two lines of Cb and Cr.  Edges are 8 samples apart and a filter reads 4 and writes 3 samples
// per side, so the work items of one launch touch disjoint samples.
static __constant__ unsigned char kDbkTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                         2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static __constant__ unsigned char kDbkBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                                           16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static __constant__ unsigned char kChromaScale[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
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


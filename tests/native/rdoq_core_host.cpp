// tests/native/rdoq_core_host.cpp -- TEST INFRASTRUCTURE.  The lane decomposition of thevc_amd/csrc/hmx_rdoq_core.h run on
// the CPU, one "lane" after the other in exactly the steps k_rdoq_wave takes on the device, against the oracle's sequential
// restatement (oracle/hmx_oracle.c hmo_xRateDistOptQuant) on random blocks: every size, scan, texture type, cbf branch,
// bit depth, with coefficient statistics from sparse to dense.  Built and run by tests/test_rdoq_core.py:
//   g++ -O2 -ffp-contract=off -I thevc_amd/csrc -I oracle tests/native/rdoq_core_host.cpp -L oracle -lhmx_oracle
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "hmx_oracle.h"
#include "hmx_rdoq_core.h"

using namespace hmx;

static int diag_xy(int W, int i) {
  int c = 0;
  for (int d = 0; d <= 2 * W - 2; d++)
    for (int x = (d < W ? 0 : d - W + 1); x <= d && x < W; x++) {
      if (c == i) return (d - x) * W + x;
      c++;
    }
  return 0;
}
static std::vector<unsigned> make_scan(int N, int sc) { // the library's ScanTab (hmx_device.h), host side
  const int G = N / 4;
  std::vector<unsigned> t(N * N);
  for (int g = 0; g < G * G; g++)
    for (int i = 0; i < 16; i++) {
      int gy, gx, y, x;
      if (sc == 1) gy = g / G, gx = g % G, y = i >> 2, x = i & 3;
      else if (sc == 2) gx = g / G, gy = g % G, x = i >> 2, y = i & 3;
      else {
        int gp = diag_xy(G, g), ip = diag_xy(4, i);
        gy = gp / G, gx = gp % G, y = ip >> 2, x = ip & 3;
      }
      t[g * 16 + i] = (unsigned)((gy * 4 + y) * N + gx * 4 + x);
    }
  return t;
}

static long g_zeroed_groups = 0, g_carried_groups = 0, g_sign_hidden = 0; // coverage of the harness
static const int kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
static const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};

// the decomposition, lane by lane
static void rdoq_block_lanes(const int *src, int *dst, int N, int B, const hmo_rdoq_cfg &cfg, const EstBitsDev &E, uint32_t *abs_sum) {
  const int lg = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5, nn = N * N, n_cg = nn / 16, inc = B - 8;
  const int tshift = 15 - B - lg;
  RdoqConst C{};
  C.lg = lg, C.scan_idx = cfg.scan_idx == 3 ? 0 : cfg.scan_idx, C.is_luma = cfg.is_luma;
  C.q = kQuantScales[cfg.rem], C.qbits = 14 + cfg.per + tshift;
  C.root_cbf = cfg.root_cbf, C.cbf_ctx = cfg.cbf_ctx, C.sign_hide = cfg.sign_hide;
  C.lambda = cfg.lambda;
  double e = (double)(1 << 15);
  e = e * ldexp(1.0, -2 * tshift);
  e = e / (double)C.q / (double)C.q / (double)(1 << (2 * inc));
  C.err_scale = e;
  const int iq = kInvQuantScales[cfg.rem];
  C.rd_factor = (long long)((double)iq * (double)iq * (double)(1 << (2 * cfg.per)) / cfg.lambda / 16 / (double)(1 << (2 * inc)) + 0.5);
  const std::vector<unsigned> scan = make_scan(N, C.scan_idx);
  auto bp_of = [&](int sp) { return scan[sp]; };
  auto gpos_of = [&](int cg) { const unsigned p0 = scan[cg * 16]; return ((p0 & (unsigned)(N - 1)) >> 2) | (((p0 >> lg) >> 2) << 8); };
  std::vector<int> ld(nn), lev(nn), ru(nn), rd(nn), sd(nn), du(nn);
  std::vector<double> cz(nn), cc(nn), cs(nn), cgs(64);
  int last_pos = -1;
  for (int sp = 0; sp < nn; sp++) { // step 0: one lane per position
    rdoq_prep(src[scan[sp]], C, ld[sp], cz[sp]);
    if (rdoq_max_level(ld[sp], C.qbits) > 0) last_pos = sp;
  }
  memset(dst, 0, sizeof(int) * nn);
  *abs_sum = 0;
  if (last_pos < 0) return;
  const int last_cg = last_pos >> 4;
  std::vector<RdoqSpec> spec((size_t)n_cg * 8);
  for (int task = 0; task < (last_cg + 1) * 8; task++) { // step 1: one lane per (group, carry, pattern)
    const int cg = task >> 3, v = task & 7;
    RdoqSpecSink sink{&spec[task]};
    spec[task].S = rdoq_walk_cg(C, E, cg, bp_of, src, N, v & 3, v >> 2, last_pos, sink);
  }
  std::vector<unsigned char> sel(n_cg);
  RdoqRun R;
  rdoq_resolve(C, E, n_cg, last_cg, gpos_of, cz.data(), spec.data(), sel.data(), cgs.data(), R); // step 2: one lane
  for (int cg = 0; cg <= last_cg; cg++) g_zeroed_groups += (R.zeroed >> cg) & 1, g_carried_groups += cg < last_cg && (sel[cg] >> 2);
  for (int cg = 0; cg < n_cg; cg++) { // step 3: one lane per group
    if (cg > last_cg) {
      for (int k = 0; k < 16; k++) lev[cg * 16 + k] = 0, cc[cg * 16 + k] = 0, cs[cg * 16 + k] = 0, ru[cg * 16 + k] = rd[cg * 16 + k] = sd[cg * 16 + k] = du[cg * 16 + k] = 0;
      continue;
    }
    RdoqFullSink sink{lev.data(), cc.data(), cs.data(), ru.data(), rd.data(), sd.data(), du.data(), cg * 16};
    rdoq_walk_cg(C, E, cg, bp_of, src, N, sel[cg] & 3, sel[cg] >> 2, last_pos, sink);
    if ((R.zeroed >> cg) & 1) rdoq_apply_zeroed_cg(cg, cz.data(), lev.data(), cc.data(), cs.data());
  }
  const int best_last_p1 = rdoq_phase_b(C, E, last_pos, last_cg, gpos_of, bp_of, R, cz.data(), lev.data(), cc.data(), cs.data(), cgs.data()); // one lane
  uint32_t sum = 0;
  for (int sp = 0; sp < nn; sp++) { // final levels: one lane per position
    int l = sp < best_last_p1 ? lev[sp] : 0;
    sum += (uint32_t)l;
    lev[sp] = src[scan[sp]] < 0 ? -l : l;
  }
  *abs_sum = sum;
  if (C.sign_hide && sum >= 2) {
    int top = -1; // the highest group that holds a level
    for (int cg = n_cg - 1; cg >= 0 && top < 0; cg--)
      for (int k = 0; k < 16; k++)
        if (lev[cg * 16 + k]) top = cg;
    for (int cg = 0; cg < n_cg; cg++) { // one lane per group
      unsigned neg = 0;
      for (int k = 0; k < 16; k++) neg |= (src[scan[cg * 16 + k]] < 0 ? 1u : 0u) << k;
      int before[16];
      memcpy(before, &lev[cg * 16], sizeof(before));
      rdoq_phase_c_cg(C, cg == top, &lev[cg * 16], neg, &ru[cg * 16], &rd[cg * 16], &sd[cg * 16], &du[cg * 16]);
      g_sign_hidden += memcmp(before, &lev[cg * 16], sizeof(before)) != 0;
    }
  }
  for (int sp = 0; sp < nn; sp++) dst[scan[sp]] = lev[sp];
}

// The second decomposition (the one the whole-picture chain runs, rdoq_wave_tiles): nothing per coefficient is stored.
// The variants of a few groups at a time (a "round" of 64 lanes) go through a small buffer and are resolved before the next
// round; the chosen variant of every group is walked AGAIN for its levels (the upper groups also leave the two costs per entry
// the last-position search needs), and sign hiding walks a group a third time with a sink that does its arithmetic.
struct LevelSink {
  int *out;
  void add(int, double) {}
  void pos(int k, unsigned, int level, double, double, int, int, int, int, double) { out[k] = level; }
};
static long g_rewalk_last = 0, g_rewalk_hide = 0, g_rounds = 0, g_round_groups = 0, g_round_tasks = 0;
static void rdoq_block_rewalk(const int *src, int *dst, int N, int B, const hmo_rdoq_cfg &cfg, const EstBitsDev &E, uint32_t *abs_sum) {
  const int lg = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5, nn = N * N, n_cg = nn / 16, inc = B - 8, G = N >> 2;
  const int tshift = 15 - B - lg;
  const int groups_per_round = N == 32 ? 8 : N == 16 ? 2 : 1; // 64 lanes = blocks per wave x groups x 8 variants
  RdoqConst C{};
  C.lg = lg, C.scan_idx = cfg.scan_idx == 3 ? 0 : cfg.scan_idx, C.is_luma = cfg.is_luma;
  C.q = kQuantScales[cfg.rem], C.qbits = 14 + cfg.per + tshift;
  C.root_cbf = cfg.root_cbf, C.cbf_ctx = cfg.cbf_ctx, C.sign_hide = cfg.sign_hide;
  C.lambda = cfg.lambda;
  double e = (double)(1 << 15);
  e = e * ldexp(1.0, -2 * tshift);
  e = e / (double)C.q / (double)C.q / (double)(1 << (2 * inc));
  C.err_scale = e;
  const int iq = kInvQuantScales[cfg.rem];
  C.rd_factor = (long long)((double)iq * (double)iq * (double)(1 << (2 * cfg.per)) / cfg.lambda / 16 / (double)(1 << (2 * inc)) + 0.5);
  const std::vector<unsigned> scan = make_scan(N, C.scan_idx);
  auto bp_of = [&](int sp) { return scan[sp]; };
  auto gpos_of = [&](int cg) { const unsigned p0 = scan[cg * 16]; return ((p0 & (unsigned)(N - 1)) >> 2) | (((p0 >> lg) >> 2) << 8); };
  auto cz_at = [&](int sp) {
    int l;
    double z;
    rdoq_prep(src[scan[sp]], C, l, z);
    return z;
  };
  memset(dst, 0, sizeof(int) * nn);
  *abs_sum = 0;
  int last_pos = -1;
  for (int sp = 0; sp < nn; sp++) {
    int l;
    double z;
    rdoq_prep(src[scan[sp]], C, l, z);
    if (rdoq_max_level(l, C.qbits) > 0) last_pos = sp;
  }
  if (last_pos < 0) return;
  const int last_cg = last_pos >> 4;
  RdoqRun R;
  rdoq_run_init(R);
  for (int sp = nn - 1; sp >= (last_cg + 1) * 16; sp--) rdoq_resolve_above(R, cz_at(sp)); // the block's resolving lane
  std::vector<unsigned char> sel(n_cg, 0xff);
  std::vector<double> cgs(n_cg, 0.0);
  int carry = 0;
  // rounds composed at run time: a group gets one lane per variant it can still take (rdoq_variant_mask), a block as many
  // groups per round as its share of the 64 lanes and of the 16 cost-of-zero rows holds
  std::vector<int> gmax(n_cg, 0), ginv(G * G, 0);
  for (int sp = 0; sp < nn; sp++) {
    int l;
    double z;
    rdoq_prep(src[scan[sp]], C, l, z);
    const int m = (int)rdoq_max_level(l, C.qbits);
    gmax[sp >> 4] = std::max(gmax[sp >> 4], std::min(m, 2));
  }
  for (int cg = 0; cg < n_cg; cg++) {
    const unsigned g = gpos_of(cg);
    ginv[(g >> 8) * G + (g & 255u)] = cg;
  }
  const int slots = N == 32 ? 64 : N == 16 ? 16 : 8, groups_max = N == 32 ? 16 : N == 16 ? 4 : 2;
  (void)groups_per_round;
  auto gmax_of = [&](int cg) { return gmax[cg]; };
  auto cg_at = [&](int gy, int gx) { return ginv[gy * G + gx]; };
  for (int top = last_cg; top >= 0;) { // a round
    struct Grp {
      int cg, base;
      unsigned mask;
    } grp[16];
    int ng = 0, used = 0;
    for (int cg = top; cg >= 0 && ng < groups_max; cg--) { // the block's resolving lane plans
      const unsigned mask = rdoq_variant_mask(C, cg, last_cg, top, gpos_of(cg), R, carry, gmax_of, cg_at);
      const int cnt = __builtin_popcount(mask);
      if (used + cnt > slots) break;
      grp[ng++] = Grp{cg, used, mask};
      used += cnt;
    }
    if (ng < 1) {
      printf("round planning: no group fits\n");
      exit(3);
    }
    g_round_groups += ng, g_round_tasks += used, g_rounds++;
    RdoqSpec buf[64];
    for (int q = 0; q < ng; q++) // one lane per (group, possible variant)
      for (int v = 0, rank = 0; v < 8; v++)
        if ((grp[q].mask >> v) & 1) {
          RdoqSpecSink sink{&buf[grp[q].base + rank]};
          auto in = [&](int, unsigned bp) { return src[bp]; };
          buf[grp[q].base + rank].S = rdoq_walk_cg_in(C, E, grp[q].cg, bp_of, in, v & 3, v >> 2, last_pos, sink);
          rank++;
        }
    for (int q = 0; q < ng; q++) { // the block's resolving lane
      const int cg = grp[q].cg;
      const unsigned mask = grp[q].mask;
      const RdoqSpec *b0 = buf + grp[q].base;
      auto spec_of = [&](int v) -> const RdoqSpec & {
        if (!((mask >> v) & 1)) {
          printf("variant %d of group %d was ruled out (mask %02x)\n", v, cg, mask);
          exit(4);
        }
        return b0[__builtin_popcount(mask & ((1u << v) - 1u))];
      };
      sel[cg] = (unsigned char)rdoq_resolve_group_fn(C, E, cg, last_cg, gpos_of(cg), spec_of, [&](int k) { return cz_at(cg * 16 + k); }, R, carry, cgs[cg]);
    }
    top = grp[ng - 1].cg - 1;
  }
  // levels as walked, one lane per group; the upper groups also leave the two costs per entry the search needs
  const int REC = N == 32 ? 32 : N == 16 ? 8 : N == 8 ? 4 : 1;
  struct Rec {
    double cs[16], cc[16];
  };
  struct RecordSink {
    int *lev;
    double *cs, *cc;
    void add(int, double) {}
    void pos(int k, unsigned, int level, double c_coded, double c_sig, int, int, int, int, double) {
      lev[k] = level;
      if (cs) cs[k] = c_sig, cc[k] = c_coded;
    }
  };
  std::vector<Rec> rec(REC);
  std::vector<int> lev(nn, 0);
  auto flagged = [&](int cg) {
    const unsigned g = gpos_of(cg), gpos = (g >> 8) * (unsigned)G + (g & 255u);
    return (R.cg_flag >> gpos) & 1;
  };
  for (int cg = 0; cg <= last_cg; cg++) {
    if ((R.zeroed >> cg) & 1) continue;
    const bool r = last_cg - cg < REC && flagged(cg);
    RecordSink sink{&lev[cg * 16], r ? rec[last_cg - cg].cs : nullptr, r ? rec[last_cg - cg].cc : nullptr};
    auto in = [&](int, unsigned bp) { return src[bp]; };
    rdoq_walk_cg_in(C, E, cg, bp_of, in, sel[cg] & 3, sel[cg] >> 2, last_pos, sink);
  }
  // last position: the resolving lane, over the records; deeper groups (rare) walked once more with the search as the sink
  RdoqLast T;
  rdoq_last_init(C, E, R, T);
  for (int cg = last_cg; cg >= 0 && !T.found; cg--) {
    rdoq_last_group(T, cgs[cg]);
    if (!flagged(cg)) continue;
    if (last_cg - cg < REC) {
      for (int k = 15; k >= 0 && !T.found; k--) {
        const int sp = cg * 16 + k;
        if (sp > last_pos) continue;
        const int lv = lev[sp];
        rdoq_last_pos(C, E, T, sp, lv ? scan[sp] : 0u, lv, rec[last_cg - cg].cc[k], rec[last_cg - cg].cs[k], lv ? cz_at(sp) : 0.0);
      }
    } else {
      RdoqLastSink sink{C, E, T, cg * 16, last_pos};
      auto in = [&](int, unsigned bp) { return src[bp]; };
      rdoq_walk_cg_in(C, E, cg, bp_of, in, sel[cg] & 3, sel[cg] >> 2, last_pos, sink);
      g_rewalk_last++;
    }
  }
  // cut at the last position, signed; one lane per group
  uint32_t sum = 0;
  int top_group = -1;
  for (int cg = 0; cg <= last_cg; cg++)
    for (int k = 0; k < 16; k++) {
      const int sp = cg * 16 + k, l = sp < T.best_last_p1 ? lev[sp] : 0;
      sum += (uint32_t)l;
      if (l) top_group = cg > top_group ? cg : top_group;
      lev[sp] = src[scan[sp]] < 0 ? -l : l;
    }
  *abs_sum = sum;
  if (C.sign_hide && sum >= 2) {
    for (int cg = 0; cg <= last_cg; cg++) { // one lane per group
      unsigned neg = 0;
      for (int k = 0; k < 16; k++) neg |= (src[scan[cg * 16 + k]] < 0 ? 1u : 0u) << k;
      const int *l16 = &lev[cg * 16];
      auto lev_of = [l16](int n) { return l16[n]; };
      bool any = false;
      for (int k = 0; k < 16; k++) any |= l16[k] != 0;
      RdoqHide H;
      if (!any || !rdoq_hide_begin(cg == top_group, lev_of, H)) continue;
      // a group with levels is a coded group that was not zeroed: sel[cg] is its variant
      RdoqHideSink<decltype(lev_of)> sink{C, H, lev_of, neg};
      auto in = [&](int, unsigned bp) { return src[bp]; };
      rdoq_walk_cg_in(C, E, cg, bp_of, in, sel[cg] & 3, sel[cg] >> 2, last_pos, sink);
      g_rewalk_hide++;
      if (H.min_pos >= 0) lev[cg * 16 + H.min_pos] += rdoq_hide_change(H, (neg >> H.min_pos) & 1u);
    }
  }
  for (int sp = 0; sp < nn; sp++) dst[scan[sp]] = lev[sp];
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  std::mt19937_64 rng(20260101);
  long checked = 0, nonzero = 0, zeroed_like = 0;
  for (int it = 0; it < rounds; it++)
    for (int lg = 2; lg <= 5; lg++) {
      const int N = 1 << lg, B = (it & 1) ? 10 : 8;
      hmo_est_bits est;
      auto pair = [&](int32_t *d) {
        const double p = 0.03 + 0.94 * (rng() % 10000) / 10000.0;
        d[0] = (int32_t)lround(-log2(1 - p) * 32768), d[1] = (int32_t)lround(-log2(p) * 32768);
      };
      for (auto &x : est.sig_cg) pair(x);
      for (auto &x : est.sig) pair(x);
      for (auto &x : est.greater1) pair(x);
      for (auto &x : est.greater2) pair(x);
      for (auto &x : est.cbf) pair(x);
      for (auto &x : est.root_cbf) pair(x);
      for (int i = 0; i < 32; i++) est.last_x[i] = (int32_t)((8000 + rng() % 52000) * (1 + i / 4)), est.last_y[i] = (int32_t)((8000 + rng() % 52000) * (1 + i / 4));
      pair(est.scan_zigzag), pair(est.scan_nonzigzag);
      hmo_rdoq_cfg cfg;
      const int qp = (int)(rng() % 52) + 6 * (B - 8);
      cfg.per = qp / 6, cfg.rem = qp % 6;
      cfg.is_luma = (int)(rng() % 3 != 0);
      cfg.is_intra = (int)(rng() & 1);
      const bool multi = cfg.is_intra && (cfg.is_luma ? (N == 4 || N == 8) : N == 4);
      cfg.scan_idx = multi ? (int)(rng() % 3) : 0;
      cfg.root_cbf = (int)(rng() % 4 == 0);
      cfg.cbf_ctx = (int)(rng() % 15);
      cfg.sign_hide = (int)(rng() % 4 != 0);
      cfg.lambda = 4.0 + (rng() % 200000) / 1000.0;
      std::vector<int32_t> src(N * N), want(N * N), got(N * N);
      const int style = (int)(rng() % 5); // sparse, decaying, dense small, dense large, mostly zero with outliers
      for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
          int amp = style == 0 ? 60 : style == 1 ? 4000 / (1 + x + y) : style == 2 ? 90 : style == 3 ? 30000 : 8;
          int v = (int)(rng() % (2 * amp + 1)) - amp;
          if (style == 0 && rng() % 8) v = 0;
          if (style == 4 && rng() % 50 == 0) v = (int)(rng() % 2001) - 1000;
          src[y * N + x] = v;
        }
      if (it % 17 == 3) std::fill(src.begin(), src.end(), 0);
      uint32_t s_want = 0, s_got = 0;
      hmo_xRateDistOptQuant(src.data(), want.data(), N, B, &cfg, &est, &s_want);
      EstBitsDev E;
      static_assert(sizeof(EstBitsDev) == sizeof(hmo_est_bits), "bit-estimate tables");
      memcpy(&E, &est, sizeof(E));
      rdoq_block_lanes(src.data(), got.data(), N, B, cfg, E, &s_got);
      if (s_want != s_got || memcmp(want.data(), got.data(), sizeof(int32_t) * N * N)) {
        printf("MISMATCH round %d N %d B %d luma %d scan %d root %d lambda %.3f style %d: abs sum %u vs %u\n", it, N, B, cfg.is_luma, cfg.scan_idx,
               cfg.root_cbf, cfg.lambda, style, s_want, s_got);
        for (int i = 0; i < N * N; i++)
          if (want[i] != got[i]) {
            printf("  first difference at (%d,%d): %d vs %d\n", i / N, i % N, want[i], got[i]);
            break;
          }
        return 1;
      }
      std::vector<int32_t> got2(N * N);
      uint32_t s_got2 = 0;
      rdoq_block_rewalk(src.data(), got2.data(), N, B, cfg, E, &s_got2);
      if (s_want != s_got2 || memcmp(want.data(), got2.data(), sizeof(int32_t) * N * N)) {
        printf("MISMATCH (re-walk decomposition) round %d N %d B %d luma %d scan %d root %d lambda %.3f style %d: abs sum %u vs %u\n", it, N, B,
               cfg.is_luma, cfg.scan_idx, cfg.root_cbf, cfg.lambda, style, s_want, s_got2);
        for (int i = 0; i < N * N; i++)
          if (want[i] != got2[i]) {
            printf("  first difference at (%d,%d): %d vs %d\n", i / N, i % N, want[i], got2[i]);
            break;
          }
        return 1;
      }
      checked++;
      nonzero += s_want > 0;
      (void)zeroed_like;
    }
  printf("rdoq_core_host: %ld blocks identical to the oracle (%ld with levels; %ld groups zeroed by the group decision, %ld entered with a carry, "
         "%ld changed by sign hiding)\n", checked, nonzero, g_zeroed_groups, g_carried_groups, g_sign_hidden);
  printf("re-walk decomposition: identical too (%ld groups beyond the search records walked again for the last position, %ld for sign hiding)\n", g_rewalk_last, g_rewalk_hide);
  printf("rounds: %ld, %.2f groups and %.1f lanes per round and block\n", g_rounds, (double)g_round_groups / g_rounds, (double)g_round_tasks / g_rounds);
  if (!g_zeroed_groups || !g_carried_groups || !g_sign_hidden || !g_rewalk_hide) {
    printf("coverage hole\n");
    return 2;
  }
  return 0;
}

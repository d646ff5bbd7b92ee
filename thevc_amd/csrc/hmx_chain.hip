// hmx_chain.hip: whole-picture all-intra chain -- schedules, resident pools, entry points -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#define HMX_CHAIN_MAIN 1
#include "hmx_chain_dev.h"

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
// after a synchronisation: did a wave of the packed schedule give up waiting (its bounded spin ran out)?
int check_packed_abort(hmx_ctx *c) {
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
// Issue the launches of one whole-picture call on `main` (and the side streams).  Also used under
// stream capture to record the call as a HIP graph.
static int issue_chain_launches(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics,
                                const PicWork *d_work, bool enc, bool use_level, int groups, hipStream_t main);
static int issue_packed(hmx_ctx *c, const hmx_intra_plan *const *plans, int plan_stride, int n_pics, bool enc, hipStream_t st);

// ---- packed schedule, host side ----
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
    G.max_levels = std::max(G.max_levels, pl->n_levels);
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
    // Inside the kernel RDOQ's levels travel as 16-bit words (hmx_rdoq.h); the reference keeps TCoeff = Int.  A level can reach
    // (32768 * q) >> qbits: at the lowest QPs of deep bit depths that exceeds 32767 for the large transforms (10 bit, per = 0, 32x32:
    // ~52000) and would wrap silently -- refuse the call instead (the scalar and block-list entries carry 32-bit levels for 4x4 and
    // the same bound otherwise: include/hmx.h).
    for (int t = 0; t < 2; t++)
      for (int lg = 2; lg <= 5; lg++) {
        if (!sz[lg - 2]) continue;
        const int qbits = 14 + p0->P.qd[t].per_qbits + (15 - p0->P.bit_depth - lg);
        if (((32768ll * p0->P.qd[t].q) >> qbits) > 32767)
          return fail(c, HMX_ERR_ARG, "frame_intra: RDOQ in the chain keeps levels in 16 bits; this QP / bit depth / block size can exceed them (raise the QP or use the flat quantiser)");
      }
  }
  // 8x8 blocks: sixteen per wave-item on four lanes each (wave_chain_8x2) where throughput counts, eight on eight lanes where a
  // wave-item's latency does (and under RDOQ, whose rounds are laid out for eight)
  G.slots8 = rdoq || G.slots4 != 64 ? 8 : c->knob.slots8 ? c->knob.slots8 : (n_pics >= 640 ? 16 : 8); // measured, 2160p mix: 512 pictures -3 %, 768 +4 %, 1024 +8 %, 2048 +11 %
  const uint64_t n_rows = (uint64_t)G.max_levels * G.n_groups;
  uint64_t waves_bound = 4 * n_rows + 4;
  for (int s = 0; s < 4; s++) waves_bound += sz[s] / pack_slots(s, G.slots4, G.slots8);
  if (items >= 0xffffffffull || waves_bound >= 0x0fffffffull || n_rows >= 0x7fffffffull / 4)
    return fail(c, HMX_ERR_ARG, "frame_intra: batch too large for one packed call (split it)");
  c->tev_prep_valid = false;
  const bool same = pk.valid && pk.key == c->table_key && pk.G.n_pics == G.n_pics && pk.G.I == G.I && pk.G.slots4 == G.slots4 && pk.G.slots8 == G.slots8 &&
                    pk.G.max_levels == G.max_levels;
  if (!same) {
    pk.valid = false;
    int r = grow_dev(c, (void **)&pk.d_pics, &pk.cap_pics, sizeof(PackPic) * n_pics);
    if (!r) r = grow_dev(c, (void **)&pk.d_descs, &pk.cap_descs, sizeof(PackDesc) * waves_bound);
    if (!r) r = grow_dev(c, (void **)&pk.d_items, &pk.cap_items, sizeof(FTu) * items);
    if (!r) r = grow_dev(c, (void **)&pk.d_rows, &pk.cap_rows, sizeof(PackRow) * n_rows);
    if (!r) r = grow_dev(c, (void **)&pk.d_done, &pk.cap_done, std::max<size_t>(sizeof(uint32_t) * kDoneStride * n_rows, sizeof(uint32_t) * 2 * kPackScanWgs)); // (also k_pack_scan's scratch)
    if (!r && !pk.d_hdr) {
      if (hipMalloc((void **)&pk.d_hdr, sizeof(PackHdr)) != hipSuccess) r = fail(c, HMX_ERR_NOMEM, "hipMalloc packed header");
      else if (hipMemsetAsync(pk.d_hdr, 0, sizeof(PackHdr), st) != hipSuccess) r = fail(c, HMX_ERR_DEVICE, "hipMemsetAsync packed header");
    }
    if (r) return r;
    std::vector<PackPic> hp(n_pics);
    for (int i = 0; i < n_pics; i++) {
      const hmx_intra_plan *pl = plans[i * plan_stride];
      for (int p = 0; p < 3; p++) hp[i].lev[p] = c->call_lev[i].plane[p], hp[i].lev_stride[p] = c->call_lev[i].stride[p];
      hp[i].n_levels = pl->n_levels;
      hp[i].ltab = pl->d_ltab;
      hp[i].ltus = pl->d_ltus;
      for (int p = 0; p < 3; p++) hp[i].sse[p] = (enc && (int)c->sse_out.size() >= n_pics) ? c->sse_out[i].plane[p] : nullptr;
    }
    HIPCHK(c, hipMemcpyAsync(pk.d_pics, hp.data(), sizeof(PackPic) * n_pics, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st)); // hp goes out of scope
    HIPCHK(c, hipMemsetAsync(pk.d_hdr, 0, offsetof(PackHdr, abort), st)); // everything but the sticky abort word (below)
    const unsigned prep_waves = (unsigned)((n_rows + (uint64_t)(64 / G.I) - 1) / (uint64_t)(64 / G.I));
    hipLaunchKernelGGL(k_pack_count, dim3(prep_waves), dim3(64), 0, st, pk.d_pics, pk.d_rows, G, (int)n_rows);
    // (the completion counters are cleared behind the prep kernels: their array doubles as the scan's scratch)
    hipLaunchKernelGGL(k_pack_scan<false>, dim3(kPackScanWgs), dim3(1024), 0, st, pk.d_rows, pk.d_hdr, G, pk.d_done);
    hipLaunchKernelGGL(k_pack_scan<true>, dim3(kPackScanWgs), dim3(1024), 0, st, pk.d_rows, pk.d_hdr, G, pk.d_done);
    hipLaunchKernelGGL(k_pack_fill, dim3((unsigned)((G.max_levels + kPackFillLevels - 1) / kPackFillLevels), (unsigned)(G.n_groups * G.I)), dim3(256), 0, st, pk.d_pics, pk.d_rows, pk.d_descs, pk.d_items, G);
    HIPCHK(c, hipGetLastError());
    if (c->timing && c->tev_prep) {
      HIPCHK(c, hipEventRecord(c->tev_prep, st));
      c->tev_prep_valid = true;
    }
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
      if (packed_rdoq_max_blocks(&nb)) return fail(c, HMX_ERR_DEVICE, "hipOccupancyMaxActiveBlocksPerMultiprocessor(k_intra_packed, RDOQ)");
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
    // The table depends on hmx_set_rdoq's multipliers and the call's quantiser parameters only: it goes up when one of them changed
    // (the first call after hmx_set_rdoq, or another QP), so that steady-state calls queue behind each other without a host wait.
    uint64_t lkey = 1469598103934665603ull ^ q.serial;
    for (int t = 0; t < 2; t++)
      for (int v : {A.P.qd[t].q, A.P.qd[t].per_qbits, A.P.qd[t].iq_scale, B, q.n}) lkey = (lkey ^ (uint64_t)(uint32_t)v) * 1099511628211ull;
    if (!q.lambda_valid || q.lambda_key != lkey) {
      HIPCHK(c, hipMemcpyAsync(q.d_lambda, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipStreamSynchronize(st)); // `up` goes out of scope
      q.lambda_key = lkey, q.lambda_valid = true;
    }
    A.rq.est = q.d_est;
    A.rq.lambda = q.d_lambda;
    A.rq.rd_factor = reinterpret_cast<const long long *>(q.d_lambda + (size_t)q.n * 2);
    A.rq.pic_mul = q.n == 1 ? 0 : 1;
    A.rq.n_pics = n_pics;
  }
  const dim3 grid((unsigned)pk.n_wg), blk(64);
  if (rdoq) {
    launch_packed_rdoq(A, A.want_sse != 0, grid.x, st); // hmx_chain_rdoq.hip
  } else if (G.slots8 == 16) {
    if (A.want_sse) hipLaunchKernelGGL((k_intra_packed<true, 64, true, false, 16>), grid, blk, 0, st, A);
    else if (enc) hipLaunchKernelGGL((k_intra_packed<true, 64, false, false, 16>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((k_intra_packed<false, 64, false, false, 16>), grid, blk, 0, st, A);
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
      for (int i = first[g]; i < first[g + 1]; i++) glevels[g] = std::max(glevels[g], (size_t)plans[i * plan_stride]->n_levels);
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
int pack_group_size(const hmx_ctx *c, int n_pics) {
  if (c && c->knob.pack_group > 0) return std::min(c->knob.pack_group, n_pics);
  if (c && c->crq.n > 0) return n_pics >= 512 ? kRdoqMaxGroup : 1; // RDOQ on (hmx_set_rdoq): the tables of a group wait in LDS
  return n_pics >= 1536 ? 4 : n_pics >= 512 ? 2 : 1;
}
// ---- pictures resident in the working layout (include/hmx.h: hmx_tpool) ----
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
    if (pl->set && !packed) { // built on the device: the packed schedule's tables only
      if (!use_level) return fail(c, HMX_ERR_ARG, "frame_intra: a plan built on the device has no tables for the wave schedule (HMX_INTRA_SCHEDULE=packed|level)");
      if (int r = plan_host_tables(c, pl)) return r; // the level schedule walks the level table on the host
      c->pipeline_conv = false;                      // (no per-CTU-row level ranges either)
    }
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
    hw[i].n_levels = pl->n_levels;
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
  std::vector<unsigned char> kb; // the bytes the key is formed from: two calls are "the same" when these are, not when a hash says so
  kb.reserve(sizeof(PicWork) * n_pics + sizeof(ConvJob) * jobs.size() + 24 * (size_t)n_pics + 64);
  auto mix = [&](const void *p, size_t n) { // FNV-1a over 8-byte words (the table of a 2048-picture call is ~1 MB, hashed on every call)
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w8;
      memcpy(&w8, b + i, 8);
      key = (key ^ w8) * 1099511628211ull;
    }
    for (; i < n; i++) key = (key ^ b[i]) * 1099511628211ull;
    kb.insert(kb.end(), b, b + n);
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
  if (c->table_valid && c->table_key == key && c->table_bytes != kb) // two different calls, one hash: never mistake one for the other
    key = key * 1099511628211ull + ++c->table_salt;
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
      c->table_bytes.swap(kb);
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
  q.lambda_valid = false;
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


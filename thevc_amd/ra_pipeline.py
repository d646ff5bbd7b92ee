"""Random-access (hierarchical-B) workload over the block hot path: frame/segment sharding with an
exchange of boundary reference pictures (BASELINE.json configs[3]).

Structure (cfg/encoder_randomaccess_main.cfg: IntraPeriod 32, GOPSize 8, DecodingRefreshType 1):
the sequence is cut into intra-period SEGMENTS; segment k = pictures POC 32k+1 .. 32k+31 plus its two
bounding I pictures POC 32k and 32(k+1).  Segments are independent once those two I pictures are
reconstructed, so segment k goes to rank k mod G; the only inter-rank traffic is the reconstruction
(three planes INCLUDING the margins) of I picture 32(k+1), sent by its owner to the owner of segment
k (ncclSend/ncclRecv through torch.distributed = RCCL over xGMI; the all-intra phase needs none).

Per picture the hot path is:
  I pictures   whole-picture all-intra chain (hmx_frame_intra_encode)
  B/P pictures motion compensation of a PU list against resident references with margins
               (hmx_batch_motionCompensation_multi), residual + transformNxN
               (hmx_batch_residual_transform_recon_multi = residual, T, Q, IQ, IT, reconstruction in one pass), border extension (hmx_pic_extend_border_multi);
               the pictures at one GOP position of all segments a rank owns share each call.
Decisions (block structure, modes, PUs, MVs) are synthetic and seeded; this module is the bench/test
harness' stand-in for TEncGOP's picture loop, not part of the product library."""
import ctypes as C

import numpy as np

from . import capi, workload

MARGIN = 80  # g_uiMaxCUWidth + 16, TComPicYuv.cpp:82


def gop_order(gop=8):
    """(offset, ref0_offset, ref1_offset or None) in coding order for one hierarchical GOP."""
    out = [(gop, 0, None)]

    def rec(lo, hi):
        if hi - lo < 2:
            return
        mid = (lo + hi) // 2
        out.append((mid, lo, hi))
        rec(lo, mid)
        rec(mid, hi)

    rec(0, gop)
    return out


def exchange_plan(n_segments, world):
    """[(i_picture_index, src_rank, dst_rank)]: I picture k+1 is reconstructed by the owner of segment
    k+1 and also needed by the owner of segment k."""
    plan = []
    for k in range(n_segments):
        src, dst = (k + 1) % world, k % world
        if src != dst:
            plan.append((k + 1, src, dst))
    return plan


def run_exchange(dist, rank, world, n_segments, tensors_of):
    """Move every boundary I picture from its owner to the owner of the previous segment.
    tensors_of(i_picture_index) -> list of tensors (planes incl. margins) on this rank."""
    ops = []
    for (ki, src, dst) in exchange_plan(n_segments, world):
        if rank == src:
            ops += [dist.P2POp(dist.isend, t, dst) for t in tensors_of(ki)]
        elif rank == dst:
            ops += [dist.P2POp(dist.irecv, t, src) for t in tensors_of(ki)]
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    return len(ops)


class LoopbackDist:
    """torch.distributed's point-to-point calls for ONE process that plays every rank in turn: P2POp / isend / irecv /
    batch_isend_irecv with the signatures run_exchange uses.  Sends and receives posted for a (source, destination) pair are matched
    in posting order and carried out as device copies on the CURRENT stream when flush() is called -- so the exchange plan, the
    landing buffers (margins included), and the ordering of the copies against the kernels on the pipeline's stream are exercised
    on a single GPU.  RAPipeline(loopback_ranks=V) uses it: not a substitute for RCCL between processes (gloo rehearsals and
    the driver's N-GPU run cover that), a test of everything around it."""

    class _Req:
        def wait(self):
            return None

    def __init__(self):
        self.rank = 0
        self.sends, self.recvs = {}, {}

    def isend(self, *a, **k):  # identity objects: P2POp compares them
        raise RuntimeError("LoopbackDist: use batch_isend_irecv")

    def irecv(self, *a, **k):
        raise RuntimeError("LoopbackDist: use batch_isend_irecv")

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def batch_isend_irecv(self, ops):
        for (op, t, peer) in ops:
            if op == self.isend:
                self.sends.setdefault((self.rank, peer), []).append(t)
            else:
                self.recvs.setdefault((peer, self.rank), []).append(t)
        return [LoopbackDist._Req() for _ in ops]

    def flush(self):
        moved = 0
        assert set(self.sends) == set(self.recvs), (sorted(self.sends), sorted(self.recvs))
        for pair, src in self.sends.items():
            dst = self.recvs[pair]
            assert len(src) == len(dst), pair
            for a, b in zip(src, dst):
                b.copy_(a, non_blocking=True)
                moved += 1
        self.sends, self.recvs = {}, {}
        return moved


class TorchPicture:
    """Three Pel planes with the reference's margins, backed by torch tensors (so that RCCL can move them)."""

    def __init__(self, torch, device, w, h, m=MARGIN, zero=False):
        """zero=False: the planes are NOT cleared (every sample of a picture is written before it is read: originals by upload(),
        predictions by motion compensation, reconstructions by the chains and the border extension) -- at 2160p a fill per plane of
        several hundred pictures was a visible share of a short bench run."""
        self.w, self.h, self.m = w, h, m
        self.dims = [(w, h, m), (w // 2, h // 2, m // 2), (w // 2, h // 2, m // 2)]
        make = torch.zeros if zero else torch.empty
        self.t = [make((ph + 2 * pm, pw + 2 * pm), dtype=torch.int16, device=device) for (pw, ph, pm) in self.dims]

    def as_pic(self):
        s = capi.Pic()
        for p, (pw, ph, pm) in enumerate(self.dims):
            st = pw + 2 * pm
            s.plane[p] = self.t[p].data_ptr() + 2 * (pm * st + pm)
            s.stride[p] = st
        return s

    def upload(self, torch, planes):
        for p, (pw, ph, pm) in enumerate(self.dims):
            self.t[p][pm:pm + ph, pm:pm + pw] = torch.from_numpy(np.ascontiguousarray(planes[p])).to(self.t[p].device)

    def download(self, with_margins=False):
        out = []
        for p, (pw, ph, pm) in enumerate(self.dims):
            a = self.t[p].cpu().numpy()
            out.append(a if with_margins else a[pm:pm + ph, pm:pm + pw].copy())
        return out


class RAWorkload:
    """Synthetic decisions shared by the GPU pipeline and the oracle check."""

    def __init__(self, w, h, B, qp, intra_period=32, gop=8, n_segments=1, seed=0, bi_frac=0.5, n_lists=2, n_distinct=0,
                 structure="ra"):
        """structure "ra": hierarchical-B segments between I pictures (encoder_randomaccess_main.cfg);
        "ldp": low-delay P (encoder_lowdelay_P_main.cfg) -- every segment is an independent SEQUENCE of
        intra_period pictures, one I picture then P pictures that each reference the previous picture, so
        nothing crosses segments and a rank only batches over the sequences it owns (replicas)."""
        self.structure = structure
        self.w, self.h, self.B, self.qp = w, h, B, qp
        self.n_distinct, self._cache = n_distinct, {}  # > 0: cycle that many synthetic originals (large benches)
        self.ip, self.gop, self.n_segments, self.seed = intra_period, gop, n_segments, seed
        self.intra_tus = workload.make_tus(seed + 1, w, h, "mix")
        self.inter = []
        for i in range(n_lists):
            tus = workload.make_tus(seed + 10 + i, w, h, "mix", ts_prob=0.0)
            tus["flags"] = capi.TU_INTER
            self.inter.append({"tus": tus, "pus_b": workload.make_pus(seed + 20 + i, w, h, n_refs=2, bi_frac=bi_frac),
                               "pus_p": workload.make_pus(seed + 30 + i, w, h, n_refs=1, bi_frac=0.0)})

    def original(self, poc):
        key = poc % self.n_distinct if self.n_distinct else poc
        if key not in self._cache:
            self._cache[key] = workload.make_planes(self.seed * 1000 + key, self.w, self.h, self.B, "texture")
            if not self.n_distinct:
                return self._cache.pop(key)
        return self._cache[key]

    def algorithmic_bytes(self):
        """Bytes one segment's pictures must move (SURVEY.md section 8d's accounting at the C-ABI types, Pel = 2 B, TCoeff = 4 B), by stage:
        "mc": per prediction unit and list, the (W + 7) x (H + 7) luma window and the two (W/2 + 3) x (H/2 + 3) chroma windows read, the
        prediction written once; "chain": the fused inter block chain, 10 B per sample (original, prediction, level, reconstruction);
        "border": the margin samples written by the border extension; "intra": one I picture through the intra chain (8 B per sample +
        the reference samples a block gathers).  Returns per-picture dictionaries for a B picture, a P picture (by list index) and the I picture."""
        def mc(pus):
            w, h = pus["w"].astype(np.int64), pus["h"].astype(np.int64)
            lists = 1 + (pus["ref1"] != 255).astype(np.int64)
            read = lists * 2 * ((w + 7) * (h + 7) + 2 * (w // 2 + 3) * (h // 2 + 3))
            write = 2 * (w * h + 2 * (w // 2) * (h // 2))
            return int((read + write).sum())

        def chain(tus):
            n = 1 << tus["log2n"].astype(np.int64)
            return int((10 * n * n).sum())

        W, H, m = self.w, self.h, MARGIN
        border = 2 * (((W + 2 * m) * (H + 2 * m) - W * H) + 2 * ((W // 2 + m) * (H // 2 + m) - (W // 2) * (H // 2)))
        n = 1 << self.intra_tus["log2n"].astype(np.int64)
        intra = int((8 * n * n + 2 * (4 * n + 1)).sum())
        return {"b": [{"mc": mc(d["pus_b"]), "chain": chain(d["tus"]), "border": border} for d in self.inter],
                "p": [{"mc": mc(d["pus_p"]), "chain": chain(d["tus"]), "border": border} for d in self.inter],
                "i": {"intra": intra, "border": border}}

    def segment_jobs(self, k):
        """Coding order of segment k: (poc, ref0_poc, ref1_poc|None, list_index)."""
        base = k * self.ip
        jobs = []
        if self.structure == "ldp":
            return [(base + i, base + i - 1, None, (base + i) % len(self.inter)) for i in range(1, self.ip)]
        for g in range(self.ip // self.gop):
            for (off, r0, r1) in gop_order(self.gop):
                poc = base + g * self.gop + off
                if poc == base + self.ip:
                    continue  # the next I picture: intra, reconstructed by its owner
                jobs.append((poc, base + g * self.gop + r0, None if r1 is None else base + g * self.gop + r1, poc % len(self.inter)))
        return jobs


class RAPipeline:
    """stream: the torch.cuda.Stream the context was created on (capi.Context(stream=stream.cuda_stream)).  The library's
    kernels, torch's uploads and the RCCL send/recv of the exchange must share ONE stream: the exchange sends
    reconstructions the intra chain has just written, and motion compensation reads what it received.  (A context
    created with stream handle 0 makes its own non-blocking stream, which nothing of torch's ever waits on.)"""

    def __init__(self, ctx, torch, wl, rank=0, world=1, dist=None, fused=True, stream=None, loopback_ranks=0):
        self.ctx, self.torch, self.wl, self.rank, self.world, self.dist = ctx, torch, wl, rank, world, dist
        self.fused = fused
        # loopback_ranks = V > 1 (world must be 1): this process plays V ranks.  Segment k "belongs" to rank k mod V; the closing I
        # picture of a segment owned by another rank is NOT read where it was reconstructed but from a landing buffer of its own, filled
        # by the exchange (exchange_plan / run_exchange as between processes, LoopbackDist carrying the copies out on the stream).
        self.loopback = int(loopback_ranks) if world == 1 else 0
        self.landing = {}  # (segment, poc) -> TorchPicture
        self.stream = stream if stream is not None else torch.cuda.current_stream()
        self._ev = None
        self._moved = (0, 0)
        self.L = capi.lib()
        w, h = wl.w, wl.h
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.pp_i = capi.PicParam(w, h, wl.qp, 0, capi.I_SLICE, 1)
        self.pp_b = capi.PicParam(w, h, wl.qp, 0, capi.B_SLICE, 1)
        self.plan = ctx.intra_plan(wl.intra_tus, self.pp_i)
        self.lists = [{"tu": ctx.tu_list(d["tus"]), "pus_b": ctx.to_device(d["pus_b"]), "n_b": len(d["pus_b"]),
                       "pus_p": ctx.to_device(d["pus_p"]), "n_p": len(d["pus_p"])} for d in wl.inter]
        self.my_segments = [k for k in range(wl.n_segments) if k % world == rank]
        last = wl.n_segments + (0 if wl.structure == "ldp" else 1)  # "ra": the closing I picture of the last segment
        self.my_i = sorted({k for k in range(last) if k % world == rank})
        self.rec = {}   # poc -> TorchPicture (reconstruction with margins)
        self.org = {}   # poc -> TorchPicture (original, no margins needed but same class)
        # one prediction picture and one level picture per owned segment: pictures at the same GOP position
        # of different segments are independent and go through every stage in ONE call
        self.pred = [TorchPicture(torch, self.dev, w, h, 0) for _ in self.my_segments]
        self.lev = [capi.DevPicture(ctx, w, h, dtype=np.int32) for _ in self.my_segments]
        self.lev_i = None
        self.ctx_i = self.stream_i = self.plan_i = None  # enable_overlap()
        self.rec_alt = {}

    def _pic(self, store, poc):
        if poc not in store:
            store[poc] = TorchPicture(self.torch, self.dev, self.wl.w, self.wl.h, MARGIN if store is self.rec else 0)
        return store[poc]

    def load_originals(self):
        """Synthetic originals for every picture this rank codes (resident before the timed region)."""
        with self.torch.cuda.stream(self.stream):
            return self._load_originals()

    def run(self):
        """One pass: intra pictures, exchange, inter pictures of every owned segment.  Returns pixels coded."""
        with self.torch.cuda.stream(self.stream):
            return self._run()

    def enable_overlap(self, ctx_i, stream_i, ctx_i2=None):
        """A second context on a second stream for the I pictures.  The intra chain of a handful of pictures is bound by the
        latency of its dependency levels and leaves the chip mostly idle (16 pictures of 2160p: 40 ms, 42 % of a step): in
        run_steps() the I pictures (and their exchange) of step n + 1 run BESIDE the inter pictures of step n.  Their
        reconstructions -- references of the inter pictures -- are double-buffered; call after load_originals().
        ctx_i2 (a third context, same stream): the odd steps' buffer set gets its own context, so each context sees the SAME
        picture table every time it is called and keeps its device tables (one context alternating between the two buffer sets
        re-uploads the table and rebuilds the packed schedule every step, with two host synchronisations that serialise the
        enqueueing of the two stages)."""
        assert self.loopback <= 1, "the loopback exchange keeps ONE set of landing buffers: not with the two-stage pipeline"
        self.ctx_i, self.stream_i = ctx_i, stream_i
        self.ctx_i_set = [ctx_i, ctx_i2 if ctx_i2 is not None else ctx_i]
        self.plan_i = ctx_i.intra_plan(self.wl.intra_tus, self.pp_i)
        self.plan_i_set = [self.plan_i, ctx_i2.intra_plan(self.wl.intra_tus, self.pp_i) if ctx_i2 is not None else self.plan_i]
        ip = self.wl.ip
        pocs = {k * ip for k in self.my_i}
        if self.wl.structure == "ra":
            pocs |= {(k + 1) * ip for k in self.my_segments}  # landing buffers of received I pictures
        self.rec_alt = {poc: TorchPicture(self.torch, self.dev, self.wl.w, self.wl.h, MARGIN) for poc in sorted(pocs)}
        self.rec_main = {poc: self.rec[poc] for poc in self.rec_alt}

    def run_steps(self, steps):
        """`steps` passes of run(); with enable_overlap() as a two-stage software pipeline over the steps."""
        torch = self.torch
        if self.ctx_i is None:
            return sum(self.run() for _ in range(steps))
        torch.cuda.synchronize()  # uploads, earlier passes
        store = [self.rec_main, self.rec_alt]
        ev_i = [torch.cuda.Event(), torch.cuda.Event()]
        ev_done = [torch.cuda.Event(), torch.cuda.Event()]

        def intra(n):
            with torch.cuda.stream(self.stream_i):
                if n >= 2:
                    self.stream_i.wait_event(ev_done[n % 2])  # the inter pictures of step n - 2 have read these buffers
                for poc, t in store[n % 2].items():
                    self.rec[poc] = t
                self._run_intra(self.ctx_i_set[n % 2], self.plan_i_set[n % 2], self.stream_i)
                ev_i[n % 2].record(self.stream_i)

        pixels = 0
        intra(0)
        for n in range(steps):
            if n + 1 < steps:
                intra(n + 1)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev_i[n % 2])
                for poc, t in store[n % 2].items():
                    self.rec[poc] = t
                pixels += len(self.my_i) * self.wl.w * self.wl.h + self._run_inter()
                ev_done[n % 2].record(self.stream)
        for poc, t in self.rec_main.items():
            self.rec[poc] = t
        return pixels

    def step_bytes(self):
        """Algorithmic bytes of ONE step of this rank (every owned I picture and segment), by stage."""
        ab = self.wl.algorithmic_bytes()
        out = {"intra": len(self.my_i) * ab["i"]["intra"], "mc": 0, "chain": 0, "border": len(self.my_i) * ab["i"]["border"]}
        for k in self.my_segments:
            for (_poc, _r0, r1, li) in self.wl.segment_jobs(k):
                d = ab["p" if r1 is None else "b"][li]
                for key in ("mc", "chain", "border"):
                    out[key] += d[key]
        return out

    def check(self):
        """After the caller's fence: read the packed schedule's abort word of every context this pipeline launched on (a
        dependency wait that timed out fails HERE instead of passing as a finished step)."""
        self.ctx.sync()
        for c in dict.fromkeys(getattr(self, "ctx_i_set", [])):
            c.sync()

    def exchange_stats(self):
        """What the boundary-picture exchange of the last run() moved on this rank, and how long it took on the stream."""
        w, h = self.wl.w, self.wl.h
        per_pic = 2 * ((w + 2 * MARGIN) * (h + 2 * MARGIN) + 2 * (w // 2 + MARGIN) * (h // 2 + MARGIN))
        out = {"pictures_sent": self._moved[0], "pictures_received": self._moved[1], "bytes_per_picture": per_pic,
               "bytes_per_step": per_pic * (self._moved[0] + self._moved[1]), "ms": None}
        if self._ev is not None:
            self._ev[1].synchronize()
            out["ms"] = round(self._ev[0].elapsed_time(self._ev[1]), 3)
        return out

    def free(self):
        L = self.L
        for l in self.lev + (self.lev_i or []):
            l.free()
        for d in self.lists:
            L.hmx_tu_list_destroy(self.ctx.h, d["tu"])
            d["pus_b"].free()
            d["pus_p"].free()
        L.hmx_intra_plan_destroy(self.ctx.h, self.plan)
        if self.plan_i is not None:
            L.hmx_intra_plan_destroy(self.ctx_i.h, self.plan_i)
            if self.plan_i_set[1] is not self.plan_i:
                L.hmx_intra_plan_destroy(self.ctx_i_set[1].h, self.plan_i_set[1])
            self.plan_i = None
        self.rec_alt = {}
        self.rec.clear()
        self.org.clear()
        self.pred = []

    def _load_originals(self):
        pocs = {k * self.wl.ip for k in self.my_i}
        for k in self.my_segments:
            pocs |= {j[0] for j in self.wl.segment_jobs(k)}
        for poc in sorted(pocs):
            self._pic(self.org, poc).upload(self.torch, self.wl.original(poc))
            self._pic(self.rec, poc)
        if self.wl.structure == "ra":
            for k in self.my_segments:  # landing buffers for the I pictures received from other ranks
                self._pic(self.rec, (k + 1) * self.wl.ip)
                if self.loopback > 1 and (k + 1) % self.loopback != k % self.loopback:
                    self.landing[(k, (k + 1) * self.wl.ip)] = TorchPicture(self.torch, self.dev, self.wl.w, self.wl.h, MARGIN)
        n_i = len(self.my_i)
        self.lev_i = [capi.DevLevelsZ(self.ctx, self.wl.w, self.wl.h) for _ in range(n_i)]
        return len(pocs)

    def _run(self):
        pixels = self._run_intra(self.ctx, self.plan, self.stream)
        return pixels + self._run_inter()

    def _run_intra(self, ctx, plan, stream):
        L, wl = self.L, self.wl
        w, h = wl.w, wl.h
        # phase 1: all my I pictures in one whole-picture call, then their borders
        ipocs = [k * wl.ip for k in self.my_i]
        n = len(ipocs)
        if n:
            org = (capi.Pic * n)(*[self.org[p].as_pic() for p in ipocs])
            rec = (capi.Pic * n)(*[self.rec[p].as_pic() for p in ipocs])
            lev = (capi.Levels * n)(*[l.as_pic() for l in self.lev_i])
            ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, n, org, rec, lev))
            ctx._chk(L.hmx_pic_extend_border_multi(ctx.h, n, rec, w, h, MARGIN, MARGIN))
        # phase 2: boundary I pictures travel to the owner of the previous segment (RCCL send/recv)
        if self.loopback > 1 and wl.structure == "ra":
            V, loop = self.loopback, LoopbackDist()
            ops = 0
            for vr in range(V):  # every virtual rank posts its sends and receives, as its process would
                loop.rank = vr
                ops += run_exchange(loop, vr, V, wl.n_segments,
                                    lambda ki, vr=vr: self.rec[ki * wl.ip].t if ki % V == vr else self.landing[(ki - 1, ki * wl.ip)].t)
            moved = loop.flush()  # device copies on this stream, behind the chain that reconstructed the pictures
            assert 2 * moved == ops
            self._moved = (moved // 3, moved // 3)
        if self.world > 1 and wl.structure == "ra":
            if self._ev is None:
                self._ev = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
            self._ev[0].record(stream)
            run_exchange(self.dist, self.rank, self.world, wl.n_segments, lambda ki: self.rec[ki * wl.ip].t)
            self._ev[1].record(stream)
            plan_x = exchange_plan(wl.n_segments, self.world)
            self._moved = (sum(1 for (_, s_, _d) in plan_x if s_ == self.rank), sum(1 for (_, _s, d_) in plan_x if d_ == self.rank))
        return n * w * h

    def _run_inter(self):
        # phase 3: inter pictures in coding order; position j of every owned segment in one call per stage
        ctx, L, wl = self.ctx, self.L, self.wl
        w, h = wl.w, wl.h
        pixels = 0
        S = len(self.my_segments)
        if not S:
            return pixels
        jobs = [wl.segment_jobs(k) for k in self.my_segments]
        for j in range(len(jobs[0])):
            groups = {}
            for si in range(S):
                poc, r0, r1, li = jobs[si][j]
                groups.setdefault((li, r1 is None), []).append(si)
            for (li, is_p), members in groups.items():
                d = self.lists[li]
                m = len(members)
                keep = []  # ctypes arrays referenced by pointer from the job table
                mc = (capi.McJob * m)()
                pred, rec, org, lev = (capi.Pic * m)(), (capi.Pic * m)(), (capi.Pic * m)(), (capi.Levels * m)()
                for q, si in enumerate(members):
                    poc, r0, r1, _ = jobs[si][j]
                    seg = self.my_segments[si]
                    refs = [self.landing.get((seg, r0), self.rec[r0])] + ([self.landing.get((seg, r1), self.rec[r1])] if r1 is not None else [])
                    ref_arr = (capi.Pic * len(refs))(*[t.as_pic() for t in refs])
                    pred[q], rec[q], org[q] = self.pred[si].as_pic(), self.rec[poc].as_pic(), self.org[poc].as_pic()
                    lev[q] = self.lev[si].as_pic()
                    keep.append(ref_arr)
                    pus, npu = (d["pus_p"], d["n_p"]) if is_p else (d["pus_b"], d["n_b"])
                    mc[q].d_pus, mc[q].n_pus, mc[q].refs, mc[q].n_refs = pus.ptr, npu, ref_arr, len(refs)
                    mc[q].dst = C.pointer(pred[q])
                    mc[q].pic_w, mc[q].pic_h = w, h
                ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, m, mc))
                if self.fused:
                    ctx._chk(L.hmx_batch_residual_transform_recon_multi(ctx.h, d["tu"], m, org, pred, lev, rec, None, C.byref(self.pp_b)))
                else:  # the two reference-shaped calls (kept for the parity test of both routes)
                    ctx._chk(L.hmx_batch_residual_transformNxN_multi(ctx.h, d["tu"], m, org, pred, lev, None, C.byref(self.pp_b)))
                    ctx._chk(L.hmx_batch_invtransformNxN_multi(ctx.h, d["tu"], m, lev, pred, rec, C.byref(self.pp_b)))
                ctx._chk(L.hmx_pic_extend_border_multi(ctx.h, m, rec, w, h, MARGIN, MARGIN))
                pixels += m * w * h
        return pixels

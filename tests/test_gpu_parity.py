"""GPU parity: libhmx (HIP, through the C-ABI) vs the CPU oracle, bit-exact.  Run with -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from thevc_amd import capi, workload

pytestmark = pytest.mark.gpu
REG_DCT = 65535


@pytest.fixture(scope="module", params=[8, 10])
def ctx(request):
    c = capi.Context(bit_depth=request.param)
    yield c
    c.close()


def _qparam(qpy, text_type, B, slice_type, sign_hide, is_intra, mode):
    qp = capi.qp_for(qpy, text_type, B)
    return capi.QuantParam(qp, -1, slice_type, sign_hide, is_intra, mode), qp


@pytest.mark.parametrize("N", [4, 8, 16, 32])
def test_scalar_transforms(ctx, N):
    O, B = ol.oracle(), ctx.bit_depth
    rng = np.random.default_rng(N + B)
    mx = (1 << B) - 1
    for it in range(12):
        mode = [REG_DCT, 0, 1, 10, 26, 34][it % 6]
        stride = N + 5
        amp = mx if it % 2 == 0 else 32767
        resi = rng.integers(-amp, amp + 1, N * stride).astype(np.int16)
        ref = np.zeros(N * N, np.int32)
        O.hmo_xT(mode, resi, stride, ref, N, B)
        assert np.array_equal(ctx.xT(mode, resi, stride, N), ref)
        coef = rng.integers(-70000, 70000, N * N).astype(np.int32)
        rr = np.zeros(N * stride, np.int16)
        O.hmo_xIT(mode, coef, rr, stride, N, B)
        got = ctx.xIT(mode, coef, stride, N)
        assert np.array_equal(got.reshape(N, stride)[:, :N], rr.reshape(N, stride)[:, :N])
        O.hmo_xTransformSkip(resi, stride, ref, N, B)
        assert np.array_equal(ctx.xTransformSkip(resi, stride, N), ref)
        O.hmo_xITransformSkip(coef, rr, stride, N, B)
        got = ctx.xITransformSkip(coef, stride, N)
        assert np.array_equal(got.reshape(N, stride)[:, :N], rr.reshape(N, stride)[:, :N])


@pytest.mark.parametrize("N", [4, 8, 16, 32])
def test_scalar_quant_paths(ctx, N):
    O, B = ol.oracle(), ctx.bit_depth
    rng = np.random.default_rng(50 + N + B)
    mx = (1 << B) - 1
    n_sbh = 0
    for it in range(40):
        tt = (capi.TEXT_LUMA, capi.TEXT_CHROMA_U, capi.TEXT_CHROMA_V)[it % 3]
        is_intra = int(it % 5 != 4)
        mode = int(rng.integers(0, 35))
        ts = int(N == 4 and it % 7 == 3)
        qpy = int(rng.choice([12, 22, 27, 32, 37, 45]))
        st = capi.I_SLICE if is_intra else (capi.P_SLICE, capi.B_SLICE)[it % 2]
        qpar, qp = _qparam(qpy, tt, B, st, 1, is_intra, mode)
        scan = O.hmo_coef_scan_idx(N, int(tt == 0), is_intra, mode)
        cfg = ol.quant_cfg(qp.per, qp.rem, intra_slice=int(st == capi.I_SLICE), sign_hide=1, scan_idx=scan)
        amp = int(rng.choice([3, 20, 80, mx]))
        stride = N + 3
        resi = rng.integers(-amp, amp + 1, N * stride).astype(np.int16)
        tmode = mode if (tt == 0 and is_intra) else REG_DCT
        ref = np.zeros(N * N, np.int32)
        s = C.c_uint32(0)
        O.hmo_transformNxN(resi, stride, ref, N, B, tmode, ts, 0, C.byref(cfg), C.byref(s))
        lvl, asum = ctx.transformNxN(resi, stride, N, tt, qpar, ts)
        assert np.array_equal(lvl, ref), (it, N, B)
        assert asum == s.value
        cfg0 = ol.quant_cfg(qp.per, qp.rem, intra_slice=int(st == capi.I_SLICE), sign_hide=0, scan_idx=scan)
        ref0 = np.zeros(N * N, np.int32)
        O.hmo_transformNxN(resi, stride, ref0, N, B, tmode, ts, 0, C.byref(cfg0), C.byref(s))
        n_sbh += int(not np.array_equal(ref, ref0))
        # inverse
        rr = np.zeros(N * stride, np.int16)
        O.hmo_invtransformNxN(0, tmode, rr, stride, ref, N, B, qp.per, qp.rem, ts)
        got = ctx.invtransformNxN(ref, stride, N, tt, tmode, qp, ts)
        assert np.array_equal(got.reshape(N, stride)[:, :N], rr.reshape(N, stride)[:, :N])
        # xQuant / xDeQuant on raw Int coefficients
        coef = rng.integers(-amp * 64, amp * 64 + 1, N * N).astype(np.int32)
        refq = np.zeros(N * N, np.int32)
        s2 = C.c_uint32(7)
        O.hmo_xQuant(coef, refq, N, B, C.byref(cfg), C.byref(s2))
        q, acs = ctx.xQuant(coef, N, tt, qpar, ac_sum=7)
        assert np.array_equal(q, refq) and acs == s2.value
        lv = rng.integers(-40000, 40000, N * N).astype(np.int32)
        refd = np.zeros(N * N, np.int32)
        O.hmo_xDeQuant(lv, refd, N, B, qp.per, qp.rem)
        assert np.array_equal(ctx.xDeQuant(lv, N, qp), refd)
    assert n_sbh > 0


def test_scalar_intra(ctx):
    O, B = ol.oracle(), ctx.bit_depth
    w, h = 192, 136
    planes = workload.make_planes(3 + B, w, h, B)
    rng = np.random.default_rng(B)
    flags = np.zeros(65, np.uint8)
    for N in (4, 8, 16, 32, 64):  # 64: the luma prediction unit of a 64x64 coding unit (no transform of that size)
        W = 2 * N + 1
        for chroma in (0, 1):
            if chroma and N >= 32:
                continue
            pl = planes[1] if chroma else planes[0]
            pw, ph = pl.shape[1], pl.shape[0]
            flat = pl.reshape(-1).copy()
            for it in range(14 if N < 64 else 6):
                bx = int(rng.integers(0, (pw - N) // N + 1)) * N
                by = int(rng.integers(0, (ph - N) // N + 1)) * N
                if it == 0:
                    bx = by = 0
                if it == 1:
                    bx, by = (pw - N, ph - N) if N < 64 else ((pw - N) // N * N, (ph - N) // N * N)  # a 64x64 unit is a whole CTU
                adi = ctx.initAdiPattern(flat, pw, bx, by, N, chroma, w, h)
                ref = np.zeros(2 * W * W, np.int32)
                nav = O.hmo_intra_avail(bx << chroma, by << chroma, N << chroma, w, h, 64, flags)
                O.hmo_fillReferenceSamples(ol.ptr(flat, by * pw + bx), pw, flags, nav, 2 if chroma else 4, N, B, ref)
                if not chroma:
                    O.hmo_filterAdi(ref, N)
                assert np.array_equal(adi, ref), (N, chroma, bx, by)
                stride = N + 2
                for mode in range(35):
                    rp = np.zeros(N * stride, np.int16)
                    if chroma:
                        O.hmo_predIntraChromaAng(ref, mode, rp, stride, N, B)
                        got = ctx.predIntraChromaAng(adi, mode, stride, N)
                    else:
                        O.hmo_predIntraLumaAng(ref, mode, rp, stride, N, B)
                        got = ctx.predIntraLumaAng(adi, mode, stride, N)
                    assert np.array_equal(got.reshape(N, stride)[:, :N], rp.reshape(N, stride)[:, :N]), (N, chroma, mode)


def test_scalar_interpolation(ctx):
    O, B = ol.oracle(), ctx.bit_depth
    rng = np.random.default_rng(900 + B)
    for it in range(60):
        w = int(rng.choice([2, 4, 8, 12, 16, 24, 32, 64]))
        h = int(rng.choice([2, 4, 8, 12, 16, 24, 32, 64]))
        ss, ds = w + 16, w + 3
        first_stage = it % 2 == 0
        src = (rng.integers(0, 1 << B, (h + 16) * ss) if first_stage else rng.integers(-32768, 32768, (h + 16) * ss)).astype(np.int16)
        org = 8 * ss + 8
        for chroma in (0, 1):
            frac = int(rng.integers(0, 8 if chroma else 4))
            sfx = "Chroma" if chroma else "Luma"
            for last in (0, 1):
                b = np.zeros(h * ds, np.int16)
                if first_stage:
                    getattr(O, "hmo_filterHor" + sfx)(ol.ptr(src, org), ss, ol.ptr(b), ds, w, h, frac, last, B)
                    got = ctx.filter("filterHor" + sfx, src, org, ss, ds, w, h, frac, is_last=last)
                    assert np.array_equal(got.reshape(h, ds)[:, :w], b.reshape(h, ds)[:, :w])
                first = 1 if first_stage else 0
                getattr(O, "hmo_filterVer" + sfx)(ol.ptr(src, org), ss, ol.ptr(b), ds, w, h, frac, first, last, B)
                got = ctx.filter("filterVer" + sfx, src, org, ss, ds, w, h, frac, is_first=first, is_last=last)
                assert np.array_equal(got.reshape(h, ds)[:, :w], b.reshape(h, ds)[:, :w])
    for (w, h) in ((8, 8), (64, 64), (16, 4), (4, 16)):
        a = rng.integers(-16384, 16384, w * h).astype(np.int16)
        b = rng.integers(-16384, 16384, w * h).astype(np.int16)
        ref = np.zeros(w * h, np.int16)
        O.hmo_addAvg(a, w, b, w, ref, w, w, h, B)
        assert np.array_equal(ctx.addAvg(a, b, w, h), ref)


def _oracle_frame(tus, w, h, B, qp, org, sign_hide=1, decode_levels=None):
    O = ol.oracle()
    cfg = ol.FrameCfg(w, h, 64, B, qp, 0, sign_hide)
    rec = [np.zeros_like(p) for p in org]
    lev = [np.zeros(p.shape, np.int32) for p in org] if decode_levels is None else decode_levels
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    strides = I3(w, w // 2, w // 2)
    t = np.ascontiguousarray(tus, ol.TU_DTYPE)
    if decode_levels is None:
        O.hmo_intra_frame_encode(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in org]), strides,
                                 P3(*[p.ctypes.data for p in rec]), strides, P3(*[p.ctypes.data for p in lev]))
    else:
        O.hmo_intra_frame_decode(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in rec]), strides,
                                 P3(*[p.ctypes.data for p in lev]))
    return rec, lev


@pytest.mark.parametrize("schedule", ["wave", "level", "packed"])
@pytest.mark.parametrize("pic,tiling", [((192, 128), "mix"), ((416, 240), "mix"), ((128, 64), 4), ((128, 64), 8),
                                        ((128, 128), 16), ((128, 128), 32), ((200, 136), "mix")])
def test_frame_intra_encode_decode(ctx, pic, tiling, schedule, hmx_opts):
    """Whole-picture all-intra chain (refs <- recon, pred, T, Q, IQ, IT, recon) incl. pictures whose
    right/bottom edge cuts the last CTU; 3 pictures per call share one plan.  Both dependency
    schedules of the library (CTU-diagonal waves / picture-wide levels) must give the same bits."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE=schedule)
    B = ctx.bit_depth
    w, h = pic
    tus = workload.make_tus(7, w, h, tiling)
    L = capi.lib()
    n_pics = 3
    pp = capi.PicParam(w, h, 30, 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(tus, pp)
    orgs = [workload.make_planes(10 + i, w, h, B, "texture" if i else "noise") for i in range(n_pics)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n_pics)]
    A = lambda lst, T: (T * n_pics)(*[x.as_pic() for x in lst])
    ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, n_pics, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n_pics):
        rec_ref, lev_ref = _oracle_frame(tus, w, h, B, 30, orgs[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        for p in range(3):
            assert np.array_equal(lev[p], lev_ref[p]), ("levels", i, p)
            assert np.array_equal(rec[p], rec_ref[p]), ("recon", i, p)
    # decode direction: levels -> recon must reproduce the encoder's reconstruction
    d_rec2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
    ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, n_pics, A(d_rec2, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n_pics):
        a, b = d_rec2[i].download(), d_rec[i].download()
        for p in range(3):
            assert np.array_equal(a[p], b[p]), ("decode", i, p)
    L.hmx_intra_plan_destroy(ctx.h, plan)
    for d in d_org + d_rec + d_rec2 + d_lev:
        d.free()


@pytest.mark.parametrize("B,qp,ctu,sign_hide,cqo", [(12, 51, 64, 1, 0), (8, 0, 64, 1, 0), (10, 51, 32, 0, 0), (8, 22, 16, 1, 3),
                                                   (12, 4, 64, 0, -5), (10, 37, 32, 1, 12), (8, 51, 64, 1, -12)])
def test_frame_intra_parameter_corners(B, qp, ctu, sign_hide, cqo):
    """The corners of the parameter space the reference allows: 12-bit samples, QP 0 and 51, chroma QP offsets
    through the chroma table's clip, sign hiding off, CTU 32 and 16 (availability and Z-order geometry change),
    both level layouts, the across-pictures and the per-picture level kernels."""
    ctx = capi.Context(bit_depth=B, ctu_size=ctu)
    try:
        L = capi.lib()
        w, h, n_pics = 152, 88, 5
        tus = workload.make_tus(31 + qp, w, h, "mix", ctu=ctu)
        pp = capi.PicParam(w, h, qp, cqo, capi.I_SLICE, sign_hide)
        plan = ctx.intra_plan(tus, pp)
        orgs = [workload.make_planes(200 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n_pics)]
        O = ol.oracle()
        refs = []
        for o in orgs:
            cfg = ol.FrameCfg(w, h, ctu, B, qp, cqo, sign_hide)
            rec = [np.zeros_like(p) for p in o]
            lev = [np.zeros(p.shape, np.int32) for p in o]
            P3, I3 = C.c_void_p * 3, C.c_int * 3
            st = I3(w, w // 2, w // 2)
            t = np.ascontiguousarray(tus, ol.TU_DTYPE)
            O.hmo_intra_frame_encode(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in o]), st,
                                     P3(*[p.ctypes.data for p in rec]), st, P3(*[p.ctypes.data for p in lev]))
            refs.append((rec, lev))
        A = lambda lst, T: (T * n_pics)(*[x.as_pic() for x in lst])
        for multi in (False, True):  # one plan for the call / one plan pointer per picture
            d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
            d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
            d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n_pics)]
            if multi:
                plans = (C.c_void_p * n_pics)(*[plan.value] * n_pics)
                ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, plans, n_pics, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
            else:
                ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, n_pics, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
            ctx.sync()
            for i in range(n_pics):
                rec, lev = d_rec[i].download(), d_lev[i].download()
                for p in range(3):
                    assert np.array_equal(lev[p], refs[i][1][p]), ("levels", multi, i, p)
                    assert np.array_equal(rec[p], refs[i][0][p]), ("recon", multi, i, p)
            for d in d_org + d_rec + d_lev:
                d.free()
        L.hmx_intra_plan_destroy(ctx.h, plan)
    finally:
        ctx.close()


@pytest.mark.parametrize("across", ["1", "groups3", "pipelined", "pipelined3", "0"])
@pytest.mark.parametrize("pic,n_pics", [((64, 64), 70), ((136, 72), 9), ((200, 264), 5)])
def test_frame_intra_many_pictures_one_plan(ctx, pic, n_pics, across, hmx_opts):
    """Pictures that follow one plan run in SIMD across pictures (one wave = one block of 16/8/4/1 pictures):
    more pictures than a wave has slots (70 > 64), ragged last chunks (9 = 8 + 1 for 8x8 blocks, 70 = 4 x 16 + 6), and the
    per-picture level kernel (HMX_INTRA_ACROSS=0) must all give the oracle's bits."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE="level")
    # "pipelined": the layout conversions run CTU row by CTU row on their own stream, overlapped with the chain
    # (the default from 256 pictures up); "pipelined3": the same with three picture groups on three streams
    hmx_opts(ctx, HMX_INTRA_ACROSS="0" if across == "0" else "1")
    hmx_opts(ctx, HMX_PIPELINE_CONV="1" if across.startswith("pipelined") else "0")
    if across in ("pipelined3", "groups3"):  # "groups3": what a call of 640 pictures or more does by default
        hmx_opts(ctx, HMX_INTRA_STREAMS="3")
    B = ctx.bit_depth
    w, h = pic
    tus = workload.make_tus(21, w, h, "mix")
    L = capi.lib()
    pp = capi.PicParam(w, h, 27, 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(tus, pp)
    orgs = [workload.make_planes(100 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n_pics)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
    d_lev = [capi.DevLevelsZ(ctx, w, h) if i % 2 else capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for i in range(n_pics)]
    A = lambda lst, T: (T * n_pics)(*[x.as_pic() for x in lst])
    ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, n_pics, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    recs = []
    for i in range(n_pics):
        rec_ref, lev_ref = _oracle_frame(tus, w, h, B, 27, orgs[i])
        rec = d_rec[i].download()
        recs.append(rec)
        for p in range(3):
            assert np.array_equal(rec[p], rec_ref[p]), ("recon", i, p)
        if i % 2 == 0:
            lev = d_lev[i].download()
            for p in range(3):
                assert np.array_equal(lev[p], lev_ref[p]), ("levels", i, p)
    d_rec2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
    ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, n_pics, A(d_rec2, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n_pics):
        a = d_rec2[i].download()
        for p in range(3):
            assert np.array_equal(a[p], recs[i][p]), ("decode", i, p)
    L.hmx_intra_plan_destroy(ctx.h, plan)
    for d in d_org + d_rec + d_rec2 + d_lev:
        d.free()


def test_batch_pred64(ctx):
    """HOT LOOP A at the size of a 64x64 coding unit (TEncSearch.cpp:2509-2540: initAdiPattern, 35 x predIntraLumaAng + calcHAD
    with uiWidth = 64): hmx_batch_predIntra / _cost on lists that hold 64x64 luma prediction units next to smaller blocks, at
    every CTU position of a picture whose right edge cuts the last CTU column (above-right whole / cut / missing)."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_calcHAD.restype = C.c_uint32
    w, h = 232, 136
    rec = workload.make_planes(17 + B, w, h, B)
    org = workload.make_planes(18 + B, w, h, B, "texture")
    flat = rec[0].reshape(-1).copy()
    tus = np.zeros(8, capi.TU_DTYPE)
    pos = [(0, 0, 6), (64, 0, 6), (128, 0, 6), (0, 64, 6), (64, 64, 6), (128, 64, 6), (192, 32, 5), (200, 128, 3)]
    for i, (x, y, lg) in enumerate(pos):
        tus[i]["x"], tus[i]["y"], tus[i]["log2n"], tus[i]["plane"], tus[i]["mode"] = x, y, lg, 0, (7 * i + 3) % 35
    lst = ctx.tu_list(tus)
    pp = capi.PicParam(w, h, 30, 0, capi.I_SLICE, 1)
    d_rec = capi.DevPicture(ctx, w, h).upload(rec)
    d_org = capi.DevPicture(ctx, w, h).upload(org)
    d_pr = capi.DevPicture(ctx, w, h).zero()
    # the block's own mode
    ctx._chk(L.hmx_batch_predIntra(ctx.h, lst, C.byref(d_rec.as_pic()), C.byref(d_pr.as_pic()), C.byref(pp), None, 0, None))
    ctx.sync()
    got = d_pr.download()[0]
    for i, t in enumerate(tus):
        N, x, y = 1 << int(t["log2n"]), int(t["x"]), int(t["y"])
        ref = ol.o_intra_pred(flat, w, x, y, N, int(t["mode"]), B, w, h, False)
        assert np.array_equal(got[y:y + N, x:x + N], ref), ("own mode", i, N)
    # the 35-mode fan-out, written out and costed in place
    modes = np.arange(35, dtype=np.uint8)
    d_modes = ctx.to_device(modes)
    fan = [ctx.alloc(2 * 35 * w * h), ctx.alloc(4), ctx.alloc(4)]
    fp = capi.Pic()
    fp.plane[0], fp.plane[1], fp.plane[2] = fan[0].ptr, fan[1].ptr, fan[2].ptr
    fp.stride[0], fp.stride[1], fp.stride[2] = w, w // 2, w // 2
    elems = (C.c_size_t * 3)(w * h, 0, 0)
    ctx._chk(L.hmx_batch_predIntra(ctx.h, lst, C.byref(d_rec.as_pic()), C.byref(fp), C.byref(pp), d_modes.ptr, 35, C.byref(elems)))
    d_cost = ctx.alloc(4 * len(tus) * 35)
    ctx._chk(L.hmx_batch_predIntra_cost(ctx.h, lst, C.byref(d_rec.as_pic()), C.byref(d_org.as_pic()), C.byref(pp), d_modes.ptr, 35, d_cost.ptr))
    ctx.sync()
    cand = fan[0].download(np.int16).reshape(35, h, w)
    cost = d_cost.download(np.uint32).reshape(len(tus), 35)
    for i, t in enumerate(tus):
        N, x, y = 1 << int(t["log2n"]), int(t["x"]), int(t["y"])
        ob = np.ascontiguousarray(org[0][y:y + N, x:x + N])
        for m in range(35):
            ref = np.ascontiguousarray(ol.o_intra_pred(flat, w, x, y, N, m, B, w, h, False))
            assert np.array_equal(cand[m, y:y + N, x:x + N], ref), ("fan-out", i, N, m)
            want = O.hmo_calcHAD(ob.ctypes.data_as(C.c_void_p), N, ref.ctypes.data_as(C.c_void_p), N, N, N, B)
            assert cost[i, m] == want, ("satd", i, N, m)
    # a list with a 64x64 block is a prediction list: the transform entry points refuse it (there is no 64x64 transform)
    d_lev = capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
    assert L.hmx_batch_transformNxN(ctx.h, lst, C.byref(d_rec.as_pic()), C.byref(d_lev.as_pic()), None, C.byref(pp)) == -1
    # a 64x64 block must be a whole luma CTU
    bad = tus[:1].copy()
    bad["x"] = 32
    out = C.c_void_p()
    assert L.hmx_tu_list_create(ctx.h, bad.ctypes.data_as(C.c_void_p), 1, C.byref(out)) == -1
    L.hmx_tu_list_destroy(ctx.h, lst)
    for d in (d_rec, d_org, d_pr, d_lev):
        d.free()


def test_batch_lists(ctx):
    """transformNxN / invtransformNxN (+recon) / predIntra (+35-mode fan-out) over block lists."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    w, h = 192, 128
    tus = workload.make_tus(21, w, h, "mix")
    inter = np.random.default_rng(4).random(len(tus)) < 0.3
    tus["flags"] = np.where(inter, tus["flags"] | capi.TU_INTER, tus["flags"]).astype(np.uint8)
    tus["flags"] = np.where(inter, tus["flags"] & ~np.uint8(capi.TU_TRANSFORM_SKIP), tus["flags"])
    lst = ctx.tu_list(tus)
    mx = (1 << B) - 1
    rng = np.random.default_rng(B)
    resi = [rng.integers(-mx // 4, mx // 4 + 1, p).astype(np.int16) for p in ((h, w), (h // 2, w // 2), (h // 2, w // 2))]
    pred = workload.make_planes(5, w, h, B)
    d_resi = capi.DevPicture(ctx, w, h).upload(resi)
    d_pred = capi.DevPicture(ctx, w, h).upload(pred)
    d_out = capi.DevPicture(ctx, w, h).zero()
    d_lev = capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
    d_sum = ctx.alloc(4 * len(tus))
    pp = capi.PicParam(w, h, 27, 0, capi.P_SLICE, 1)
    ctx._chk(L.hmx_batch_transformNxN(ctx.h, lst, C.byref(d_resi.as_pic()), C.byref(d_lev.as_pic()), d_sum.ptr, C.byref(pp)))
    ctx._chk(L.hmx_batch_invtransformNxN(ctx.h, lst, C.byref(d_lev.as_pic()), C.byref(d_pred.as_pic()),
                                         C.byref(d_out.as_pic()), C.byref(pp)))
    ctx.sync()
    lev, out, sums = d_lev.download(), d_out.download(), d_sum.download(np.uint32)
    for i, t in enumerate(tus):
        N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        is_inter = bool(t["flags"] & capi.TU_INTER)
        ts = int(t["flags"] & capi.TU_TRANSFORM_SKIP)
        qp = O.hmo_setQPforQuant(27, int(p != 0), 6 * (B - 8), 0)
        scan = O.hmo_coef_scan_idx(N, int(p == 0), int(not is_inter), int(t["mode"]))
        cfg = ol.quant_cfg(qp.per, qp.rem, intra_slice=0, sign_hide=1, scan_idx=scan)
        tmode = int(t["mode"]) if (p == 0 and not is_inter) else REG_DCT
        blk = np.ascontiguousarray(resi[p][y:y + N, x:x + N])
        ref, s = ol.o_transformNxN(blk, N, B, tmode, ts, cfg)
        assert np.array_equal(lev[p][y:y + N, x:x + N], ref), i
        assert sums[i] == s
        r = ol.o_invtransformNxN(ref, N, B, tmode, qp.per, qp.rem, ts)
        rec = np.clip(pred[p][y:y + N, x:x + N].astype(np.int32) + r, 0, mx).astype(np.int16)
        assert np.array_equal(out[p][y:y + N, x:x + N], rec), i
    # prediction from a fixed reconstructed picture (every neighbour taken from it)
    tus2 = workload.make_tus(22, w, h, "mix")
    lst2 = ctx.tu_list(tus2)
    d_pr = capi.DevPicture(ctx, w, h).zero()
    ctx._chk(L.hmx_batch_predIntra(ctx.h, lst2, C.byref(d_pred.as_pic()), C.byref(d_pr.as_pic()), C.byref(pp), None, 0, None))
    ctx.sync()
    got = d_pr.download()
    flat = [p.reshape(-1).copy() for p in pred]
    for i, t in enumerate(tus2):
        N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        ref = ol.o_intra_pred(flat[p], pred[p].shape[1], x, y, N, int(t["mode"]), B, w, h, p != 0)
        assert np.array_equal(got[p][y:y + N, x:x + N], ref), (i, t)
    # 35-mode fan-out on luma blocks
    luma = tus2[tus2["plane"] == 0][:200]
    lst3 = ctx.tu_list(luma)
    modes = np.arange(35, dtype=np.uint8)
    d_modes = ctx.to_device(modes)
    fan = [ctx.alloc(2 * 35 * w * h), ctx.alloc(4), ctx.alloc(4)]
    fp = capi.Pic()
    fp.plane[0], fp.plane[1], fp.plane[2] = fan[0].ptr, fan[1].ptr, fan[2].ptr
    fp.stride[0], fp.stride[1], fp.stride[2] = w, w // 2, w // 2
    elems = (C.c_size_t * 3)(w * h, 0, 0)
    ctx._chk(L.hmx_batch_predIntra(ctx.h, lst3, C.byref(d_pred.as_pic()), C.byref(fp), C.byref(pp), d_modes.ptr, 35, C.byref(elems)))
    ctx.sync()
    cand = fan[0].download(np.int16).reshape(35, h, w)
    for t in luma[::7]:
        N, x, y = 1 << int(t["log2n"]), int(t["x"]), int(t["y"])
        for m in range(35):
            ref = ol.o_intra_pred(flat[0], w, x, y, N, m, B, w, h, False)
            assert np.array_equal(cand[m, y:y + N, x:x + N], ref), (t, m)
    # the same fan-out costed in place: calcHAD(original, prediction) for every (block, mode), nothing written back
    O = ol.oracle()
    O.hmo_calcHAD.restype = C.c_uint32
    orgp = workload.make_planes(91, w, h, B, "texture")
    d_orgp = capi.DevPicture(ctx, w, h).upload(orgp)
    d_cost = ctx.alloc(4 * len(luma) * 35)
    ctx._chk(L.hmx_batch_predIntra_cost(ctx.h, lst3, C.byref(d_pred.as_pic()), C.byref(d_orgp.as_pic()), C.byref(pp), d_modes.ptr, 35, d_cost.ptr))
    ctx.sync()
    cost = d_cost.download(np.uint32).reshape(len(luma), 35)
    for i, t in enumerate(luma):
        N, x, y = 1 << int(t["log2n"]), int(t["x"]), int(t["y"])
        ob = np.ascontiguousarray(orgp[0][y:y + N, x:x + N])
        for m in range(0, 35, 3) if i % 7 else range(35):
            pb = np.ascontiguousarray(ol.o_intra_pred(flat[0], w, x, y, N, m, B, w, h, False))
            want = O.hmo_calcHAD(ob.ctypes.data_as(C.c_void_p), N, pb.ctypes.data_as(C.c_void_p), N, N, N, B)
            assert cost[i, m] == want, ("satd", i, N, m)
    # chroma and the block's own mode (d_modes NULL), all sizes
    d_cost2 = ctx.alloc(4 * len(tus2))
    ctx._chk(L.hmx_batch_predIntra_cost(ctx.h, lst2, C.byref(d_pred.as_pic()), C.byref(d_orgp.as_pic()), C.byref(pp), None, 0, d_cost2.ptr))
    ctx.sync()
    cost2 = d_cost2.download(np.uint32)
    for i, t in enumerate(tus2):
        N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        ob = np.ascontiguousarray(orgp[p][y:y + N, x:x + N])
        pb = np.ascontiguousarray(got[p][y:y + N, x:x + N])
        assert cost2[i] == O.hmo_calcHAD(ob.ctypes.data_as(C.c_void_p), N, pb.ctypes.data_as(C.c_void_p), N, N, N, B), ("satd own mode", i)
    for l in (lst, lst2, lst3):
        L.hmx_tu_list_destroy(ctx.h, l)


def test_distortion_dropins(ctx):
    """calcHAD / getDistPart(SSE) drop-ins vs the oracle over block shapes incl. non-square ones."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_calcHAD.restype = O.hmo_getSSE.restype = C.c_uint32
    rng = np.random.default_rng(33 + B)
    mx = (1 << B) - 1
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (16, 8), (12, 16), (64, 32)):
        so, sc = w + 3, w + 1
        org = rng.integers(0, mx + 1, so * h).astype(np.int16)
        cur = rng.integers(0, mx + 1, sc * h).astype(np.int16)
        po, pc = org.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p)
        v = C.c_uint32(0)
        ctx._chk(L.hmx_calcHAD(ctx.h, po, so, pc, sc, w, h, C.byref(v)))
        assert v.value == O.hmo_calcHAD(po, so, pc, sc, w, h, B), ("calcHAD", w, h)
        ctx._chk(L.hmx_getSSE(ctx.h, pc, sc, po, so, w, h, C.byref(v)))
        assert v.value == O.hmo_getSSE(po, so, pc, sc, w, h, B), ("SSE", w, h)


def test_batch_motion_compensation(ctx):
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    w, h, m = 256, 192, 80
    n_refs = 2
    refs = [workload.make_planes(40 + i, w, h, B) for i in range(n_refs)]
    d_refs = [capi.DevPicture(ctx, w, h, m, m).upload(r) for r in refs]
    for d in d_refs:
        ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d.as_pic()), w, h, m, m))
    ctx.sync()
    # border extension parity + oracle-side extended planes
    ext = []
    for i in range(n_refs):
        full = d_refs[i].download(with_margins=True)
        planes = []
        for p in range(3):
            pw, ph, pmx, pmy = d_refs[i].dims[p]
            st = pw + 2 * pmx
            e = np.zeros((ph + 2 * pmy, st), np.int16)
            e[pmy:pmy + ph, pmx:pmx + pw] = refs[i][p]
            flat = e.reshape(-1)
            O.hmo_extendPicBorder(ol.ptr(flat, pmy * st + pmx), st, pw, ph, pmx, pmy)
            assert np.array_equal(full[p], e), ("border", i, p)
            planes.append(flat)
        ext.append(planes)
    for bi_frac in (0.0, 0.5):
        pus = workload.make_pus(9, w, h, n_refs=n_refs, bi_frac=bi_frac)
        d_pus = ctx.to_device(pus)
        d_dst = capi.DevPicture(ctx, w, h).zero()
        ref_arr = (capi.Pic * n_refs)(*[d.as_pic() for d in d_refs])
        ctx._chk(L.hmx_batch_motionCompensation(ctx.h, d_pus.ptr, len(pus), ref_arr, n_refs, C.byref(d_dst.as_pic())))
        ctx.sync()
        got = d_dst.download()
        # oracle
        dst = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        P3, I3 = C.c_void_p * 3, C.c_int * 3
        ptrs = (C.c_void_p * (3 * n_refs))()
        for i in range(n_refs):
            for p in range(3):
                pw, ph, pmx, pmy = d_refs[i].dims[p]
                ptrs[i * 3 + p] = ext[i][p].ctypes.data + 2 * (pmy * (pw + 2 * pmx) + pmx)
        rs = I3(w + 2 * m, w // 2 + m, w // 2 + m)
        t = np.ascontiguousarray(pus, ol.PU_DTYPE)
        O.hmo_mc_frame(t.ctypes.data, len(t), B, ptrs, rs, P3(*[d.ctypes.data for d in dst]), I3(w, w // 2, w // 2))
        for p in range(3):
            assert np.array_equal(got[p], dst[p]), ("mc", bi_frac, p)
        # the multi-picture entry point with the picture size given: cell-map path, two jobs in one call
        d_dst2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(2)]
        dst_pics = (capi.Pic * 2)(*[d.as_pic() for d in d_dst2])
        jobs = (capi.McJob * 2)()
        for q in range(2):
            jobs[q].d_pus, jobs[q].n_pus, jobs[q].refs, jobs[q].n_refs = d_pus.ptr, len(pus), ref_arr, n_refs
            jobs[q].dst, jobs[q].pic_w, jobs[q].pic_h = C.pointer(dst_pics[q]), w, h
        ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, 2, jobs))
        ctx.sync()
        for q in range(2):
            got2 = d_dst2[q].download()
            for p in range(3):
                assert np.array_equal(got2[p], dst[p]), ("mc multi", bi_frac, q, p)


@pytest.mark.parametrize("schedule", ["wave", "level", "packed"])
def test_frame_intra_multi_plan(ctx, schedule, hmx_opts):
    """Every picture with its own block structure and modes (hmx_frame_intra_encode_multi)."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE=schedule)
    B, L = ctx.bit_depth, capi.lib()
    w, h, n = 256, 192, 4
    pp = capi.PicParam(w, h, 27, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(30 + i, w, h, "mix" if i % 2 == 0 else (4, 16)[i // 2 % 2]) for i in range(n)]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(60 + i, w, h, B, "texture") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode(tus[i], w, h, B, 27, orgs[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        for p in range(3):
            assert np.array_equal(lev[p], lr[p]) and np.array_equal(rec[p], rr[p]), (i, p)
    d_rec2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    ctx._chk(L.hmx_frame_intra_decode_multi(ctx.h, parr, n, A(d_rec2, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        a, b = d_rec2[i].download(), d_rec[i].download()
        assert all(np.array_equal(a[p], b[p]) for p in range(3))
    for p in plans:
        L.hmx_intra_plan_destroy(ctx.h, p)


@pytest.mark.parametrize("knobs", [{}, {"HMX_PACK_SLOTS4": "64"}, {"HMX_PACK_SLOTS4": "16", "HMX_PACK_GROUP": "64"},
                                   {"HMX_PACK_WAVES": "3", "HMX_PACK_GROUP": "7"}, {"HMX_PACK_WAVES": "4096", "HMX_PACK_SLOTS4": "64", "HMX_PACK_GROUP": "3"},
                                   {"HMX_PACK_SLOTS8": "16"}, {"HMX_PACK_SLOTS8": "16", "HMX_PACK_GROUP": "5", "HMX_PACK_WAVES": "7"}])
@pytest.mark.parametrize("pic,n", [((136, 72), 70), ((256, 192), 9), ((64, 64), 130)])
def test_frame_intra_packed_own_plans(ctx, pic, n, knobs, hmx_opts):
    """The packed schedule (one persistent launch, k_intra_packed): n pictures, EVERY ONE with its own block structure
    and modes, packed into waves across pictures -- more pictures than a group holds (70 and 130 > 64: a full group
    plus a ragged one with 64-picture groups; groups of 3, 7, 9 and 17 pictures otherwise), plans with different numbers
    of dependency levels (uniform 4x4 / 16x16 / 32x32 tilings next to mixed ones), both 4x4 wave shapes, both 8x8 shapes (eight
    lanes per block, or four lanes with two rows each and sixteen blocks per wave: HMX_PACK_SLOTS8), three
    persistent waves only (at most three XCDs own all the shards; every wave-item waits behind tickets drawn much
    earlier) and far more waves than work.  Bit-exact vs the oracle, encoder and decoder direction, both level layouts."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE="packed", **knobs)
    B, L = ctx.bit_depth, capi.lib()
    w, h = pic
    qp = 30
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tilings = ["mix", "mix", "mix", 4, "mix", 16, "mix", 32, 8]
    tus = [workload.make_tus(500 + i, w, h, tilings[i % len(tilings)]) for i in range(n)]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(700 + i, w, h, B, "texture" if i % 3 else "noise") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevLevelsZ(ctx, w, h) if i % 2 else capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for i in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    for rep in range(2):  # the second call re-uses the schedule tables of the first (same pictures, same plans)
        ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    sched, groups = C.c_int(), C.c_int()
    L.hmx_last_call_shape(ctx.h, C.byref(sched), C.byref(groups))
    assert sched.value == 3
    recs = []
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i])
        rec = d_rec[i].download()
        recs.append(rec)
        lev = d_lev[i].to_planes(tus[i]) if i % 2 else d_lev[i].download()
        for p in range(3):
            assert np.array_equal(rec[p], rr[p]), ("recon", i, p)
            assert np.array_equal(lev[p], lr[p]), ("levels", i, p)
    d_rec2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    ctx._chk(L.hmx_frame_intra_decode_multi(ctx.h, parr, n, A(d_rec2, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        a = d_rec2[i].download()
        assert all(np.array_equal(a[p], recs[i][p]) for p in range(3)), ("decode", i)
    for p in plans:
        L.hmx_intra_plan_destroy(ctx.h, p)
    for d in d_org + d_rec + d_rec2 + d_lev:
        d.free()


@pytest.mark.parametrize("schedule", ["wave", "level", "packed"])
@pytest.mark.parametrize("pic,tiling", [((416, 240), "mix"), ((200, 136), "mix"), ((128, 128), 32)])
def test_frame_intra_zorder_levels(ctx, pic, tiling, schedule, hmx_opts):
    """Levels in the reference's Z-order coefficient layout (hmx_levels.stride == 0), encode + decode."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE=schedule)
    B, L = ctx.bit_depth, capi.lib()
    w, h = pic
    tus = workload.make_tus(17, w, h, tiling)
    pp = capi.PicParam(w, h, 32, 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(tus, pp)
    org = workload.make_planes(18, w, h, B, "texture")
    d_org = capi.DevPicture(ctx, w, h).upload(org)
    d_rec = capi.DevPicture(ctx, w, h).zero()
    d_lev = capi.DevLevelsZ(ctx, w, h).zero()
    ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, 1, C.byref(d_org.as_pic()), C.byref(d_rec.as_pic()), C.byref(d_lev.as_pic())))
    ctx.sync()
    rr, lr = ol.o_intra_frame_encode(tus, w, h, B, 32, org)
    rec, lev = d_rec.download(), d_lev.to_planes(tus)
    for p in range(3):
        assert np.array_equal(lev[p], lr[p]) and np.array_equal(rec[p], rr[p]), p
    d_rec2 = capi.DevPicture(ctx, w, h).zero()
    ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, 1, C.byref(d_rec2.as_pic()), C.byref(d_lev.as_pic())))
    ctx.sync()
    rec2 = d_rec2.download()
    assert all(np.array_equal(rec2[p], rr[p]) for p in range(3))
    L.hmx_intra_plan_destroy(ctx.h, plan)


def test_rdoq_batch_vs_oracle(ctx):
    """RDOQ over a block list (one lane per block): coefficients of random residuals in level-plane geometry,
    several bit-estimate tables, luma and chroma QPs and lambdas, intra scans and inter blocks, both cbf
    branches -> levels and absolute sums bit-exact with the oracle (doubles in the reference's order)."""
    B, L, O = ctx.bit_depth, capi.lib(), ol.oracle()
    rng = np.random.default_rng(77 + B)
    w, h, qp, cqo = 128, 96, 30, 2
    tus = workload.make_tus(17, w, h, "mix")
    inter = rng.random(len(tus)) < 0.3
    tus["flags"] = np.where(inter, capi.TU_INTER, 0).astype(np.uint8)
    n = len(tus)
    n_est = 5
    ests = [ol.make_est_bits(rng) for _ in range(n_est)]
    est_arr = (capi.EstBits * n_est)(*[capi.EstBits.from_buffer_copy(bytes(e)) for e in ests])
    side = (capi.RdoqSide * n)()
    coef = [np.zeros((h, w), np.int32), np.zeros((h // 2, w // 2), np.int32), np.zeros((h // 2, w // 2), np.int32)]
    mx = (1 << B) - 1
    for i, t in enumerate(tus):
        N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        amp = int(rng.choice([20, 60, 200, mx]))
        resi = rng.integers(-amp, amp + 1, N * N).astype(np.int16)
        c = np.zeros(N * N, np.int32)
        tmode = int(t["mode"]) if (p == 0 and not inter[i]) else REG_DCT
        O.hmo_xT(tmode, resi, N, c, N, B)
        coef[p][y:y + N, x:x + N] = c.reshape(N, N)
        side[i].est_idx = int(rng.integers(0, n_est))
        side[i].root_cbf = int(inter[i] and p == 0 and rng.random() < 0.5)
        side[i].cbf_ctx = int(rng.integers(0, 5)) + (5 if p else 0)
    lam = (41.5, 33.25)
    d_coef = capi.DevPicture(ctx, w, h, dtype=np.int32).upload(coef)
    d_lev = capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
    d_sum = ctx.alloc(4 * n)
    pp = capi.PicParam(w, h, qp, cqo, capi.I_SLICE, 1)
    t_c = np.ascontiguousarray(tus, capi.TU_DTYPE)
    ctx._chk(L.hmx_batch_xRateDistOptQuant(ctx.h, t_c.ctypes.data, side, n, C.byref(d_coef.as_pic()), C.byref(d_lev.as_pic()),
                                           d_sum.ptr, C.byref(pp), est_arr, n_est, lam[0], lam[1]))
    ctx.sync()
    lev = d_lev.download()
    sums = d_sum.download(np.uint32, n)
    n_nz = 0
    for i, t in enumerate(tus):
        N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        q = O.hmo_setQPforQuant(qp, int(p != 0), 6 * (B - 8), cqo if p else 0)
        scan = O.hmo_coef_scan_idx(N, int(p == 0), int(not inter[i]), int(t["mode"]))
        cfg = ol.RdoqCfg(q.per, q.rem, int(p == 0), int(not inter[i]), scan, side[i].root_cbf, side[i].cbf_ctx, 1, lam[1 if p else 0])
        lo, so = ol.o_rdoq(coef[p][y:y + N, x:x + N], N, B, cfg, ests[side[i].est_idx])
        assert np.array_equal(lev[p][y:y + N, x:x + N], lo), ("levels", i, N, p)
        assert int(sums[i]) == so, ("abs_sum", i)
        n_nz += int(so > 0)
    assert n_nz > n // 4
    d_coef.free(), d_lev.free(), d_sum.free()


def test_intra_building_blocks(ctx):
    """predIntraGetPredValDC / xPredIntraPlanar / xPredIntraAng drop-ins on random border buffers vs the oracle."""
    O, B = ol.oracle(), ctx.bit_depth
    O.hmo_predIntraGetPredValDC.restype = C.c_int16
    rng = np.random.default_rng(61 + B)
    for N in (4, 8, 16, 32):
        W = 2 * N + 1
        adi = rng.integers(0, 1 << B, W * W).astype(np.int32)
        src = ol.ptr(adi, W + 1)
        for above in (0, 1):
            for left in (0, 1):
                want = O.hmo_predIntraGetPredValDC(src, W, N, above, left)
                assert ctx.predIntraGetPredValDC(adi, N, above, left) == want, (N, above, left)
                got = ctx.xPredIntraAng(adi, N, 1, above, left, 0)
                assert (got == want).all(), ("dc fill", N, above, left)
        ref = np.zeros(N * N, np.int16)
        O.hmo_xPredIntraPlanar(src, W, ref.ctypes.data_as(C.c_void_p), N, N)
        assert np.array_equal(ctx.xPredIntraPlanar(adi, N), ref), ("planar", N)
        for mode in range(2, 35):
            for filt in (0, 1):
                O.hmo_xPredIntraAng(src, W, ref.ctypes.data_as(C.c_void_p), N, N, mode, filt, B)
                assert np.array_equal(ctx.xPredIntraAng(adi, N, mode, 1, 1, filt), ref), ("ang", N, mode, filt)


def test_yuv_files_on_device(ctx, tmp_path):
    """hmx_yuv_unpack / hmx_yuv_pack (through thevc_amd.yuvio, real files) vs the oracle's TVideoIOYuv restatement:
    8- and 16-bit files, scaling up and down to the context's bit depth, right/bottom padding, cropping."""
    from thevc_amd import yuvio
    O, B = ol.oracle(), ctx.bit_depth
    rng = np.random.default_rng(5 + B)
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    for file_bits in (8, 10, 12):
        w, h, px, py = 72, 40, 24, 8
        wf, hf = w + px, h + py
        wide = file_bits > 8
        frames = []
        path = str(tmp_path / f"in{file_bits}.yuv")
        with open(path, "wb") as f:
            for k in range(3):
                vals = rng.integers(0, 1 << file_bits, w * h * 3 // 2)
                raw = vals.astype("<u2").tobytes() if wide else vals.astype(np.uint8).tobytes()
                frames.append(np.frombuffer(raw, np.uint8))
                f.write(raw)
        rd = yuvio.YuvReader(ctx, path, w, h, file_bits)
        wr = yuvio.YuvWriter(ctx, str(tmp_path / f"out{file_bits}.yuv"), file_bits)
        pic = capi.DevPicture(ctx, wf, hf)
        rd.skip_frames(1)
        want_bytes = b""
        for k in (1, 2):
            assert rd.read(pic, px, py)
            ctx.sync()
            got = pic.download()
            oy, ocb, ocr = np.zeros((hf, wf), np.int16), np.zeros((hf // 2, wf // 2), np.int16), np.zeros((hf // 2, wf // 2), np.int16)
            O.hmo_yuv_unpack(frames[k].ctypes.data_as(C.c_void_p), file_bits, B, wf, hf, px, py,
                             P3(oy.ctypes.data, ocb.ctypes.data, ocr.ctypes.data), I3(wf, wf // 2, wf // 2))
            for p, o in enumerate((oy, ocb, ocr)):
                assert np.array_equal(got[p], o), ("unpack", file_bits, k, p)
            wr.write(pic, wf, hf, px, py)
            mine = np.zeros(w * h * 3 // 2 * (2 if wide else 1), np.uint8)
            O.hmo_yuv_pack(P3(oy.ctypes.data, ocb.ctypes.data, ocr.ctypes.data), I3(wf, wf // 2, wf // 2), wf, hf, px, py, B,
                           file_bits, mine.ctypes.data_as(C.c_void_p))
            want_bytes += mine.tobytes()
        assert not rd.read(pic, px, py)  # end of file
        rd.close(), wr.close()
        assert open(str(tmp_path / f"out{file_bits}.yuv"), "rb").read() == want_bytes, ("pack", file_bits)
        pic.free()


@pytest.mark.parametrize("pic,n", [((136, 72), 70), ((416, 240), 5), ((64, 64), 1)])
def test_resident_pictures(ctx, pic, n):
    """Pictures resident in the working layout (hmx_tpool): planes -> pool -> planes is the identity (a ragged last
    group, pictures that cut the last CTU), and the all-intra chain on resident pools -- no layout conversion inside the
    call -- writes the oracle's levels and, exported, its reconstruction; decoder direction too."""
    B, L = ctx.bit_depth, capi.lib()
    w, h = pic
    qp = 31
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(900 + i, w, h, "mix") for i in range(min(n, 7))]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(950 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h, pad=(i % 2)).upload(o) for i, o in enumerate(orgs)]
    d_out = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    p_org, p_rec = capi.ResidentPool(ctx, w, h, n), capi.ResidentPool(ctx, w, h, n)
    p_org.import_planes(0, d_org)
    p_org.export_planes(0, d_out)
    ctx.sync()
    for i in range(n):
        got = d_out[i].download()
        assert all(np.array_equal(got[p], orgs[i][p]) for p in range(3)), ("round trip", i)
    d_lev = [capi.DevLevelsZ(ctx, w, h) for _ in range(n)]
    lev_arr = (capi.Levels * n)(*[d.as_pic() for d in d_lev])
    parr = (C.c_void_p * n)(*[plans[i % len(plans)].value for i in range(n)])
    ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, parr, 1, n, p_org.h_, p_rec.h_, lev_arr))
    p_rec.export_planes(0, d_out)
    ctx.sync()
    recs = []
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode(tus[i % len(tus)], w, h, B, qp, orgs[i])
        rec, lev = d_out[i].download(), d_lev[i].to_planes(tus[i % len(tus)])
        recs.append(rec)
        for p in range(3):
            assert np.array_equal(rec[p], rr[p]), ("recon", i, p)
            assert np.array_equal(lev[p], lr[p]), ("levels", i, p)
    p_dec = capi.ResidentPool(ctx, w, h, n)
    ctx._chk(L.hmx_frame_intra_decode_resident(ctx.h, parr, 1, n, p_dec.h_, lev_arr))
    for d in d_out:
        d.zero()
    p_dec.export_planes(0, d_out)
    ctx.sync()
    for i in range(n):
        got = d_out[i].download()
        assert all(np.array_equal(got[p], recs[i][p]) for p in range(3)), ("decode", i)
    # a pool of another size is refused, not mis-addressed
    p_bad = capi.ResidentPool(ctx, w + 64, h, n)
    assert L.hmx_frame_intra_decode_resident(ctx.h, parr, 1, n, p_bad.h_, lev_arr) != 0
    for x in (p_org, p_rec, p_dec, p_bad):
        x.free()
    for pl in plans:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_out + d_lev:
        d.free()


@pytest.mark.parametrize("slots4", ["16", "64", "64+16"])
def test_frame_intra_sse_output(ctx, slots4, hmx_opts):
    """hmx_set_sse_output: the whole-picture encode also writes xGetSSE(org, rec) of every block (getDistPart right
    behind the reconstruction, TEncSearch.cpp:1163) at the block's first 4x4 unit in partition order.  All four block
    sizes, both 4x4 wave shapes, pictures with their own plans; vs the oracle's getSSE on the oracle's reconstruction
    (which the GPU's equals).  Switching the output off restores the plain kernel."""
    if "+" in slots4:  # the four-lane 8x8 shape beside the lane-per-block 4x4 one
        hmx_opts(ctx, HMX_PACK_SLOTS4="64", HMX_PACK_SLOTS8="16")
    else:
        hmx_opts(ctx, HMX_PACK_SLOTS4=slots4)
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_getSSE.restype = C.c_uint32
    w, h, n, qp = 200, 136, 5, 30
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(1300 + i, w, h, "mix") for i in range(n)]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(1400 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevLevelsZ(ctx, w, h) for _ in range(n)]
    cw, ch = -(-w // 64), -(-h // 64)
    units = [cw * ch * 256, cw * ch * 64, cw * ch * 64]
    d_sse = [[ctx.alloc(4 * u).zero() for u in units] for _ in range(n)]
    sse_arr = (capi.Sse * n)()
    for i in range(n):
        for p in range(3):
            sse_arr[i].plane[p] = d_sse[i][p].ptr
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    ctx._chk(L.hmx_set_sse_output(ctx.h, sse_arr, n))
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    checked = 0
    for i in range(n):
        rr, _ = ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i])
        rec = d_rec[i].download()
        got = [d_sse[i][p].download(np.uint32) for p in range(3)]
        for p in range(3):
            assert np.array_equal(rec[p], rr[p])
        for t in tus[i]:
            N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = np.ascontiguousarray(orgs[i][p][y:y + N, x:x + N])
            r = np.ascontiguousarray(rr[p][y:y + N, x:x + N])
            want = O.hmo_getSSE(o.ctypes.data_as(C.c_void_p), N, r.ctypes.data_as(C.c_void_p), N, N, N, B)
            u = d_lev[i].block_offset(p, x, y) // 16
            assert int(got[p][u]) == want, ("sse", i, p, x, y, N)
            checked += 1
    assert checked == sum(len(t) for t in tus)
    ctx._chk(L.hmx_set_sse_output(ctx.h, None, 0))
    for b in d_sse[0]:
        b.zero()
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    assert not d_sse[0][0].download(np.uint32).any()
    for pl in plans:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_rec + d_lev + [b for row in d_sse for b in row]:
        d.free()


def test_inter_chain_sse(ctx):
    """hmx_batch_residual_transform_recon_sse_multi: the fused inter chain also returns getDistPart(rec, org, DF_SSE) per
    block (all four block sizes: lane-per-block 4x4 kernel, list kernels, one wave per 32x32 block), in the caller's
    block order, for every picture of the call; same levels and reconstruction as the call without it."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_getSSE.restype = C.c_uint32
    w, h, n = 192, 128, 2
    mx = (1 << B) - 1
    rng = np.random.default_rng(640 + B)
    tus = workload.make_tus(41, w, h, "mix", ts_prob=0.0)
    tus["flags"] = capi.TU_INTER
    tl = ctx.tu_list(tus)
    pp = capi.PicParam(w, h, 29, 0, capi.P_SLICE, 1)
    orgs = [workload.make_planes(80 + i, w, h, B, "texture") for i in range(n)]
    preds = [[np.clip(o.astype(np.int32) + rng.integers(-40, 41, o.shape), 0, mx).astype(np.int16) for o in orgs[i]] for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_pred = [capi.DevPicture(ctx, w, h).upload(p) for p in preds]
    outs = []
    for with_sse in (True, False):
        d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
        d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
        A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
        d_sse = ctx.alloc(4 * n * len(tus)).zero()
        if with_sse:
            ctx._chk(L.hmx_batch_residual_transform_recon_sse_multi(ctx.h, tl, n, A(d_org, capi.Pic), A(d_pred, capi.Pic), A(d_lev, capi.Levels),
                                                                    A(d_rec, capi.Pic), None, d_sse.ptr, C.byref(pp)))
        else:
            ctx._chk(L.hmx_batch_residual_transform_recon_multi(ctx.h, tl, n, A(d_org, capi.Pic), A(d_pred, capi.Pic), A(d_lev, capi.Levels),
                                                                A(d_rec, capi.Pic), None, C.byref(pp)))
        ctx.sync()
        outs.append(([d.download() for d in d_rec], [d.download() for d in d_lev], d_sse.download(np.uint32)))
        for d in d_rec + d_lev + [d_sse]:
            d.free()
    (rec, lev, sse), (rec0, lev0, _) = outs
    for i in range(n):
        for p in range(3):
            assert np.array_equal(rec[i][p], rec0[i][p]) and np.array_equal(lev[i][p], lev0[i][p]), (i, p)
        for k, t in enumerate(tus):
            N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = np.ascontiguousarray(orgs[i][p][y:y + N, x:x + N])
            r = np.ascontiguousarray(rec[i][p][y:y + N, x:x + N])
            assert int(sse[i * len(tus) + k]) == O.hmo_getSSE(o.ctypes.data_as(C.c_void_p), N, r.ctypes.data_as(C.c_void_p), N, N, N, B), (i, k, N)
    assert sse.any()
    L.hmx_tu_list_destroy(ctx.h, tl)
    for d in d_org + d_pred:
        d.free()


@pytest.mark.parametrize("use_had", [1, 0])
def test_batch_subpel_cost(ctx, use_had):
    """hmx_batch_subpel_cost, the encoder's sub-pel refinement fan-out (xExtDIFUpSamplingH / Q + xPatternRefinement): the
    distortion of every unit at every candidate displacement -- the reference's two search stages (nine half-sample,
    nine quarter-sample candidates) and the whole 7 x 7 neighbourhood -- vs the same made with the oracle's filters the
    way the reference makes its planes: filterHorLuma(isLast = false) over height + 7 rows, filterVerLuma(isFirst = false,
    isLast = true), then calcHAD / SAD.  All unit shapes (8x8 and 4x4 Hadamard sub-blocks), two references."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_calcHAD.restype = C.c_uint32
    rng = np.random.default_rng(9000 + B + use_had)
    w, h, m = 192, 128, 80
    refs_h = [workload.make_planes(70 + i, w, h, B, "texture") for i in range(2)]
    org_h = workload.make_planes(75, w, h, B, "texture")
    refs = [capi.DevPicture(ctx, w, h, m, m).upload(r) for r in refs_h]
    for r in refs:
        ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(r.as_pic()), w, h, m, m))
    org = capi.DevPicture(ctx, w, h).upload(org_h)
    ctx.sync()
    full = [r.download(with_margins=True)[0] for r in refs]  # luma with margins, as the device holds it
    pus = workload.make_pus(77, w, h, n_refs=2, bi_frac=0.0, mv_range=12)
    pus = pus[::3][:40].copy()
    pus["mv0x"] &= ~3
    pus["mv0y"] &= ~3
    half = [(0, 0), (0, -2), (0, 2), (-2, 0), (2, 0), (-2, -2), (2, -2), (-2, 2), (2, 2)]           # s_acMvRefineH x 2
    quarter = [(2 + dx, -2 + dy) for (dx, dy) in ((0, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (1, 1))]  # around one half-sample winner
    allpos = [(dx, dy) for dy in range(-3, 4) for dx in range(-3, 4)]
    ref_arr = (capi.Pic * 2)(*[r.as_pic() for r in refs])
    for cands in (half, quarter, allpos):
        offs = np.array(cands, np.int8)
        d_cost = ctx.alloc(4 * len(pus) * len(cands))
        ctx._chk(L.hmx_batch_subpel_cost(ctx.h, pus.ctypes.data, len(pus), ref_arr, 2, C.byref(org.as_pic()), offs.ctypes.data, len(cands), use_had, d_cost.ptr))
        got = d_cost.download(np.uint32).reshape(len(pus), len(cands))
        d_cost.free()
        st = w + 2 * m
        for i, u in enumerate(pus[:12] if cands is allpos else pus):
            x, y, pw, ph = int(u["x"]), int(u["y"]), int(u["w"]), int(u["h"])
            plane = full[int(u["ref0"])].reshape(-1)
            ob = np.ascontiguousarray(org_h[0][y:y + ph, x:x + pw])
            for k, (dx, dy) in enumerate(cands):
                mvx, mvy = int(u["mv0x"]) + dx, int(u["mv0y"]) + dy
                o0 = (m + y + (mvy >> 2) - 3) * st + m + x + (mvx >> 2)
                tmp = np.zeros((ph + 7, pw), np.int16)
                O.hmo_filterHorLuma(ol.ptr(plane, o0), st, ol.ptr(tmp.reshape(-1)), pw, pw, ph + 7, mvx & 3, 0, B)
                pred = np.zeros((ph, pw), np.int16)
                O.hmo_filterVerLuma(ol.ptr(tmp.reshape(-1), 3 * pw), pw, ol.ptr(pred.reshape(-1)), pw, pw, ph, mvy & 3, 0, 1, B)
                if use_had:
                    want = O.hmo_calcHAD(ob.ctypes.data_as(C.c_void_p), pw, pred.ctypes.data_as(C.c_void_p), pw, pw, ph, B)
                else:
                    want = int(np.abs(ob.astype(np.int32) - pred).sum()) >> (B - 8)
                assert int(got[i, k]) == want, ("subpel", i, (pw, ph), (dx, dy))
    for d in refs + [org]:
        d.free()


def test_yuv_resident(ctx):
    """hmx_yuv_unpack_resident / hmx_yuv_pack_resident: a file's frame straight into a pool picture and back, vs the
    plane-geometry entry points (which are held against the oracle's TVideoIOYuv restatement above)."""
    B, L = ctx.bit_depth, capi.lib()
    rng = np.random.default_rng(77 + B)
    for file_bits in (8, 10):
        w, h, px, py = 72, 40, 24, 8
        wf, hf = w + px, h + py
        wide = file_bits > 8
        pool = capi.ResidentPool(ctx, wf, hf, 3)
        pic, pic2 = capi.DevPicture(ctx, wf, hf), capi.DevPicture(ctx, wf, hf).zero()
        for k in range(3):
            vals = rng.integers(0, 1 << file_bits, w * h * 3 // 2)
            raw = np.frombuffer(vals.astype("<u2").tobytes() if wide else vals.astype(np.uint8).tobytes(), np.uint8)
            d_file = ctx.to_device(raw)
            p = pic.as_pic()
            ctx._chk(L.hmx_yuv_unpack(ctx.h, d_file.ptr, file_bits, C.byref(p), wf, hf, px, py))
            ctx._chk(L.hmx_yuv_unpack_resident(ctx.h, d_file.ptr, file_bits, pool.h_, k, px, py))
            pool.export_planes(k, [pic2])
            ctx.sync()
            a, b = pic.download(), pic2.download()
            assert all(np.array_equal(a[q], b[q]) for q in range(3)), ("unpack", file_bits, k)
            n_out = L.hmx_yuv_frame_bytes(w, h, file_bits)
            o1, o2 = ctx.alloc(n_out), ctx.alloc(n_out)
            ctx._chk(L.hmx_yuv_pack(ctx.h, C.byref(p), wf, hf, px, py, file_bits, o1.ptr))
            ctx._chk(L.hmx_yuv_pack_resident(ctx.h, pool.h_, k, px, py, file_bits, o2.ptr))
            assert np.array_equal(o1.download(np.uint8, n_out), o2.download(np.uint8, n_out)), ("pack", file_bits, k)
            assert np.array_equal(o1.download(np.uint8, n_out), raw) or file_bits != B  # same depth: the file comes back as it was
            for x in (d_file, o1, o2):
                x.free()
        pool.free(), pic.free(), pic2.free()


def test_scalar_motion_compensation_dropins(ctx):
    """hmx_xPredInterLumaBlk / hmx_xPredInterChromaBlk / hmx_motionCompensation (host pointers, one prediction unit per
    call: what TEncCu.cpp:1299 / TDecCu.cpp:452 reach through TComPrediction::motionCompensation) vs the oracle's
    restatement of TComPrediction.cpp:410-642: all 16 luma / 64 chroma phases, uni- and bi-prediction, AMP shapes."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    rng = np.random.default_rng(4100 + B)
    mx = (1 << B) - 1
    W, H, M = 160, 96, 16
    st, stc = W + 2 * M, W // 2 + M
    refs = []
    for _ in range(2):
        y = rng.integers(0, mx + 1, (H + 2 * M, st)).astype(np.int16)
        cb = rng.integers(0, mx + 1, (H // 2 + M, stc)).astype(np.int16)
        cr = rng.integers(0, mx + 1, (H // 2 + M, stc)).astype(np.int16)
        refs.append((y, cb, cr))

    def pic_of(planes):
        p = capi.Pic()
        for k, a in enumerate(planes):
            m = M if k == 0 else M // 2
            p.plane[k] = a.ctypes.data + 2 * (m * a.shape[1] + m)
            p.stride[k] = a.shape[1]
        return p

    shapes = [(8, 8), (16, 4), (4, 16), (32, 24), (64, 16), (12, 16), (16, 64), (8, 4)]
    n = 0
    for it in range(48):
        w, h = shapes[it % len(shapes)]
        x, y = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
        mv = [[int(rng.integers(-4 * (M - 8), 4 * (M - 8) + 1)) for _ in range(2)] for _ in range(2)]
        if it < 16:
            mv[0] = [(mv[0][0] & ~3) | (it & 3), (mv[0][1] & ~3) | (it >> 2)]  # every luma phase once
        use = [(1, 0), (0, 1), (1, 1)][it % 3]
        # the two block functions, list 0, both values of bi
        for bi in (0, 1):
            ry = refs[0][0]
            o = M * st + M + y * st + x
            want = np.zeros((h, w), np.int16)
            O.hmo_predInterLumaBlk(ol.ptr(ry.reshape(-1), o), st, mv[0][0], mv[0][1], w, h, want.reshape(-1), w, bi, B)
            got = np.zeros((h, w + 3), np.int16)
            ctx._chk(L.hmx_xPredInterLumaBlk(ctx.h, ry.ctypes.data + 2 * o, st, mv[0][0], mv[0][1], w, h, got.ctypes.data, w + 3, bi))
            assert np.array_equal(got[:, :w], want), ("luma", it, bi)
            rc = refs[0][1]
            oc = (M // 2) * stc + M // 2 + (y // 2) * stc + x // 2
            wantc = np.zeros((h // 2, w // 2), np.int16)
            O.hmo_predInterChromaBlk(ol.ptr(rc.reshape(-1), oc), stc, mv[0][0], mv[0][1], w, h, wantc.reshape(-1), w // 2, bi, B)
            gotc = np.zeros((h // 2, w // 2), np.int16)
            ctx._chk(L.hmx_xPredInterChromaBlk(ctx.h, rc.ctypes.data + 2 * oc, stc, mv[0][0], mv[0][1], w, h, gotc.ctypes.data, w // 2, bi))
            assert np.array_equal(gotc, wantc), ("chroma", it, bi)
        # the whole unit
        dst = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        dp = capi.Pic()
        for k in range(3):
            dp.plane[k], dp.stride[k] = dst[k].ctypes.data, dst[k].shape[1]
        r0, r1 = pic_of(refs[0]), pic_of(refs[1])
        m0, m1 = (C.c_int * 2)(*mv[0]), (C.c_int * 2)(*mv[1])
        ctx._chk(L.hmx_motionCompensation(ctx.h, C.byref(r0) if use[0] else None, m0 if use[0] else None,
                                          C.byref(r1) if use[1] else None, m1 if use[1] else None, x, y, w, h, C.byref(dp)))
        bi = int(use[0] and use[1])
        for k in range(3):
            ch = 1 if k else 0
            pw, ph, sk, mk = w >> ch, h >> ch, (stc if k else st), (M // 2 if k else M)
            parts = []
            for l in range(2):
                if not use[l]:
                    continue
                a = refs[l][k]
                o = mk * sk + mk + (y >> ch) * sk + (x >> ch)
                t = np.zeros((ph, pw), np.int16)
                (O.hmo_predInterChromaBlk if k else O.hmo_predInterLumaBlk)(ol.ptr(a.reshape(-1), o), sk, mv[l][0], mv[l][1], w, h,
                                                                            t.reshape(-1), pw, bi, B)
                parts.append(t)
            want = parts[0]
            if bi:
                want = np.zeros((ph, pw), np.int16)
                O.hmo_addAvg(parts[0].reshape(-1), pw, parts[1].reshape(-1), pw, want.reshape(-1), pw, pw, ph, B)
            assert np.array_equal(dst[k], want), ("motionCompensation", it, k, use)
        n += 1
    assert n == 48


def test_deblock_picture_vs_oracle(ctx):
    """The deblocking application over a 416x240 picture (not a multiple of the CTU size: the device path has no
    LCU structure), random boundary strengths, per-8x8 QPs, no-filter units, non-zero beta/tc offsets."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    rng = np.random.default_rng(818 + B)
    w, h = 416, 240
    mx = (1 << B) - 1
    uw, uh = w // 4, h // 4
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for (boff, toff, use_nof) in ((0, 0, 0), (2, -1, 1), (-4, 4, 1)):
        ramp = (np.arange(w)[None, :] // 8 + np.arange(h)[:, None] // 8) * (1 << (B - 8))
        y = np.clip(rng.integers(0, 30 << (B - 8), (h // 8, w // 8)).repeat(8, 0).repeat(8, 1) + rng.integers(-2, 3, (h, w)) + (60 << (B - 8)) + ramp, 0, mx).astype(np.int16)
        cb = np.clip(rng.integers(0, mx // 3, (h // 16, w // 16)).repeat(8, 0).repeat(8, 1) + rng.integers(0, 6, (h // 2, w // 2)), 0, mx).astype(np.int16)
        cr = np.clip(rng.integers(0, mx // 3, (h // 16, w // 16)).repeat(8, 0).repeat(8, 1) + rng.integers(0, 6, (h // 2, w // 2)), 0, mx).astype(np.int16)
        bsv, bsh = rng.integers(0, 3, (uh, uw)).astype(np.uint8), rng.integers(0, 3, (uh, uw)).astype(np.uint8)
        bsv[:, 0] = 0
        bsh[0, :] = 0
        qp = rng.integers(10, 52, (uh // 2, uw // 2)).repeat(2, 0).repeat(2, 1).astype(np.int8)
        nof = (rng.random((uh // 2, uw // 2)) < 0.15).repeat(2, 0).repeat(2, 1).astype(np.uint8)
        pic = capi.DevPicture(ctx, w, h).upload([y, cb, cr])
        d = [ctx.to_device(a) for a in (bsv, bsh, qp, nof)]
        p = pic.as_pic()
        ctx._chk(L.hmx_deblock_picture(ctx.h, C.byref(p), w, h, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr if use_nof else None, boff, toff))
        ctx.sync()
        got = pic.download()
        oy, ocb, ocr = y.copy(), cb.copy(), cr.copy()
        O.hmo_deblock_picture(P3(oy.ctypes.data, ocb.ctypes.data, ocr.ctypes.data), I3(w, w // 2, w // 2), w, h, B, vp(bsv), vp(bsh), vp(qp),
                              vp(nof) if use_nof else None, boff, toff)
        for a, b in zip(got, (oy, ocb, ocr)):
            assert np.array_equal(a, b), (boff, toff, use_nof)
        assert (oy != y).sum() > 1000 and (ocb != cb).sum() > 100
        pic.free()
        for x in d:
            x.free()


def test_deblock_strengths_vs_oracle(ctx):
    """Boundary strengths on the device vs the oracle (pinned against xGetBoundaryStrengthSingle), P and B slices,
    then strengths -> hmx_deblock_picture end to end."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    import test_oracle_vs_ref as T
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    w, h = 256, 192
    uw, uh = w // 4, h // 4
    for is_b in (0, 1):
        rng = np.random.default_rng(404 + is_b + B)
        units, ev, eh = T._dbk_units(rng, uw, uh, is_b)
        d_u, d_ev, d_eh = ctx.to_device(units), ctx.to_device(ev), ctx.to_device(eh)
        d_bv, d_bh = ctx.alloc(uw * uh), ctx.alloc(uw * uh)
        ctx._chk(L.hmx_deblock_strengths(ctx.h, d_u.ptr, d_ev.ptr, d_eh.ptr, w, h, is_b, d_bv.ptr, d_bh.ptr))
        ctx.sync()
        ov, oh = np.zeros((uh, uw), np.uint8), np.zeros((uh, uw), np.uint8)
        O.hmo_deblock_strengths(vp(units), vp(ev), vp(eh), w, h, 64, is_b, vp(ov), vp(oh))
        assert np.array_equal(d_bv.download(np.uint8).reshape(uh, uw), ov)
        assert np.array_equal(d_bh.download(np.uint8).reshape(uh, uw), oh)
        assert (ov == 1).sum() > 100 and (ov == 2).sum() > 100
        for d in (d_u, d_ev, d_eh, d_bv, d_bh):
            d.free()


def test_example_end_to_end(tmp_path):
    """examples/all_intra_reconstruct.py (file -> unpack -> intra chain -> strengths -> deblock -> SAO -> pack -> file)
    against the same chain of oracle functions, on a picture whose size is not a multiple of 8."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                                                                      "all_intra_reconstruct.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    O = ol.oracle()
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    w, h, B, qp, n_frames = 100, 52, 10, 30, 2
    rng = np.random.default_rng(12)
    frames = [np.clip(rng.integers(0, 200, w * h * 3 // 2) + 300, 0, 1023).astype("<u2") for _ in range(n_frames)]
    inp, outp = str(tmp_path / "in.yuv"), str(tmp_path / "out.yuv")
    open(inp, "wb").write(b"".join(f.tobytes() for f in frames))
    args = argparse.Namespace(input=inp, output=outp, width=w, height=h, file_bits=10, bit_depth=B, qp=qp, frames=0, seed=4)
    n, tus, ev, eh, sao = ex.run(args)
    assert n == n_frames
    got = np.frombuffer(open(outp, "rb").read(), np.uint8)
    pw, ph = 104, 56
    uw, uh = pw // 4, ph // 4
    want = b""
    for f in frames:
        raw = np.frombuffer(f.tobytes(), np.uint8)
        pl = [np.zeros((ph, pw), np.int16), np.zeros((ph // 2, pw // 2), np.int16), np.zeros((ph // 2, pw // 2), np.int16)]
        O.hmo_yuv_unpack(vp(raw), 10, B, pw, ph, pw - w, ph - h, P3(*[p.ctypes.data for p in pl]), I3(pw, pw // 2, pw // 2))
        rec, _ = ol.o_intra_frame_encode(tus, pw, ph, B, qp, pl)
        units = np.zeros(uw * uh, np.dtype([("intra", "u1"), ("cbf", "u1"), ("ref", "i1", 2), ("mv", "<i2", (2, 2))]))
        units["intra"] = 1
        bv, bh = np.zeros(uw * uh, np.uint8), np.zeros(uw * uh, np.uint8)
        O.hmo_deblock_strengths(vp(units), vp(ev), vp(eh), pw, ph, 64, 0, vp(bv), vp(bh))
        qpm = np.full(uw * uh, qp, np.int8)
        O.hmo_deblock_picture(P3(*[p.ctypes.data for p in rec]), I3(pw, pw // 2, pw // 2), pw, ph, B, vp(bv), vp(bh), vp(qpm), None, 0, 0)
        so = [np.zeros_like(p) for p in rec]
        prm = np.ascontiguousarray(sao)
        O.hmo_sao_picture(P3(*[p.ctypes.data for p in rec]), P3(*[p.ctypes.data for p in so]), I3(pw, pw // 2, pw // 2), pw, ph, B, 64,
                          P3(prm[0].ctypes.data, prm[1].ctypes.data, prm[2].ctypes.data))
        packed = np.zeros(w * h * 3, np.uint8)
        O.hmo_yuv_pack(P3(*[p.ctypes.data for p in so]), I3(pw, pw // 2, pw // 2), pw, ph, pw - w, ph - h, B, 10, vp(packed))
        want += packed.tobytes()
    assert got.tobytes() == want


def test_rdoq_lane_kernel_still_agrees(ctx, hmx_opts):
    """The block-list RDOQ runs 8x8 and larger blocks through the wave-cooperative routine (k_rdoq_tiles, hmx_rdoq_core.h) since round 2;
    round 1's one-lane-per-block kernel (all sizes; HMX_RDOQ_LANE) is kept as a cross-check of the same vectors."""
    hmx_opts(ctx, HMX_RDOQ_LANE="1")
    test_rdoq_batch_vs_oracle(ctx)


def test_cell_map_growth_keeps_rdoq_workspace(ctx):
    """Regression: growing the motion-compensation cell map once released the RDOQ workspace of the same context.
    RDOQ, then a mapped MC call on a picture larger than any before it (the map is re-allocated), then RDOQ again."""
    L, B = capi.lib(), ctx.bit_depth
    test_rdoq_batch_vs_oracle(ctx)
    w, h, m = 640, 384, 80
    d_ref = capi.DevPicture(ctx, w, h, m, m).upload(workload.make_planes(3, w, h, B))
    ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d_ref.as_pic()), w, h, m, m))
    pus = workload.make_pus(4, w, h, n_refs=1)
    d_pus = ctx.to_device(pus)
    d_a, d_b = capi.DevPicture(ctx, w, h).zero(), capi.DevPicture(ctx, w, h).zero()
    ref_arr = (capi.Pic * 1)(d_ref.as_pic())
    ctx._chk(L.hmx_batch_motionCompensation(ctx.h, d_pus.ptr, len(pus), ref_arr, 1, C.byref(d_a.as_pic())))  # wave per PU
    pic_b = d_b.as_pic()
    job = (capi.McJob * 1)()
    job[0].d_pus, job[0].n_pus, job[0].refs, job[0].n_refs = d_pus.ptr, len(pus), ref_arr, 1
    job[0].dst, job[0].pic_w, job[0].pic_h = C.pointer(pic_b), w, h
    ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, 1, job))  # cell map
    ctx.sync()
    for a, b in zip(d_a.download(), d_b.download()):
        assert np.array_equal(a, b)
    test_rdoq_batch_vs_oracle(ctx)
    for d in (d_ref, d_a, d_b):
        d.free()


@pytest.mark.parametrize("w,h,m", [(64, 48, 80), (72, 40, 16), (64, 48, 12), (36, 20, 6)])
def test_extend_border_shapes(ctx, w, h, m):
    """extendPicBorder (TComPicYuv.cpp:248-286) through both kernels: four samples per thread (widths and margins that
    are multiples of 8 luma samples) and the one-sample fallback."""
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    planes = workload.make_planes(5, w, h, B)
    d = capi.DevPicture(ctx, w, h, m, m).upload(planes)
    ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d.as_pic()), w, h, m, m))
    ctx.sync()
    full = d.download(with_margins=True)
    for p in range(3):
        pw, ph, pmx, pmy = d.dims[p]
        st = pw + 2 * pmx
        e = np.zeros((ph + 2 * pmy, st), np.int16)
        e[pmy:pmy + ph, pmx:pmx + pw] = planes[p]
        flat = e.reshape(-1)
        O.hmo_extendPicBorder(ol.ptr(flat, pmy * st + pmx), st, pw, ph, pmx, pmy)
        assert np.array_equal(full[p], e), ("border", p)
    d.free()


def test_inter_path_on_unaligned_planes(ctx):
    """The inter kernels read window and block rows with multi-dword accesses: motion compensation, the fused residual
    chain (lane per 4x4 block, list kernels, one wave per 32x32 block) and the border extension on planes with an ODD
    stride that start on an odd sample (2-byte-aligned addresses) must give what they give on aligned planes (which
    the other tests pin to the oracle)."""
    L, B = capi.lib(), ctx.bit_depth
    w, h, m = 192, 128, 80
    pus = workload.make_pus(21, w, h, n_refs=2, bi_frac=0.5)
    d_pus = ctx.to_device(pus)
    tus = workload.make_tus(22, w, h, "mix", ts_prob=0.0)
    tus["flags"] = capi.TU_INTER
    tl = ctx.tu_list(tus)
    pp = capi.PicParam(w, h, 30, 1, capi.B_SLICE, 1)
    planes = [workload.make_planes(60 + i, w, h, B, "texture") for i in range(3)]
    got = []
    for pad, skew in ((0, 0), (1, 1)):
        refs = [capi.DevPicture(ctx, w, h, m, m, pad=pad, skew=skew).upload(planes[i]) for i in range(2)]
        for r in refs:
            ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(r.as_pic()), w, h, m, m))
        org = capi.DevPicture(ctx, w, h, pad=pad, skew=skew).upload(planes[2])
        pred = capi.DevPicture(ctx, w, h, pad=pad, skew=skew).zero()
        rec = capi.DevPicture(ctx, w, h, m, m, pad=pad, skew=skew).zero()
        lev = capi.DevPicture(ctx, w, h, dtype=np.int32, pad=pad).zero()
        ref_arr = (capi.Pic * 2)(*[r.as_pic() for r in refs])
        a_pred, a_rec, a_org, a_lev = (capi.Pic * 1)(pred.as_pic()), (capi.Pic * 1)(rec.as_pic()), (capi.Pic * 1)(org.as_pic()), (capi.Levels * 1)(lev.as_pic())
        job = (capi.McJob * 1)()
        job[0].d_pus, job[0].n_pus, job[0].refs, job[0].n_refs = d_pus.ptr, len(pus), ref_arr, 2
        job[0].dst, job[0].pic_w, job[0].pic_h = C.pointer(a_pred[0]), w, h
        ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, 1, job))
        ctx._chk(L.hmx_batch_residual_transform_recon_multi(ctx.h, tl, 1, a_org, a_pred, a_lev, a_rec, None, C.byref(pp)))
        ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(rec.as_pic()), w, h, m, m))
        ctx.sync()
        got.append((refs[0].download(True), pred.download(), lev.download(), rec.download(True)))
        for d in refs + [org, pred, rec, lev]:
            d.free()
    names = ("reference with border", "prediction", "levels", "reconstruction with border")
    for k, name in enumerate(names):
        for p in range(3):
            assert np.array_equal(got[0][k][p], got[1][k][p]), (name, p)
    assert any(np.count_nonzero(a) for a in got[0][2])


@pytest.mark.parametrize("n_pics,schedule", [(1, "wave"), (5, "level"), (70, "level"), (1, "packed"), (5, "packed"), (70, "packed")])
def test_frame_intra_decode_onto(ctx, n_pics, schedule, hmx_opts):
    """hmx_frame_intra_decode_onto: the plan lists only SOME blocks (those of an inter picture's intra coding units);
    they are reconstructed onto what the pictures already hold, and everything else stays.  Both schedules, the
    across-pictures pool included (70 pictures), vs the oracle decoding the same blocks onto the same pictures."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE=schedule)
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    w, h, qp = 136, 72, 29
    tus_all = workload.make_tus(31, w, h, "mix")
    sh = (tus_all["plane"] != 0).astype(np.int64)
    region = ((tus_all["x"].astype(np.int64) << sh) // 16) * 7 + ((tus_all["y"].astype(np.int64) << sh) // 16) * 3
    tus = np.ascontiguousarray(tus_all[region % 3 != 0])  # two thirds of the 16x16 regions are "intra", the rest stays
    assert 0 < len(tus) < len(tus_all)
    rng = np.random.default_rng(5 + n_pics)
    held = [workload.make_planes(400 + i, w, h, B, "texture") for i in range(n_pics)]  # "inter reconstruction"
    orgs = [workload.make_planes(500 + i, w, h, B, "noise" if i % 2 else "texture") for i in range(n_pics)]
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    st = I3(w, w // 2, w // 2)
    cfg = ol.frame_cfg(w, h, B, qp)
    plan = ctx.intra_plan(tus, capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1))
    d_rec, d_lev, want = [], [], []
    for i in range(n_pics):
        # levels that make sense for these blocks: encode the blocks against an original, onto the held picture
        rec = [np.ascontiguousarray(a, np.int16).copy() for a in held[i]]
        lev = [np.zeros(a.shape, np.int32) for a in rec]
        t = np.ascontiguousarray(tus, ol.TU_DTYPE)
        oo = [np.ascontiguousarray(a, np.int16) for a in orgs[i]]
        O.hmo_intra_frame_encode(C.byref(cfg), t.ctypes.data, len(t), P3(*[a.ctypes.data for a in oo]), st,
                                 P3(*[a.ctypes.data for a in rec]), st, P3(*[a.ctypes.data for a in lev]))
        rec2 = [np.ascontiguousarray(a, np.int16).copy() for a in held[i]]
        O.hmo_intra_frame_decode(C.byref(cfg), t.ctypes.data, len(t), P3(*[a.ctypes.data for a in rec2]), st, P3(*[a.ctypes.data for a in lev]))
        assert all(np.array_equal(a, b) for a, b in zip(rec, rec2))
        want.append(rec2)
        d_rec.append(capi.DevPicture(ctx, w, h).upload(held[i]))
        d_lev.append(capi.DevPicture(ctx, w, h, dtype=np.int32).upload(lev))
    A = lambda lst, T: (T * n_pics)(*[x.as_pic() for x in lst])
    ctx._chk(L.hmx_frame_intra_decode_onto(ctx.h, plan, n_pics, A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    untouched = np.ones((h, w), bool)
    for t in tus:
        if t["plane"] == 0:
            n = 1 << int(t["log2n"])
            untouched[int(t["y"]):int(t["y"]) + n, int(t["x"]):int(t["x"]) + n] = False
    assert untouched.any()
    for i in range(n_pics):
        got = d_rec[i].download()
        for p in range(3):
            assert np.array_equal(got[p], want[i][p]), ("onto", i, p)
        assert np.array_equal(got[0][untouched], np.asarray(held[i][0])[untouched])
    # encoder side: the same blocks from the originals onto the held pictures -> the oracle's levels and pictures
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec2 = [capi.DevPicture(ctx, w, h).upload(held[i]) for i in range(n_pics)]
    d_lev2 = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n_pics)]
    ctx._chk(L.hmx_frame_intra_encode_onto(ctx.h, plan, n_pics, A(d_org, capi.Pic), A(d_rec2, capi.Pic), A(d_lev2, capi.Levels)))
    ctx.sync()
    for i in range(n_pics):
        got, lv, lv_want = d_rec2[i].download(), d_lev2[i].download(), d_lev[i].download()
        for p in range(3):
            assert np.array_equal(got[p], want[i][p]), ("encode onto", i, p)
            assert np.array_equal(lv[p], lv_want[p]), ("encode onto levels", i, p)
    L.hmx_intra_plan_destroy(ctx.h, plan)
    for d in d_rec + d_lev + d_org + d_rec2 + d_lev2:
        d.free()


def test_set_rdoq_keeps_its_state_when_it_refuses(ctx):
    """hmx_set_rdoq checks every input before it touches the context (round-2 advisor finding): a good set, then a call with a
    non-positive multiplier is REFUSED and the good set stays in force -- same levels as before the refused call, also when the encode
    calls are queued back to back (the multiplier table goes up once, not per call).  And the chain refuses a QP at which RDOQ's
    16-bit levels could overflow."""
    B, L = ctx.bit_depth, capi.lib()
    w, h, n, qp = 136, 72, 2, 30
    rng = np.random.default_rng(5)
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = [workload.with_cbf_ctx(workload.make_tus(3300 + i, w, h, "mix")) for i in range(n)]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(3400 + i, w, h, B, "texture") for i in range(n)]
    ests = [[ol.make_est_bits(rng) for _ in range(8)] for _ in range(n)]
    lams = [(40.0 + i, 30.0 + i) for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    ctx.set_rdoq([(ests[i], lams[i][0], lams[i][1]) for i in range(n)])
    with pytest.raises(capi.HmxError):
        ctx.set_rdoq([(ests[0], 10.0, 0.0)])  # one picture, chroma multiplier 0: refused
    for rep in range(3):  # queued back to back
        ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        rr, ll = ol.o_intra_frame_encode_rdoq(tus[i], w, h, B, qp, orgs[i], ests[i], lams[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        assert all(np.array_equal(lev[p], ll[p]) and np.array_equal(rec[p], rr[p]) for p in range(3)), i
    if B == 10:  # QP -12 at 10 bit (the lowest the bit depth allows: per = 0): a 32x32 level can reach ~52000
        pp0 = capi.PicParam(w, h, -12, 0, capi.I_SLICE, 1)
        big = workload.with_cbf_ctx(workload.make_tus(1, w, h, 32))
        pl0 = ctx.intra_plan(big, pp0)
        ctx.set_rdoq([(ests[0], 40.0, 30.0)])  # one set for every picture
        rc = L.hmx_frame_intra_encode(ctx.h, pl0, 1, C.byref(d_org[0].as_pic()), C.byref(d_rec[0].as_pic()), C.byref(d_lev[0].as_pic()))
        assert rc == -1 and b"16 bits" in L.hmx_last_error(ctx.h)
        L.hmx_intra_plan_destroy(ctx.h, pl0)
    ctx.set_rdoq(None)
    for pl in plans:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_rec + d_lev:
        d.free()


@pytest.mark.parametrize("shared", [False, True])
def test_frame_intra_rdoq_in_chain(ctx, shared):
    """hmx_set_rdoq: xRateDistOptQuant as the quantiser INSIDE the whole-picture chain (what TEncSearch::xIntraCodingLumaBlk /
    ChromaBlk run with RDOQ on, TComTrQuant.cpp:1121-1122): the levels of a block decide its reconstruction, which later
    blocks predict from.  All four block sizes, transform-skip blocks (flat quantiser), own plans, per-picture tables and
    multipliers (or one set for all), cbf contexts in the flags; levels and reconstruction vs the oracle's chain."""
    B, L = ctx.bit_depth, capi.lib()
    w, h, n, qp = 200, 136, 4, 27
    rng = np.random.default_rng(77)
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = []
    for i in range(n):
        t = workload.make_tus(2300 + i, w, h, "mix")
        depth = rng.integers(0, 3, len(t))
        t["flags"] = (t["flags"] & 1) | (np.where(t["plane"] == 0, np.minimum(depth, 1), 5 + depth).astype(np.uint8) << 4)
        tus.append(t)
    assert any((t["flags"] & 1).any() for t in tus) and all(len({int(x) for x in t["log2n"]}) == 4 for t in tus)
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(2400 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n)]
    n_sets = 1 if shared else n
    ests = [[ol.make_est_bits(rng) for _ in range(8)] for _ in range(n_sets)]
    lams = [(float(rng.uniform(20, 120)), float(rng.uniform(15, 90))) for _ in range(n_sets)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    ctx.set_rdoq([(ests[i], lams[i][0], lams[i][1]) for i in range(n_sets)])
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    flat_differs = 0
    for i in range(n):
        k = 0 if shared else i
        rr, ll = ol.o_intra_frame_encode_rdoq(tus[i], w, h, B, qp, orgs[i], ests[k], lams[k])
        _, lf = ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        for p in range(3):
            assert np.array_equal(lev[p], ll[p]), ("levels", i, p, np.argwhere(lev[p] != ll[p])[:4])
            assert np.array_equal(rec[p], rr[p]), ("rec", i, p)
            flat_differs += int((ll[p] != lf[p]).sum())
    assert flat_differs > 100  # RDOQ did take other decisions than the flat quantiser
    ctx.set_rdoq(None)  # off again: the flat quantiser's result
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    _, lf = ol.o_intra_frame_encode(tus[0], w, h, B, qp, orgs[0])
    assert all(np.array_equal(d_lev[0].download()[p], lf[p]) for p in range(3))
    for pl in plans:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_rec + d_lev:
        d.free()


def test_frame_intra_rdoq_resident_with_sse(ctx, hmx_opts):
    """RDOQ inside the chain together with the distortion output (the kernel variant of its own), on pictures RESIDENT in the
    working layout, packing groups of two pictures with a ragged last group (the tables of a group wait in LDS: slot =
    picture within the group), per-picture tables and multipliers; levels, reconstruction and xGetSSE vs the oracle.  A
    packing group the tables do not fit is refused."""
    hmx_opts(ctx, HMX_PACK_GROUP="2")
    O, B, L = ol.oracle(), ctx.bit_depth, capi.lib()
    O.hmo_getSSE.restype = C.c_uint32
    w, h, n, qp = 320, 192, 5, 30
    rng = np.random.default_rng(99)
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = [workload.with_cbf_ctx(workload.make_tus(3300 + i, w, h, "mix")) for i in range(n)]
    plans = [ctx.intra_plan(t, pp) for t in tus]
    orgs = [workload.make_planes(3400 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n)]
    ests = [[workload.make_est_bits(4000 + 8 * i + k) for k in range(8)] for i in range(n)]
    lams = [(float(rng.uniform(20, 120)), float(rng.uniform(15, 90))) for _ in range(n)]
    ctx.set_rdoq([(ests[i], lams[i][0], lams[i][1]) for i in range(n)])  # before the pools: it bounds the packing group
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_out = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    p_org, p_rec = capi.ResidentPool(ctx, w, h, n), capi.ResidentPool(ctx, w, h, n)
    p_org.import_planes(0, d_org)
    d_lev = [capi.DevLevelsZ(ctx, w, h) for _ in range(n)]
    lev_arr = (capi.Levels * n)(*[d.as_pic() for d in d_lev])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    cw, ch = -(-w // 64), -(-h // 64)
    units = [cw * ch * 256, cw * ch * 64, cw * ch * 64]
    d_sse = [[ctx.alloc(4 * u).zero() for u in units] for _ in range(n)]
    sse_arr = (capi.Sse * n)()
    for i in range(n):
        for p in range(3):
            sse_arr[i].plane[p] = d_sse[i][p].ptr
    ctx._chk(L.hmx_set_sse_output(ctx.h, sse_arr, n))
    ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, parr, 1, n, p_org.h_, p_rec.h_, lev_arr))
    p_rec.export_planes(0, d_out)
    ctx.sync()
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode_rdoq(tus[i], w, h, B, qp, orgs[i], ests[i], lams[i])
        rec, lev = d_out[i].download(), d_lev[i].to_planes(tus[i])
        got = [d_sse[i][p].download(np.uint32) for p in range(3)]
        for p in range(3):
            assert np.array_equal(lev[p], lr[p]), ("levels", i, p)
            assert np.array_equal(rec[p], rr[p]), ("recon", i, p)
        for t in tus[i][:: 7]:
            N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = np.ascontiguousarray(orgs[i][p][y:y + N, x:x + N])
            r = np.ascontiguousarray(rr[p][y:y + N, x:x + N])
            want = O.hmo_getSSE(o.ctypes.data_as(C.c_void_p), N, r.ctypes.data_as(C.c_void_p), N, N, N, B)
            assert int(got[p][d_lev[i].block_offset(p, x, y) // 16]) == want, ("sse", i, p, x, y, N)
    ctx._chk(L.hmx_set_sse_output(ctx.h, None, 0))
    # groups of four: the tables of a group would not fit
    hmx_opts(ctx, HMX_PACK_GROUP="4")
    p4_org, p4_rec = capi.ResidentPool(ctx, w, h, n), capi.ResidentPool(ctx, w, h, n)
    assert L.hmx_frame_intra_encode_resident(ctx.h, parr, 1, n, p4_org.h_, p4_rec.h_, lev_arr) != 0
    ctx.set_rdoq(None)
    for x in (p_org, p_rec, p4_org, p4_rec):
        x.free()
    for pl in plans:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_out + d_lev + [b for row in d_sse for b in row]:
        d.free()


def test_intra_plan_create_multi(ctx):
    """hmx_intra_plan_create_multi: the host-side analysis of several pictures on host threads gives the plans of the one-picture
    call (same schedule sizes level by level) and the same results through the packed path; a bad picture leaves no plan."""
    B, L = ctx.bit_depth, capi.lib()
    w, h, n, qp = 200, 136, 9, 29
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(5100 + i, w, h, "mix") for i in range(n)]
    single = [ctx.intra_plan(t, pp) for t in tus]
    multi = ctx.intra_plans(tus, pp)
    for a, b in zip(single, multi):
        ia, ib = [(C.c_int(), C.c_int(), C.c_int()) for _ in range(2)]
        L.hmx_intra_plan_info(a, *[C.byref(x) for x in ia])
        L.hmx_intra_plan_info(b, *[C.byref(x) for x in ib])
        assert [x.value for x in ia] == [x.value for x in ib]
        for lvl in range(ia[1].value):
            ca, cb, wa, wb = (C.c_uint32 * 4)(), (C.c_uint32 * 4)(), C.c_uint32(), C.c_uint32()
            L.hmx_intra_plan_level(a, lvl, ca, C.byref(wa))
            L.hmx_intra_plan_level(b, lvl, cb, C.byref(wb))
            assert list(ca) == list(cb) and wa.value == wb.value
    orgs = [workload.make_planes(5200 + i, w, h, B, "texture") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in multi])
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        for p in range(3):
            assert np.array_equal(rec[p], rr[p]) and np.array_equal(lev[p], lr[p]), (i, p)
    bad = [t.copy() for t in tus[:3]]
    bad[1]["x"][5] = 4000  # outside the picture
    arrs = [np.ascontiguousarray(t, capi.TU_DTYPE) for t in bad]
    ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs])
    cnts = (C.c_int * 3)(*[len(a) for a in arrs])
    out = (C.c_void_p * 3)()
    assert L.hmx_intra_plan_create_multi(ctx.h, ptrs, cnts, 3, C.byref(pp), out) != 0
    assert all(not out[i] for i in range(3))
    for pl in single + multi:
        L.hmx_intra_plan_destroy(ctx.h, pl)
    for d in d_org + d_rec + d_lev:
        d.free()

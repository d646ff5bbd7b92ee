"""Golden vectors captured from the compiled reference (tests/golden/make_golden.py).
CPU part: the oracle reproduces them.  GPU part (-m gpu): libhmx reproduces them through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REG_DCT = 65535


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------ CPU: oracle vs golden
@pytest.mark.parametrize("B", [8, 10])
def test_oracle_transforms(B):
    g, O = load(f"transforms_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, mode in enumerate(g[f"tr{N}_mode"]):
            blk = np.ascontiguousarray(g[f"tr{N}_in"][k])
            out = np.zeros(N * N, np.int16)
            O.hmo_xTrMxN(blk, out, N, int(mode), B)
            assert np.array_equal(out, g[f"tr{N}_fwd"][k])
            O.hmo_xITrMxN(blk, out, N, int(mode), B)
            assert np.array_equal(out, g[f"tr{N}_inv"][k])


def _quant_case(O, B, N, par):
    qpy, st, ttype, is_intra, mode, ts = (int(v) for v in par)
    q = O.hmo_setQPforQuant(qpy, int(ttype != 0), 6 * (B - 8), 0)
    scan = O.hmo_coef_scan_idx(N, int(ttype == 0), is_intra, mode)
    cfg = ol.quant_cfg(q.per, q.rem, intra_slice=int(st == 2), sign_hide=1, scan_idx=scan)
    tmode = mode if (ttype == 0 and is_intra) else REG_DCT
    return q, cfg, tmode, ts


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_quant(B):
    g, O = load(f"quant_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"q{N}_par"]):
            q, cfg, tmode, ts = _quant_case(O, B, N, par)
            lev, s = ol.o_transformNxN(g[f"q{N}_resi"][k], N, B, tmode, ts, cfg)
            assert np.array_equal(lev.reshape(-1), g[f"q{N}_lev"][k]) and s == g[f"q{N}_sum"][k]
            r = ol.o_invtransformNxN(g[f"q{N}_lev"][k], N, B, tmode, q.per, q.rem, ts)
            assert np.array_equal(r.reshape(-1), g[f"q{N}_inv"][k])


def _rdoq_case(O, B, N, par, lam, est_words):
    qpy, st, ttype, is_intra, mode, tr_idx, cbf_ctx = (int(v) for v in par)
    q = O.hmo_setQPforQuant(qpy, int(ttype != 0), 6 * (B - 8), 0)
    scan = O.hmo_coef_scan_idx(N, int(ttype == 0), is_intra, mode)
    root = int((not is_intra) and ttype == 0 and tr_idx == 0)
    cfg = ol.RdoqCfg(q.per, q.rem, int(ttype == 0), is_intra, scan, root, cbf_ctx, 1, float(lam))
    est = ol.EstBits.from_buffer_copy(np.ascontiguousarray(est_words, np.int32).tobytes())
    return cfg, est


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_rdoq(B):
    g, O = load(f"rdoq_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"r{N}_par"]):
            cfg, est = _rdoq_case(O, B, N, par, g[f"r{N}_lambda"][k], g[f"r{N}_est"][k])
            lev, s = ol.o_rdoq(g[f"r{N}_coef"][k], N, B, cfg, est)
            assert np.array_equal(lev.reshape(-1), g[f"r{N}_lev"][k]) and s == g[f"r{N}_sum"][k], (N, k)


def _quant_scaled_case(O, B, N, par):
    qpy, st, ttype, is_intra, mode, tr_idx, cbf_ctx, rdoq = (int(v) for v in par)
    q = O.hmo_setQPforQuant(qpy, int(ttype != 0), 6 * (B - 8), 0)
    scan = O.hmo_coef_scan_idx(N, int(ttype == 0), is_intra, mode)
    fc = ol.quant_cfg(q.per, q.rem, intra_slice=int(st == 2), sign_hide=1, scan_idx=scan)
    root = int((not is_intra) and ttype == 0 and tr_idx == 0)
    return fc, (q, scan, root, cbf_ctx), rdoq


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_quant_scaled(B):
    """The quantisers under a scaling list (tables in): the oracle against the reference's levels, uiAcSum and pArlDes."""
    g, O = load(f"quant_scaled_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"q{N}_par"]):
            fc, (q, scan, root, cbf_ctx), rdoq = _quant_scaled_case(O, B, N, par)
            coef, qtab, estab = g[f"q{N}_coef"][k], g[f"q{N}_qtab"][k], g[f"q{N}_estab"][k]
            assert np.array_equal(ol.o_arl(coef, N, B, fc, rdoq, qtab).reshape(-1), g[f"q{N}_arl"][k]), (N, k, "arl")
            if rdoq:
                cfg = ol.RdoqCfg(q.per, q.rem, int(par[2] == 0), int(par[3]), scan, root, cbf_ctx, 1, float(g[f"q{N}_lambda"][k]))
                est = ol.EstBits.from_buffer_copy(np.ascontiguousarray(g[f"q{N}_est"][k], np.int32).tobytes())
                lev, s = ol.o_rdoq_scaled(coef, N, B, cfg, est, qtab, estab)
            else:
                lev, s = ol.o_quant_scaled(coef, N, B, fc, qtab)
            assert np.array_equal(lev.reshape(-1), g[f"q{N}_lev"][k]) and s == g[f"q{N}_sum"][k], (N, k, rdoq)


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_dequant_scaled(B):
    g, O = load(f"dequant_scaled_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, (qpy, _lt) in enumerate(g[f"s{N}_par"]):
            q = O.hmo_setQPforQuant(int(qpy), 0, 6 * (B - 8), 0)
            o = np.zeros(N * N, np.int32)
            O.hmo_xDeQuant_scaled(np.ascontiguousarray(g[f"s{N}_lev"][k]), o, N, B, q.per, np.ascontiguousarray(g[f"s{N}_tab"][k]))
            assert np.array_equal(o, g[f"s{N}_out"][k]), (N, k)


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_arl(B):
    """pArlDes (AdaptiveQpSelection) of the reference's xQuant / xRateDistOptQuant, and the flat branch's levels when the slice's
    base QP is not the block's."""
    g, O = load(f"arl_b{B}.npz"), ol.oracle()
    bd = 6 * (B - 8)
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"a{N}_par"]):
            qpy, qp_base, st, ttype, is_intra, mode, rdoq = (int(v) for v in par)
            q, qb = O.hmo_setQPforQuant(qpy, int(ttype != 0), bd, 0), O.hmo_setQPforQuant(qp_base, int(ttype != 0), bd, 0)
            scan = O.hmo_coef_scan_idx(N, int(ttype == 0), is_intra, mode)
            fc = ol.quant_cfg(q.per, q.rem, intra_slice=int(st == 2), sign_hide=1, scan_idx=scan, per_qbits=qb.per)
            assert np.array_equal(ol.o_arl(g[f"a{N}_coef"][k], N, B, fc, rdoq).reshape(-1), g[f"a{N}_arl"][k]), (N, k)
            if not rdoq:
                lev, s = np.zeros(N * N, np.int32), C.c_uint32(0)
                O.hmo_xQuant(np.ascontiguousarray(g[f"a{N}_coef"][k]), lev, N, B, C.byref(fc), C.byref(s))
                assert np.array_equal(lev, g[f"a{N}_lev"][k]) and s.value == g[f"a{N}_sum"][k], (N, k)


def _deblock_case(g, k):
    par = g[f"d{k}_par"]
    ins = [np.ascontiguousarray(g[f"d{k}_{n}"]) for n in ("y", "cb", "cr", "bsv", "bsh", "qp", "nof")]
    outs = [g[f"d{k}_{n}"] for n in ("oy", "ocb", "ocr")]
    return int(par[0]), int(par[1]), int(par[2]), ins, outs


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_deblock(B):
    g, O = load(f"deblock_b{B}.npz"), ol.oracle()
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for k in range(2):
        boff, toff, use_nof, (y, cb, cr, bsv, bsh, qp, nof), outs = _deblock_case(g, k)
        h, w = y.shape
        y, cb, cr = y.copy(), cb.copy(), cr.copy()
        O.hmo_deblock_picture(P3(y.ctypes.data, cb.ctypes.data, cr.ctypes.data), I3(w, w // 2, w // 2), w, h, B, vp(bsv), vp(bsh), vp(qp),
                              vp(nof) if use_nof else None, boff, toff)
        for a, b in zip((y, cb, cr), outs):
            assert np.array_equal(a, b), k


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_sao(B):
    g, O = load(f"sao_b{B}.npz"), ol.oracle()
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    y, cb, cr = (np.ascontiguousarray(g[k]) for k in ("y", "cb", "cr"))
    prm = np.ascontiguousarray(g["prm"])
    h, w = y.shape
    oy, ocb, ocr = np.zeros_like(y), np.zeros_like(cb), np.zeros_like(cr)
    O.hmo_sao_picture(P3(y.ctypes.data, cb.ctypes.data, cr.ctypes.data), P3(oy.ctypes.data, ocb.ctypes.data, ocr.ctypes.data),
                      I3(w, w // 2, w // 2), w, h, B, 64, P3(prm[0].ctypes.data, prm[1].ctypes.data, prm[2].ctypes.data))
    assert np.array_equal(oy, g["oy"]) and np.array_equal(ocb, g["ocb"]) and np.array_equal(ocr, g["ocr"])


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_intra(B):
    g, O = load(f"intra_b{B}.npz"), ol.oracle()
    w, h = g["pic_y"].shape[1], g["pic_y"].shape[0]
    flags = np.zeros(65, np.uint8)
    planes = [np.ascontiguousarray(g[k]).reshape(-1) for k in ("pic_y", "pic_cb", "pic_cr")]
    for N in (4, 8, 16, 32):
        W = 2 * N + 1
        for k, (bx, by) in enumerate(g[f"il{N}_pos"]):
            adi = np.zeros(2 * W * W, np.int32)
            nav = O.hmo_intra_avail(int(bx), int(by), N, w, h, 64, flags)
            O.hmo_fillReferenceSamples(ol.ptr(planes[0], int(by) * w + int(bx)), w, flags, nav, 4, N, B, adi)
            O.hmo_filterAdi(adi, N)
            assert np.array_equal(adi, g[f"il{N}_adi"][k])
            for m in range(35):
                p = np.zeros(N * N, np.int16)
                O.hmo_predIntraLumaAng(adi, m, p, N, N, B)
                assert np.array_equal(p, g[f"il{N}_pred"][k][m])
    for Nc in (4, 8, 16):
        W = 2 * Nc + 1
        for k, (bx, by) in enumerate(g[f"ic{Nc}_pos"]):
            nav = O.hmo_intra_avail(int(bx), int(by), 2 * Nc, w, h, 64, flags)
            for c, pl in ((0, planes[1]), (1, planes[2])):
                adi = np.zeros(2 * W * W, np.int32)
                O.hmo_fillReferenceSamples(ol.ptr(pl, (int(by) // 2) * (w // 2) + int(bx) // 2), w // 2, flags, nav, 2, Nc, B, adi)
                assert np.array_equal(adi[:W * W], g[f"ic{Nc}_adi"][k][c * W * W:(c + 1) * W * W])
                if c == 0:
                    for m in range(35):
                        p = np.zeros(Nc * Nc, np.int16)
                        O.hmo_predIntraChromaAng(adi, m, p, Nc, Nc, B)
                        assert np.array_equal(p, g[f"ic{Nc}_pred"][k][m])


def _ext_planes(g, O):
    planes = []
    for key, m in (("pic_y", 80), ("pic_cb", 40), ("pic_cr", 40)):
        pl = g[key]
        ph, pw = pl.shape
        st = pw + 2 * m
        e = np.zeros((ph + 2 * m, st), np.int16)
        e[m:m + ph, m:m + pw] = pl
        flat = e.reshape(-1)
        O.hmo_extendPicBorder(ol.ptr(flat, m * st + m), st, pw, ph, m, m)
        planes.append((flat, st, m))
    return planes


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_inter(B):
    g, O = load(f"inter_b{B}.npz"), ol.oracle()
    planes = _ext_planes(g, O)
    assert np.array_equal(planes[0][0].reshape(g["ext_y"].shape), g["ext_y"])
    h, w = g["pic_y"].shape
    for k, (px, py, pw, ph, mvx, mvy, cmx, cmy, bi) in enumerate(g["pu"].tolist()):
        cx, cy = C.c_int(mvx), C.c_int(mvy)
        O.hmo_clipMv(C.byref(cx), C.byref(cy), px, py, w, h, 64)
        assert (cx.value, cy.value) == (cmx, cmy)
        e, st, m = planes[0]
        o = np.zeros(pw * ph, np.int16)
        O.hmo_predInterLumaBlk(ol.ptr(e, (m + py) * st + m + px), st, cmx, cmy, pw, ph, o, pw, bi, B)
        assert np.array_equal(o, g["pu_y"][k][:pw * ph])
        for c in (1, 2):
            e, st, m = planes[c]
            o = np.zeros(pw * ph // 4, np.int16)
            O.hmo_predInterChromaBlk(ol.ptr(e, (m + py // 2) * st + m + px // 2), st, cmx, cmy, pw, ph, o, pw // 2, bi, B)
            assert np.array_equal(o, g["pu_c"][k][(c - 1) * 1024:(c - 1) * 1024 + pw * ph // 4])
    o = np.zeros(256, np.int16)
    O.hmo_addAvg(np.ascontiguousarray(g["avg_a"]), 16, np.ascontiguousarray(g["avg_b"]), 16, o, 16, 16, 16, B)
    assert np.array_equal(o, g["avg_o"])


FRAMES = [("frame_416x240_mix_b8.npz", 8), ("frame_200x136_mix_b10.npz", 10)]


@pytest.mark.parametrize("name,B", FRAMES)
def test_oracle_frame(name, B):
    g = load(name)
    org = [np.ascontiguousarray(g[k]) for k in ("org_y", "org_cb", "org_cr")]
    h, w = org[0].shape
    rec, lev = ol.o_intra_frame_encode(g["tus"], w, h, B, int(g["qp"]), org)
    for p, k in enumerate(("y", "cb", "cr")):
        assert np.array_equal(rec[p], g["rec_" + k])
        assert np.array_equal(lev[p], g["lev_" + k].astype(np.int32))


# ------------------------------------------------------------------ GPU: libhmx vs golden
@pytest.fixture(scope="module", params=[8, 10])
def gctx(request):
    from thevc_amd import capi
    c = capi.Context(bit_depth=request.param)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_transforms_quant(gctx):
    from thevc_amd import capi
    B = gctx.bit_depth
    g = load(f"transforms_b{B}.npz")
    for N in (4, 8, 16, 32):
        for k, mode in enumerate(g[f"tr{N}_mode"]):
            blk = np.ascontiguousarray(g[f"tr{N}_in"][k])
            assert np.array_equal(gctx.xT(int(mode), blk, N, N), g[f"tr{N}_fwd"][k].astype(np.int32))
            assert np.array_equal(gctx.xIT(int(mode), blk.astype(np.int32), N, N), g[f"tr{N}_inv"][k])
    g = load(f"quant_b{B}.npz")
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"q{N}_par"]):
            qpy, st, ttype, is_intra, mode, ts = (int(v) for v in par)
            qp = capi.qp_for(qpy, ttype, B)
            qpar = capi.QuantParam(qp, -1, st, 1, is_intra, mode)
            lev, s = gctx.transformNxN(g[f"q{N}_resi"][k], N, N, ttype, qpar, ts)
            assert np.array_equal(lev, g[f"q{N}_lev"][k]) and s == g[f"q{N}_sum"][k]
            tmode = mode if (ttype == 0 and is_intra) else REG_DCT
            r = gctx.invtransformNxN(g[f"q{N}_lev"][k], N, N, ttype, tmode, qp, ts)
            assert np.array_equal(r, g[f"q{N}_inv"][k])


@pytest.mark.parametrize("B", [8, 10])
def test_oracle_intra64(B):
    """64x64 luma prediction units (TEncSearch.cpp:2509-2540 runs initAdiPattern and the 35 modes at the PU size)."""
    g, O = load(f"intra64_b{B}.npz"), ol.oracle()
    h, w = g["pic_y"].shape
    flags = np.zeros(65, np.uint8)
    y = np.ascontiguousarray(g["pic_y"]).reshape(-1)
    N, W = 64, 129
    for k, (bx, by) in enumerate(g["il64_pos"]):
        adi = np.zeros(2 * W * W, np.int32)
        nav = O.hmo_intra_avail(int(bx), int(by), N, w, h, 64, flags)
        O.hmo_fillReferenceSamples(ol.ptr(y, int(by) * w + int(bx)), w, flags, nav, 4, N, B, adi)
        O.hmo_filterAdi(adi, N)
        assert np.array_equal(adi, g["il64_adi"][k])
        for m in range(35):
            p = np.zeros(N * N, np.int16)
            O.hmo_predIntraLumaAng(adi, m, p, N, N, B)
            assert np.array_equal(p, g["il64_pred"][k][m]), (k, m)


@pytest.mark.gpu
def test_gpu_intra64(gctx):
    B = gctx.bit_depth
    g = load(f"intra64_b{B}.npz")
    h, w = g["pic_y"].shape
    y = np.ascontiguousarray(g["pic_y"]).reshape(-1)
    for k, (bx, by) in enumerate(g["il64_pos"]):
        adi = gctx.initAdiPattern(y, w, int(bx), int(by), 64, 0, w, h)
        assert np.array_equal(adi, g["il64_adi"][k]), (k, "initAdiPattern")
        for m in range(35):
            assert np.array_equal(gctx.predIntraLumaAng(adi, m, 64, 64), g["il64_pred"][k][m]), (k, m)


@pytest.mark.gpu
def test_gpu_intra(gctx):
    B = gctx.bit_depth
    g = load(f"intra_b{B}.npz")
    h, w = g["pic_y"].shape
    planes = [np.ascontiguousarray(g[k]).reshape(-1) for k in ("pic_y", "pic_cb", "pic_cr")]
    for N in (4, 8, 16, 32):
        for k, (bx, by) in enumerate(g[f"il{N}_pos"]):
            adi = gctx.initAdiPattern(planes[0], w, int(bx), int(by), N, 0, w, h)
            assert np.array_equal(adi, g[f"il{N}_adi"][k])
            for m in range(35):
                assert np.array_equal(gctx.predIntraLumaAng(adi, m, N, N), g[f"il{N}_pred"][k][m])
    for Nc in (4, 8, 16):
        W = 2 * Nc + 1
        for k, (bx, by) in enumerate(g[f"ic{Nc}_pos"]):
            for c in (0, 1):
                adi = gctx.initAdiPattern(planes[1 + c], w // 2, int(bx) // 2, int(by) // 2, Nc, 1, w, h)
                assert np.array_equal(adi[:W * W], g[f"ic{Nc}_adi"][k][c * W * W:(c + 1) * W * W])
                if c == 0:
                    for m in range(35):
                        assert np.array_equal(gctx.predIntraChromaAng(adi, m, Nc, Nc), g[f"ic{Nc}_pred"][k][m])


@pytest.mark.gpu
def test_gpu_inter(gctx):
    from thevc_amd import capi
    B, L = gctx.bit_depth, capi.lib()
    g = load(f"inter_b{B}.npz")
    h, w = g["pic_y"].shape
    d_ref = capi.DevPicture(gctx, w, h, 80, 80).upload([g["pic_y"], g["pic_cb"], g["pic_cr"]])
    gctx._chk(L.hmx_pic_extend_border(gctx.h, C.byref(d_ref.as_pic()), w, h, 80, 80))
    gctx.sync()
    assert np.array_equal(d_ref.download(with_margins=True)[0], g["ext_y"])
    # one PU per call: the golden PUs overlap each other
    for k, (px, py, pw, ph, mvx, mvy, cmx, cmy, bi) in enumerate(g["pu"].tolist()):
        cx, cy = C.c_int(mvx), C.c_int(mvy)
        L.hmx_clipMv(C.byref(cx), C.byref(cy), px, py, w, h, 64)
        assert (cx.value, cy.value) == (cmx, cmy)
        pu = np.zeros(1, capi.PU_DTYPE)
        pu["x"], pu["y"], pu["w"], pu["h"] = px, py, pw, ph
        pu["ref0"], pu["ref1"] = 0, 255
        pu["mv0x"], pu["mv0y"] = cmx, cmy
        if bi:  # golden "bi" outputs are the 14-bit intermediates of ONE list: average a list with itself
            pu["ref1"], pu["mv1x"], pu["mv1y"] = 0, cmx, cmy
        d_pu = gctx.to_device(pu)
        d_dst = capi.DevPicture(gctx, w, h).zero()
        ref_arr = (capi.Pic * 1)(d_ref.as_pic())
        gctx._chk(L.hmx_batch_motionCompensation(gctx.h, d_pu.ptr, 1, ref_arr, 1, C.byref(d_dst.as_pic())))
        gctx.sync()
        got = d_dst.download()
        ey = g["pu_y"][k][:pw * ph].reshape(ph, pw).astype(np.int32)
        ec = [g["pu_c"][k][c * 1024:c * 1024 + pw * ph // 4].reshape(ph // 2, pw // 2).astype(np.int32) for c in (0, 1)]
        if bi:
            sh, off, mx = 15 - B, (1 << (14 - B)) + 2 * 8192, (1 << B) - 1
            ey = np.clip((2 * ey + off) >> sh, 0, mx)
            ec = [np.clip((2 * e + off) >> sh, 0, mx) for e in ec]
        assert np.array_equal(got[0][py:py + ph, px:px + pw], ey), k
        for c in (0, 1):
            assert np.array_equal(got[1 + c][py // 2:(py + ph) // 2, px // 2:(px + pw) // 2], ec[c]), (k, c)
        d_pu.free()
        d_dst.free()
    assert np.array_equal(gctx.addAvg(g["avg_a"], g["avg_b"], 16, 16), g["avg_o"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", FRAMES)
def test_gpu_frame(name, B):
    from thevc_amd import capi
    g = load(name)
    ctx = capi.Context(bit_depth=B)
    L = capi.lib()
    org = [np.ascontiguousarray(g[k]) for k in ("org_y", "org_cb", "org_cr")]
    h, w = org[0].shape
    pp = capi.PicParam(w, h, int(g["qp"]), 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(g["tus"], pp)
    d_org = capi.DevPicture(ctx, w, h).upload(org)
    d_rec = capi.DevPicture(ctx, w, h).zero()
    d_lev = capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
    ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, 1, C.byref(d_org.as_pic()), C.byref(d_rec.as_pic()), C.byref(d_lev.as_pic())))
    ctx.sync()
    rec, lev = d_rec.download(), d_lev.download()
    for p, k in enumerate(("y", "cb", "cr")):
        assert np.array_equal(rec[p], g["rec_" + k]), k
        assert np.array_equal(lev[p], g["lev_" + k].astype(np.int32)), k
    L.hmx_intra_plan_destroy(ctx.h, plan)
    ctx.close()


@pytest.mark.gpu
def test_gpu_quant_scaled(gctx):
    """hmx_xQuant_scaled, hmx_xRateDistOptQuant_scaled and hmx_arlCoeff with a table vs the reference's vectors."""
    from thevc_amd import capi
    B = gctx.bit_depth
    g = load(f"quant_scaled_b{B}.npz")
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"q{N}_par"]):
            qpy, st, ttype, is_intra, mode, tr_idx, cbf_ctx, rdoq = (int(v) for v in par)
            qp = capi.qp_for(qpy, ttype, B)
            coef, qtab, estab = g[f"q{N}_coef"][k], g[f"q{N}_qtab"][k], g[f"q{N}_estab"][k]
            p = capi.QuantParam(qp, -1, st, 1, is_intra, mode)
            assert np.array_equal(gctx.arlCoeff(coef, N, ttype, p, rdoq, qtab), g[f"q{N}_arl"][k]), (N, k, "arl")
            if rdoq:
                root = int((not is_intra) and ttype == 0 and tr_idx == 0)
                rp = capi.RdoqParam(qp, 1, is_intra, mode, root, cbf_ctx, float(g[f"q{N}_lambda"][k]))
                est = capi.EstBits.from_buffer_copy(np.ascontiguousarray(g[f"q{N}_est"][k], np.int32).tobytes())
                lev, s = gctx.xRateDistOptQuant_scaled(coef, N, ttype, rp, est, qtab, estab)
            else:
                lev, s = gctx.xQuant_scaled(coef, N, ttype, p, qtab)
            assert np.array_equal(lev, g[f"q{N}_lev"][k]) and s == g[f"q{N}_sum"][k], (N, k, rdoq)


@pytest.mark.gpu
def test_gpu_dequant_scaled(gctx):
    """hmx_xDeQuant_scaled vs the reference's vectors (both shift directions, extreme levels)."""
    from thevc_amd import capi
    B = gctx.bit_depth
    g = load(f"dequant_scaled_b{B}.npz")
    for N in (4, 8, 16, 32):
        for k, (qpy, _lt) in enumerate(g[f"s{N}_par"]):
            o = gctx.xDeQuant_scaled(g[f"s{N}_lev"][k], N, capi.qp_for(int(qpy), 0, B), g[f"s{N}_tab"][k])
            assert np.array_equal(o, g[f"s{N}_out"][k]), (N, k)


@pytest.mark.gpu
def test_gpu_arl(gctx):
    """hmx_arlCoeff (both forms) and hmx_xQuant with a base QP of its own vs the reference's vectors."""
    from thevc_amd import capi
    B = gctx.bit_depth
    g = load(f"arl_b{B}.npz")
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"a{N}_par"]):
            qpy, qp_base, st, ttype, is_intra, mode, rdoq = (int(v) for v in par)
            qp, qb = capi.qp_for(qpy, ttype, B), capi.qp_for(qp_base, ttype, B)
            p = capi.QuantParam(qp, qb.per, st, 1, is_intra, mode)
            assert np.array_equal(gctx.arlCoeff(g[f"a{N}_coef"][k], N, ttype, p, rdoq), g[f"a{N}_arl"][k]), (N, k, rdoq)
            if not rdoq:
                lev, s = gctx.xQuant(g[f"a{N}_coef"][k], N, ttype, p)
                assert np.array_equal(lev, g[f"a{N}_lev"][k]) and s == g[f"a{N}_sum"][k], (N, k)


@pytest.mark.gpu
def test_gpu_rdoq(gctx):
    """libhmx's xRateDistOptQuant drop-in vs the reference's vectors (coefficients, bit estimates, lambda in)."""
    from thevc_amd import capi
    B = gctx.bit_depth
    g, O = load(f"rdoq_b{B}.npz"), ol.oracle()
    for N in (4, 8, 16, 32):
        for k, par in enumerate(g[f"r{N}_par"]):
            qpy, st, ttype, is_intra, mode, tr_idx, cbf_ctx = (int(v) for v in par)
            qp = capi.qp_for(qpy, ttype, B)
            root = int((not is_intra) and ttype == 0 and tr_idx == 0)
            rp = capi.RdoqParam(qp, 1, is_intra, mode, root, cbf_ctx, float(g[f"r{N}_lambda"][k]))
            est = capi.EstBits.from_buffer_copy(np.ascontiguousarray(g[f"r{N}_est"][k], np.int32).tobytes())
            lev, s = gctx.xRateDistOptQuant(g[f"r{N}_coef"][k], N, ttype, rp, est)
            assert np.array_equal(lev, g[f"r{N}_lev"][k]) and s == g[f"r{N}_sum"][k], (N, k)


@pytest.mark.gpu
def test_gpu_deblock(gctx):
    """hmx_deblock_picture vs the reference's edge filters (golden), luma and chroma, with and without no-filter units."""
    from thevc_amd import capi
    B, L = gctx.bit_depth, capi.lib()
    g = load(f"deblock_b{B}.npz")
    for k in range(2):
        boff, toff, use_nof, (y, cb, cr, bsv, bsh, qp, nof), outs = _deblock_case(g, k)
        h, w = y.shape
        pic = capi.DevPicture(gctx, w, h).upload([y, cb, cr])
        d = [gctx.to_device(a) for a in (bsv, bsh, qp, nof)]
        p = pic.as_pic()
        gctx._chk(L.hmx_deblock_picture(gctx.h, C.byref(p), w, h, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr if use_nof else None, boff, toff))
        gctx.sync()
        got = pic.download()
        for a, b in zip(got, outs):
            assert np.array_equal(a, b), k
        pic.free()
        for x in d:
            x.free()


@pytest.mark.gpu
def test_gpu_sao(gctx):
    """hmx_sao_picture vs the reference's SAOProcess (golden): all edge classes, band offset, cut CTUs, chroma."""
    from thevc_amd import capi
    B, L = gctx.bit_depth, capi.lib()
    g = load(f"sao_b{B}.npz")
    y, cb, cr = g["y"], g["cb"], g["cr"]
    h, w = y.shape
    src = capi.DevPicture(gctx, w, h).upload([y, cb, cr])
    dst = capi.DevPicture(gctx, w, h).zero()
    d_prm = gctx.to_device(np.ascontiguousarray(g["prm"]))
    a, b = src.as_pic(), dst.as_pic()
    gctx._chk(L.hmx_sao_picture(gctx.h, C.byref(a), C.byref(b), w, h, d_prm.ptr, g["prm"].shape[1]))
    gctx.sync()
    got = dst.download()
    for x, k in zip(got, ("oy", "ocb", "ocr")):
        assert np.array_equal(x, g[k]), k
    src.free(), dst.free(), d_prm.free()

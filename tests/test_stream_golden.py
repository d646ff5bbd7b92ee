"""Real reference streams: decision lists that oracle/ref_decision_tap.cpp took from the
reference DECODER while it decoded streams made by the reference ENCODER (tests/golden/make_stream_golden.py).
The block path must turn every picture's decisions + levels into exactly the samples the reference decoder
reconstructed (its picture-digest SEI check passed when the fixture was made)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from thevc_amd.decisions import (MARGIN, TU_INTER, deblock_maps, has_sao, is_deblocked, levels_to_planes,  # noqa: F401
                                 load_pictures as pictures, prediction_units, reference_pocs, split_blocks, strength_inputs, z_offset)

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "stream_*.npz")))


def oracle_decode_sequence(pics, hook=None):
    """Every picture of a stream, in decoding order, from its decisions; references are this function's own outputs."""
    O = ol.oracle()
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    ext, out = {}, []
    for p in pics:
        w, h, B, m = p["w"], p["h"], p["B"], MARGIN
        st = I3(w, w // 2, w // 2)
        lev = levels_to_planes(p)
        rec = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        intra_tus, inter_tus = split_blocks(p)
        if len(p["pus"]):
            pocs = reference_pocs(p)
            pus = prediction_units(p, {poc: i for i, poc in enumerate(pocs)})
            ptrs = (C.c_void_p * (3 * len(pocs)))()
            for i, poc in enumerate(pocs):
                for k in range(3):
                    pm, pw = (m, w) if k == 0 else (m // 2, w // 2)
                    ptrs[i * 3 + k] = ext[poc][k].ctypes.data + 2 * (pm * (pw + 2 * pm) + pm)
            O.hmo_mc_frame(pus.ctypes.data, len(pus), B, ptrs, I3(w + 2 * m, w // 2 + m, w // 2 + m), P3(*[a.ctypes.data for a in rec]), st)
            if hook:
                hook(p, [a.copy() for a in rec], lev, inter_tus)
            mx = (1 << B) - 1
            for t in inter_tus:  # residual of the inter coding units onto their prediction (invRecurTransformNxN + addClip)
                n, k, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
                q = O.hmo_setQPforQuant(p["qp"], int(k != 0), 6 * (B - 8), 0)
                r = ol.o_invtransformNxN(lev[k][y:y + n, x:x + n], n, B, 65535, q.per, q.rem, int(t["flags"]) & 1)
                rec[k][y:y + n, x:x + n] = np.clip(rec[k][y:y + n, x:x + n].astype(np.int32) + r, 0, mx)
        if len(intra_tus):  # intra blocks, in list order, onto what is there (DEC/TDecCu.cpp:469-687)
            cfg = ol.frame_cfg(w, h, B, p["qp"], 1, 0, p["ctu"])
            O.hmo_intra_frame_decode(C.byref(cfg), intra_tus.ctypes.data, len(intra_tus), P3(*[a.ctypes.data for a in rec]), st,
                                     P3(*[a.ctypes.data for a in lev]))
        if is_deblocked(p):
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            bsv, bsh, qpm = deblock_maps(p)
            if len(p["pus"]):  # boundary strengths from motion, coded-block flags and intra flags (:444-569)
                units, ev, eh = strength_inputs(p)
                O.hmo_deblock_strengths(vp(units), vp(ev), vp(eh), w, h, p["ctu"], int(p["slice_type"] == 0), vp(bsv), vp(bsh))
            O.hmo_deblock_picture(P3(*[a.ctypes.data for a in rec]), st, w, h, B, vp(bsv), vp(bsh), vp(qpm), None, p["dbk"][1], p["dbk"][2])
        if has_sao(p):
            flt = [np.zeros_like(a) for a in rec]
            prm = p["sao"]
            O.hmo_sao_picture(P3(*[a.ctypes.data for a in rec]), P3(*[a.ctypes.data for a in flt]), st, w, h, B, p["ctu"],
                              P3(prm[0].ctypes.data, prm[1].ctypes.data, prm[2].ctypes.data))
            rec = flt
        planes = []
        for k, a in enumerate(rec):  # reference picture with margins (TComPicYuv::extendPicBorder)
            pm = m if k == 0 else m // 2
            ph, pw = a.shape
            e = np.zeros((ph + 2 * pm, pw + 2 * pm), np.int16)
            e[pm:pm + ph, pm:pm + pw] = a
            flat = e.reshape(-1)
            O.hmo_extendPicBorder(ol.ptr(flat, pm * (pw + 2 * pm) + pm), pw + 2 * pm, pw, ph, pm, pm)
            planes.append(flat)
        ext[p["poc"]] = planes
        out.append(rec)
    return out


def test_fixtures_present():
    assert len(FIXTURES) >= 14 and len(ENC_FIXTURES) >= 3
    allp = [p for f in FIXTURES for p in pictures(f)]
    assert any(is_deblocked(p) for p in allp) and any(not is_deblocked(p) for p in allp)
    sao_types = set(int(t) for p in allp for t in p["sao"]["type"].reshape(-1))
    assert {-1, 4} <= sao_types and sao_types & {0, 1, 2, 3}, sao_types  # off, band offset, edge offsets
    for f in FIXTURES:
        pics = list(pictures(f))
        assert pics and all(len(p["tus"]) > 20 for p in pics)
        sizes = set(int(s) for p in pics for s in p["tus"]["log2n"])
        assert {2, 3, 4} <= sizes, sizes
        assert any((p["tus"]["flags"] & 1).any() for p in pics), "no transform-skip block in the stream"


@pytest.mark.skipif(not os.path.isdir("/root/reference/source/Lib/TLibDecoder"), reason="needs the reference sources (not on the GPU box)")
def test_fixture_comes_from_the_reference(tmp_path):
    """Provenance: re-make one fixture with the reference encoder + decoder tap and compare it with the committed file."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stream_golden", os.path.join(HERE, "golden", "make_stream_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.make("lowdelay_P_main_q30", 17, 192, 128, 4, 8, 30, "encoder_lowdelay_P_main.cfg", mod.PURE, motion=True, out_dir=str(tmp_path))
    new, old = np.load(tmp_path / "stream_lowdelay_P_main_q30.npz"), np.load(os.path.join(HERE, "golden", "stream_lowdelay_P_main_q30.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in new.files:
        assert np.array_equal(new[k], old[k]), k


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_oracle_reconstructs_reference_streams(path):
    """CPU restatement (decoder direction: DEC/TDecCu.cpp:384-687, motion compensation, loop filters) vs the reference
    decoder's own output, picture by picture in decoding order."""
    pics = list(pictures(path))
    for p, rec in zip(pics, oracle_decode_sequence(pics)):
        for k in range(3):
            bad = np.argwhere(rec[k] != p["rec"][k])
            assert not len(bad), (os.path.basename(path), "poc", p["poc"], "plane", k, "first mismatch (y, x)", bad[0].tolist(), len(bad))


ENC_FIXTURES = [f for f in FIXTURES if "rdoq0" in f]


@pytest.mark.parametrize("path", ENC_FIXTURES, ids=[os.path.basename(f)[:-4] for f in ENC_FIXTURES])
def test_oracle_encodes_like_the_reference_encoder(path):
    """ENCODER direction with the flat quantiser + sign-bit hiding.  Intra pictures (ENC/TEncSearch.cpp:1006-1390): the
    reference encoder's decisions and its input picture must give the levels it wrote into the stream and its
    reconstruction.  Inter pictures (:4526-4990): wherever the encoder kept a transform block's residual (its rate-
    distortion check may zero a block, which is a decision), the stream's levels must be the quantised transform of
    input - prediction."""
    O = ol.oracle()
    pics = list(pictures(path))
    checked = [0, 0]

    def inter_residuals(p, pred, lev, inter_tus):
        B = p["B"]
        for t in inter_tus:
            n, k, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            want = lev[k][y:y + n, x:x + n]
            checked[1] += 1
            if not want.any():
                continue
            q = O.hmo_setQPforQuant(p["qp"], int(k != 0), 6 * (B - 8), 0)
            cfg = ol.quant_cfg(q.per, q.rem, intra_slice=0, sign_hide=1, scan_idx=0)
            resi = (p["org"][k][y:y + n, x:x + n].astype(np.int32) - pred[k][y:y + n, x:x + n]).astype(np.int16)
            got, _ = ol.o_transformNxN(resi, n, B, 65535, int(t["flags"]) & 1, cfg)
            assert np.array_equal(np.asarray(got).reshape(n, n), want), (os.path.basename(path), p["poc"], "inter levels", k, x, y, n)
            checked[0] += 1

    for p in pics:
        assert p["org"] is not None
        if p["slice_type"] != 2:
            continue
        rec, lev = ol.o_intra_frame_encode(np.ascontiguousarray(p["tus"], ol.TU_DTYPE), p["w"], p["h"], p["B"], p["qp"], p["org"])
        want = levels_to_planes(p)
        for k in range(3):
            assert np.array_equal(lev[k], want[k]), (os.path.basename(path), p["poc"], "levels", k, int((lev[k] != want[k]).sum()))
            assert np.array_equal(rec[k], p["rec"][k]), (os.path.basename(path), p["poc"], "reconstruction", k)
    if any(p["slice_type"] != 2 for p in pics):
        oracle_decode_sequence(pics, inter_residuals)
        assert checked[0] > 150, checked
        n_intra = 0
        for p in pics:  # the intra coding units the encoder chose inside inter pictures: encoded onto the picture
            intra_tus, _ = split_blocks(p)
            if p["slice_type"] == 2 or not len(intra_tus):
                continue
            w, h = p["w"], p["h"]
            P3, I3 = C.c_void_p * 3, C.c_int * 3
            st = I3(w, w // 2, w // 2)
            rec = [np.ascontiguousarray(a, np.int16).copy() for a in p["rec"]]  # holds the inter reconstruction
            lev = [np.zeros(a.shape, np.int32) for a in rec]
            org = [np.ascontiguousarray(a, np.int16) for a in p["org"]]
            cfg = ol.frame_cfg(w, h, p["B"], p["qp"], 1, 0, p["ctu"], inter_slice=1)  # rounding of a P/B slice
            O.hmo_intra_frame_encode(C.byref(cfg), intra_tus.ctypes.data, len(intra_tus), P3(*[a.ctypes.data for a in org]), st,
                                     P3(*[a.ctypes.data for a in rec]), st, P3(*[a.ctypes.data for a in lev]))
            want = levels_to_planes(p)
            for t in intra_tus:
                n, k, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
                assert np.array_equal(lev[k][y:y + n, x:x + n], want[k][y:y + n, x:x + n]), (p["poc"], "intra-in-inter levels", k, x, y, n)
                n_intra += 1
            for k in range(3):
                assert np.array_equal(rec[k], p["rec"][k]), (p["poc"], "intra-in-inter reconstruction", k)
        assert n_intra > 100, n_intra


@pytest.mark.gpu
@pytest.mark.parametrize("path", ENC_FIXTURES, ids=[os.path.basename(f)[:-4] for f in ENC_FIXTURES])
def test_gpu_encodes_like_the_reference_encoder(path):
    """hmx_frame_intra_encode on the reference encoder's decisions and input: its stream's levels, its reconstruction."""
    from thevc_amd import capi
    L = capi.lib()
    pics = list(pictures(path))
    ctx = capi.Context(bit_depth=pics[0]["B"], ctu_size=pics[0]["ctu"])
    try:
        refs = {}
        for p in pics:
            w, h = p["w"], p["h"]
            if p["slice_type"] != 2:  # inter picture: residual chain of the kept blocks on libhmx's own prediction
                m = MARGIN
                intra_tus, inter_tus = split_blocks(p)
                pocs = reference_pocs(p)
                pus = prediction_units(p, {poc: i for i, poc in enumerate(pocs)})
                d_pus = ctx.to_device(pus)
                ref_arr = (capi.Pic * len(pocs))(*[refs[poc].as_pic() for poc in pocs])
                d_pred, d_org = capi.DevPicture(ctx, w, h).zero(), capi.DevPicture(ctx, w, h).upload(p["org"])
                d_rec = capi.DevPicture(ctx, w, h, m, m).upload(p["rec"])  # the reference decoder's picture as next reference
                d_tmp, d_lp = capi.DevPicture(ctx, w, h).zero(), capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
                pred_arr = (capi.Pic * 1)(d_pred.as_pic())
                job = (capi.McJob * 1)()
                job[0].d_pus, job[0].n_pus, job[0].refs, job[0].n_refs = d_pus.ptr, len(pus), ref_arr, len(pocs)
                job[0].dst, job[0].pic_w, job[0].pic_h = C.pointer(pred_arr[0]), w, h
                ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, 1, job))
                tl = ctx.tu_list(inter_tus)
                pp = capi.PicParam(w, h, p["qp"], 0, capi.P_SLICE if p["slice_type"] == 1 else capi.B_SLICE, 1)
                ctx._chk(L.hmx_batch_residual_transform_recon_multi(ctx.h, tl, 1, (capi.Pic * 1)(d_org.as_pic()), pred_arr,
                                                                    (capi.Levels * 1)(d_lp.as_pic()), (capi.Pic * 1)(d_tmp.as_pic()), None, C.byref(pp)))
                ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d_rec.as_pic()), w, h, m, m))
                ctx.sync()
                got, want, kept = d_lp.download(), levels_to_planes(p), 0
                for t in inter_tus:
                    n, k, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
                    if want[k][y:y + n, x:x + n].any():
                        assert np.array_equal(got[k][y:y + n, x:x + n], want[k][y:y + n, x:x + n]), (p["poc"], "inter levels", k, x, y, n)
                        kept += 1
                assert kept > 50, kept
                if len(intra_tus):  # the intra coding units of this inter picture, encoder side, onto the picture
                    plan = ctx.intra_plan(intra_tus, capi.PicParam(w, h, p["qp"], 0, capi.P_SLICE, 1))  # rounding of a P slice
                    d_held = capi.DevPicture(ctx, w, h).upload(p["rec"])
                    d_lz = capi.DevLevelsZ(ctx, w, h, p["ctu"]).zero()
                    ctx._chk(L.hmx_frame_intra_encode_onto(ctx.h, plan, 1, (capi.Pic * 1)(d_org.as_pic()), (capi.Pic * 1)(d_held.as_pic()),
                                                           (capi.Levels * 1)(d_lz.as_pic())))
                    ctx.sync()
                    held = d_held.download()
                    raw = [d_lz.bufs[k].download(np.int32) for k in range(3)]
                    for k in range(3):
                        assert np.array_equal(held[k], p["rec"][k]), (p["poc"], "intra-in-inter reconstruction", k)
                    for t in intra_tus:
                        n, k, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
                        o = z_offset(k, x, y, w, p["ctu"])
                        assert np.array_equal(raw[k][o:o + n * n], p["lev"][k][o:o + n * n]), (p["poc"], "intra-in-inter levels", k, x, y, n)
                    L.hmx_intra_plan_destroy(ctx.h, plan)
                refs[p["poc"]] = d_rec
                continue
            plan = ctx.intra_plan(p["tus"], capi.PicParam(w, h, p["qp"], 0, capi.I_SLICE, 1))
            d_org = capi.DevPicture(ctx, w, h).upload(p["org"])
            d_rec = capi.DevPicture(ctx, w, h).zero()
            d_lev = capi.DevLevelsZ(ctx, w, h, p["ctu"]).zero()
            ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, 1, (capi.Pic * 1)(d_org.as_pic()), (capi.Pic * 1)(d_rec.as_pic()),
                                              (capi.Levels * 1)(d_lev.as_pic())))
            ctx.sync()
            got = d_rec.download()
            for k in range(3):
                assert np.array_equal(d_lev.bufs[k].download(np.int32), p["lev"][k]), (os.path.basename(path), p["poc"], "levels", k)
                assert np.array_equal(got[k], p["rec"][k]), (os.path.basename(path), p["poc"], "reconstruction", k)
            L.hmx_intra_plan_destroy(ctx.h, plan)
            d_ref = capi.DevPicture(ctx, w, h, MARGIN, MARGIN).upload(got)  # reference picture for the stream's inter pictures
            ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d_ref.as_pic()), w, h, MARGIN, MARGIN))
            refs[p["poc"]] = d_ref
            d_org.free(), d_rec.free(), d_lev.free()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_gpu_reconstructs_reference_streams(path):
    """libhmx through the C-ABI (thevc_amd/decisions.py: motion compensation, inter residual, intra blocks [onto the
    inter reconstruction], deblocking with strengths from motion, SAO, border extension; references = libhmx's own
    earlier outputs) vs the reference decoder, picture by picture in decoding order."""
    from thevc_amd.decisions import decode_sequence
    pics = list(pictures(path))
    for p, got in zip(pics, decode_sequence(pics)):
        for k in range(3):
            bad = np.argwhere(got[k] != p["rec"][k])
            assert not len(bad), (os.path.basename(path), "poc", p["poc"], "plane", k, "first mismatch (y, x)", bad[0].tolist(), len(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["level", "packed"])
@pytest.mark.parametrize("n_pics", [70, 400])
def test_gpu_real_decisions_across_pictures(n_pics, schedule, monkeypatch):
    """The across-pictures level schedule (one wave = one block of 16/8/4/1 pictures; two picture groups on two streams
    from 384 pictures) on the block structure of a real stream: n_pics copies of one picture's decisions and levels
    must all come out as the reference decoder's picture."""
    from thevc_amd import capi
    L = capi.lib()
    path = os.path.join(HERE, "golden", "stream_intra_main_q37_416x240_dbk.npz")
    p = next(iter(pictures(path)))
    monkeypatch.setenv("HMX_INTRA_SCHEDULE", schedule)  # read once, when the context is created
    ctx = capi.Context(bit_depth=p["B"], ctu_size=p["ctu"])
    try:
        w, h = p["w"], p["h"]
        plan = ctx.intra_plan(p["tus"], capi.PicParam(w, h, p["qp"], 0, capi.I_SLICE, 1))
        d_lev = capi.DevLevelsZ(ctx, w, h, p["ctu"])
        for k in range(3):
            d_lev.bufs[k].upload(np.ascontiguousarray(p["lev"][k], np.int32))
        d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
        rec_arr = (capi.Pic * n_pics)(*[d.as_pic() for d in d_rec])
        lev_arr = (capi.Levels * n_pics)(*[d_lev.as_pic() for _ in range(n_pics)])  # every picture reads the same levels
        ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, n_pics, rec_arr, lev_arr))
        ctx.sync()
        want = list(oracle_decode_sequence([dict(p, dbk=[1, 0, 0])]))[0]  # before the loop filter
        for i in sorted({0, 1, 15, 16, n_pics // 2, n_pics - 1}):
            got = d_rec[i].download()
            for k in range(3):
                assert np.array_equal(got[k], want[k]), (i, k)
        L.hmx_intra_plan_destroy(ctx.h, plan)
    finally:
        ctx.close()

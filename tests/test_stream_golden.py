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

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "stream_*.npz")))


def _spread4(v):
    v = (v | (v << 2)) & 0x33
    return (v | (v << 1)) & 0x55


def z_offset(plane, x, y, w, ctu):
    """Offset of a block in the reference's coefficient layout: CTUs in raster order, 16 ints per 4x4 unit in Z order."""
    c = ctu >> (1 if plane else 0)
    cw = -(-(w >> (1 if plane else 0)) // c)
    z = _spread4((x & (c - 1)) >> 2) | (_spread4((y & (c - 1)) >> 2) << 1)
    return ((y // c) * cw + (x // c)) * c * c + z * 16


def pictures(path):
    d = np.load(path)
    for i in range(int(d["n"])):
        poc, w, h, B, qp, ctu, slice_type = (int(v) for v in d[f"hdr{i}"])
        yield dict(poc=poc, w=w, h=h, B=B, qp=qp, ctu=ctu, slice_type=slice_type, pus=d[f"pus{i}"], cus=d[f"cus{i}"], tus=d[f"tus{i}"], lev=[d[f"lev{i}_{k}"] for k in range(3)],
                   rec=[d[f"rec{i}_{k}"] for k in range(3)], org=[d[f"org{i}_{k}"] for k in range(3)] if f"org{i}_0" in d else None, sao=np.ascontiguousarray(d[f"sao{i}"]), dbk=[int(v) for v in d[f"dbk{i}"]])


def deblock_maps(p):
    """Boundary strengths of an all-intra picture (xGetBoundaryStrengthSingle :444-470: 2 wherever an edge is
    filtered): the left / top sides of the luma transform blocks that lie on the 8x8 grid, not on the picture boundary
    (xSetEdgefilterTU, xSetEdgefilterPU :264-330; every coding-unit edge is also a transform-block edge)."""
    uw, uh = p["w"] // 4, p["h"] // 4
    bsv, bsh = np.zeros((uh, uw), np.uint8), np.zeros((uh, uw), np.uint8)
    for t in p["tus"]:
        if t["plane"]:
            continue
        n, x, y = (1 << int(t["log2n"])) // 4, int(t["x"]) // 4, int(t["y"]) // 4
        if x and x % 2 == 0:
            bsv[y:y + n, x] = 2
        if y and y % 2 == 0:
            bsh[y, x:x + n] = 2
    return bsv, bsh, np.full((uh, uw), p["qp"], np.int8)


DBK_UNIT = np.dtype([("intra", "u1"), ("cbf", "u1"), ("ref", "i1", 2), ("mv", "<i2", (2, 2))])


def strength_inputs(p):
    """What xGetBoundaryStrengthSingle reads, per 4x4 unit, and the edge maps (hmx_deblock_strengths): intra flag from the
    coding units, luma coded-block flag from the transform blocks, reference picture (its POC; -1 = list unused) and
    vector from the prediction units; edges: coding-unit and transform-block sides = 3, prediction-unit sides inside
    a coding unit = 1 (xSetEdgefilterTU / xSetEdgefilterPU, COM/TComLoopFilter.cpp:264-330)."""
    uw, uh = p["w"] // 4, p["h"] // 4
    units = np.zeros((uh, uw), DBK_UNIT)
    units["ref"][:] = -1
    ev, eh = np.zeros((uh, uw), np.uint8), np.zeros((uh, uw), np.uint8)

    def sides(x, y, wd, ht, v):
        if x:
            ev[y:y + ht, x] |= v
        if y:
            eh[y, x:x + wd] |= v

    for c in p["cus"]:
        n, x, y = (1 << int(c["log2size"])) // 4, int(c["x"]) // 4, int(c["y"]) // 4
        units["intra"][y:y + n, x:x + n] = c["intra"]
        sides(x, y, n, n, 3)
    for t in p["tus"]:
        if t["plane"] == 0:
            n, x, y = (1 << int(t["log2n"])) // 4, int(t["x"]) // 4, int(t["y"]) // 4
            units["cbf"][y:y + n, x:x + n] = 1 if int(t["flags"]) & 0x80 else 0
            sides(x, y, n, n, 3)
    for u in p["pus"]:
        x, y, wd, ht = int(u["x"]) // 4, int(u["y"]) // 4, int(u["w"]) // 4, int(u["h"]) // 4
        for l in (0, 1):
            if u[f"poc{l}"] > -32768:
                units["ref"][y:y + ht, x:x + wd, l] = int(u[f"poc{l}"])
                units["mv"][y:y + ht, x:x + wd, l, 0] = int(u[f"mv{l}x"])
                units["mv"][y:y + ht, x:x + wd, l, 1] = int(u[f"mv{l}y"])
        sides(x, y, wd, ht, 1)
    return np.ascontiguousarray(units), ev, eh


def is_deblocked(p):
    return not p["dbk"][0]


def has_sao(p):
    return bool((p["sao"]["type"] >= 0).any())


MARGIN = 80  # luma margin of reference pictures (TComPicYuv: g_uiMaxCUWidth + 16)
TU_INTER = 2


def levels_to_planes(p):
    """The picture's levels from the reference's per-CTU layout into plane geometry, following the block list."""
    w, h = p["w"], p["h"]
    lev = [np.zeros((h, w), np.int32), np.zeros((h // 2, w // 2), np.int32), np.zeros((h // 2, w // 2), np.int32)]
    for t in p["tus"]:
        n, pl, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
        o = z_offset(pl, x, y, w, p["ctu"])
        lev[pl][y:y + n, x:x + n] = p["lev"][pl][o:o + n * n].reshape(n, n)
    return lev


def prediction_units(p, slot):
    """The tap's prediction units as hmx_pu / hmo_pu records: reference POC -> slot of the reference table, vectors
    clipped as motion compensation clips them (TComDataCU::clipMv, COM/TComDataCU.cpp:3505-3517: relative to the
    coding unit's origin, 8 samples + one CTU beyond the picture)."""
    src, ctu = p["pus"], p["ctu"]
    out = np.zeros(len(src), ol.PU_DTYPE)
    for k in ("x", "y", "w", "h"):
        out[k] = src[k]
    cx, cy = src["cu_x"].astype(np.int64), src["cu_y"].astype(np.int64)
    for l in (0, 1):
        used = src[f"poc{l}"] > -32768
        out[f"ref{l}"] = [slot[int(v)] if u else 255 for v, u in zip(src[f"poc{l}"], used)]
        out[f"mv{l}x"] = np.clip(src[f"mv{l}x"].astype(np.int64), (-ctu - 8 - cx + 1) * 4, (p["w"] + 8 - cx - 1) * 4)
        out[f"mv{l}y"] = np.clip(src[f"mv{l}y"].astype(np.int64), (-ctu - 8 - cy + 1) * 4, (p["h"] + 8 - cy - 1) * 4)
    return out


def reference_pocs(p):
    return sorted({int(v) for l in (0, 1) for v in p["pus"][f"poc{l}"] if v > -32768})


def split_blocks(p):
    inter = (p["tus"]["flags"] & TU_INTER) != 0
    return np.ascontiguousarray(p["tus"][~inter], ol.TU_DTYPE), np.ascontiguousarray(p["tus"][inter], ol.TU_DTYPE)


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
    """libhmx through the C-ABI vs the reference decoder, picture by picture in decoding order, references = libhmx's
    own earlier outputs: hmx_batch_motionCompensation_multi + hmx_batch_invtransformNxN_multi (inter coding units),
    hmx_frame_intra_decode[_onto] with the levels in the reference's own coefficient layout (intra coding units),
    hmx_deblock_picture, hmx_sao_picture, hmx_pic_extend_border."""
    from thevc_amd import capi
    L = capi.lib()
    pics = list(pictures(path))
    ctx = capi.Context(bit_depth=pics[0]["B"], ctu_size=pics[0]["ctu"])
    refs, m = {}, MARGIN
    try:
        for p in pics:
            w, h = p["w"], p["h"]
            intra_tus, inter_tus = split_blocks(p)
            d_rec = capi.DevPicture(ctx, w, h, m, m).zero()
            rec_arr = (capi.Pic * 1)(d_rec.as_pic())
            keep = []
            if len(p["pus"]):
                pocs = reference_pocs(p)
                pus = prediction_units(p, {poc: i for i, poc in enumerate(pocs)})
                d_pus = ctx.to_device(pus)
                ref_arr = (capi.Pic * len(pocs))(*[refs[poc].as_pic() for poc in pocs])
                d_pred = capi.DevPicture(ctx, w, h).zero()
                pred_arr = (capi.Pic * 1)(d_pred.as_pic())
                for dst in (pred_arr, rec_arr):  # the prediction, and the reconstruction of units without residual
                    job = (capi.McJob * 1)()
                    job[0].d_pus, job[0].n_pus, job[0].refs, job[0].n_refs = d_pus.ptr, len(pus), ref_arr, len(pocs)
                    job[0].dst, job[0].pic_w, job[0].pic_h = C.pointer(dst[0]), w, h
                    ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, 1, job))
                if len(inter_tus):
                    tl = ctx.tu_list(inter_tus)
                    d_lp = capi.DevPicture(ctx, w, h, dtype=np.int32).upload(levels_to_planes(p))
                    lp_arr = (capi.Levels * 1)(d_lp.as_pic())
                    pp = capi.PicParam(w, h, p["qp"], 0, capi.B_SLICE, 1)
                    ctx._chk(L.hmx_batch_invtransformNxN_multi(ctx.h, tl, 1, lp_arr, pred_arr, rec_arr, C.byref(pp)))
                    keep += [d_lp]
                keep += [d_pred, d_pus]
            if len(intra_tus):
                plan = ctx.intra_plan(intra_tus, capi.PicParam(w, h, p["qp"], 0, capi.I_SLICE, 1))
                d_lev = capi.DevLevelsZ(ctx, w, h, p["ctu"])
                for k in range(3):
                    assert d_lev.elems[k] == len(p["lev"][k])
                    d_lev.bufs[k].upload(np.ascontiguousarray(p["lev"][k], np.int32))
                lev_arr = (capi.Levels * 1)(d_lev.as_pic())
                fn = L.hmx_frame_intra_decode_onto if len(p["pus"]) else L.hmx_frame_intra_decode
                ctx._chk(fn(ctx.h, plan, 1, rec_arr, lev_arr))
                ctx.sync()
                L.hmx_intra_plan_destroy(ctx.h, plan)
                d_lev.free()
            if is_deblocked(p):
                bsv, bsh, qpm = deblock_maps(p)
                d_bv, d_bh, d_qp = ctx.to_device(bsv), ctx.to_device(bsh), ctx.to_device(qpm)
                if len(p["pus"]):
                    units, ev, eh = strength_inputs(p)
                    d_u, d_ev, d_eh = ctx.to_device(units), ctx.to_device(ev), ctx.to_device(eh)
                    ctx._chk(L.hmx_deblock_strengths(ctx.h, d_u.ptr, d_ev.ptr, d_eh.ptr, w, h, int(p["slice_type"] == 0), d_bv.ptr, d_bh.ptr))
                ctx._chk(L.hmx_deblock_picture(ctx.h, C.byref(rec_arr[0]), w, h, d_bv.ptr, d_bh.ptr, d_qp.ptr, None, p["dbk"][1], p["dbk"][2]))
            d_out = d_rec
            if has_sao(p):
                d_out = capi.DevPicture(ctx, w, h, m, m).zero()
                d_prm = ctx.to_device(p["sao"])
                a, b = d_rec.as_pic(), d_out.as_pic()
                ctx._chk(L.hmx_sao_picture(ctx.h, C.byref(a), C.byref(b), w, h, d_prm.ptr, p["sao"].shape[1]))
            ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(d_out.as_pic()), w, h, m, m))
            ctx.sync()
            refs[p["poc"]] = d_out
            got = d_out.download()
            for k in range(3):
                bad = np.argwhere(got[k] != p["rec"][k])
                assert not len(bad), (os.path.basename(path), "poc", p["poc"], "plane", k, "first mismatch (y, x)", bad[0].tolist(), len(bad))
            for d in keep:
                d.free()
    finally:
        ctx.close()

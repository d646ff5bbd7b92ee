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
        poc, w, h, B, qp, ctu = (int(v) for v in d[f"hdr{i}"])
        yield dict(poc=poc, w=w, h=h, B=B, qp=qp, ctu=ctu, tus=d[f"tus{i}"], lev=[d[f"lev{i}_{k}"] for k in range(3)],
                   rec=[d[f"rec{i}_{k}"] for k in range(3)], sao=np.ascontiguousarray(d[f"sao{i}"]), dbk=[int(v) for v in d[f"dbk{i}"]])


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


def is_deblocked(p):
    return not p["dbk"][0]


def has_sao(p):
    return bool((p["sao"]["type"] >= 0).any())


def test_fixtures_present():
    assert len(FIXTURES) >= 6
    allp = [p for f in FIXTURES for p in pictures(f)]
    assert any(is_deblocked(p) for p in allp) and any(not is_deblocked(p) for p in allp)
    sao_types = set(int(t) for p in allp for t in p["sao"]["type"].reshape(-1))
    assert {-1, 4} <= sao_types and sao_types & {0, 1, 2, 3}, sao_types  # off, band offset, edge offsets
    for f in FIXTURES:
        pics = list(pictures(f))
        assert pics and all(len(p["tus"]) > 100 for p in pics)
        sizes = set(int(s) for p in pics for s in p["tus"]["log2n"])
        assert {2, 3, 4} <= sizes, sizes
        assert any((p["tus"]["flags"] & 1).any() for p in pics), "no transform-skip block in the stream"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_oracle_reconstructs_reference_streams(path):
    """CPU restatement (decoder direction, DEC/TDecCu.cpp:469-687) vs the reference decoder's own output."""
    O = ol.oracle()
    for p in pictures(path):
        w, h, tus = p["w"], p["h"], np.ascontiguousarray(p["tus"], ol.TU_DTYPE)
        lev = [np.zeros((h, w), np.int32), np.zeros((h // 2, w // 2), np.int32), np.zeros((h // 2, w // 2), np.int32)]
        for t in tus:
            n, pl, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = z_offset(pl, x, y, w, p["ctu"])
            lev[pl][y:y + n, x:x + n] = p["lev"][pl][o:o + n * n].reshape(n, n)
        rec = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        cfg = ol.frame_cfg(w, h, p["B"], p["qp"], 1, 0, p["ctu"])
        P3, I3 = C.c_void_p * 3, C.c_int * 3
        st = I3(w, w // 2, w // 2)
        O.hmo_intra_frame_decode(C.byref(cfg), tus.ctypes.data, len(tus), P3(*[a.ctypes.data for a in rec]), st,
                                 P3(*[a.ctypes.data for a in lev]))
        if is_deblocked(p):
            bsv, bsh, qpm = deblock_maps(p)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            O.hmo_deblock_picture(P3(*[a.ctypes.data for a in rec]), st, w, h, p["B"], vp(bsv), vp(bsh), vp(qpm), None, p["dbk"][1], p["dbk"][2])
        if has_sao(p):
            out = [np.zeros_like(a) for a in rec]
            prm = p["sao"]
            O.hmo_sao_picture(P3(*[a.ctypes.data for a in rec]), P3(*[a.ctypes.data for a in out]), st, w, h, p["B"], p["ctu"],
                              P3(prm[0].ctypes.data, prm[1].ctypes.data, prm[2].ctypes.data))
            rec = out
        for k in range(3):
            bad = np.argwhere(rec[k] != p["rec"][k])
            assert not len(bad), (os.path.basename(path), p["poc"], "plane", k, "first mismatch (y, x)", bad[0].tolist(), len(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_gpu_reconstructs_reference_streams(path):
    """libhmx (hmx_frame_intra_decode, levels in the reference's own coefficient layout) vs the reference decoder."""
    from thevc_amd import capi
    L = capi.lib()
    pics = list(pictures(path))
    ctx = capi.Context(bit_depth=pics[0]["B"], ctu_size=pics[0]["ctu"])
    try:
        for p in pics:
            w, h = p["w"], p["h"]
            plan = ctx.intra_plan(p["tus"], capi.PicParam(w, h, p["qp"], 0, capi.I_SLICE, 1))
            d_lev = capi.DevLevelsZ(ctx, w, h, p["ctu"])
            for k in range(3):
                assert d_lev.elems[k] == len(p["lev"][k])
                d_lev.bufs[k].upload(np.ascontiguousarray(p["lev"][k], np.int32))
            d_rec = capi.DevPicture(ctx, w, h).zero()
            rec_arr, lev_arr = (capi.Pic * 1)(d_rec.as_pic()), (capi.Levels * 1)(d_lev.as_pic())
            ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, 1, rec_arr, lev_arr))
            if is_deblocked(p):
                bsv, bsh, qpm = deblock_maps(p)
                d_bv, d_bh, d_qp = ctx.to_device(bsv), ctx.to_device(bsh), ctx.to_device(qpm)
                ctx._chk(L.hmx_deblock_picture(ctx.h, C.byref(rec_arr[0]), w, h, d_bv.ptr, d_bh.ptr, d_qp.ptr, None, p["dbk"][1], p["dbk"][2]))
            d_out = d_rec
            if has_sao(p):
                d_out = capi.DevPicture(ctx, w, h).zero()
                d_prm = ctx.to_device(p["sao"])
                a, b = d_rec.as_pic(), d_out.as_pic()
                ctx._chk(L.hmx_sao_picture(ctx.h, C.byref(a), C.byref(b), w, h, d_prm.ptr, p["sao"].shape[1]))
            ctx.sync()
            got = d_out.download()
            for k in range(3):
                bad = np.argwhere(got[k] != p["rec"][k])
                assert not len(bad), (os.path.basename(path), p["poc"], "plane", k, "first mismatch (y, x)", bad[0].tolist(), len(bad))
            L.hmx_intra_plan_destroy(ctx.h, plan)
            d_rec.free(), d_lev.free()
    finally:
        ctx.close()

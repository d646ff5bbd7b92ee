"""Decision lists of real streams -> pictures: the host-side driver that turns what a decoder has PARSED (the
reference's, via oracle/ref_decision_tap.cpp) into libhmx calls, picture by picture in decoding
order, with libhmx's own earlier outputs as reference pictures:

    inter coding units   hmx_batch_motionCompensation_multi, hmx_batch_invtransformNxN_multi
    intra coding units   hmx_frame_intra_decode (intra pictures) / hmx_frame_intra_decode_onto (inside inter pictures)
    loop filters         hmx_deblock_strengths (inter pictures; intra pictures have strength 2 on every edge),
                         hmx_deblock_picture, hmx_sao_picture
    reference pictures   hmx_pic_extend_border

A picture is a dict: poc, w, h, B, qp, ctu, slice_type (0 B, 1 P, 2 I), tus (hmx_tu records; flags bit 1 = block of
an inter coding unit, bit 7 = luma coded-block flag), pus / cus (records of the tap), lev (three arrays in the
reference's per-CTU coefficient layout), sao ([3][n_ctu] hmx_sao_lcu records), dbk ([disabled, beta_offset_div2,
tc_offset_div2]).  load_pictures() reads the .npz fixtures of tests/golden/make_stream_golden.py.  The numpy helpers
import nothing of the GPU side; decode_sequence() needs libhmx."""
import ctypes as C

import numpy as np

MARGIN = 80  # luma margin of reference pictures (TComPicYuv: g_uiMaxCUWidth + 16)
TU_INTER = 2
TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2n", "u1"), ("plane", "u1"), ("mode", "u1"), ("flags", "u1")])
PU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("ref0", "u1"), ("ref1", "u1"), ("mv0x", "<i2"),
                     ("mv0y", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2")])  # hmx_pu


def _spread4(v):
    v = (v | (v << 2)) & 0x33
    return (v | (v << 1)) & 0x55


def z_offset(plane, x, y, w, ctu):
    """Offset of a block in the reference's coefficient layout: CTUs in raster order, 16 ints per 4x4 unit in Z order."""
    c = ctu >> (1 if plane else 0)
    cw = -(-(w >> (1 if plane else 0)) // c)
    z = _spread4((x & (c - 1)) >> 2) | (_spread4((y & (c - 1)) >> 2) << 1)
    return ((y // c) * cw + (x // c)) * c * c + z * 16


def load_pictures(path):
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
    out = np.zeros(len(src), PU_DTYPE)
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
    return np.ascontiguousarray(p["tus"][~inter], TU_DTYPE), np.ascontiguousarray(p["tus"][inter], TU_DTYPE)


def decode_sequence(pics):
    """Every picture of the stream through libhmx, in decoding order; returns the output pictures (three planes each)."""
    from thevc_amd import capi
    L = capi.lib()
    pics = list(pics)
    ctx = capi.Context(bit_depth=pics[0]["B"], ctu_size=pics[0]["ctu"])
    out = []
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
                    assert d_lev.elems[k] == len(p["lev"][k]), "levels are not in the per-CTU layout of this picture size"
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
            out.append(d_out.download())
            for d in keep:
                d.free()
    finally:
        ctx.close()
    return out

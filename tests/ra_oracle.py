"""CPU oracle of the random-access pipeline (thevc_amd/ra_pipeline.py).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import oracle_lib as ol
from thevc_amd.ra_pipeline import MARGIN


def oracle_segment(wl, k, i_recs):
    """CPU oracle of segment k given the reconstructed bounding I pictures {poc: [3 planes]}: returns
    {poc: [3 planes]} for every inter picture (test infrastructure; imports the oracle)."""
    O = ol.oracle()
    w, h, B = wl.w, wl.h, wl.B
    m = MARGIN
    recs = dict(i_recs)
    ext = {}

    def extended(poc):
        if poc not in ext:
            planes = []
            for p, pl in enumerate(recs[poc]):
                pm = m if p == 0 else m // 2
                ph, pw = pl.shape
                e = np.zeros((ph + 2 * pm, pw + 2 * pm), np.int16)
                e[pm:pm + ph, pm:pm + pw] = pl
                flat = e.reshape(-1)
                O.hmo_extendPicBorder(ol.ptr(flat, pm * (pw + 2 * pm) + pm), pw + 2 * pm, pw, ph, pm, pm)
                planes.append(flat)
            ext[poc] = planes
        return ext[poc]

    P3, I3 = C.c_void_p * 3, C.c_int * 3
    for (poc, r0, r1, li) in wl.segment_jobs(k):
        d = wl.inter[li]
        refs = [extended(r0)] + ([extended(r1)] if r1 is not None else [])
        pus = d["pus_b"] if r1 is not None else d["pus_p"]
        pred = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        ptrs = (C.c_void_p * (3 * len(refs)))()
        for i, r in enumerate(refs):
            for p in range(3):
                pm = m if p == 0 else m // 2
                pw = w if p == 0 else w // 2
                ptrs[i * 3 + p] = r[p].ctypes.data + 2 * (pm * (pw + 2 * pm) + pm)
        t = np.ascontiguousarray(pus, ol.PU_DTYPE)
        O.hmo_mc_frame(t.ctypes.data, len(t), B, ptrs, I3(w + 2 * m, w // 2 + m, w // 2 + m), P3(*[x.ctypes.data for x in pred]),
                       I3(w, w // 2, w // 2))
        org = wl.original(poc)
        rec = [np.zeros_like(x) for x in pred]
        mx = (1 << B) - 1
        for tu in d["tus"]:
            N, p, x, y = 1 << int(tu["log2n"]), int(tu["plane"]), int(tu["x"]), int(tu["y"])
            qp = O.hmo_setQPforQuant(wl.qp, int(p != 0), 6 * (B - 8), 0)
            cfg = ol.quant_cfg(qp.per, qp.rem, intra_slice=0, sign_hide=1, scan_idx=0)
            resi = (org[p][y:y + N, x:x + N].astype(np.int32) - pred[p][y:y + N, x:x + N]).astype(np.int16)
            lvl, _ = ol.o_transformNxN(resi, N, B, 65535, 0, cfg)
            r = ol.o_invtransformNxN(lvl, N, B, 65535, qp.per, qp.rem, 0)
            rec[p][y:y + N, x:x + N] = np.clip(pred[p][y:y + N, x:x + N].astype(np.int32) + r, 0, mx)
        recs[poc] = rec
    return recs

"""RDOQ held against what the REFERENCE ENCODER itself fed it: tests/golden/rdoq_enc_tap.npz holds 1150 calls of the reference's
xRateDistOptQuant recorded inside its encoder (tests/golden/make_rdoq_enc_tap.py) -- the bit-estimate table as the live CABAC
state left it before THAT block (TEncSearch.cpp:1101, 1304, 4882; no two blocks share one), the multiplier of the slice and
component, the coefficients, the block's scan / cbf context / root-cbf branch -- and the levels and absolute sum it returned.
Intra 8- and 10-bit and low-delay-P clips, luma and chroma, 4x4 ... 32x32.  The oracle (CPU) and libhmx (GPU: the scalar drop-in
and the block-list entry with its per-block table index) must return the same levels."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
COLS = ["clip", "depth", "magic", "w", "ttype", "qp", "per", "rem", "bits", "sign_hide", "intra", "scan", "root_cbf", "cbf_ctx", "cu_qp", "poc",
        "abs_sum", "table_bytes"]


def _load():
    g = np.load(os.path.join(HERE, "golden", "rdoq_enc_tap.npz"))
    hdr = {c: g["hdr"][:, i] for i, c in enumerate(COLS)}
    return g, hdr


def test_oracle_rdoq_on_the_encoders_tables():
    g, h = _load()
    assert g["tables"].shape[1] == C.sizeof(ol.EstBits)
    n = len(g["lam"])
    assert n > 1000 and len({bytes(t) for t in g["tables"]}) > n // 2  # real, distinct encoder states
    nz = 0
    for i in range(n):
        N, B = int(h["w"][i]), int(h["depth"][i])
        est = ol.EstBits.from_buffer_copy(g["tables"][i].tobytes())
        cfg = ol.RdoqCfg(int(h["per"][i]), int(h["rem"][i]), int(h["ttype"][i] == 0), int(h["intra"][i]), int(h["scan"][i]), int(h["root_cbf"][i]),
                         int(h["cbf_ctx"][i]), int(h["sign_hide"][i]), float(g["lam"][i]))
        a, b = int(g["off"][i]), int(g["off"][i + 1])
        lev, s = ol.o_rdoq(g["coef"][a:b], N, B, cfg, est)
        assert np.array_equal(lev.reshape(-1), g["lev"][a:b]), ("levels", i, N, int(h["ttype"][i]), int(h["clip"][i]))
        assert s == int(h["abs_sum"][i]), ("abs_sum", i)
        nz += s > 0
    assert nz > n // 2


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 10])
def test_gpu_rdoq_on_the_encoders_tables(B):
    from thevc_amd import capi
    g, h = _load()
    L = capi.lib()
    ctx = capi.Context(bit_depth=B)
    sel = np.flatnonzero(h["depth"] == B)
    assert len(sel) > 100
    # (a) the scalar drop-in, as the encoder shim calls it: every fourth recorded call
    for i in sel[::4]:
        N = int(h["w"][i])
        rp = capi.RdoqParam(capi.Qp(int(h["qp"][i]), int(h["per"][i]), int(h["rem"][i]), int(h["bits"][i])), int(h["sign_hide"][i]), int(h["intra"][i]),
                            {0: 0, 1: 26, 2: 10}[int(h["scan"][i])], int(h["root_cbf"][i]), int(h["cbf_ctx"][i]), float(g["lam"][i]))
        est = capi.EstBits.from_buffer_copy(g["tables"][i].tobytes())
        a, b = int(g["off"][i]), int(g["off"][i + 1])
        lev, s = ctx.xRateDistOptQuant(g["coef"][a:b], N, int(h["ttype"][i]), rp, est)
        assert np.array_equal(lev, g["lev"][a:b]) and s == int(h["abs_sum"][i]), ("scalar", int(i), N)
    # (b) the block-list entry: the calls of one slice (same CU QP and multipliers) as ONE list, every block with ITS OWN table
    groups = {}
    for i in sel:
        groups.setdefault((int(h["clip"][i]), int(h["poc"][i]), int(h["cu_qp"][i])), []).append(int(i))
    checked = 0
    for (clip, poc, cu_qp), idx in sorted(groups.items()):
        lam = [None, None]
        for i in idx:
            c = 0 if h["ttype"][i] == 0 else 1
            assert lam[c] in (None, float(g["lam"][i]))
            lam[c] = float(g["lam"][i])
        lam = [x if x is not None else 1.0 for x in lam]
        # lay the blocks out on a canvas in plane geometry: 32x32 cells, luma plane 0, Cb / Cr planes 1 / 2
        n = len(idx)
        cols = 8
        rows = (n + cols - 1) // cols
        w, hh = cols * 64, rows * 64
        coef = [np.zeros((hh, w), np.int32), np.zeros((hh // 2, w // 2), np.int32), np.zeros((hh // 2, w // 2), np.int32)]
        tus = np.zeros(n, capi.TU_DTYPE)
        side = (capi.RdoqSide * n)()
        ests = (capi.EstBits * n)()
        for k, i in enumerate(idx):
            N, tt = int(h["w"][i]), int(h["ttype"][i])
            p = 0 if tt == 0 else (2 if tt == 3 else 1)
            x, y = (k % cols) * (64 if p == 0 else 32), (k // cols) * (64 if p == 0 else 32)
            a, b = int(g["off"][i]), int(g["off"][i + 1])
            coef[p][y:y + N, x:x + N] = g["coef"][a:b].reshape(N, N)
            tus[k]["x"], tus[k]["y"], tus[k]["log2n"], tus[k]["plane"] = x, y, int(np.log2(N)), p
            tus[k]["mode"] = {0: 0, 1: 26, 2: 10}[int(h["scan"][i])]
            tus[k]["flags"] = 0 if h["intra"][i] else capi.TU_INTER
            side[k].est_idx, side[k].root_cbf, side[k].cbf_ctx = k, int(h["root_cbf"][i]), int(h["cbf_ctx"][i])
            C.memmove(C.byref(ests[k]), g["tables"][i].tobytes(), C.sizeof(capi.EstBits))
            q = capi.qp_for(cu_qp, 0 if p == 0 else 1, B, 0)
            assert (q.per, q.rem) == (int(h["per"][i]), int(h["rem"][i])), "the recorded m_cQP is the CU QP's"
        d_coef = capi.DevPicture(ctx, w, hh, dtype=np.int32).upload(coef)
        d_lev = capi.DevPicture(ctx, w, hh, dtype=np.int32).zero()
        d_sum = ctx.alloc(4 * n)
        pp = capi.PicParam(w, hh, cu_qp, 0, capi.I_SLICE, int(h["sign_hide"][idx[0]]))
        ctx._chk(L.hmx_batch_xRateDistOptQuant(ctx.h, np.ascontiguousarray(tus).ctypes.data, side, n, C.byref(d_coef.as_pic()), C.byref(d_lev.as_pic()),
                                               d_sum.ptr, C.byref(pp), ests, n, lam[0], lam[1]))
        ctx.sync()
        lev, sums = d_lev.download(), d_sum.download(np.uint32, n)
        for k, i in enumerate(idx):
            N, p, x, y = int(h["w"][i]), int(tus[k]["plane"]), int(tus[k]["x"]), int(tus[k]["y"])
            a, b = int(g["off"][i]), int(g["off"][i + 1])
            assert np.array_equal(lev[p][y:y + N, x:x + N].reshape(-1), g["lev"][a:b]), ("list", clip, poc, k, N, p)
            assert int(sums[k]) == int(h["abs_sum"][i])
        checked += n
        d_coef.free(), d_lev.free(), d_sum.free()
    assert checked == len(sel)
    ctx.close()

"""The C++ host mirror (thevc_amd/host/hmx_hm.hpp): same member names as the reference classes over
the C-ABI.  CPU: it compiles and links.  GPU: one encoder-style block call chain vs the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "thevc_amd", "host", "hm_mirror_test")


def _build():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)


def test_mirror_builds():
    _build()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,qp,mode", [(8, 4, 32, 26), (8, 8, 27, 10), (10, 16, 37, 0), (8, 32, 22, 18)])
def test_mirror_transform_chain(B, N, qp, mode):
    _build()
    out = subprocess.run([EXE, str(B), str(N), str(qp), str(mode), "7"], capture_output=True, text=True, check=True).stdout
    lines = out.strip().split("\n")
    abs_sum = int(lines[0])
    resi, lev, rec = (np.array(l.split(), np.int64) for l in lines[1:4])
    O = ol.oracle()
    q = O.hmo_setQPforQuant(qp, 0, 6 * (B - 8), 0)
    cfg = ol.quant_cfg(q.per, q.rem, 1, 1, O.hmo_coef_scan_idx(N, 1, 1, mode))
    ref_lev, s = ol.o_transformNxN(resi.astype(np.int16), N, B, mode, 0, cfg)
    assert s == abs_sum and np.array_equal(ref_lev.reshape(-1), lev)
    ref_rec = ol.o_invtransformNxN(ref_lev, N, B, mode, q.per, q.rem, 0)
    assert np.array_equal(ref_rec.reshape(-1), rec)


@pytest.mark.gpu
@pytest.mark.parametrize("B,qp", [(8, 30), (10, 24)])
def test_mirror_invRecurTransformNxN(B, qp):
    """The transform-quadtree walk of an inter CU (TComTrQuant.cpp:1452): a 32x32 CU with a 16x16 leaf, four
    8x8 leaves, an uncoded 16x16 and a quadrant with two coded 8x8 leaves.  The mirror's walk must place every
    coded leaf's inverse transform (REG_DCT) where the reference's recursion does and leave the rest zero;
    leaves are recomputed with the oracle from the z-order coefficient buffer."""
    _build()
    out = subprocess.run([EXE, "recur", str(B), str(qp), "11"], capture_output=True, text=True, check=True).stdout
    lines = out.strip().split("\n")
    coef, resi = (np.array(l.split(), np.int64) for l in lines[:2])
    resi = resi.reshape(32, 32)
    O = ol.oracle()
    q = O.hmo_setQPforQuant(qp, 0, 6 * (B - 8), 0)
    want = np.zeros((32, 32), np.int64)
    off = 0
    for quad in range(4):
        qx, qy = (quad & 1) * 16, (quad >> 1) * 16
        if quad in (0, 2):  # one 16x16 node: coded in quadrant 0 only
            if quad == 0:
                want[qy:qy + 16, qx:qx + 16] = ol.o_invtransformNxN(coef[off:off + 256], 16, B, 65535, q.per, q.rem, 0)
            off += 256
        else:
            for sub in range(4):
                sx, sy = qx + (sub & 1) * 8, qy + (sub >> 1) * 8
                if quad == 1 or sub in (0, 3):
                    want[sy:sy + 8, sx:sx + 8] = ol.o_invtransformNxN(coef[off:off + 64], 8, B, 65535, q.per, q.rem, 0)
                off += 64
    assert np.array_equal(resi, want)
    assert np.abs(want).sum() > 0

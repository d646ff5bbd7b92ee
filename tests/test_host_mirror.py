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

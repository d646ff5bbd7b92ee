"""hmx_intra_dependency_mask (the dependency order of a plan follows what a mode READS, not which neighbours exist) against the
oracle, by perturbation: for every block size, luma / chroma, all 35 modes and a spread of availability patterns, the
oracle's prediction (fillReferenceSamples -> smoothing -> predIntra*Ang) must not change when every sample of an AVAILABLE
neighbour unit outside the mask is replaced by another value -- those units may hold a stale reconstruction when the block
runs.  No GPU: the function is host code of the library."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from thevc_amd import capi


def _predict(O, plane, x0, y0, N, luma, mode, flags, B):
    """oracle_lib's argument types: numpy arrays (flags uint8, adi int32, prediction int16), the plane by address"""
    W = 2 * N + 1
    adi = np.zeros(2 * W * W, np.int32)
    fl = np.ascontiguousarray(flags, np.uint8)
    flat = plane.reshape(-1)
    O.hmo_fillReferenceSamples(ol.ptr(flat, y0 * plane.shape[1] + x0), plane.shape[1], fl, int(fl.sum()), 4 if luma else 2, N, B, adi)
    pred = np.zeros(N * N, np.int16)
    if luma:
        O.hmo_filterAdi(adi, N)
        O.hmo_predIntraLumaAng(adi, mode, pred, N, N, B)
    else:
        O.hmo_predIntraChromaAng(adi, mode, pred, N, N, B)
    return pred.reshape(N, N)


@pytest.mark.parametrize("luma", [True, False])
@pytest.mark.parametrize("N", [4, 8, 16, 32])
def test_dependency_mask_covers_what_the_prediction_reads(N, luma):
    O, L = ol.oracle(), capi.lib()
    if not luma and N == 32:
        pytest.skip("chroma blocks are at most 16x16 in 4:2:0 with 32x32 luma transforms")
    B, U = 10, 4 if luma else 2
    n = N // U
    rng = np.random.default_rng(N * 2 + luma)
    x0 = y0 = 2 * N + 8
    side = 4 * N + 32
    patterns = [[1] * (4 * n + 1), [0] * n + [1] * (3 * n + 1), [1] * (3 * n + 1) + [0] * n, [0] * n + [1] * (2 * n + 1) + [0] * n,
                [0] * (2 * n + 1) + [1] * (2 * n), [1] * (2 * n) + [0] * (2 * n + 1), [0] * (2 * n) + [1] + [0] * (2 * n)]
    patterns += [list(rng.integers(0, 2, 4 * n + 1)) for _ in range(6)]
    pruned = 0
    for flags in patterns:
        avail = sum(int(b) << u for u, b in enumerate(flags))
        for mode in range(35):
            dep = L.hmx_intra_dependency_mask(N, int(luma), mode, avail)
            assert dep & ~avail == 0
            pruned += bin(avail & ~dep).count("1")
            for trial in range(3):
                plane = rng.integers(0, 1 << B, (side, side)).astype(np.int16)
                want = _predict(O, plane, x0, y0, N, luma, mode, flags, B)
                other = plane.copy()
                for u in range(4 * n + 1):
                    if not flags[u] or (dep >> u) & 1:
                        continue
                    for j in range(1 if u == 2 * n else U):  # the samples of unit u, by line position p
                        p = u * U + j if u < 2 * n else (2 * N if u == 2 * n else 2 * N + 1 + (u - 2 * n - 1) * U + j)
                        xx, yy = (-1, 2 * N - 1 - p) if p < 2 * N else ((-1, -1) if p == 2 * N else (p - 2 * N - 1, -1))
                        other[y0 + yy, x0 + xx] = rng.integers(0, 1 << B)
                got = _predict(O, other, x0, y0, N, luma, mode, flags, B)
                assert np.array_equal(got, want), (N, luma, mode, flags, hex(dep))
    assert pruned > 0  # the mask does rule units out


@pytest.mark.parametrize("pic", [(416, 240), (200, 136), (64, 64), (1920, 1080), (72, 200)])
def test_availability_closed_form(pic):
    """The closed form of the neighbour-availability mask (a Z-order argument on the block's position, used by the device plan
    builder) against the unit-by-unit rule the kernels use and against the oracle's rule (pinned on the reference's getPU* walk
    by test_oracle_vs_ref.py), at EVERY aligned block position and size of pictures that cut the last CTU row and column."""
    O, L = ol.oracle(), capi.lib()
    w, h = pic
    flags = np.zeros(65, np.uint8)
    step = 1 if w * h < 300000 else 5  # the large picture: every fifth position per size (all CTU columns and rows still occur)
    for size in (4, 8, 16, 32):
        n = size // 4
        pos = [(x, y) for y in range(0, h - size + 1, size) for x in range(0, w - size + 1, size)]
        for (x, y) in pos[::step] + pos[-3:]:
            a = L.hmx_intra_avail_mask(x, y, size, w, h, 0)
            b = L.hmx_intra_avail_mask(x, y, size, w, h, 1)
            assert a == b, (pic, size, x, y, hex(a), hex(b))
            O.hmo_intra_avail(x, y, size, w, h, 64, flags)
            o = sum(int(flags[u]) << u for u in range(4 * n + 1))
            assert a == o, ("oracle", pic, size, x, y, hex(a), hex(o))

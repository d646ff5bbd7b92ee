"""Synthetic workloads for the parity tests and bench.py: seeded pictures,
transform-block (TU) tilings in the reference's coding order, PU lists with clipped MVs.

Blocks follow the reference's structure for 4:2:0, CTU 64, CU 64..8, TU 32..4
(cfg/encoder_intra_main.cfg: QuadtreeTULog2MaxSize 5, MinSize 2): per CU the luma TUs in Z-order, then
the Cb TUs, then the Cr TUs (ENC/TEncSearch.cpp:1394-1700, 2160-2400); an 8x8 CU with four 4x4 luma
TUs carries one 4x4 TU per chroma plane (TComTrQuant.cpp:1467-1476)."""
import numpy as np

from .capi import PU_DTYPE, TU_DTYPE, TU_TRANSFORM_SKIP, EstBits

CHROMA_MODES = (0, 26, 10, 1)  # planar, vertical, horizontal, DC (+ DM = luma mode)


def _cu_blocks(x, y, cu, tu, out):
    """One CU at (x,y) of size cu with luma TU size tu: append (x, y, log2n, plane) in coding order."""
    n = cu // tu
    lg = int(np.log2(tu))
    zs = _zorder(n)
    for (j, i) in zs:
        out.append((x + i * tu, y + j * tu, lg, 0))
    ctu_c = max(tu // 2, 4)
    nc = (cu // 2) // ctu_c
    lgc = int(np.log2(ctu_c))
    for pl in (1, 2):
        for (j, i) in _zorder(nc):
            out.append((x // 2 + i * ctu_c, y // 2 + j * ctu_c, lgc, pl))


_zcache = {}


def _zorder(n):
    if n not in _zcache:
        if n == 1:
            _zcache[n] = [(0, 0)]
        else:
            h = _zorder(n // 2)
            m = n // 2
            _zcache[n] = ([(j, i) for (j, i) in h] + [(j, i + m) for (j, i) in h] + [(j + m, i) for (j, i) in h] +
                          [(j + m, i + m) for (j, i) in h])
    return _zcache[n]


def _quadtree(rng, x, y, size, wa, ha, out, tiling):
    """Recursive CU quadtree of one CTU; wa/ha = picture extent relative to the CTU origin."""
    if x >= wa or y >= ha:
        return
    crosses = x + size > wa or y + size > ha
    if tiling == "mix":
        split = size > 8 and (crosses or rng.random() < {64: 0.85, 32: 0.6, 16: 0.5}[size])
    else:
        split = size > max(int(tiling), 8) or (crosses and size > 8)
    if split:
        h = size // 2
        for (dy, dx) in ((0, 0), (0, h), (h, 0), (h, h)):
            _quadtree(rng, x + dx, y + dy, h, wa, ha, out, tiling)
        return
    if tiling == "mix":
        choices = [t for t in (32, 16, 8, 4) if t <= size and t >= size // 4]
        tu = int(rng.choice(choices))
    else:
        tu = min(int(tiling), size, 32)
    _cu_blocks(x, y, size, tu, out)


def make_tus(seed, pic_w, pic_h, tiling="mix", n_patterns=6, ts_prob=0.15, ctu=64):
    """TU list of one picture in coding order.  tiling: 4/8/16/32 (uniform) or 'mix' (random quadtrees,
    n_patterns distinct CTU patterns reused over the picture; edge CTUs get their own)."""
    assert pic_w % 8 == 0 and pic_h % 8 == 0
    rng = np.random.default_rng(seed)
    cw, ch = -(-pic_w // ctu), -(-pic_h // ctu)
    full = []
    for _ in range(n_patterns if tiling == "mix" else 1):
        o = []
        _quadtree(rng, 0, 0, ctu, ctu, ctu, o, tiling)
        full.append(np.array(o, np.int32))
    parts = []
    for Y in range(ch):
        for X in range(cw):
            wa, ha = min(ctu, pic_w - X * ctu), min(ctu, pic_h - Y * ctu)
            if wa == ctu and ha == ctu:
                pat = full[int(rng.integers(len(full)))]
            else:
                o = []
                _quadtree(rng, 0, 0, ctu, wa, ha, o, tiling)
                pat = np.array(o, np.int32)
            p = pat.copy()
            sh = (p[:, 3] != 0).astype(np.int32)
            p[:, 0] += (X * ctu) >> sh
            p[:, 1] += (Y * ctu) >> sh
            parts.append(p)
    a = np.concatenate(parts)
    tus = np.zeros(len(a), TU_DTYPE)
    tus["x"], tus["y"], tus["log2n"], tus["plane"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    luma = tus["plane"] == 0
    modes = rng.integers(0, 35, len(a)).astype(np.uint8)
    cm = np.array(CHROMA_MODES + (34,), np.uint8)[rng.integers(0, 5, len(a))]
    dm = rng.random(len(a)) < 0.3
    tus["mode"] = np.where(luma, modes, np.where(dm, modes, cm))
    ts = (tus["log2n"] == 2) & (rng.random(len(a)) < ts_prob)
    tus["flags"] = np.where(ts, TU_TRANSFORM_SKIP, 0).astype(np.uint8)
    return tus


def make_planes(seed, w, h, bit_depth, kind="noise"):
    """Seeded 4:2:0 picture as three int16 arrays (h x w, h/2 x w/2 x2).  'texture' = smooth
    sinusoid + noise (gives mostly small residuals), 'noise' = uniform over the legal range."""
    rng = np.random.default_rng(seed)
    mx = (1 << bit_depth) - 1
    out = []
    for (pw, ph) in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
        if kind == "noise":
            p = rng.integers(0, mx + 1, (ph, pw))
        else:
            # sin(x / 23 + seed) * cos(y / 17) as an outer product of the two 1-D factors: the same doubles multiplied,
            # the same samples as the element-wise form over a 2-D grid, at a fraction of the time for a 2160p picture
            sx = np.sin(np.arange(pw) / 23.0 + seed)[None, :]
            cy = np.cos(np.arange(ph) / 17.0)[:, None]
            p = (mx / 2) * (1 + 0.6 * sx * cy) + rng.normal(0, 6 * mx / 255, (ph, pw))
            p = np.clip(np.rint(p), 0, mx)
        out.append(p.astype(np.int16))
    return out


PU_SHAPES = ((64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16), (8, 8), (8, 4),
             (4, 8), (64, 16), (64, 48), (16, 64), (48, 64), (32, 8), (32, 24), (8, 32), (24, 32), (16, 4), (16, 12),
             (4, 16), (12, 16))


def make_pus(seed, pic_w, pic_h, n_refs=1, bi_frac=0.0, mv_range=64, ctu=64):
    """One PU list covering the picture: every CTU is cut into rows of one random PU shape
    (AMP shapes included); MVs uniform in +-mv_range pixels at quarter-pel, clipped like
    TComDataCU::clipMv (TComDataCU.cpp:3505-3517)."""
    rng = np.random.default_rng(seed)
    rows = []
    for Y in range(0, pic_h, ctu):
        for X in range(0, pic_w, ctu):
            pw, ph = PU_SHAPES[int(rng.integers(len(PU_SHAPES)))]
            for y in range(Y, min(Y + ctu, pic_h), ph):
                for x in range(X, min(X + ctu, pic_w), pw):
                    w = min(pw, X + ctu - x, pic_w - x)  # PUs partition the CTU (no overlap)
                    h = min(ph, Y + ctu - y, pic_h - y)
                    rows.append((x, y, w, h))
    a = np.array(rows, np.int32)
    n = len(a)
    pus = np.zeros(n, PU_DTYPE)
    pus["x"], pus["y"], pus["w"], pus["h"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    bi = rng.random(n) < bi_frac
    pus["ref0"] = rng.integers(0, n_refs, n)
    pus["ref1"] = np.where(bi, rng.integers(0, n_refs, n), 255)
    for k in ("mv0", "mv1"):
        mvx = rng.integers(-4 * mv_range, 4 * mv_range + 1, n)
        mvy = rng.integers(-4 * mv_range, 4 * mv_range + 1, n)
        # clipMv with the PU origin as CU origin (a CU's PUs share the CU origin; using the PU's is
        # stricter and keeps every read inside the 80/40-sample margins)
        pus[k + "x"] = np.clip(mvx, (-ctu - 8 - a[:, 0] + 1) * 4, (pic_w + 8 - a[:, 0] - 1) * 4)
        pus[k + "y"] = np.clip(mvy, (-ctu - 8 - a[:, 1] + 1) * 4, (pic_h + 8 - a[:, 1] - 1) * 4)
    return pus


# --- what RDOQ takes from the encoder's live state (hmx_set_rdoq), synthesised --------------------------------------
def make_est_bits(seed):
    """A plausible bit-estimate table (estBitsSbacStruct): every context holds a probability p of the bin being 1,
    bits[0] = -log2(1 - p), bits[1] = -log2(p) in 1/32768 bit (the scale of TEncBinCABAC's entropy bits); the last-position
    prefixes cost more the further out they reach."""
    rng = np.random.default_rng(seed)
    e = EstBits()

    def pair(dst):
        p = float(rng.uniform(0.03, 0.97))
        dst[0] = int(round(-np.log2(1 - p) * 32768))
        dst[1] = int(round(-np.log2(p) * 32768))

    for name, n in (("significantCoeffGroupBits", 2), ("significantBits", 42), ("greaterOneBits", 24), ("levelAbsBits", 6),
                    ("blockCbpBits", 15), ("blockRootCbpBits", 4)):
        arr = getattr(e, name)
        for i in range(n):
            pair(arr[i])
    for i in range(32):
        e.lastXBits[i] = int(rng.integers(8000, 60000) * (1 + i // 4))
        e.lastYBits[i] = int(rng.integers(8000, 60000) * (1 + i // 4))
    pair(e.scanZigzag)
    pair(e.scanNonZigzag)
    return e


def rdoq_lambdas(qp):
    """m_dLambda of an I slice as TEncSlice::initEncSlice forms it for the all-intra cfgs (TEncSlice.cpp:260-330: QP factor
    0.57, no B pictures), luma; the chroma blocks' multiplier is the luma one divided by the chroma weight
    2^((qp - qp_chroma) / 3) (TEncSlice.cpp:380-395)."""
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    mid = (29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37)
    qpc = qp if qp < 30 else (qp - 6 if qp >= 43 else mid[qp - 30])
    return lam, lam / 2.0 ** ((qp - qpc) / 3.0)


def with_cbf_ctx(tus):
    """The context of every block's coded-block flag in hmx_tu::flags bits 4..7 (TComDataCU::getCtxQtCbf: luma 1 at
    transform depth 0 else 0; chroma = the transform depth, behind the five luma contexts).  The synthetic decisions do
    not carry CU sizes: blocks of 16 and more count as depth 0, 8 as depth 1, 4 as depth 2."""
    t = tus.copy()
    depth = np.where(t["log2n"] >= 4, 0, np.where(t["log2n"] == 3, 1, 2))
    ctx = np.where(t["plane"] == 0, (depth == 0).astype(np.int64), 5 + depth)
    t["flags"] = (t["flags"] & 15) | (ctx.astype(np.uint8) << 4)
    return t

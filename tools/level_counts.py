"""Dependency levels of the first picture of the REAL reference streams under tests/golden (decisions of the reference encoder):
with every available neighbour as a dependency (round 1) and with what the block's mode reads (hmx_intra_dependency_mask).
No GPU.  python3 tools/level_counts.py"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from thevc_amd import capi  # noqa: E402
from thevc_amd.decisions import load_pictures, split_blocks  # noqa: E402

O, L = ol.oracle(), capi.lib()


def levels(tus, w, h, by_mode):
    uw, uh = (w + 63) // 64 * 16 + 2, (h + 63) // 64 * 16 + 2
    g = [np.zeros((uh, uw), np.int32) for _ in range(3)]
    top = 0
    for t in tus:
        pl = int(t["plane"])
        sh = 1 if pl else 0
        N = 1 << int(t["log2n"])
        lx, ly, ls = int(t["x"]) << sh, int(t["y"]) << sh, N << sh
        n, ux, uy = ls // 4, lx // 4, ly // 4
        flags = np.zeros(4 * n + 1, np.uint8)
        O.hmo_intra_avail(lx, ly, ls, w, h, 64, flags)
        m = sum(int(b) << u for u, b in enumerate(flags))
        if by_mode:
            m = L.hmx_intra_dependency_mask(N, int(pl == 0), int(t["mode"]), m)
        lv = 0
        for u in range(4 * n + 1):
            if (m >> u) & 1:
                qx, qy = (ux - 1, uy + 2 * n - 1 - u) if u < 2 * n else ((ux - 1, uy - 1) if u == 2 * n else (ux + (u - 2 * n - 1), uy - 1))
                lv = max(lv, int(g[pl][qy, qx]))
        g[pl][uy:uy + n, ux:ux + n] = lv + 1
        top = max(top, lv + 1)
    return top


for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stream_intra*.npz"))):
    p = next(iter(load_pictures(path)))
    tus, _ = split_blocks(p)
    a, b = levels(tus, p["w"], p["h"], False), levels(tus, p["w"], p["h"], True)
    print(f"{os.path.basename(path)}: {p['w']}x{p['h']}, {len(tus)} blocks: {a} -> {b} levels ({a / b:.2f}x)")

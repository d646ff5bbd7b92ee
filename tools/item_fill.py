#!/usr/bin/env python3
"""tools/item_fill.py -- how full the packed schedule's wave-items are: for groups of I pictures (each its own plan), wave-items per
picture against blocks / slots (the count with every item full).  Needs a GPU (the plans are analysed through the library)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thevc_amd import capi, workload  # noqa: E402

ctx = capi.Context(bit_depth=10)
w, h, n = 3840, 2160, 8
pp = capi.PicParam(w, h, 32, 0, capi.I_SLICE, 1)
tus = [workload.make_tus(100 + i, w, h, "mix") for i in range(n)]
plans = ctx.intra_plans(tus, pp)
tabs = [ctx.plan_tables(p)[1] for p in plans]  # per level: start[4], count[4]
slots = [64, 16, 4, 1]
print("levels per picture", [len(t) for t in tabs])
for I in (1, 2, 4, 8):
    items, ideal, per_class = 0, 0.0, np.zeros(4)
    for g in range(0, n, I):
        L = max(len(t) for t in tabs[g:g + I])
        cnt = np.zeros((L, 4), np.int64)
        for t in tabs[g:g + I]:
            cnt[:len(t)] += t[:, 4:8]
        for s in range(4):
            k = int(np.ceil(cnt[:, s] / slots[s]).sum())
            items += k
            per_class[s] += k
            ideal += cnt[:, s].sum() / slots[s]
    print(f"group of {I}: wave-items per picture {items / n:.0f} (by class {np.round(per_class / n)}), with every item full {ideal / n:.0f}, ratio {items / ideal:.3f}")

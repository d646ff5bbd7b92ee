"""Shape of the level schedule of a synthetic picture: blocks and wavefronts per dependency level."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from thevc_amd import capi, workload  # noqa: E402

w, h, tiling = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
tiling = tiling if tiling == "mix" else int(tiling)
ctx = capi.Context(bit_depth=10)
tus = workload.make_tus(1, w, h, tiling)
plan = ctx.intra_plan(tus, capi.PicParam(w, h, 32, 0, capi.I_SLICE, 1))
L = capi.lib()
nb, nl, nd = C.c_int(), C.c_int(), C.c_int()
L.hmx_intra_plan_info(plan, C.byref(nb), C.byref(nl), C.byref(nd))
cnt = np.zeros((nl.value, 4), np.int64)
waves = np.zeros(nl.value, np.int64)
for l in range(nl.value):
    c4, nw = (C.c_uint32 * 4)(), C.c_uint32()
    L.hmx_intra_plan_level(plan, l, C.byref(c4), C.byref(nw))
    cnt[l] = list(c4)
    waves[l] = nw.value
ideal = (cnt / np.array([64, 8, 4, 1])).sum()
print(f"{w}x{h} tiling {tiling}: blocks {nb.value} levels {nl.value} diagonals {nd.value}")
print("blocks by size", cnt.sum(0).tolist(), "waves/picture", int(waves.sum()), "ideal", round(float(ideal)))
print("waves per level: min/median/mean/max", int(waves.min()), int(np.median(waves)), round(float(waves.mean()), 1), int(waves.max()))
q = np.percentile(waves, [10, 25, 75, 90]).tolist()
print("percentiles 10/25/75/90", q)

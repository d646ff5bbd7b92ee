"""35-mode intra fan-out of one 1080p picture's luma blocks: candidates written (hmx_batch_predIntra) vs
costed in registers (hmx_batch_predIntra_cost).  Numbers go to DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thevc_amd import capi, workload  # noqa: E402

B, w, h = 8, 1920, 1088
ctx = capi.Context(bit_depth=B)
L = capi.lib()
tus = workload.make_tus(3, w, h, "mix")
luma = tus[tus["plane"] == 0]
lst = ctx.tu_list(luma)
rec = capi.DevPicture(ctx, w, h).upload(workload.make_planes(1, w, h, B, "texture"))
org = capi.DevPicture(ctx, w, h).upload(workload.make_planes(2, w, h, B, "texture"))
pp = capi.PicParam(w, h, 32, 0, capi.I_SLICE, 1)
d_modes = ctx.to_device(np.arange(35, dtype=np.uint8))
d_cost = ctx.alloc(4 * len(luma) * 35)
fan = ctx.alloc(2 * 35 * w * h)
fp = capi.Pic()
fp.plane[0], fp.stride[0] = fan.ptr, w
elems = (C.c_size_t * 3)(w * h, 0, 0)


def timed(fn, n=5):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n


t_cost = timed(lambda: ctx._chk(L.hmx_batch_predIntra_cost(ctx.h, lst, C.byref(rec.as_pic()), C.byref(org.as_pic()), C.byref(pp), d_modes.ptr, 35, d_cost.ptr)))
t_pred = timed(lambda: ctx._chk(L.hmx_batch_predIntra(ctx.h, lst, C.byref(rec.as_pic()), C.byref(fp), C.byref(pp), d_modes.ptr, 35, C.byref(elems))))
px = w * h
print(f"{len(luma)} luma blocks x 35 modes: cost only {t_cost * 1e3:.2f} ms ({35 * px / t_cost / 1e9:.1f} G predicted samples/s), "
      f"candidates written {t_pred * 1e3:.2f} ms")

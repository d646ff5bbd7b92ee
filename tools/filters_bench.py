"""Throughput of the picture-level kernels either side of the chain on one 2160p picture set: deblocking (strengths +
two passes), SAO, YUV unpack / pack.  Numbers go to DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thevc_amd import capi, workload  # noqa: E402

B, w, h, NP = 10, 3840, 2160, 16
ctx = capi.Context(bit_depth=B)
L = capi.lib()
rng = np.random.default_rng(1)
pics = [capi.DevPicture(ctx, w, h).upload(workload.make_planes(i, w, h, B, "texture")) for i in range(2)]
pics += [capi.DevPicture(ctx, w, h).zero() for _ in range(NP - 2)]
outs = [capi.DevPicture(ctx, w, h).zero() for _ in range(NP)]
uw, uh = w // 4, h // 4
units = np.zeros(uw * uh, np.dtype([("intra", "u1"), ("cbf", "u1"), ("ref", "i1", 2), ("mv", "<i2", (2, 2))]))
units["intra"] = rng.random(uw * uh) < 0.2
units["cbf"] = rng.random(uw * uh) < 0.4
units["mv"] = rng.integers(-8, 9, (uw * uh, 2, 2))
edge = ((np.arange(uw)[None, :] % 2 == 0) * 3).astype(np.uint8).repeat(uh, 0).copy()
d_units, d_ev, d_eh = ctx.to_device(units), ctx.to_device(edge), ctx.to_device(edge)
d_bv, d_bh = ctx.alloc(uw * uh), ctx.alloc(uw * uh)
d_qp = ctx.to_device(np.full(uw * uh, 32, np.int8))
n_lcu = 60 * 34
sao = np.zeros((3, n_lcu), np.dtype([("type", "i1"), ("band", "u1"), ("offset", "i1", 4)]))
sao["type"] = rng.integers(-1, 5, (3, n_lcu))
sao["offset"] = rng.integers(-3, 4, (3, n_lcu, 4))
d_sao = ctx.to_device(np.ascontiguousarray(sao))
nbytes = L.hmx_yuv_frame_bytes(w, h, 10)
d_file = ctx.alloc(nbytes)


def timed(fn, reps=3):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps / NP


def per_pic(f):
    def g():
        for i in range(NP):
            f(i)
    return g


px = w * h
t = timed(per_pic(lambda i: ctx._chk(L.hmx_deblock_strengths(ctx.h, d_units.ptr, d_ev.ptr, d_eh.ptr, w, h, 1, d_bv.ptr, d_bh.ptr))))
print(f"deblock strengths   {t * 1e6:8.1f} us/picture  {px / t / 1e9:6.1f} Gpx/s")
t = timed(per_pic(lambda i: ctx._chk(L.hmx_deblock_picture(ctx.h, C.byref(pics[i].as_pic()), w, h, d_bv.ptr, d_bh.ptr, d_qp.ptr, None, 0, 0))))
print(f"deblock two passes  {t * 1e6:8.1f} us/picture  {px / t / 1e9:6.1f} Gpx/s")
t = timed(per_pic(lambda i: ctx._chk(L.hmx_sao_picture(ctx.h, C.byref(pics[i].as_pic()), C.byref(outs[i].as_pic()), w, h, d_sao.ptr, n_lcu))))
print(f"SAO                 {t * 1e6:8.1f} us/picture  {px / t / 1e9:6.1f} Gpx/s  ({6 * px / t / 1e9:.0f} GB/s of 2 B read + 2 B written per sample)")
t = timed(per_pic(lambda i: ctx._chk(L.hmx_yuv_pack(ctx.h, C.byref(pics[i].as_pic()), w, h, 0, 0, 10, d_file.ptr))))
print(f"YUV pack (10-bit)   {t * 1e6:8.1f} us/picture  {px / t / 1e9:6.1f} Gpx/s")
t = timed(per_pic(lambda i: ctx._chk(L.hmx_yuv_unpack(ctx.h, d_file.ptr, 10, C.byref(outs[i].as_pic()), w, h, 0, 0))))
print(f"YUV unpack (10-bit) {t * 1e6:8.1f} us/picture  {px / t / 1e9:6.1f} Gpx/s")

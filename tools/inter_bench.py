"""Throughput of the inter-picture stages alone on a batch of 1080p pictures: motion compensation over PU lists and the
fused residual transform + reconstruction over a TU list (the two calls RAPipeline makes per GOP position).
  python tools/inter_bench.py [n_pictures] [mc|tq|all]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thevc_amd import capi, workload  # noqa: E402

NP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
what = sys.argv[2] if len(sys.argv) > 2 else "all"
B, w, h, M = 8, 1920, 1080, 80
ctx = capi.Context(bit_depth=B)
L = capi.lib()
refs = [capi.DevPicture(ctx, w, h, M, M).upload(workload.make_planes(i, w, h, B, "texture")) for i in range(2)]
for r in refs:
    ctx._chk(L.hmx_pic_extend_border(ctx.h, C.byref(r.as_pic()), w, h, M, M))
org = capi.DevPicture(ctx, w, h).upload(workload.make_planes(7, w, h, B, "texture"))
pred = [capi.DevPicture(ctx, w, h).zero() for _ in range(NP)]
rec = [capi.DevPicture(ctx, w, h, M, M).zero() for _ in range(NP)]
lev = [capi.DevPicture(ctx, w, h, dtype=np.int32) for _ in range(NP)]
pus = workload.make_pus(3, w, h, n_refs=2, bi_frac=0.5)
d_pus = ctx.to_device(pus)
tus = workload.make_tus(5, w, h, "mix", ts_prob=0.0)
tus["flags"] = capi.TU_INTER
tl = ctx.tu_list(tus)
pp = capi.PicParam(w, h, 32, 0, capi.B_SLICE, 1)

ref_arr = (capi.Pic * 2)(*[r.as_pic() for r in refs])
mc = (capi.McJob * NP)()
a_pred, a_rec, a_org, a_lev = (capi.Pic * NP)(), (capi.Pic * NP)(), (capi.Pic * NP)(), (capi.Levels * NP)()
for q in range(NP):
    a_pred[q], a_rec[q], a_org[q], a_lev[q] = pred[q].as_pic(), rec[q].as_pic(), org.as_pic(), lev[q].as_pic()
    mc[q].d_pus, mc[q].n_pus, mc[q].refs, mc[q].n_refs = d_pus.ptr, len(pus), ref_arr, 2
    mc[q].dst = C.pointer(a_pred[q])
    mc[q].pic_w, mc[q].pic_h = w, h


def timed(fn, reps=5):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


px = NP * w * h * 1.5
if what in ("mc", "all"):
    t = timed(lambda: ctx._chk(L.hmx_batch_motionCompensation_multi(ctx.h, NP, mc)))
    print(f"motion compensation (50% bi)  {t * 1e3:7.3f} ms / {NP} pictures  {px / t / 1e9:6.1f} Gsamples/s")
if what in ("tq", "all"):
    t = timed(lambda: ctx._chk(L.hmx_batch_residual_transform_recon_multi(ctx.h, tl, NP, a_org, a_pred, a_lev, a_rec, None, C.byref(pp))))
    print(f"residual T/Q/IQ/IT + recon    {t * 1e3:7.3f} ms / {NP} pictures  {px / t / 1e9:6.1f} Gsamples/s")
if what in ("border", "all"):
    t = timed(lambda: ctx._chk(L.hmx_pic_extend_border_multi(ctx.h, NP, a_rec, w, h, M, M)))
    print(f"border extension              {t * 1e3:7.3f} ms / {NP} pictures  {px / t / 1e9:6.1f} Gsamples/s")

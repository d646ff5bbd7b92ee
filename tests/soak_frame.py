"""Soak test of the whole-picture chain against the oracle: random picture sizes (multiples of 8, cut CTUs), bit depths,
QPs, CTU sizes, picture counts, both level kernels.  python tests/soak_frame.py [iterations] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from thevc_amd import capi, workload  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O, L = ol.oracle(), capi.lib()
P3, I3 = C.c_void_p * 3, C.c_int * 3
bad = 0
for it in range(iters):
    B = int(rng.choice([8, 10, 12]))
    ctu = int(rng.choice([64, 64, 32, 16]))
    w, h = 8 * int(rng.integers(2, 40)), 8 * int(rng.integers(2, 30))
    qp = int(rng.integers(0, 52))
    cqo = int(rng.integers(-12, 13))
    sh = int(rng.integers(0, 2))
    n_pics = int(rng.choice([1, 2, 5, 9, 66]))
    os.environ["HMX_INTRA_ACROSS"] = str(int(rng.integers(0, 2)))
    os.environ["HMX_PIPELINE_CONV"] = str(int(rng.integers(0, 2)))
    ctx = capi.Context(bit_depth=B, ctu_size=ctu)
    tus = workload.make_tus(int(rng.integers(1 << 30)), w, h, "mix", ctu=ctu)
    plan = ctx.intra_plan(tus, capi.PicParam(w, h, qp, cqo, capi.I_SLICE, sh))
    orgs = [workload.make_planes(int(rng.integers(1 << 30)), w, h, B, "texture" if i % 2 else "noise") for i in range(n_pics)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n_pics)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n_pics)]
    A = lambda lst, T: (T * n_pics)(*[x.as_pic() for x in lst])
    ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, n_pics, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    ok = True
    for i in list(range(min(n_pics, 3))) + ([n_pics - 1] if n_pics > 3 else []):
        cfg = ol.FrameCfg(w, h, ctu, B, qp, cqo, sh)
        rec = [np.zeros_like(p) for p in orgs[i]]
        lev = [np.zeros(p.shape, np.int32) for p in orgs[i]]
        t = np.ascontiguousarray(tus, ol.TU_DTYPE)
        st = I3(w, w // 2, w // 2)
        O.hmo_intra_frame_encode(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in orgs[i]]), st,
                                 P3(*[p.ctypes.data for p in rec]), st, P3(*[p.ctypes.data for p in lev]))
        g_rec, g_lev = d_rec[i].download(), d_lev[i].download()
        ok = ok and all(np.array_equal(g_rec[p], rec[p]) and np.array_equal(g_lev[p], lev[p]) for p in range(3))
    print(f"{it:3d} B={B} ctu={ctu} {w}x{h} qp={qp} cqo={cqo} sbh={sh} pics={n_pics} across={os.environ['HMX_INTRA_ACROSS']} "
          f"pipe={os.environ['HMX_PIPELINE_CONV']} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += int(not ok)
    L.hmx_intra_plan_destroy(ctx.h, plan)
    for d in d_org + d_rec + d_lev:
        d.free()
    ctx.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)

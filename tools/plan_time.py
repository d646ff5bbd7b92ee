"""How long do the host-side steps around a whole-picture call take?  hmx_intra_plan_create (dependency analysis of one picture's
decisions + upload) and the first call's table build, 2160p.  python3 tools/plan_time.py [pictures]"""
import sys, time, ctypes as C
sys.path.insert(0, ".")
from thevc_amd import capi, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, B, qp = 3840, 2160, 10, 32
ctx = capi.Context(bit_depth=B)
pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
tus = [workload.make_tus(500 + i, w, h, "mix") for i in range(n)]
t0 = time.perf_counter()
plans = [ctx.intra_plan(t, pp) for t in tus]
ctx.sync()
dt = time.perf_counter() - t0
print(f"hmx_intra_plan_create: {dt / n * 1e3:.1f} ms per 2160p picture ({sum(len(t) for t in tus) / n / 1e3:.0f}k blocks)")
for p in plans:
    capi.lib().hmx_intra_plan_destroy(ctx.h, p)
m = max(n, 32)
tus = [tus[i % n] for i in range(m)]
t0 = time.perf_counter()
plans = ctx.intra_plans(tus, pp)
ctx.sync()
dt = time.perf_counter() - t0
import os
print(f"hmx_intra_plan_create_multi: {dt / m * 1e3:.1f} ms per picture over {m} pictures on {os.cpu_count()} logical cores")

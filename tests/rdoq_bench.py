"""Throughput of the RDOQ block-list entry (8x8 and larger blocks shared by the lanes of a wave in LDS, k_rdoq_tiles; HMX_RDOQ_LANE=1: one lane per block) on one 1080p picture's worth of blocks,
next to the CPU oracle on a sample.  Not part of bench.py's headline; numbers go to DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from thevc_amd import capi, workload  # noqa: E402

B, w, h, qp = 8, 1920, 1080, 32
h = 1088 if len(sys.argv) < 2 else int(sys.argv[1])
tiling = "mix" if len(sys.argv) < 3 else (sys.argv[2] if sys.argv[2] == "mix" else int(sys.argv[2]))
ctx = capi.Context(bit_depth=B)
L, O = capi.lib(), ol.oracle()
rng = np.random.default_rng(5)
tus = workload.make_tus(3, w, h, tiling)
n = len(tus)
coef = [rng.integers(-400, 401, (h >> (1 if p else 0), w >> (1 if p else 0))).astype(np.int32) for p in range(3)]
for c in coef:  # transform coefficients decay with frequency: thin out the tail
    c[rng.random(c.shape) < 0.7] //= 16
ests = [ol.make_est_bits(rng) for _ in range(4)]
est_arr = (capi.EstBits * 4)(*[capi.EstBits.from_buffer_copy(bytes(e)) for e in ests])
side = (capi.RdoqSide * n)()
for i, t in enumerate(tus):
    side[i].est_idx = int(t["log2n"]) - 2
    side[i].cbf_ctx = 1 + (5 if t["plane"] else 0)
d_coef = capi.DevPicture(ctx, w, h, dtype=np.int32).upload(coef)
d_lev = capi.DevPicture(ctx, w, h, dtype=np.int32).zero()
pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
t_c = np.ascontiguousarray(tus, capi.TU_DTYPE)


def run():
    ctx._chk(L.hmx_batch_xRateDistOptQuant(ctx.h, t_c.ctypes.data, side, n, C.byref(d_coef.as_pic()), C.byref(d_lev.as_pic()), None,
                                           C.byref(pp), est_arr, 4, 58.0, 47.0))


run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(3):
    run()
ctx.sync()
gpu = (time.perf_counter() - t0) / 3
# the same blocks K times in one list: enough 32x32 blocks to fill the chip (one picture has 700, the chip runs 4096 waves)
K = 1 if len(sys.argv) < 4 else int(sys.argv[3])
if K > 1:
    t_k = np.ascontiguousarray(np.tile(tus, K), capi.TU_DTYPE)
    side_k = (capi.RdoqSide * (n * K))()
    for r in range(K):
        C.memmove(C.addressof(side_k) + r * C.sizeof(side), side, C.sizeof(side))

    def run_k():
        ctx._chk(L.hmx_batch_xRateDistOptQuant(ctx.h, t_k.ctypes.data, side_k, n * K, C.byref(d_coef.as_pic()), C.byref(d_lev.as_pic()), None,
                                               C.byref(pp), est_arr, 4, 58.0, 47.0))

    run_k()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        run_k()
    ctx.sync()
    gk = (time.perf_counter() - t0) / 3
    print(f"{K} pictures' blocks in one list: {gk * 1e3:.2f} ms = {K * w * h / gk / 1e6:.1f} Mpx/s")
samples = sum((1 << (2 * int(t["log2n"]))) for t in tus)
# CPU oracle on the first 2000 blocks
m = min(n, 2000)
t0 = time.perf_counter()
cs = 0
for i in range(m):
    t = tus[i]
    N, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
    q = O.hmo_setQPforQuant(qp, int(p != 0), 0, 0)
    cfg = ol.RdoqCfg(q.per, q.rem, int(p == 0), 1, O.hmo_coef_scan_idx(N, int(p == 0), 1, int(t["mode"])), 0, side[i].cbf_ctx, 1,
                     58.0 if p == 0 else 47.0)
    ol.o_rdoq(coef[p][y:y + N, x:x + N], N, B, cfg, ests[side[i].est_idx])
    cs += N * N
cpu = time.perf_counter() - t0
print(f"blocks {n} samples {samples}: GPU {gpu * 1e3:.2f} ms/picture = {samples / gpu / 1e6:.1f} Msamples/s "
      f"({w * h / gpu / 1e6:.1f} Mpx/s); CPU oracle {cs / cpu / 1e6:.2f} Msamples/s on {m} blocks (python call overhead included)")

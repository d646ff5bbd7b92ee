"""CPU-side checks: the C-ABI library exports every symbol include/hmx.h declares (no compute calls),
host-side helpers, and the multi-rank plumbing of bench.py rehearsed with gloo at world_size 2."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(ROOT, "include", "hmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(hmx_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 35
    lib = C.CDLL(os.path.join(ROOT, "thevc_amd", "libhmx.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_capi_binds_and_fails_loudly_without_gpu():
    from thevc_amd import capi
    L = capi.lib()  # every bound symbol resolves
    q = L.hmx_setQPforQuant(32, capi.TEXT_LUMA, 0, 0)
    assert (q.qp, q.per, q.rem, q.bits) == (32, 5, 2, 20)
    q = L.hmx_setQPforQuant(32, capi.TEXT_CHROMA, 12, 0)  # 10-bit chroma: table 32 -> 31, + 12
    assert (q.qp, q.per, q.rem) == (43, 7, 1)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(capi.HmxError):
            capi.Context(bit_depth=8)  # no device: must raise, never fall back to the CPU


def test_clip_mv_host_helper_matches_oracle():
    import oracle_lib as ol
    from thevc_amd import capi
    L, O = capi.lib(), ol.oracle()
    rng = np.random.default_rng(0)
    for _ in range(500):
        mv = rng.integers(-5000, 5000, 2)
        cu = rng.integers(0, 1900, 2)
        a, b = C.c_int(int(mv[0])), C.c_int(int(mv[1]))
        c, d = C.c_int(int(mv[0])), C.c_int(int(mv[1]))
        L.hmx_clipMv(C.byref(a), C.byref(b), int(cu[0]), int(cu[1]), 1920, 1080, 64)
        O.hmo_clipMv(C.byref(c), C.byref(d), int(cu[0]), int(cu[1]), 1920, 1080, 64)
        assert (a.value, b.value) == (c.value, d.value)


def test_workload_generators_are_deterministic_and_tile_the_picture():
    from thevc_amd import workload
    for (w, h, tiling) in ((416, 240, "mix"), (128, 64, 4), (200, 136, "mix"), (256, 128, 32)):
        t1, t2 = workload.make_tus(3, w, h, tiling), workload.make_tus(3, w, h, tiling)
        assert np.array_equal(t1, t2)
        cover = [np.zeros((h, w), np.int32), np.zeros((h // 2, w // 2), np.int32), np.zeros((h // 2, w // 2), np.int32)]
        for t in t1:
            n = 1 << int(t["log2n"])
            cover[int(t["plane"])][int(t["y"]):int(t["y"]) + n, int(t["x"]):int(t["x"]) + n] += 1
        assert all((c == 1).all() for c in cover)  # every sample of every plane in exactly one block
    pus = workload.make_pus(1, 256, 192, 2, 0.5)
    cover = np.zeros((192, 256), np.int32)
    for u in pus:
        cover[int(u["y"]):int(u["y"]) + int(u["h"]), int(u["x"]):int(u["x"]) + int(u["w"])] += 1
    assert (cover == 1).all()


def _rank_main(rank, world, port, out_q):
    """One rank of the rehearsal: gloo backend, the CPU oracle stands in for the GPU step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch.distributed as dist
    import bench
    import oracle_lib as ol
    from thevc_amd import workload
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, B, qp, F = 128, 64, 8, 32, 3
    tus = workload.make_tus(1, w, h, "mix")
    seeds = bench.rank_picture_seeds(rank, F)
    pics = [workload.make_planes(s, w, h, B, "texture") for s in seeds]
    dist.barrier()
    t0 = time.perf_counter()
    recs = [ol.o_intra_frame_encode(tus, w, h, B, qp, p)[0] for p in pics]
    time.sleep(0.05 * rank)  # uneven ranks: the slowest one must set the time
    dist.barrier()
    dt_local = time.perf_counter() - t0
    dt = bench.max_over_ranks(dt_local, world, "cpu")
    value = bench.whole_job_value(w * h * F, 1, world, dt)
    digest = int(sum(int(r[0].astype(np.int64).sum()) for r in recs))
    out_q.put((rank, seeds, dt_local, dt, value, digest))
    dist.destroy_process_group()


def test_two_rank_rehearsal_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, s0, l0, d0, v0, g0), (r1, s1, l1, d1, v1, g1) = res
    assert not set(s0) & set(s1)            # ranks own disjoint pictures (no data-path collective)
    assert abs(d0 - d1) < 1e-9              # both ranks agree on the MAX time
    assert d0 >= max(l0, l1) - 1e-9
    assert abs(v0 - 2 * 128 * 64 * 3 / d0 / 1e6) < 1e-6  # value is the whole-job aggregate
    assert g0 != g1                          # different pictures were really processed


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its ranks itself (a child torch.distributed.run, before the
    parent touches any GPU).  Rehearsed on the CPU: gloo, the CPU oracle for the device step, the random-access
    boundary-picture exchange through the same ra_pipeline.run_exchange the GPU path calls; rank 0 prints ONE JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["rehearsal"] == {"ranks": 2, "exchange_ops_rank0": d["rehearsal"]["exchange_ops_rank0"], "boundary_pictures_arrived": True}
    assert d["rehearsal"]["exchange_ops_rank0"] > 0

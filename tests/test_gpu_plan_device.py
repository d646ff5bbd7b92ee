"""Plans built on the device (hmx_intra_plan_create_device) against the host's analysis (hmx_intra_plan_create): the same
tables entry for entry, and the same pictures through the whole-picture chain.  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from thevc_amd import capi, workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[8, 10])
def ctx(request):
    c = capi.Context(bit_depth=request.param)
    yield c
    c.close()


def _device_plans(ctx, tus_list, pp):
    cat = np.ascontiguousarray(np.concatenate(tus_list), capi.TU_DTYPE)
    offs = np.concatenate([[0], np.cumsum([len(t) for t in tus_list])])
    d = ctx.to_device(cat)
    plans = ctx.intra_plans_device(d.ptr, offs, pp)
    d.free()  # the lists may go as soon as the call returns
    return plans


def _same_tables(ctx, a, b, what):
    L = capi.lib()
    ia, ib = [(C.c_int(), C.c_int(), C.c_int()) for _ in range(2)]
    L.hmx_intra_plan_info(a, *[C.byref(x) for x in ia])
    L.hmx_intra_plan_info(b, *[C.byref(x) for x in ib])
    assert [x.value for x in ia] == [x.value for x in ib], (what, "blocks / levels / diagonals")
    ba, la = ctx.plan_tables(a)
    bb, lb = ctx.plan_tables(b)
    assert np.array_equal(la, lb), (what, "level table")
    assert np.array_equal(ba, bb), (what, "sorted block list", int(np.flatnonzero(ba != bb)[0]))
    for lvl in (0, ia[1].value // 2, ia[1].value - 1):  # the host-side view of a device plan is fetched on demand
        ca, cb, wa, wb = (C.c_uint32 * 4)(), (C.c_uint32 * 4)(), C.c_uint32(), C.c_uint32()
        assert L.hmx_intra_plan_level(a, lvl, ca, C.byref(wa)) == 0 and L.hmx_intra_plan_level(b, lvl, cb, C.byref(wb)) == 0
        assert list(ca) == list(cb) and wa.value == wb.value, (what, "level", lvl)


@pytest.mark.parametrize("pic", [(416, 240), (200, 136), (64, 64), (1280, 720)])
def test_device_plans_equal_host_plans(ctx, pic):
    """Every table entry: levels, bucket starts and counts, the order of the blocks inside a bucket and their availability
    masks -- mixed and uniform tilings, pictures that cut the last CTU row and column, transform-skip blocks, all 35 modes."""
    L = capi.lib()
    w, h = pic
    pp = capi.PicParam(w, h, 31, 0, capi.I_SLICE, 1)
    tilings = ["mix", 4, "mix", 8, 16, 32, "mix"] if w < 1000 else ["mix", 4]
    tus = [workload.make_tus(900 + 13 * i + w, w, h, t) for i, t in enumerate(tilings)]
    host = ctx.intra_plans(tus, pp)
    dev = _device_plans(ctx, tus, pp)
    for i, (a, b) in enumerate(zip(host, dev)):
        _same_tables(ctx, a, b, (pic, tilings[i]))
    for p in host + dev:
        L.hmx_intra_plan_destroy(ctx.h, p)
    # a second call re-uses the first call's memory and gives the same tables
    host = ctx.intra_plans(tus[:2], pp)
    dev = _device_plans(ctx, tus[:2], pp)
    for a, b in zip(host, dev):
        _same_tables(ctx, a, b, (pic, "second call"))
    for p in host + dev:
        L.hmx_intra_plan_destroy(ctx.h, p)


def test_device_plan_more_levels_than_rows(ctx, hmx_opts):
    """The level walk counts blocks into a table laid out before the number of levels is known; a picture with more levels than the
    table has rows makes the builder start over with the format's limit -- same plans."""
    L = capi.lib()
    w, h = 200, 136
    pp = capi.PicParam(w, h, 30, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(4400 + i, w, h, t) for i, t in enumerate(["mix", 4, 8])]
    host = ctx.intra_plans(tus, pp)
    hmx_opts(ctx, HMX_PLAN_ROWS="16")
    dev = _device_plans(ctx, tus, pp)
    for i, (a, b) in enumerate(zip(host, dev)):
        _same_tables(ctx, a, b, ("few rows", i))
    for p in host + dev:
        L.hmx_intra_plan_destroy(ctx.h, p)


def test_device_plan_walks_one_after_the_other(ctx, hmx_opts):
    """The luma and the chroma level walk are independent launch chains on two streams; HMX_PLAN_STREAMS=1 runs them in sequence
    on the context's own stream -- same plans either way."""
    L = capi.lib()
    w, h = 416, 240
    pp = capi.PicParam(w, h, 30, 0, capi.I_SLICE, 1)
    tus = [workload.make_tus(5200 + i, w, h, t) for i, t in enumerate(["mix", 8, "mix", 4])]
    host = ctx.intra_plans(tus, pp)
    hmx_opts(ctx, HMX_PLAN_STREAMS="1")
    dev = _device_plans(ctx, tus, pp)
    for i, (a, b) in enumerate(zip(host, dev)):
        _same_tables(ctx, a, b, ("one stream", i))
    for p in host + dev:
        L.hmx_intra_plan_destroy(ctx.h, p)


def test_device_plan_sparse_and_errors(ctx):
    """A plan that lists only part of a picture's blocks (the intra coding units of an inter picture: whole CTUs and parts of
    CTUs missing), and the argument checks of the host analysis."""
    L = capi.lib()
    w, h = 416, 240
    pp = capi.PicParam(w, h, 30, 0, capi.P_SLICE, 1)
    tus = workload.make_tus(77, w, h, "mix")
    rng = np.random.default_rng(5)
    # keep whole 16x16 luma areas (and the chroma that belongs to them) at random: coding order is preserved
    sh = (tus["plane"] != 0).astype(np.int32)
    key = ((tus["y"].astype(np.int32) << sh) // 16) * 64 + (tus["x"].astype(np.int32) << sh) // 16
    keep_area = rng.random(64 * 64) < 0.35
    keep_area[((np.arange(64 * 64) // 64) // 4 == 1) & ((np.arange(64 * 64) % 64) // 4 == 2)] = False  # one CTU without any block
    sparse = tus[keep_area[key]]
    assert 0 < len(sparse) < len(tus)
    a = ctx.intra_plan(sparse, pp)
    (b,) = _device_plans(ctx, [sparse], pp)
    _same_tables(ctx, a, b, "sparse")
    L.hmx_intra_plan_destroy(ctx.h, a)
    L.hmx_intra_plan_destroy(ctx.h, b)

    def fails(lst):
        cat = np.ascontiguousarray(lst, capi.TU_DTYPE)
        d = ctx.to_device(cat)
        off = (C.c_uint32 * 2)(0, len(cat))
        out = (C.c_void_p * 1)()
        rc = L.hmx_intra_plan_create_device(ctx.h, d.ptr, off, 1, C.byref(pp), out)
        d.free()
        return rc == -1 and not out[0]

    bad = tus.copy()
    bad["x"][5] = 4000  # outside the picture
    assert fails(bad)
    bad = tus.copy()
    bad["log2n"][7] = 7
    assert fails(bad)
    bad = tus.copy()
    first_ctu = int(np.flatnonzero((tus["plane"] == 0) & (tus["x"] >= 64))[0])
    bad = np.concatenate([tus[first_ctu:], tus[:first_ctu]])  # the first CTU's blocks at the end: not coding order
    assert fails(bad)
    (ok,) = _device_plans(ctx, [tus], pp)  # and the context still works
    L.hmx_intra_plan_destroy(ctx.h, ok)


@pytest.mark.parametrize("schedule", ["packed", "level"])
def test_device_plans_through_the_chain(ctx, schedule, hmx_opts):
    """Whole pictures coded on device-built plans, every picture its own decisions: levels and reconstruction equal the
    oracle's, encoder and decoder direction; packed schedule and (the level table fetched on demand) the level schedule."""
    hmx_opts(ctx, HMX_INTRA_SCHEDULE=schedule)
    B, L = ctx.bit_depth, capi.lib()
    w, h, n, qp = 200, 136, 11, 28
    pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
    tilings = ["mix", "mix", 4, "mix", 16, 8, "mix", 32]
    tus = [workload.make_tus(6100 + i, w, h, tilings[i % len(tilings)]) for i in range(n)]
    plans = _device_plans(ctx, tus, pp)
    orgs = [workload.make_planes(6200 + i, w, h, B, "texture" if i % 2 else "noise") for i in range(n)]
    d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
    d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    d_lev = [capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
    A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
    parr = (C.c_void_p * n)(*[p.value for p in plans])
    ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    recs = []
    for i in range(n):
        rr, lr = ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i])
        rec, lev = d_rec[i].download(), d_lev[i].download()
        recs.append(rec)
        for p in range(3):
            assert np.array_equal(rec[p], rr[p]) and np.array_equal(lev[p], lr[p]), (i, p)
    d_rec2 = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
    ctx._chk(L.hmx_frame_intra_decode_multi(ctx.h, parr, n, A(d_rec2, capi.Pic), A(d_lev, capi.Levels)))
    ctx.sync()
    for i in range(n):
        a = d_rec2[i].download()
        assert all(np.array_equal(a[p], recs[i][p]) for p in range(3)), ("decode", i)
    for p in plans:
        L.hmx_intra_plan_destroy(ctx.h, p)
    for d in d_org + d_rec + d_rec2 + d_lev:
        d.free()

"""The benched size and shape on the GPU: 640 pictures of 3840x2160 10-bit, each following one of 8 distinct decision
structures, resident in the working layout, through the packed schedule (bench.py's default path) -- and a picture of
every other packing group held against the CPU oracle, levels and reconstruction, encoder and decoder direction; and the
same path with RDOQ as the quantiser (bench.py --rdoq) on 96 pictures.
Run with -m gpu (needs ~70 GB of HBM and ~1 minute each)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from thevc_amd import capi, workload

pytestmark = pytest.mark.gpu


def test_packed_2160p_640_pictures_vs_oracle():
    w, h, B, qp, F, n_plans, n_src = 3840, 2160, 10, 32, 640, 8, 8
    ctx = capi.Context(bit_depth=B)
    try:
        L = capi.lib()
        pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
        tus = [workload.make_tus(101 + j, w, h, "mix") for j in range(n_plans)]
        plans = [ctx.intra_plan(t, pp) for t in tus]
        src = [workload.make_planes(300 + j, w, h, B, "texture") for j in range(n_src)]
        p_org, p_rec, p_dec = (capi.ResidentPool(ctx, w, h, F) for _ in range(3))
        stage = [capi.DevPicture(ctx, w, h) for _ in range(n_src)]
        for k, d in enumerate(stage):
            d.upload(src[k])
        for i0 in range(0, F, n_src):  # picture i holds source i mod n_src
            p_org.import_planes(i0, stage[:min(n_src, F - i0)])
        lev = capi.DevLevelsZSlab(ctx, w, h, F).zero()
        lev_arr = (capi.Levels * F)(*[lev.as_pic(i) for i in range(F)])
        # picture i follows plan (i * 3) mod 8: the pictures of a group (40 consecutive ones) mix all eight structures
        plan_of = [(3 * i) % n_plans for i in range(F)]
        parr = (C.c_void_p * F)(*[plans[j].value for j in plan_of])
        ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, parr, 1, F, p_org.h_, p_rec.h_, lev_arr))
        ctx._chk(L.hmx_frame_intra_decode_resident(ctx.h, parr, 1, F, p_dec.h_, lev_arr))
        ctx.sync()
        sched, groups = C.c_int(), C.c_int()
        L.hmx_last_call_shape(ctx.h, C.byref(sched), C.byref(groups))
        assert sched.value == 3
        cache = {}
        checked = 0
        for i in (0, 39, 40, 127, 255, 256, 383, 500, 599, 639):  # first / last picture of groups, first and last group
            key = (plan_of[i], i % n_src)
            if key not in cache:
                cache[key] = ol.o_intra_frame_encode(tus[plan_of[i]], w, h, B, qp, src[i % n_src])
            rr, lr = cache[key]
            p_rec.export_planes(i, stage[:1])
            rec = stage[0].download()
            p_dec.export_planes(i, stage[:1])
            dec = stage[0].download()
            got_lev = lev.picture(i).to_planes(tus[plan_of[i]])
            for p in range(3):
                assert np.array_equal(rec[p], rr[p]), ("reconstruction", i, p)
                assert np.array_equal(dec[p], rr[p]), ("decoder direction", i, p)
                assert np.array_equal(got_lev[p], lr[p]), ("levels", i, p)
            checked += 1
        assert checked == 10
        for x in (p_org, p_rec, p_dec):
            x.free()
        lev.free()
        for d in stage:
            d.free()
        for pl in plans:
            L.hmx_intra_plan_destroy(ctx.h, pl)
    finally:
        ctx.close()


def test_packed_rdoq_2160p_96_pictures_vs_oracle():
    """RDOQ as the quantiser of the chain at the benched picture size: 96 pictures of 3840x2160 10-bit, 8 decision structures,
    per-picture bit-estimate tables and multipliers, resident pools, packing groups as the library picks them with RDOQ on
    (bench.py --rdoq's path); six pictures against the oracle's chain with xRateDistOptQuant, levels and reconstruction."""
    w, h, B, qp, F, n_plans, n_src = 3840, 2160, 10, 32, 96, 8, 8
    ctx = capi.Context(bit_depth=B)
    try:
        L = capi.lib()
        pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
        tus = [workload.with_cbf_ctx(workload.make_tus(151 + j, w, h, "mix")) for j in range(n_plans)]
        plans = [ctx.intra_plan(t, pp) for t in tus]
        src = [workload.make_planes(350 + j, w, h, B, "texture") for j in range(n_src)]
        plan_of = [(5 * i) % n_plans for i in range(F)]
        lams = workload.rdoq_lambdas(qp)
        sets = [[workload.make_est_bits(9000 + 8 * i + k) for k in range(8)] for i in range(12)]  # picture i: set i mod 12
        ctx.set_rdoq([(sets[i % 12], lams[0] * (1 + 0.01 * (i % 7)), lams[1] * (1 + 0.02 * (i % 5))) for i in range(F)])
        p_org, p_rec = capi.ResidentPool(ctx, w, h, F), capi.ResidentPool(ctx, w, h, F)
        stage = [capi.DevPicture(ctx, w, h) for _ in range(n_src)]
        for k, d in enumerate(stage):
            d.upload(src[k])
        for i0 in range(0, F, n_src):
            p_org.import_planes(i0, stage[:min(n_src, F - i0)])
        lev = capi.DevLevelsZSlab(ctx, w, h, F).zero()
        lev_arr = (capi.Levels * F)(*[lev.as_pic(i) for i in range(F)])
        parr = (C.c_void_p * F)(*[plans[j].value for j in plan_of])
        ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, parr, 1, F, p_org.h_, p_rec.h_, lev_arr))
        ctx.sync()
        for i in (0, 1, 37, 64, 94, 95):
            lam_i = (lams[0] * (1 + 0.01 * (i % 7)), lams[1] * (1 + 0.02 * (i % 5)))
            rr, lr = ol.o_intra_frame_encode_rdoq(tus[plan_of[i]], w, h, B, qp, src[i % n_src], sets[i % 12], lam_i)
            p_rec.export_planes(i, stage[:1])
            rec = stage[0].download()
            got_lev = lev.picture(i).to_planes(tus[plan_of[i]])
            for p in range(3):
                assert np.array_equal(got_lev[p], lr[p]), ("levels", i, p)
                assert np.array_equal(rec[p], rr[p]), ("reconstruction", i, p)
        ctx.set_rdoq(None)
        for x in (p_org, p_rec):
            x.free()
        lev.free()
        for d in stage:
            d.free()
        for pl in plans:
            L.hmx_intra_plan_destroy(ctx.h, pl)
    finally:
        ctx.close()

"""Random-access pipeline (segment sharding + reference-picture exchange): GPU result vs the CPU oracle
on a small sequence, and the exchange itself rehearsed with gloo at world_size 2 on the CPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gop_order_and_exchange_plan():
    from thevc_amd import ra_pipeline as ra
    assert [o[0] for o in ra.gop_order(8)] == [8, 4, 2, 1, 3, 6, 5, 7]  # cfg/encoder_randomaccess_main.cfg Frame1-8
    assert ra.gop_order(8)[1] == (4, 0, 8) and ra.gop_order(8)[5] == (6, 4, 8)
    # 8 segments on 8 ranks: every rank receives exactly one picture, from its right neighbour
    plan = ra.exchange_plan(8, 8)
    assert sorted(d for (_, _, d) in plan) == list(range(8))
    assert all(s == (d + 1) % 8 for (_, s, d) in plan)
    assert ra.exchange_plan(4, 1) == []  # one rank: nothing travels
    wl = ra.RAWorkload(128, 64, 8, 32, intra_period=8, gop=4, n_segments=2)
    jobs = wl.segment_jobs(1)
    assert [j[0] for j in jobs] == [12, 10, 9, 11, 14, 13, 15] and jobs[4][1:3] == (12, 16)


def _exchange_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from thevc_amd import ra_pipeline as ra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_seg = 5
    # every I picture k has a known content on its owner, zeros elsewhere
    pics = {k: [torch.full((6, 8), 100 * k + p, dtype=torch.int16) if k % world == rank else torch.zeros((6, 8), dtype=torch.int16)
                for p in range(3)] for k in range(n_seg + 1)}
    ra.run_exchange(dist, rank, world, n_seg, lambda ki: pics[ki])
    ok = True
    for k in range(n_seg):  # the owner of segment k must now hold I(k) and I(k+1)
        if k % world == rank:
            for ki in (k, k + 1):
                ok &= all(int(pics[ki][p][0, 0]) == 100 * ki + p for p in range(3))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_reference_picture_exchange_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_exchange_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("B", [8, 10])
def test_ra_pipeline_vs_oracle(B, fused):
    import torch
    import oracle_lib as ol
    import ra_oracle
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, qp = 192, 128, 30
    wl = ra.RAWorkload(w, h, B, qp, intra_period=8, gop=4, n_segments=2, seed=3)
    stream = torch.cuda.Stream()  # the library's kernels and torch's copies share one stream (ra_pipeline.RAPipeline)
    ctx = capi.Context(bit_depth=B, stream=stream.cuda_stream)
    pipe = ra.RAPipeline(ctx, torch, wl, fused=fused, stream=stream)  # one-pass inter chain / the two reference-shaped calls
    pipe.load_originals()
    px = pipe.run()
    torch.cuda.synchronize()
    pipe.check()  # the packed schedule's abort word
    assert px == 17 * w * h  # 3 I pictures + 2 x 7 inter pictures
    i_recs = {}
    for k in range(3):
        rr, _ = ol.o_intra_frame_encode(wl.intra_tus, w, h, B, qp, wl.original(k * 8))
        got = pipe.rec[k * 8].download()
        assert all(np.array_equal(got[p], rr[p]) for p in range(3)), ("I picture", k)
        i_recs[k * 8] = rr
    for k in range(2):
        recs = ra_oracle.oracle_segment(wl, k, {p: i_recs[p] for p in (k * 8, k * 8 + 8)})
        for (poc, _, _, _) in wl.segment_jobs(k):
            got = pipe.rec[poc].download()
            for p in range(3):
                assert np.array_equal(got[p], recs[poc][p]), ("inter picture", poc, p)
    # the margins of a referenced picture equal the oracle's border extension
    full = pipe.rec[4].download(with_margins=True)[0]
    assert (full[:80, 80:80 + w] == full[80, 80:80 + w]).all()
    ctx.close()


@pytest.mark.gpu
def test_ra_pipeline_loopback_exchange():
    """The boundary-picture exchange on ONE GPU: the process plays two ranks (RAPipeline(loopback_ranks=2)); every segment's closing
    I picture belongs to the other rank, so the inter pictures read it from a landing buffer that only the exchange fills --
    exchange_plan / run_exchange as between processes, the copies issued on the pipeline's stream behind the chain that
    reconstructed the pictures.  Landing buffers start as garbage: every inter picture still equals the oracle's, and the
    landing buffers equal the owners' pictures margins included (a missing picture, a copy that overtakes the chain or the border
    extension, or a wrong plan fails here)."""
    import torch
    import oracle_lib as ol
    import ra_oracle
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, qp, B = 192, 128, 31, 8
    wl = ra.RAWorkload(w, h, B, qp, intra_period=8, gop=4, n_segments=3, seed=11)
    stream = torch.cuda.Stream()
    ctx = capi.Context(bit_depth=B, stream=stream.cuda_stream)
    pipe = ra.RAPipeline(ctx, torch, wl, stream=stream, loopback_ranks=2)
    pipe.load_originals()
    assert sorted(pipe.landing) == [(0, 8), (1, 16), (2, 24)]
    for t in pipe.landing.values():
        for pl in t.t:
            pl.fill_(777)
    for rep in range(2):  # the second pass overwrites landing buffers the first one filled
        px = pipe.run()
    torch.cuda.synchronize()
    pipe.check()
    assert px == (4 + 3 * 7) * w * h and pipe._moved == (3, 3)
    i_recs = {k * 8: ol.o_intra_frame_encode(wl.intra_tus, w, h, B, qp, wl.original(k * 8))[0] for k in range(4)}
    for (k, poc), t in pipe.landing.items():
        got, own = t.download(with_margins=True), pipe.rec[poc].download(with_margins=True)
        assert all(np.array_equal(got[p], own[p]) for p in range(3)), ("landing buffer", k, poc)
        assert all(np.array_equal(t.download()[p], i_recs[poc][p]) for p in range(3))
    for k in range(3):
        recs = ra_oracle.oracle_segment(wl, k, {p: i_recs[p] for p in (k * 8, k * 8 + 8)})
        for (poc, _, _, _) in wl.segment_jobs(k):
            got = pipe.rec[poc].download()
            for p in range(3):
                assert np.array_equal(got[p], recs[poc][p]), ("inter picture", poc, p)
    pipe.free()
    ctx.close()


@pytest.mark.gpu
def test_ldp_pipeline_vs_oracle():
    """Low-delay P (configs[2]): independent sequences, every P picture references the previous one; the
    pipeline batches position j of all sequences into one call per stage."""
    import torch
    import oracle_lib as ol
    import ra_oracle
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, qp, B, ip = 192, 128, 33, 8, 5
    wl = ra.RAWorkload(w, h, B, qp, intra_period=ip, n_segments=3, seed=5, structure="ldp")
    assert wl.segment_jobs(1) == [(ip + i, ip + i - 1, None, (ip + i) % 2) for i in range(1, ip)]
    stream = torch.cuda.Stream()
    ctx = capi.Context(bit_depth=B, stream=stream.cuda_stream)
    pipe = ra.RAPipeline(ctx, torch, wl, stream=stream)
    pipe.load_originals()
    px = pipe.run()
    torch.cuda.synchronize()
    assert px == 3 * ip * w * h
    for k in range(3):
        rr, _ = ol.o_intra_frame_encode(wl.intra_tus, w, h, B, qp, wl.original(k * ip))
        got = pipe.rec[k * ip].download()
        assert all(np.array_equal(got[p], rr[p]) for p in range(3)), ("I picture", k)
        recs = ra_oracle.oracle_segment(wl, k, {k * ip: rr})
        for (poc, _, _, _) in wl.segment_jobs(k):
            got = pipe.rec[poc].download()
            for p in range(3):
                assert np.array_equal(got[p], recs[poc][p]), ("P picture", poc, p)
    ctx.close()


@pytest.mark.gpu
def test_ra_pipeline_overlapped_steps():
    """run_steps() with enable_overlap(): the I pictures (second context, second stream, double-buffered reconstructions) of
    step n + 1 run beside the inter pictures of step n.  Every step codes the same pictures, so after three pipelined steps
    every picture must equal what one sequential pass leaves -- including the last step's I pictures in the alternate
    buffers having been used as references."""
    import torch
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, qp, B = 192, 128, 30, 8
    wl = ra.RAWorkload(w, h, B, qp, intra_period=8, gop=4, n_segments=3, seed=9)
    stream = torch.cuda.Stream()
    ctx = capi.Context(bit_depth=B, stream=stream.cuda_stream)
    ref_pipe = ra.RAPipeline(ctx, torch, wl, stream=stream)
    ref_pipe.load_originals()
    ref_pipe.run()
    torch.cuda.synchronize()
    want = {poc: t.download() for poc, t in ref_pipe.rec.items()}
    stream_i = torch.cuda.Stream()
    ctx_i = capi.Context(bit_depth=B, stream=stream_i.cuda_stream)
    ctx_i2 = capi.Context(bit_depth=B, stream=stream_i.cuda_stream)  # one context per buffer set: each keeps its device tables
    pipe = ra.RAPipeline(ctx, torch, wl, stream=stream)
    pipe.load_originals()
    pipe.enable_overlap(ctx_i, stream_i, ctx_i2)
    for steps in (3, 2):  # odd and even: the last step's I pictures sit in either buffer set
        for t in pipe.rec.values():
            for pl in t.t:
                pl.zero_()
        px = pipe.run_steps(steps)
        torch.cuda.synchronize()
        pipe.check()
        assert px == steps * 25 * w * h  # 4 I pictures + 3 x 7 inter pictures per step
        last = (pipe.rec_main, pipe.rec_alt)[(steps - 1) % 2]
        for poc in want:
            got = (last[poc] if poc in last else pipe.rec[poc]).download()
            for p in range(3):
                assert np.array_equal(got[p], want[poc][p]), ("picture", poc, p, steps)
    pipe.free()
    ref_pipe.free()
    ctx_i.close()
    ctx_i2.close()
    ctx.close()

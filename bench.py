#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HM block hot path on MI355X.

Metric (BASELINE.json): Mpixels/sec through transform + pred (+MC), all-intra, bit-exact vs HM.
One "step" = one pass of the all-intra chain (refs <- recon, intra prediction, residual, T, Q, IQ,
IT, reconstruction) over a batch of synthetic pictures that is resident in HBM before the timed
region starts.  Weak scaling: every rank (one process per GPU) owns its own batch of pictures
(IntraPeriod 1 => pictures are independent, no data-path collective).

    python bench.py --gpus N --steps K --warmup W [--workload ai2160p10|ai2160p8|ai1080p8] [--frames F]
    python bench.py --workload ra2160p8|ra1080p8|ldp1080p8 [--segments S]     (secondary workloads)

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (k_intra_level_across) against the
HBM peak with ALGORITHMIC bytes; `cpu_baseline` times the reference's own CPU
functions (oracle/_ref, kind "reference") or, if that library is absent, the CPU oracle (kind "port")
on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, bit depth, qp, BASELINE.json config it stands for)
    "ai2160p10": (3840, 2160, 10, 32, "configs[4] All-intra he10 3840x2160 10-bit"),
    "ai2160p8": (3840, 2160, 8, 32, "all-intra main 3840x2160 8-bit (metric's 2160p all-intra, main profile)"),
    "ai1080p8": (1920, 1080, 8, 32, "configs[1] All-intra main 1920x1080 8-bit"),
    # random access: intra-period segments sharded over ranks, boundary I pictures exchanged over RCCL
    "ra2160p8": (3840, 2160, 8, 32, "configs[3] Random-access main 3840x2160 8-bit (hierarchical B, IntraPeriod 32, GOP 8)"),
    "ra1080p8": (1920, 1080, 8, 32, "random-access main 1920x1080 8-bit (configs[3] shape at 1080p)"),
    # low-delay P: a strict chain inside a sequence, so a rank batches over independent sequences (replicas)
    "ldp1080p8": (1920, 1080, 8, 32, "configs[2] Low-delay-P main 1920x1080 8-bit, 64 pictures per sequence"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(tus, n_pics, decode=False):
    """Bytes the all-intra chain must move per step: per sample 2 (original) +
    4 (level written) + 2 (reconstruction written), plus the 4N+1 reference samples (2 B) each block
    gathers from the reconstruction.  Decoder direction: 4 (level read) + 2 (reconstruction written)
    + the reference samples."""
    n = (1 << tus["log2n"].astype(np.int64))
    return int(((n * n) * (6 if decode else 8) + (4 * n + 1) * 2).sum()) * n_pics


def cpu_baseline(tus, w, h, B, qp, checks, seconds_target=12.0):
    """Time the CPU path on one picture of the same workload (one thread): the batch's first picture, so that what the
    CPU computes doubles as a check of what the GPU wrote.  checks = [(picture index, original planes, (GPU
    reconstruction planes, GPU level planes))]; the first entry is timed, further ones (--verify: a picture of every
    picture group) are computed once."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    kind = "reference" if ol.have_ref() else "port"
    fn = ol.r_intra_frame_encode if kind == "reference" else ol.o_intra_frame_encode

    def identical(result, gpu):
        return all(np.array_equal(gpu[0][p], result[0][p]) and np.array_equal(gpu[1][p], result[1][p]) for p in range(3))

    org = checks[0][1]
    t0 = time.perf_counter()
    n, same = 0, None
    while True:
        rec, lev = fn(tus, w, h, B, qp, org)
        if n == 0:
            same = identical((rec, lev), checks[0][2])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or n >= 256:
            break
    out = {"value": round(n * w * h / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
           "sample": f"{n} picture(s) {w}x{h} of the same block structure, single thread, "
                     + ("HM's own functions from oracle/_ref" if kind == "reference" else "CPU oracle (oracle/hmx_oracle.c)")}
    out["gpu_picture_0_identical"] = bool(same)  # levels and reconstruction of the batch's first picture
    if len(checks) > 1:
        cache = {}
        for (i, o, gpu) in checks[1:]:
            key = tuple(a.tobytes()[:64] for a in o)
            if key not in cache:
                cache[key] = fn(tus, w, h, B, qp, o)
            same = same and identical(cache[key], gpu)
        out["gpu_pictures_checked"] = [c[0] for c in checks]
        out["gpu_pictures_identical"] = bool(same)
    return out


def rank_picture_seeds(rank, n_pics, n_distinct=4):
    """Sharding rule of the all-intra path: pictures are independent, rank r owns its
    own batch; picture i of rank r is synthetic picture seed 1000*r + (i mod n_distinct)."""
    return [1000 * rank + (i % min(n_pics, n_distinct)) for i in range(n_pics)]


def max_over_ranks(seconds, world, device):
    """Wall time of the slowest rank (the contract's MAX over ranks)."""
    if world <= 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_value(pixels_per_rank_step, steps, world, seconds):
    """Mpixels/s of the WHOLE job: every rank processed pixels_per_rank_step per step."""
    return pixels_per_rank_step * steps * world / seconds / 1e6


def bench_random_access(args, torch, dist, rank, local_rank, world):
    """configs[3]: one step = every rank codes its intra-period segments (I pictures through the intra
    chain, B/P pictures through MC + residual transform + reconstruction); boundary I pictures travel
    between ranks by RCCL send/recv.  Weak scaling: --segments segments PER RANK."""
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, B, qp, cfg_name = WORKLOADS[args.workload]
    stream = torch.cuda.current_stream().cuda_stream
    ctx = capi.Context(bit_depth=B, device=local_rank, stream=stream)
    n_seg = args.segments * world
    ldp = args.workload.startswith("ldp")
    wl = ra.RAWorkload(w, h, B, qp, n_segments=n_seg, seed=7, n_distinct=4, structure="ldp" if ldp else "ra",
                       intra_period=64 if ldp else 32)
    pipe = ra.RAPipeline(ctx, torch, wl, rank, world, dist if world > 1 else None)
    n_pics = pipe.load_originals()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.run()
    fence()
    t0 = time.perf_counter()
    px = 0
    for _ in range(args.steps):
        px += pipe.run()
    fence()
    dt = max_over_ranks(time.perf_counter() - t0, world, "cuda")
    if world > 1:
        t = torch.tensor([px], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        px = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "Mpixels/sec transform+pred+MC, 2160p all-intra, 1/2/4/8 MI355X; bit-exact vs HM",
            "value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {cfg_name}; {args.segments} independent sequence(s) per GPU, I picture through the "
                                    f"intra chain, P pictures (each references the previous one) MC + residual T/Q + IQ/IT + recon; "
                                    f"no exchange between sequences" if ldp else
                                    f"{args.workload}: {cfg_name}; {args.segments} segment(s) of 32 pictures per GPU, I pictures through "
                                    f"the intra chain, B/P pictures MC (50% bi-pred) + residual T/Q + IQ/IT + recon, boundary I pictures "
                                    f"exchanged by RCCL send/recv"), "segments_per_gpu": args.segments, "pictures_per_gpu": n_pics,
                       "width": w, "height": h, "bit_depth": B, "qp": qp},
            "roofline": None, "note": "secondary workload (SURVEY.md 8e); the roofline line is reported for the all-intra default"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ai2160p10", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None,
                    help="pictures per GPU per step (150 MB of HBM each at 2160p: planes, levels and the working pool); "
                         "default 1792 at 2160p (272 GB of the 309 GB) and 4096 at 1080p, reduced to what the free HBM holds")
    ap.add_argument("--tiling", default="mix", help="mix | 4 | 8 | 16 | 32 (uniform transform size)")
    ap.add_argument("--segments", type=int, default=16,
                    help="random-access workloads: intra-period segments (32 pictures each) per GPU; the I pictures of all "
                         "segments share one whole-picture call, so few segments are dominated by its 4844 dependent launches")
    ap.add_argument("--decode", action="store_true",
                    help="time the decoder direction of the chain (levels -> reconstruction, DEC/TDecCu.cpp:469-687) instead of the "
                         "encoder direction; the levels come from one untimed encode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="the cpu_baseline leg also checks a picture of every picture group")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from thevc_amd import capi, workload

    if args.workload.startswith(("ra", "ldp")):
        return bench_random_access(args, torch, dist, rank, local_rank, world)
    w, h, B, qp, cfg_name = WORKLOADS[args.workload]
    h_c = h - (h % 8)  # pictures are coded in multiples of the minimum CU (8): 1080 -> 1072 + cropped row
    tiling = args.tiling if args.tiling == "mix" else int(args.tiling)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = capi.Context(bit_depth=B, device=local_rank, stream=stream)
    L = capi.lib()
    tus = workload.make_tus(1, w, h_c, tiling)
    pp = capi.PicParam(w, h_c, qp, 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(tus, pp)

    # Batch size: the chain is latency-bound per dependency level, so throughput comes from pictures in flight;
    # the default fills most of the HBM (18.2 bytes per luma sample: planes 6 + levels 6 + tiled working pool 6.2).
    F = args.frames if args.frames else (1792 if w >= 3840 else 4096)
    free_b, _total_b = torch.cuda.mem_get_info()
    per_pic = int(18.3 * w * h_c)
    if not args.frames and F * per_pic > 0.92 * free_b:
        F = max(8, int(0.92 * free_b / per_pic))
    seeds = rank_picture_seeds(rank, F)  # distinct synthetic pictures, cycled over the batch
    cache = {}
    for sd in seeds:
        if sd not in cache:
            cache[sd] = workload.make_planes(sd, w, h_c, B, "texture")
    src = [cache[sd] for sd in seeds]
    d_org = [capi.DevPicture(ctx, w, h_c).upload(src[i]) for i in range(F)]
    d_rec = [capi.DevPicture(ctx, w, h_c).zero() for _ in range(F)]
    d_lev = [capi.DevLevelsZ(ctx, w, h_c).zero() for _ in range(F)]  # the reference's own coefficient layout
    org_arr = (capi.Pic * F)(*[d.as_pic() for d in d_org])
    rec_arr = (capi.Pic * F)(*[d.as_pic() for d in d_rec])
    lev_arr = (capi.Levels * F)(*[d.as_pic() for d in d_lev])

    def step():
        if args.decode:
            ctx._chk(L.hmx_frame_intra_decode(ctx.h, plan, F, rec_arr, lev_arr))
        else:
            ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, F, org_arr, rec_arr, lev_arr))

    if args.decode:  # produce the levels the decoder direction consumes
        ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, F, org_arr, rec_arr, lev_arr))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx._chk(L.hmx_set_timing(ctx.h, 1))
    ev0, ev1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(ev0)
    for _ in range(args.steps):
        step()
    ctx.record(ev1)
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = ctx.elapsed_ms(ev0, ev1)  # HIP events on the stream the kernels run on
    ta, tb, tc = C.c_float(), C.c_float(), C.c_float()
    if L.hmx_last_call_timing(ctx.h, C.byref(ta), C.byref(tb), C.byref(tc)):  # last step: convert / chain / convert
        # graph replay (HMX_GRAPH=1) records no inner events: the whole step stands in for the chain
        ta.value, tb.value, tc.value = 0.0, kernel_ms / args.steps, 0.0
    sched, groups = C.c_int(), C.c_int()
    L.hmx_last_call_shape(ctx.h, C.byref(sched), C.byref(groups))  # how the library issued the timed calls
    dt = max_over_ranks(dt, world, "cuda")

    if rank == 0:
        px_step = w * h_c * F
        nb, nl, nd = C.c_int(), C.c_int(), C.c_int()
        L.hmx_intra_plan_info(plan, C.byref(nb), C.byref(nl), C.byref(nd))
        level_sched = sched.value > 0
        n_levels = nl.value if level_sched else nd.value   # dependent steps of the chain
        n_launch = n_levels * groups.value                 # launches of the dominant kernel per step
        kernel = ["k_intra_wave<%s>", "k_intra_level<%s>", "k_intra_level_across<%s>"][sched.value] % ("false" if args.decode else "true")
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and args.workload == "ai2160p10" and args.tiling == "mix" and level_sched and not args.decode:
            t = json.load(open(tj))
            if t.get("frames") == F:
                # PMC bytes per launch (profiles/r01_traffic.json): gfx950 FETCH_SIZE counts 64 B per
                # 128-B request (MI355X_MICROARCH.md, HBM), hence the factor 2 on the read side
                traffic = round((2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024 / t["launches"])
        bytes_step = algorithmic_bytes(tus, F, args.decode)
        # the dominant kernel alone: HIP events around the chain launches of the last step (the two
        # layout-conversion launches are timed separately); algorithmic bytes / that time
        ach = bytes_step / (tb.value * 1e-3) / 1e9
        out = {
            "metric": "Mpixels/sec transform+pred+MC, 2160p all-intra, 1/2/4/8 MI355X; bit-exact vs HM",
            "value": round(whole_job_value(px_step, args.steps, world, dt), 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg_name}; all-intra chain " +
                                   ("DECODER direction (intra refs+pred, IQ, IT, recon from levels), " if args.decode else
                                    "(intra refs+pred, T, flat Q+SBH, IQ, IT, recon), ") +
                                   f"{F} pictures {w}x{h_c} per GPU per step, QP {qp}, TU tiling '{args.tiling}' "
                                   f"({len(tus)} blocks/picture), frames sharded over ranks, no collective",
                       "pictures_per_gpu": F, "width": w, "height": h_c, "bit_depth": B, "qp": qp, "tiling": str(args.tiling)},
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         # `achieved` is the device-level rate of the chain: the library runs `concurrent_launches`
                         # picture groups on separate streams, so that many launches of the kernel share the GPU
                         # at any time and each lasts about chain time / dependency levels
                         "kernel": kernel, "launches_per_step": n_launch, "concurrent_launches": groups.value,
                         "algorithmic_bytes_per_launch": round(bytes_step / n_launch),
                         "avg_launch_us": round(tb.value * 1e3 / n_levels, 2),
                         "achieved_per_launch": round(bytes_step / n_launch / (tb.value * 1e-3 / n_levels) / 1e9, 2),
                         "step_ms_events": round(kernel_ms / args.steps, 3),
                         # conversion in / out as phases of their own (0 when HMX_PIPELINE_CONV=1 overlaps them with the chain)
                         "layout_conversion_ms": [round(ta.value, 3), round(tc.value, 3)]},
        }
        if world == 1 and (args.verify or not args.no_cpu_baseline):
            # --verify: one picture of every picture group (the groups are separate interleave domains of the pool)
            idx = sorted({0, F // 3, (2 * F) // 3, F // 2, F - 1}) if args.verify else [0]
            checks = [(i, src[i], (d_rec[i].download(), d_lev[i].to_planes(tus))) for i in idx]
            out["cpu_baseline"] = cpu_baseline(tus, w, h_c, B, qp, checks, 0.0 if args.no_cpu_baseline else 12.0)
            if args.verify:
                cb = out["cpu_baseline"]
                out["verified_bit_exact_vs_oracle"] = cb.get("gpu_pictures_identical", cb["gpu_picture_0_identical"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()

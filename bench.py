#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HM block hot path on MI355X.

Metric (BASELINE.json): Mpixels/sec through transform + pred (+MC), all-intra, bit-exact vs HM.
One "step" = one pass of the all-intra chain (refs <- recon, intra prediction, residual, T, Q, IQ,
IT, reconstruction) over a batch of synthetic pictures that is resident in HBM before the timed
region starts.  Every picture follows its own decisions: by default 64 distinct block structures
(seeded quadtrees + modes, picture i follows plan i mod 64, so the 64 pictures of a packing group are
all different) and 64 distinct source pictures.  Weak scaling: every rank (one process per GPU) owns
its own batch (IntraPeriod 1 => pictures are independent, no data-path collective).

    python bench.py --gpus N --steps K --warmup W [--workload ai2160p10|ai2160p8|ai1080p8] [--frames F]
    python bench.py --workload ra2160p8|ra1080p8|ldp1080p8 [--segments S]     (secondary workloads)

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process touches the GPU).  Prints ONE JSON line on rank 0.
`roofline` prices the dominant kernel (k_intra_packed, one persistent launch per step) against the HBM
peak with ALGORITHMIC bytes; `cpu_baseline` times the reference's own CPU functions
(oracle/_ref, kind "reference") or, if that library is absent, the CPU oracle (kind "port") on a bounded
sample of the same workload, rank 0 at N=1 only: one core in-process (which doubles as a bit-exactness
check of what the GPU wrote), then all host cores as independent worker processes.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
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
METRIC = "Mpixels/sec transform+pred+MC, 2160p all-intra, 1/2/4/8 MI355X; bit-exact vs HM"


def algorithmic_bytes(tus, decode=False):
    """Bytes the all-intra chain must move for ONE picture: per sample 2 (original) +
    4 (level written) + 2 (reconstruction written), plus the 4N+1 reference samples (2 B) each block
    gathers from the reconstruction.  Decoder direction: 4 (level read) + 2 (reconstruction written)
    + the reference samples."""
    n = (1 << tus["log2n"].astype(np.int64))
    return int(((n * n) * (6 if decode else 8) + (4 * n + 1) * 2).sum())


def rdoq_inputs(plan_seed, qp):
    """--rdoq: what hmx_set_rdoq takes per picture -- eight bit-estimate tables [luma, chroma][4 sizes] (synthetic, a set per
    decision structure) and the I-slice multipliers of the workload's QP."""
    from thevc_amd import workload
    return [workload.make_est_bits(7000 + 8 * plan_seed + k) for k in range(8)], workload.rdoq_lambdas(qp)


def _cpu_fn(rdoq=None):
    """the CPU side of cpu_baseline: HM's own functions (oracle/_ref) chained over the decisions, or the oracle; with RDOQ as
    the quantiser (rdoq = (tables, multipliers)) the oracle's chain (the compiled reference exposes RDOQ per block only)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    if rdoq is not None:
        ests, lams = rdoq
        return "port", lambda tus, w, h, B, qp, org: ol.o_intra_frame_encode_rdoq(tus, w, h, B, qp, org, ests, lams)
    kind = "reference" if ol.have_ref() else "port"
    return kind, (ol.r_intra_frame_encode if kind == "reference" else ol.o_intra_frame_encode)


def cpu_worker(argv):
    """`bench.py --cpu-worker W H B QP TILING PLAN_SEED PIC_SEED SECONDS [rdoq]`: one host core codes one synthetic picture of the
    workload over and over for SECONDS and prints the pictures it finished and the time it took (a worker process of
    the all-cores leg of cpu_baseline; no GPU, no torch)."""
    from thevc_amd import workload
    w, h, B, qp = (int(v) for v in argv[:4])
    tiling = argv[4] if argv[4] == "mix" else int(argv[4])
    plan_seed, pic_seed, seconds = int(argv[5]), int(argv[6]), float(argv[7])
    rdoq = len(argv) > 8 and argv[8] == "rdoq"
    _kind, fn = _cpu_fn(rdoq_inputs(plan_seed, qp) if rdoq else None)
    tus = workload.make_tus(plan_seed, w, h, tiling)
    if rdoq:
        tus = workload.with_cbf_ctx(tus)
    org = workload.make_planes(pic_seed, w, h, B, "texture")
    fn(tus, w, h, B, qp, org)  # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        fn(tus, w, h, B, qp, org)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    print(json.dumps({"pictures": n, "seconds": dt}), flush=True)


def cpu_baseline(w, h, B, qp, tiling, checks, seconds_target=10.0, all_cores=True, max_cores=16, rdoq=False):
    """checks = [(picture index, decisions (tus), plan seed, picture seed, original planes, (GPU reconstruction planes,
    GPU level planes))].  Leg 1: ONE core, in this process, codes the first entry over and over for seconds_target (what
    the CPU computes doubles as a check of what the GPU wrote; further entries -- --verify: a picture of every packing
    group -- are computed once).  Leg 2: every host core this process may run on, one worker process per core, each
    coding its own picture of the workload for the same time: the node's CPU throughput on independent pictures."""
    kind, fn = _cpu_fn(rdoq_inputs(checks[0][2], qp) if rdoq else None)

    def identical(result, gpu):
        return all(np.array_equal(gpu[0][p], result[0][p]) and np.array_equal(gpu[1][p], result[1][p]) for p in range(3))

    _i, tus, _ps, _is, org, gpu = checks[0]
    t0 = time.perf_counter()
    n, same = 0, None
    while True:
        rec, lev = fn(tus, w, h, B, qp, org)
        if n == 0:
            same = identical((rec, lev), gpu)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or n >= 256:
            break
    one = n * w * h / dt / 1e6
    src = "HM's own functions from oracle/_ref" if kind == "reference" else "CPU oracle (oracle/hmx_oracle.c)"
    if rdoq:
        src += ", xRateDistOptQuant as the quantiser"
    out = {"value": round(one, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
           "sample": f"{n} picture(s) {w}x{h} of the workload's block structure, single thread, {src}"}
    out["gpu_picture_0_identical"] = bool(same)  # levels and reconstruction of the batch's first picture
    if len(checks) > 1:
        for (_i, tus_i, _ps, _is, o, g) in checks[1:]:
            fn_i = _cpu_fn(rdoq_inputs(_ps, qp))[1] if rdoq else fn
            same = same and identical(fn_i(tus_i, w, h, B, qp, o), g)
        out["gpu_pictures_checked"] = [c[0] for c in checks]
        out["gpu_pictures_identical"] = bool(same)
    if all_cores and seconds_target > 0:
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(avail, max_cores)  # --cpu-cores (default 16 = the host-core share of one GPU of the pool, whatever the affinity mask shows)
        out["cores_in_affinity_mask"] = avail
        procs = []
        for k in range(cores):
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(w), str(h), str(B), str(qp), str(tiling),
                   str(checks[0][2]), str(checks[0][3] + k), str(seconds_target)] + (["rdoq"] if rdoq else [])
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True))
        rate, done = 0.0, 0
        for pr in procs:
            so, _ = pr.communicate()
            try:
                r = json.loads(so.strip().splitlines()[-1])
                rate += r["pictures"] * w * h / r["seconds"] / 1e6
                done += r["pictures"]
            except (ValueError, IndexError, KeyError):
                pass
        # the contract's fields describe the node: all cores; the single-thread figure stays beside them
        out.update({"value": round(rate, 3), "cores": cores, "one_core_value": round(one, 3),
                    "sample": f"{done} pictures {w}x{h} of the workload's block structure in {seconds_target:.0f} s on {cores} "
                              f"worker processes (one per host core, independent pictures: HM is single-threaded), {src}; "
                              f"one core alone: {one:.1f} Mpixels/s"})
    return out


def fresh_variant(tus, k):
    """Decisions of picture k of the fresh-decisions leg: the block structure of its plan with every LUMA mode rotated by 5k
    (mod 35) -- another set of dependencies, levels and code paths per picture.  The same formula runs on the device
    (fresh_decision_lists) and here for the pictures the CPU checks."""
    t = tus.copy()
    luma = t["plane"] == 0
    t["mode"] = np.where(luma, (t["mode"].astype(np.int32) + 5 * k) % 35, t["mode"]).astype(np.uint8)
    return t


def fresh_decision_lists(torch, tus_list, F, device):
    """The decision lists of F pictures back to back in HBM (hmx_tu, 8 bytes per block): picture i = plan i mod P with its luma
    modes rotated by 5 * (i div P) -- F DISTINCT decision structures, formed on the device from the P uploaded ones."""
    base = [torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(-1, 8).copy()).to(device) for t in tus_list]
    P = len(base)
    offs = np.zeros(F + 1, np.int64)
    for i in range(F):
        offs[i + 1] = offs[i] + len(tus_list[i % P])
    out = torch.empty((int(offs[F]), 8), dtype=torch.uint8, device=device)
    for i in range(F):
        b = base[i % P]
        dst = out[int(offs[i]):int(offs[i + 1])]
        dst.copy_(b)
        k = i // P
        if k:
            luma = b[:, 5] == 0
            dst[:, 6] = torch.where(luma, ((b[:, 6].to(torch.int32) + 5 * k) % 35).to(torch.uint8), b[:, 6])
    return out, offs


def rank_picture_seeds(rank, n_pics, n_distinct=64):
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


def launch_ranks(n):
    """Start the N ranks of a multi-GPU run: a CHILD process running torch.distributed.run (this process has not
    touched the GPU and never will; nothing is exec'ed over a process that has).  Returns the child's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def run_random_access(args, torch, dist, rank, local_rank, world, workload_name, segments, steps, warmup):
    """configs[3]: one step = every rank codes its intra-period segments (I pictures through the intra
    chain, B/P pictures through MC + residual transform + reconstruction); boundary I pictures travel
    between ranks by RCCL send/recv.  Weak scaling: `segments` segments PER RANK.  Returns the result object."""
    from thevc_amd import capi
    from thevc_amd import ra_pipeline as ra
    w, h, B, qp, cfg_name = WORKLOADS[workload_name]
    # ONE stream for the library's kernels, torch's uploads and the RCCL exchange: the exchange sends reconstructions
    # the intra chain has just written and motion compensation reads what it received
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = capi.Context(bit_depth=B, device=local_rank, stream=stream.cuda_stream)
        n_seg = segments * world
        ldp = workload_name.startswith("ldp")
        wl = ra.RAWorkload(w, h, B, qp, n_segments=n_seg, seed=7, n_distinct=4, structure="ldp" if ldp else "ra",
                           intra_period=64 if ldp else 32)
        pipe = ra.RAPipeline(ctx, torch, wl, rank, world, dist if world > 1 else None, stream=stream)
        n_pics = pipe.load_originals()
        ctx_i = None
        if not ldp:
            # the I pictures of the next step (few pictures: latency-bound, the chip mostly idle) beside this step's inter pictures
            stream_i = torch.cuda.Stream()
            ctx_i = capi.Context(bit_depth=B, device=local_rank, stream=stream_i.cuda_stream)
            ctx_i2 = capi.Context(bit_depth=B, device=local_rank, stream=stream_i.cuda_stream)  # one context per buffer set
            pipe.enable_overlap(ctx_i, stream_i, ctx_i2)

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        if warmup:
            pipe.run_steps(warmup)
        fence()
        t0 = time.perf_counter()
        px = pipe.run_steps(steps)
        fence()
        dt = max_over_ranks(time.perf_counter() - t0, world, "cuda")
        if world > 1:
            t = torch.tensor([px], dtype=torch.float64, device="cuda")
            dist.all_reduce(t)
            px = float(t.item())
        exch = pipe.exchange_stats()
        pipe.check()  # the abort word of every context the steps ran on
        sb = pipe.step_bytes()
    out = {
        "metric": METRIC,
        "value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": (f"{workload_name}: {cfg_name}; {segments} independent sequence(s) per GPU, I picture through the "
                                f"intra chain, P pictures (each references the previous one) MC + residual T/Q + IQ/IT + recon; "
                                f"no exchange between sequences" if ldp else
                                f"{workload_name}: {cfg_name}; {segments} segment(s) of 32 pictures per GPU, I pictures through "
                                f"the intra chain, B/P pictures MC (50% bi-pred) + residual T/Q + IQ/IT + recon, boundary I pictures "
                                f"exchanged by RCCL send/recv" + ("; the I pictures of step n+1 run beside the inter pictures of step n (second stream)" if ctx_i is not None else "")), "segments_per_gpu": segments, "pictures_per_gpu": n_pics,
                   "width": w, "height": h, "bit_depth": B, "qp": qp},
        "exchange": exch,
    }
    # Roofline of the whole step: the inter workloads have no single dominant kernel (motion compensation, the inter block chain, the
    # border extension and the I pictures' intra chain share the step; profiles/r03_ra2160p8_kernel_stats.csv has their split), so the
    # algorithmic bytes of every stage are priced against the step the driver clocks.
    tot = sum(sb.values())
    ach = tot / (dt / steps) / 1e9
    out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                       "traffic": None, "kernel": "whole step (k_mc_cells + k_list / k_inter4 / k_inter32 + k_border + k_intra_packed)",
                       "algorithmic_bytes_per_step": tot, "algorithmic_bytes_by_stage": sb}
    pipe.free()
    ctx.close()
    if ctx_i is not None:
        ctx_i.close()
        ctx_i2.close()
    return out


def rehearse_cpu(args):
    """`bench.py --gpus N --rehearse-cpu`: the multi-rank plumbing of this file WITHOUT a GPU -- the launcher above, one
    process per rank, gloo instead of RCCL, the CPU oracle standing in for the device step.  Every rank codes its own
    all-intra pictures (rank_picture_seeds), the ranks exchange the boundary I pictures of a random-access split
    (ra_pipeline.exchange_plan / run_exchange, the calls the GPU path makes) and agree on the slowest rank's time; rank 0
    prints the same one-line JSON.  What the driver's N-GPU run exercises, minus the kernels (tests/test_abi_and_ranks.py)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from thevc_amd import ra_pipeline as ra
    from thevc_amd import workload
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    w, h, B, qp, F = 128, 64, 8, 32, 3
    tus = workload.make_tus(1, w, h, "mix")
    seeds = rank_picture_seeds(rank, F)
    pics = [workload.make_planes(sd, w, h, B, "texture") for sd in seeds]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    recs = [ol.o_intra_frame_encode(tus, w, h, B, qp, p)[0] for p in pics]
    # the random-access split: segment k on rank k mod N, I picture k+1 travels to the owner of segment k
    n_seg = 2 * world + 1
    i_pics = {k: [torch.from_numpy(np.ascontiguousarray(recs[k % F][p])).clone() if k % world == rank else
                  torch.zeros(recs[0][p].shape, dtype=torch.int16) for p in range(3)] for k in range(n_seg + 1)}
    moved = ra.run_exchange(dist, rank, world, n_seg, lambda ki: i_pics[ki]) if world > 1 else 0
    if world > 1:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, "cpu")
    ok = all(int(i_pics[ki][0].abs().sum()) > 0 for k in range(n_seg) if k % world == rank for ki in (k, k + 1))
    if world > 1:
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(whole_job_value(w * h * F, 1, world, dt), 3), "unit": "Mpixels/s", "n_gpus": world,
                          "steps": 1, "warmup": 0, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                          "config": {"workload": "CPU rehearsal of the rank plumbing (gloo, CPU oracle): NOT a measurement"},
                          "rehearsal": {"ranks": world, "exchange_ops_rank0": moved, "boundary_pictures_arrived": ok}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ai2160p10", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None,
                    help="pictures per GPU per step (110 MB of HBM each at 2160p: resident original and reconstruction, levels and the "
                         "packed schedule's tables); default 2048 at 2160p and 8192 at 1080p, reduced to what the free HBM holds")
    ap.add_argument("--plans", type=int, default=64,
                    help="distinct decision structures (block quadtrees + modes) in the batch; picture i follows plan i mod PLANS. "
                         "1 = every picture shares one structure (the best case of round 1's headline)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic source pictures, cycled over the batch")
    ap.add_argument("--tiling", default="mix", help="mix | 4 | 8 | 16 | 32 (uniform transform size)")
    ap.add_argument("--segments", type=int, default=16,
                    help="random-access workloads: intra-period segments (32 pictures each) per GPU; the I pictures of all "
                         "segments share one whole-picture call")
    ap.add_argument("--decode", action="store_true",
                    help="time the decoder direction of the chain (levels -> reconstruction, DEC/TDecCu.cpp:469-687) instead of the "
                         "encoder direction; the levels come from one untimed encode")
    ap.add_argument("--rdoq", action="store_true",
                    help="quantise with xRateDistOptQuant inside the chain (hmx_set_rdoq; HM's encoder default) instead of the flat "
                         "quantiser: per-picture bit-estimate tables (a set per decision structure) and multipliers")
    ap.add_argument("--planar", action="store_true",
                    help="hand the pictures over in the reference's plane geometry (TComPicYuv): every call converts them into and out "
                         "of the working layout; default: pictures resident in the working layout (hmx_tpool)")
    ap.add_argument("--no-fresh", action="store_true",
                    help="skip the fresh-decisions leg (every picture of every step brings NEW decisions: the plans are analysed on the "
                         "device and the packed schedule's tables rebuilt inside each timed step; reported as \"fresh_decisions\")")
    ap.add_argument("--fresh-steps", type=int, default=3, help="timed steps of the fresh-decisions leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-core-only", action="store_true", help="cpu_baseline: skip the all-cores leg")
    ap.add_argument("--cpu-cores", type=int, default=16, help="cpu_baseline: worker processes of the all-cores leg (the host-core share of one GPU)")
    ap.add_argument("--no-ra", action="store_true", help="skip the random-access leg (ra2160p8, segments sharded over the ranks, RCCL exchange of the "
                                                         "boundary I pictures when N > 1) that follows the all-intra measurement")
    ap.add_argument("--ra-segments", type=int, default=16, help="segments (32 pictures each) per GPU of the random-access leg")
    ap.add_argument("--verify", action="store_true", help="the cpu_baseline leg also checks a picture of every packing group")
    ap.add_argument("--rehearse-cpu", action="store_true", help="no GPU: run the rank launcher, sharding, exchange and reporting with gloo and the CPU oracle")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # before anything touches the GPU
    if args.rehearse_cpu:
        return rehearse_cpu(args)

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
        import datetime
        # a collective that never completes raises after three minutes instead of hanging the run
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))

    from thevc_amd import capi, workload

    if args.workload.startswith(("ra", "ldp")):
        out = run_random_access(args, torch, dist, rank, local_rank, world, args.workload, args.segments, args.steps, args.warmup)
        out["note"] = "secondary workload (SURVEY.md 8e): the roofline object prices the WHOLE step's algorithmic bytes (no single dominant kernel)"
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    w, h, B, qp, cfg_name = WORKLOADS[args.workload]
    h_c = h - (h % 8)  # pictures are coded in multiples of the minimum CU (8): 1080 -> 1072 + cropped row
    tiling = args.tiling if args.tiling == "mix" else int(args.tiling)
    stream = torch.cuda.Stream()  # the library's kernels, its events and torch's fences share this stream
    torch.cuda.set_stream(stream)
    ctx = capi.Context(bit_depth=B, device=local_rank, stream=stream.cuda_stream)
    L = capi.lib()
    n_plans = max(1, args.plans)
    plan_seeds = [1 + j for j in range(n_plans)]
    tus_list = [workload.make_tus(sd, w, h_c, tiling) for sd in plan_seeds]
    if args.rdoq:
        if args.decode:
            raise SystemExit("--rdoq is the encoder direction's quantiser")
        tus_list = [workload.with_cbf_ctx(t) for t in tus_list]
    pp = capi.PicParam(w, h_c, qp, 0, capi.I_SLICE, 1)
    plans = ctx.intra_plans(tus_list, pp)  # the host-side dependency analysis of the plans on all host threads

    # Batch size: throughput comes from pictures in flight; the default fills most of the HBM (planes 6 + levels 6 +
    # tiled working pool 6.2 bytes per luma sample, plus 16 B per block of the packed schedule's item table).
    F = args.frames if args.frames else ((2048 if w >= 3840 else 8192) if not args.planar else (1536 if w >= 3840 else 4096))
    free_b, _total_b = torch.cuda.mem_get_info()
    fresh_leg = not args.no_fresh and not args.planar and not args.decode and not args.rdoq
    # (+ 72 B per block for the fresh-decisions leg: the decision lists, two generations of device-built plans and the builder's work buffers)
    per_pic = int((18.3 if args.planar else 12.3) * w * h_c) + 18 * max(len(t) for t in tus_list) + (72 * int(np.mean([len(t) for t in tus_list])) if fresh_leg else 0)
    if not args.frames and F * per_pic > 0.92 * free_b:
        F = max(8, int(0.92 * free_b / per_pic) // 64 * 64 or 8)
    n_plans = min(n_plans, F)
    seeds = rank_picture_seeds(rank, F, max(1, args.distinct))  # distinct synthetic pictures, cycled over the batch
    from concurrent.futures import ThreadPoolExecutor
    uniq = sorted(set(seeds))
    with ThreadPoolExecutor(max_workers=min(16, len(uniq), os.cpu_count() or 1)) as ex:  # numpy releases the GIL in the large array ops
        cache = dict(zip(uniq, ex.map(lambda sd: workload.make_planes(sd, w, h_c, B, "texture"), uniq)))
    src = [cache[sd] for sd in seeds]
    if args.rdoq:  # before the pools are laid out: RDOQ keeps the packing groups small
        sets = [rdoq_inputs(sd, qp) for sd in plan_seeds[:n_plans]]
        ctx.set_rdoq([(sets[i % n_plans][0], sets[i % n_plans][1][0], sets[i % n_plans][1][1]) for i in range(F)])
    lev_slab = capi.DevLevelsZSlab(ctx, w, h_c, F).zero()  # the reference's own coefficient layout, one slab per plane
    lev_arr = (capi.Levels * F)(*[lev_slab.as_pic(i) for i in range(F)])
    plan_arr = (C.c_void_p * F)(*[plans[i % n_plans].value for i in range(F)])
    stride = 0 if n_plans == 1 else 1
    if args.planar:
        # the reference's plane geometry at the boundary: every call converts the originals into the working layout and
        # the reconstruction out of it (two extra passes over the pictures, reported as layout_conversion_ms)
        d_org = [capi.DevPicture(ctx, w, h_c).upload(src[i]) for i in range(F)]
        d_rec = [capi.DevPicture(ctx, w, h_c).zero() for _ in range(F)]
        org_arr = (capi.Pic * F)(*[d.as_pic() for d in d_org])
        rec_arr = (capi.Pic * F)(*[d.as_pic() for d in d_rec])
        p_org = p_rec = None
    else:
        # pictures RESIDENT in the working layout (hmx_tpool): originals are brought in once, before the timed region
        # (in a pipeline: hmx_yuv_unpack_resident straight from the file's bytes); the reconstruction stays resident
        d_org = d_rec = []
        p_org, p_rec = capi.ResidentPool(ctx, w, h_c, F), capi.ResidentPool(ctx, w, h_c, F)
        stage = [capi.DevPicture(ctx, w, h_c) for _ in range(min(F, 16))]
        for i0 in range(0, F, len(stage)):
            part = stage[:min(len(stage), F - i0)]
            for k, d in enumerate(part):
                d.upload(src[i0 + k])
            p_org.import_planes(i0, part)
        ctx.sync()

    def call(enc):
        if args.planar:
            if enc:
                ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, plan_arr, F, org_arr, rec_arr, lev_arr) if stride else
                         L.hmx_frame_intra_encode(ctx.h, plans[0], F, org_arr, rec_arr, lev_arr))
            else:
                ctx._chk(L.hmx_frame_intra_decode_multi(ctx.h, plan_arr, F, rec_arr, lev_arr) if stride else
                         L.hmx_frame_intra_decode(ctx.h, plans[0], F, rec_arr, lev_arr))
        elif enc:
            ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, plan_arr, stride, F, p_org.h_, p_rec.h_, lev_arr))
        else:
            ctx._chk(L.hmx_frame_intra_decode_resident(ctx.h, plan_arr, stride, F, p_rec.h_, lev_arr))

    def encode():
        call(True)

    def step():
        call(not args.decode)

    def reconstruction(i):
        if args.planar:
            return d_rec[i].download()
        p_rec.export_planes(i, stage[:1])
        return stage[0].download()

    if args.decode:  # produce the levels the decoder direction consumes
        encode()

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
    ctx.sync()  # also reads the packed schedule's abort word: a timed-out dependency wait fails loudly here
    kernel_ms = ctx.elapsed_ms(ev0, ev1)  # HIP events on the stream the kernels run on
    ta, tb, tc = C.c_float(), C.c_float(), C.c_float()
    if L.hmx_last_call_timing(ctx.h, C.byref(ta), C.byref(tb), C.byref(tc)):  # last step: convert / chain / convert
        ta.value, tb.value, tc.value = 0.0, kernel_ms / args.steps, 0.0
    sched, groups = C.c_int(), C.c_int()
    L.hmx_last_call_shape(ctx.h, C.byref(sched), C.byref(groups))  # how the library issued the timed calls
    dt = max_over_ranks(dt, world, "cuda")

    # ---- fresh decisions: nothing carried over from step to step (round-2 verdict, item 1) ----
    # Every picture of the batch has its own decision list in HBM (F distinct structures); a step analyses them on the device
    # (hmx_intra_plan_create_device), builds the packed schedule's tables for these plans and runs the chain.  The headline above
    # repeats one batch, so its plans and tables are built once; this is the same chain fed the way an encoder feeds it.
    fresh = None
    if fresh_leg:
        try:
            d_lists, offs = fresh_decision_lists(torch, tus_list[:n_plans], F, torch.device("cuda", local_rank))
            torch.cuda.synchronize()
            t_plan, t_tab, t_chain, fplans = [], [], [], None
            n_fresh = max(1, args.fresh_steps)
            f0 = None
            n_warm = 2  # untimed: the builder's buffers and BOTH generations of plan tables (step n's plans live until step n + 1's exist) get allocated
            for it in range(n_fresh + n_warm):
                if it == n_warm:
                    fence()
                    f0 = time.perf_counter()
                a0 = time.perf_counter()
                new_plans = ctx.intra_plans_device(d_lists.data_ptr(), offs, pp)  # returns when the tables are complete
                a1 = time.perf_counter()
                farr = (C.c_void_p * F)(*[p.value for p in new_plans])
                ctx._chk(L.hmx_frame_intra_encode_resident(ctx.h, farr, 1, F, p_org.h_, p_rec.h_, lev_arr))
                if fplans is not None:  # (synchronises the stream once: the previous step's call is long done)
                    L.hmx_intra_plan_destroy_many(ctx.h, (C.c_void_p * F)(*[p.value for p in fplans]), F)
                fplans = new_plans
                if it >= n_warm:
                    fa, fb, fc, ft = C.c_float(), C.c_float(), C.c_float(), C.c_float()
                    L.hmx_last_call_timing(ctx.h, C.byref(fa), C.byref(fb), C.byref(fc))
                    L.hmx_last_call_tables_ms(ctx.h, C.byref(ft))
                    t_plan.append((a1 - a0) * 1e3), t_tab.append(ft.value), t_chain.append(fb.value - ft.value)
            fence()
            fdt = max_over_ranks(time.perf_counter() - f0, world, "cuda")
            ctx.sync()
            lv = [C.c_int() for _ in range(3)]
            lvls = []
            for p in fplans[:: max(1, F // 16)]:
                L.hmx_intra_plan_info(p, C.byref(lv[0]), C.byref(lv[1]), C.byref(lv[2]))
                lvls.append(lv[1].value)
            fresh = {"value": round(whole_job_value(w * h_c * F, n_fresh, world, fdt), 2), "unit": "Mpixels/s", "steps": n_fresh,
                     "ms_per_step": round(fdt / n_fresh * 1e3, 3), "ms_plan": round(float(np.mean(t_plan)), 3),
                     "ms_tables": round(float(np.mean(t_tab)), 3), "ms_chain": round(float(np.mean(t_chain)), 3),
                     "distinct_plans": F, "levels_per_picture_sampled": [min(lvls), max(lvls)],
                     "what": "every step: the F pictures' decision lists (resident in HBM, F distinct structures) -> plans analysed on the device "
                             "(hmx_intra_plan_create_device) -> packed-schedule tables -> the chain; nothing re-used between steps"}
            if world == 1 and not args.no_cpu_baseline:
                # three pictures of the last fresh step against the CPU (first, one from the middle of the batch, last)
                kind, fn = _cpu_fn()
                same = True
                for i in sorted({0, F // 2 + 1, F - 1}):
                    tv = fresh_variant(tus_list[i % n_plans], i // n_plans)
                    rec_c, lev_c = fn(tv, w, h_c, B, qp, src[i])
                    rec_g, lev_g = reconstruction(i), lev_slab.picture(i).to_planes(tv)
                    same = same and all(np.array_equal(rec_g[p], rec_c[p]) and np.array_equal(lev_g[p], lev_c[p]) for p in range(3))
                fresh["gpu_pictures_checked"] = sorted({0, F // 2 + 1, F - 1})
                fresh["gpu_pictures_identical"] = bool(same)
                fresh["checked_against"] = kind
            for p in fplans:
                L.hmx_intra_plan_destroy(ctx.h, p)
            del d_lists
            # leave the pools as the headline steps left them (the cpu_baseline leg reads picture 0)
            step()
            fence()
        except Exception as e:  # noqa: BLE001  (the headline line is printed whatever happens here)
            fresh = {"error": f"{type(e).__name__}: {e}"[:300]}

    out = None
    if rank == 0:
        px_step = w * h_c * F
        levels = []
        for p in plans[:n_plans]:
            nb, nl, nd = C.c_int(), C.c_int(), C.c_int()
            L.hmx_intra_plan_info(p, C.byref(nb), C.byref(nl), C.byref(nd))
            levels.append(nl.value if sched.value > 0 else nd.value)
        n_levels = max(levels)                                       # dependent steps of the chain
        n_launch = 1 if sched.value == 3 else n_levels * groups.value  # launches of the dominant kernel per step
        kernel = ["k_intra_wave<%s>", "k_intra_level<%s>", "k_intra_level_across<%s>", "k_intra_packed<%s>"][sched.value] % (
            "false" if args.decode else "true")
        if args.rdoq:
            kernel = "k_intra_packed<true, 64, false, true>"
        per_plan_bytes = [algorithmic_bytes(t, args.decode) for t in tus_list[:n_plans]]
        bytes_step = sum(per_plan_bytes[i % n_plans] for i in range(F))
        traffic, traffic_note = None, "no PMC record for this workload / batch under profiles/ (tools/pmc.py collects one)"
        tj = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if not os.path.exists(tj):
            tj = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if os.path.exists(tj) and args.workload == "ai2160p10" and args.tiling == "mix" and not args.decode and not args.rdoq:
            t = json.load(open(tj))
            if t.get("frames") == F and t.get("plans") == n_plans and t.get("kernel", "").startswith(kernel.split("<")[0]):
                # PMC bytes per launch (tools/pmc.sh): gfx950 FETCH_SIZE counts 64 B per 128-B request
                # (MI355X_MICROARCH.md, HBM), hence the factor 2 on the read side
                traffic = round((2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024 / t["launches"])
                traffic_note = f"from {os.path.basename(tj)} (a separate rocprofv3 --pmc run of this command; FETCH_SIZE x2: it counts 128-byte lines as 64, tools/issue_probe.hip)"
        step_ms = dt / args.steps * 1e3
        # the dominant kernel: HIP events around its launch(es) in the last step (the two layout-conversion launches are
        # timed separately); algorithmic bytes / that time.  `frac_step` prices the same bytes against the WHOLE step the
        # driver clocks (conversions included).
        ach = bytes_step / (tb.value * 1e-3) / 1e9
        ach_step = bytes_step / (step_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC,
            "value": round(whole_job_value(px_step, args.steps, world, dt), 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg_name}; all-intra chain " +
                                   ("DECODER direction (intra refs+pred, IQ, IT, recon from levels), " if args.decode else
                                    "(intra refs+pred, T, RDOQ with per-picture bit-estimate tables, IQ, IT, recon), " if args.rdoq else
                                    "(intra refs+pred, T, flat Q+SBH, IQ, IT, recon), ") +
                                   f"{F} pictures {w}x{h_c} per GPU per step, QP {qp}, TU tiling '{args.tiling}', "
                                   f"{n_plans} distinct decision structures (picture i follows plan i mod {n_plans}: "
                                   f"{min(len(t) for t in tus_list[:n_plans])}-{max(len(t) for t in tus_list[:n_plans])} blocks, "
                                   f"{min(levels)}-{max(levels)} dependency levels per picture), "
                                   f"{len(cache)} distinct source pictures, frames sharded over ranks, no collective",
                       "pictures_per_gpu": F, "width": w, "height": h_c, "bit_depth": B, "qp": qp, "tiling": str(args.tiling),
                       "quantiser": "rdoq" if args.rdoq else "flat", "distinct_plans": n_plans, "distinct_pictures": len(cache), "shared_decisions": n_plans == 1,
                       "pictures": "plane geometry, converted per call" if args.planar else "resident in the working layout (hmx_tpool)"},
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kernel, "launches_per_step": n_launch, "concurrent_launches": groups.value,
                         "algorithmic_bytes_per_launch": round(bytes_step / n_launch),
                         "avg_launch_us": round(tb.value * 1e3 / n_launch, 2),
                         "achieved_step": round(ach_step, 2), "frac_step": round(ach_step / HBM_PEAK_GBS, 5),
                         "step_ms_events": round(kernel_ms / args.steps, 3),
                         # conversion in / out as phases of their own
                         "layout_conversion_ms": [round(ta.value, 3), round(tc.value, 3)]},
        }
        if fresh is not None:
            out["fresh_decisions"] = fresh
        if world == 1 and (args.verify or not args.no_cpu_baseline):
            # --verify: one picture of every few packing groups, first and last included
            # default: three pictures (the first, one from the middle of the batch -- another packing group, another plan -- and the last)
            idx = sorted({0, F - 1} | set(range(0, F, max(64, F // 6 // 64 * 64 or 64)))) if args.verify else sorted({0, F // 2 + 1, F - 1})
            checks = [(i, tus_list[i % n_plans], plan_seeds[i % n_plans], seeds[i], src[i],
                       (reconstruction(i), lev_slab.picture(i).to_planes(tus_list[i % n_plans]))) for i in idx]
            out["cpu_baseline"] = cpu_baseline(w, h_c, B, qp, args.tiling, checks, 0.0 if args.no_cpu_baseline else 10.0,
                                               all_cores=not args.one_core_only, max_cores=args.cpu_cores, rdoq=args.rdoq)
            if args.verify:
                cb = out["cpu_baseline"]
                out["verified_bit_exact_vs_oracle"] = cb.get("gpu_pictures_identical", cb["gpu_picture_0_identical"])
    # free the batch before the optional random-access leg
    for d in d_org + d_rec + ([] if args.planar else stage):
        d.free()
    lev_slab.free()
    for x in (p_org, p_rec):
        if x is not None:
            x.free()
    for p in plans:
        L.hmx_intra_plan_destroy(ctx.h, p)
    ctx.close()
    torch.cuda.set_stream(torch.cuda.default_stream())
    if not args.no_ra:
        # north_star's second claim: frame-sharded random access with the reference-picture exchange over RCCL/xGMI.
        # A secondary leg: whatever happens in it, the all-intra line above is still printed.
        try:
            ra = run_random_access(args, torch, dist, rank, local_rank, world, "ra2160p8", args.ra_segments, max(1, min(2 * args.steps, 8)), 1)
            if rank == 0:
                out["random_access"] = {k: ra[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "config", "exchange", "roofline")}
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["random_access"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


if __name__ == "__main__":
    main()

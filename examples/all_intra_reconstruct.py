"""End-to-end use of libhmx on one GPU: a planar YUV file in, the all-intra reconstruction chain, deblocking, SAO,
a planar YUV file out -- every stage through the C-ABI (thevc_amd/capi.py), nothing computed on the CPU.

    python examples/all_intra_reconstruct.py in.yuv out.yuv --width 416 --height 240 [--file-bits 8] [--bit-depth 8]
                                             [--qp 32] [--frames N]

Decisions (block structure, intra modes, SAO parameters) are synthetic and seeded: libhmx accelerates the block
path, it does not search.  Deblocking strengths follow from the decisions: every block is intra, so every
transform-block edge on the 8x8 grid has strength 2 (TComLoopFilter.cpp:466-470)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thevc_amd import capi, workload, yuvio  # noqa: E402


def edge_maps(tus, w, h):
    """3 (= filtered edge that is a transform-block edge) on the left / top side of every luma block, per 4x4 unit."""
    uw, uh = w // 4, h // 4
    ev, eh = np.zeros((uh, uw), np.uint8), np.zeros((uh, uw), np.uint8)
    for t in tus[tus["plane"] == 0]:
        n, ux, uy = (1 << int(t["log2n"])) // 4, int(t["x"]) // 4, int(t["y"]) // 4
        ev[uy:uy + n, ux] = 3
        eh[uy, ux:ux + n] = 3
    return ev, eh


def run(args):
    w, h, B = args.width, args.height, args.bit_depth
    pw, ph = -(-w // 8) * 8, -(-h // 8) * 8  # the reference pads the source to a multiple of the minimum CU size
    ctx = capi.Context(bit_depth=B)
    L = capi.lib()
    tus = workload.make_tus(args.seed, pw, ph, "mix")
    pp = capi.PicParam(pw, ph, args.qp, 0, capi.I_SLICE, 1)
    plan = ctx.intra_plan(tus, pp)
    org, rec, out = (capi.DevPicture(ctx, pw, ph) for _ in range(3))
    lev = capi.DevLevelsZ(ctx, pw, ph)
    uw, uh = pw // 4, ph // 4
    ev, eh = edge_maps(tus, pw, ph)
    units = np.zeros(uw * uh, np.dtype([("intra", "u1"), ("cbf", "u1"), ("ref", "i1", 2), ("mv", "<i2", (2, 2))]))
    units["intra"] = 1
    d_units, d_ev, d_eh = ctx.to_device(units), ctx.to_device(ev), ctx.to_device(eh)
    d_bv, d_bh = ctx.alloc(uw * uh), ctx.alloc(uw * uh)
    d_qp = ctx.to_device(np.full(uw * uh, args.qp, np.int8))
    rng = np.random.default_rng(args.seed)
    n_lcu = -(-pw // 64) * -(-ph // 64)
    sao = np.zeros((3, n_lcu), np.dtype([("type", "i1"), ("band", "u1"), ("offset", "i1", 4)]))
    sao["type"] = rng.integers(-1, 5, (3, n_lcu))
    sao["band"] = rng.integers(0, 32, (3, n_lcu))
    sao["offset"] = rng.integers(-2, 3, (3, n_lcu, 4))
    d_sao = ctx.to_device(np.ascontiguousarray(sao))
    rd = yuvio.YuvReader(ctx, args.input, w, h, args.file_bits)
    wr = yuvio.YuvWriter(ctx, args.output, args.file_bits)
    n = 0
    A = lambda x, T: (T * 1)(x.as_pic())
    while (args.frames <= 0 or n < args.frames) and rd.read(org, pw - w, ph - h):
        ctx._chk(L.hmx_frame_intra_encode(ctx.h, plan, 1, A(org, capi.Pic), A(rec, capi.Pic), A(lev, capi.Levels)))
        ctx._chk(L.hmx_deblock_strengths(ctx.h, d_units.ptr, d_ev.ptr, d_eh.ptr, pw, ph, 0, d_bv.ptr, d_bh.ptr))
        p = rec.as_pic()
        ctx._chk(L.hmx_deblock_picture(ctx.h, C.byref(p), pw, ph, d_bv.ptr, d_bh.ptr, d_qp.ptr, None, 0, 0))
        q = out.as_pic()
        ctx._chk(L.hmx_sao_picture(ctx.h, C.byref(p), C.byref(q), pw, ph, d_sao.ptr, n_lcu))
        wr.write(out, pw, ph, pw - w, ph - h)
        n += 1
    ctx.sync()
    rd.close(), wr.close()
    L.hmx_intra_plan_destroy(ctx.h, plan)
    ctx.close()
    return n, tus, ev, eh, sao


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--file-bits", type=int, default=8)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    print(f"{run(a)[0]} frame(s) written to {a.output}")

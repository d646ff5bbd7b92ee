"""Fixtures from REAL reference streams: a synthetic clip is encoded by the reference's own
encoder (oracle/_ref/TAppEncoder, built by oracle/build_ref_apps.sh), the stream is decoded by the reference's decoder
library under oracle/ref_decision_tap.cpp, and every picture's decision list (transform blocks, levels) is stored
with the reference decoder's reconstruction.  Needs /root/reference (not on the GPU box); the .npz travels.

  python tests/golden/make_stream_golden.py            # writes tests/golden/stream_*.npz
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFBIN = os.path.join(ROOT, "oracle", "_ref")
TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2n", "u1"), ("plane", "u1"), ("mode", "u1"), ("flags", "u1")])
SAO_DTYPE = np.dtype([("type", "i1"), ("band", "u1"), ("offset", "i1", 4)])
CU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2size", "u1"), ("intra", "u1"), ("skipped", "u1"), ("pad", "u1")])
PU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("cu_x", "<u2"), ("cu_y", "<u2"), ("w", "u1"), ("h", "u1"), ("poc0", "<i2"), ("poc1", "<i2"),
                     ("mv0x", "<i2"), ("mv0y", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2")])


def synthetic_clip(seed, w, h, n, B, smooth=False, motion=False):
    """Pictures with flat areas, ramps, sharp rectangles, fine noise and text-like 1-sample detail, so that the
    encoder's search picks every block size, many directions and transform skip."""
    rng = np.random.default_rng(seed)
    mx = (1 << B) - 1
    out = []
    if motion:
        # one larger canvas seen through a window that pans 4 x 2 luma samples per picture, an object that moves
        # against the pan, fresh noise every picture and a new rectangle in every picture (uncovered content: the
        # encoder codes it intra inside inter pictures)
        canvas = synthetic_clip(seed + 1000, w + 64, h + 64, 1, B)[0]
        for i in range(n):
            planes = []
            for k, c in enumerate(canvas):
                s = 1 if k else 0
                ox, oy = (8 + 4 * i) >> s, (8 + 2 * i) >> s
                p = c[oy:oy + (h >> s), ox:ox + (w >> s)].astype(np.float64)
                bx, by = (100 - 6 * i) >> s, (40 + 4 * i) >> s
                p[by:by + (24 >> s), bx:bx + (32 >> s)] = (0.8 if k == 0 else 0.3) * mx
                nx, ny = int(rng.integers(0, (w >> s) - 24)), int(rng.integers(0, (h >> s) - 24))
                p[ny:ny + (20 >> s), nx:nx + (20 >> s)] = rng.integers(0, mx + 1, ((20 >> s), (20 >> s)))
                p += rng.normal(0, 0.004 * mx, p.shape)
                planes.append(np.clip(np.rint(p), 0, mx).astype(np.uint16))
            out.append(planes)
        return out
    for i in range(n):
        planes = []
        for (pw, ph) in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
            yy, xx = np.mgrid[0:ph, 0:pw]
            p = 0.5 * mx * (1 + 0.5 * np.sin(xx / 19.0 + seed + i) * np.cos(yy / 13.0))
            p[:, : pw // 3] = 0.3 * mx + 0.4 * mx * xx[:, : pw // 3] / pw                   # ramp
            for _ in range(3 if smooth else 12):                                            # sharp rectangles
                x0, y0 = int(rng.integers(0, pw - 8)), int(rng.integers(0, ph - 8))
                p[y0:y0 + int(rng.integers(3, 24)), x0:x0 + int(rng.integers(3, 40))] = rng.integers(0, mx + 1)
            noisy = ((xx // 16 + yy // 16) % 3 == 0) & ((not smooth) | (xx > 0.7 * pw))
            p = np.where(noisy, p + rng.normal(0, 0.08 * mx, (ph, pw)), p + rng.normal(0, 0.004 * mx, (ph, pw)))
            p[ph // 2: ph // 2 + 12, ::2] = mx * (rng.random((12, (pw + 1) // 2)) < 0.5)   # 1-sample detail
            planes.append(np.clip(np.rint(p), 0, mx).astype(np.uint16))
        out.append(planes)
    return out


def parse_hmxd(path):
    b = open(path, "rb").read()
    assert b[:4] == b"HMXD"
    ver, n = np.frombuffer(b, "<i4", 2, 4)
    assert ver == 4
    off, pics = 12, []
    for _ in range(n):
        poc, w, h, B, qp, ctu, slice_type, n_tu, n_pu, n_cu = (int(v) for v in np.frombuffer(b, "<i4", 10, off))
        off += 40
        tus = np.frombuffer(b, TU_DTYPE, n_tu, off).copy()
        off += 8 * n_tu
        pus = np.frombuffer(b, PU_DTYPE, n_pu, off).copy()
        off += PU_DTYPE.itemsize * n_pu
        cus = np.frombuffer(b, CU_DTYPE, n_cu, off).copy()
        off += 8 * n_cu
        n_ctu = -(-w // ctu) * -(-h // ctu)
        lev = []
        for p in range(3):
            e = n_ctu * ctu * ctu >> (2 if p else 0)
            lev.append(np.frombuffer(b, "<i4", e, off).copy())
            off += 4 * e
        sao = np.frombuffer(b, SAO_DTYPE, 3 * n_ctu, off).reshape(3, n_ctu).copy()
        off += 6 * 3 * n_ctu
        dbk = np.frombuffer(b, "<i4", 3, off).copy()  # disabled, beta_offset_div2, tc_offset_div2
        off += 12
        rec = []
        for p in range(3):
            pw, ph = w >> (1 if p else 0), h >> (1 if p else 0)
            rec.append(np.frombuffer(b, "<i2", pw * ph, off).reshape(ph, pw).copy())
            off += 2 * pw * ph
        pics.append(dict(poc=poc, w=w, h=h, B=B, qp=qp, ctu=ctu, tus=tus, pus=pus, cus=cus, slice_type=slice_type, lev=lev, rec=rec, sao=sao, dbk=dbk))
    assert off == len(b)
    return pics


def make(name, seed, w, h, n, B, qp, cfg, extra=(), smooth=False, motion=False, keep_org=False, out_dir=HERE):
    enc, tap = os.path.join(REFBIN, "TAppEncoder"), os.path.join(REFBIN, "hm_decision_tap")
    if not (os.path.exists(enc) and os.path.exists(tap)):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref_apps.sh")])
    with tempfile.TemporaryDirectory() as d:
        yuv, bit, out = (os.path.join(d, f) for f in ("in.yuv", "str.bin", "out.hmxd"))
        clip = synthetic_clip(seed, w, h, n, B, smooth, motion)
        with open(yuv, "wb") as f:
            for planes in clip:
                for p in planes:
                    f.write(p.astype(np.uint8 if B == 8 else "<u2").tobytes())
        cmd = [enc, "-c", os.path.join("/root/reference/cfg", cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "30",
               "-f", str(n), "-q", str(qp), "-b", bit, "-o", os.path.join(d, "rec.yuv"), "--SEIpictureDigest=1", f"--InputBitDepth={B}",
               f"--InternalBitDepth={B}"] + list(extra)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        subprocess.run([tap, bit, out], check=True, stdout=subprocess.DEVNULL)
        pics = parse_hmxd(out)
        nbytes = os.path.getsize(bit)
    arrays = {"n": np.int32(len(pics)), "stream_bytes": np.int32(nbytes), "command": np.array(" ".join(c for c in cmd[1:] if not c.startswith(d)))}  # without the temporary paths
    for i, p in enumerate(pics):
        arrays[f"hdr{i}"] = np.array([p["poc"], p["w"], p["h"], p["B"], p["qp"], p["ctu"], p["slice_type"]], np.int32)
        arrays[f"pus{i}"] = p["pus"]
        arrays[f"cus{i}"] = p["cus"]
        if keep_org:  # the encoder's input picture (all-intra: coding order = input order)
            for k in range(3):
                arrays[f"org{i}_{k}"] = clip[p["poc"]][k].astype(np.int16)
        arrays[f"tus{i}"] = p["tus"]
        arrays[f"sao{i}"] = p["sao"]
        arrays[f"dbk{i}"] = p["dbk"]
        for k in range(3):
            arrays[f"lev{i}_{k}"] = p["lev"][k]
            arrays[f"rec{i}_{k}"] = p["rec"][k]
    path = os.path.join(out_dir, f"stream_{name}.npz")
    np.savez_compressed(path, **arrays)
    sizes = np.bincount(np.concatenate([p["tus"]["log2n"][p["tus"]["plane"] == 0] for p in pics]), minlength=6)[2:]
    ts = sum(int((p["tus"]["flags"] & 1).sum()) for p in pics)
    sao_on = sum(int((p["sao"]["type"] >= 0).sum()) for p in pics)
    n_pu = sum(len(p["pus"]) for p in pics)
    n_bi = sum(int(((p["pus"]["poc0"] > -32768) & (p["pus"]["poc1"] > -32768)).sum()) for p in pics)
    intra_in_inter = sum(int(((p["tus"]["flags"] & 2) == 0).sum()) for p in pics if p["slice_type"] != 2)
    print(f"{path}: {len(pics)} picture(s), {nbytes} stream bytes, luma blocks 4/8/16/32 = {sizes.tolist()}, transform-skip blocks {ts}, SAO CTU-components on {sao_on}, PUs {n_pu} ({n_bi} bi), intra blocks in inter pictures {intra_in_inter}, "
          f"{os.path.getsize(path)} bytes")


# loop filters off (the slice-level disable is only signalled with the control-present flag)
PURE = ["--DeblockingFilterControlPresent=1", "--LoopFilterDisable=1", "--SAO=0"]

if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        _make = make
        make = lambda name, *a, **k: _make(name, *a, **k) if name in only else None  # noqa: E731
    # pure reconstruction: loop filters off, so the decoder's output IS prediction + residual of the block path
    make("intra_main_q27", 11, 192, 128, 2, 8, 27, "encoder_intra_main.cfg", PURE)
    make("intra_he10_q32", 12, 128, 128, 1, 10, 32, "encoder_intra_he10.cfg", PURE)
    # deblocking as the configuration ships it (all-intra: every filtered edge has strength 2), SAO off: the decoder's
    # output is the block path followed by the deblocking filter
    make("intra_main_q34_dbk", 13, 192, 128, 1, 8, 34, "encoder_intra_main.cfg", ["--SAO=0"])
    # 416x240: 6.5 x 3.75 CTUs (coding units cut by the picture boundary), smoother content for 32x32 blocks, deblocked
    make("intra_main_q37_416x240_dbk", 14, 416, 240, 1, 8, 37, "encoder_intra_main.cfg", ["--SAO=0"], smooth=True)
    # the shipped configurations as they are: deblocking and SAO on (the decoder's output = block path, deblocking, SAO)
    make("intra_main_q32_full", 15, 256, 192, 2, 8, 32, "encoder_intra_main.cfg")
    make("intra_he10_q30_416x240_full", 16, 416, 240, 1, 10, 30, "encoder_intra_he10.cfg")
    # encoder direction: with the flat quantiser (RDOQ off; sign-bit hiding stays on) the levels in the stream are a
    # function of the decisions and the input picture alone, so the encoder-side chain can be held against them
    make("intra_main_q29_rdoq0", 22, 192, 128, 2, 8, 29, "encoder_intra_main.cfg", PURE + ["--RDOQ=0"], keep_org=True)
    make("intra_he10_q35_rdoq0", 23, 128, 128, 1, 10, 35, "encoder_intra_he10.cfg", PURE + ["--RDOQ=0"], keep_org=True)
    # the same for inter residuals (rounding offset of inter slices): wherever the encoder kept a transform block's
    # residual, its levels are the flat quantiser's of (input - prediction)
    make("lowdelay_P_main_q28_rdoq0", 24, 192, 128, 3, 8, 28, "encoder_lowdelay_P_main.cfg", PURE + ["--RDOQ=0"], motion=True, keep_org=True)
    # inter pictures (low delay P, random access), loop filters off: motion compensation + inter residual + the intra
    # blocks the encoder chose inside inter pictures
    make("lowdelay_P_main_q30", 17, 192, 128, 4, 8, 30, "encoder_lowdelay_P_main.cfg", PURE, motion=True)
    make("randomaccess_main_q32", 18, 192, 128, 9, 8, 32, "encoder_randomaccess_main.cfg", PURE, motion=True)
    # the same two structures with the loop filters of the shipped configurations (boundary strengths from motion)
    make("lowdelay_P_main_q32_full", 19, 192, 128, 4, 8, 32, "encoder_lowdelay_P_main.cfg", motion=True)
    make("randomaccess_main_q34_full", 20, 256, 192, 9, 8, 34, "encoder_randomaccess_main.cfg", motion=True)
    # 10-bit low-delay B, 416x240 (coding units cut by the picture boundary, vectors that leave the picture), filters on
    make("lowdelay_he10_q33_416x240_full", 21, 416, 240, 3, 10, 33, "encoder_lowdelay_he10.cfg", motion=True)

"""tests/golden/rdoq_enc_tap.npz: what the REFERENCE ENCODER itself fed its RDOQ -- block by block the bit-estimate table as the
live CABAC state left it (TEncSearch.cpp:1101, TEncSbac::estBit), the multiplier, the coefficients, what the function read of the
coding unit -- and the levels it returned.  Recorded by oracle/_ref/TAppEncoder_rdoqtap (the reference encoder with ONE recorder
statement at the entry of its own xRateDistOptQuant: oracle/ref_rdoq_tap.h, oracle/build_ref_enc_shim.sh) on the clips of
enc_shim_cases.py; a sample of the calls is kept: per clip, texture type and block size up to PER_CLASS calls spread over the
run (the search calls the function tens of thousands of times), inter blocks and the root-cbf branch included.
Every table in the fixture is a real encoder state, no two blocks of a clip share one.  Needs oracle/_ref (build container).

  python tests/golden/make_rdoq_enc_tap.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from enc_shim_cases import CASES, options, write_clip  # noqa: E402

ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_rdoqtap")
PER_CLASS = 40


def records(path):
    with open(path, "rb") as f:
        data = f.read()
    off, out = 0, []
    while off < len(data):
        h = struct.unpack_from("<16i", data, off)
        assert h[0] == 0x52444F51
        w, tb = h[1], h[15]
        off += 64
        lam = struct.unpack_from("<d", data, off)[0]
        off += 8
        table = data[off:off + tb]
        off += tb
        n = w * w
        coef = np.frombuffer(data, "<i4", n, off).copy()
        off += 4 * n
        lev = np.frombuffer(data, "<i4", n, off).copy()
        off += 4 * n
        out.append((h, lam, table, coef, lev))
    return out


def main():
    keep = []
    for ci, case in enumerate(CASES):
        with tempfile.TemporaryDirectory() as d:
            yuv = write_clip(case, os.path.join(d, "in.yuv"))
            tap = os.path.join(d, "tap.bin")
            subprocess.run([ENC] + options(case, yuv, os.path.join(d, "s.bin"), os.path.join(d, "rec.yuv")), check=True, stdout=subprocess.DEVNULL,
                           env=dict(os.environ, HMX_RDOQ_TAP=tap))
            recs = records(tap)
        classes = {}
        for k, r in enumerate(recs):
            h = r[0]
            classes.setdefault((h[1], h[2] != 0, h[8], h[10]), []).append(k)  # size, chroma, intra, root cbf
        for key, idx in sorted(classes.items()):
            # prefer calls that code something; spread over the run
            nz = [k for k in idx if recs[k][0][14] > 0] or idx
            pick = [nz[int(j)] for j in np.linspace(0, len(nz) - 1, min(PER_CLASS, len(nz)))]
            for k in sorted(set(pick)):
                keep.append((ci, case["bits"]) + recs[k])
        print(case["name"], len(recs), "calls,", {k: len(v) for k, v in sorted(classes.items())})
    n = len(keep)
    hdr = np.array([[k[0], k[1]] + list(k[2]) for k in keep], np.int32)  # clip, bit depth, then the recorder's 16 header words
    lam = np.array([k[3] for k in keep], np.float64)
    tables = np.frombuffer(b"".join(k[4] for k in keep), np.uint8).reshape(n, -1)
    off = np.concatenate([[0], np.cumsum([len(k[5]) for k in keep])]).astype(np.int64)
    coef = np.concatenate([k[5] for k in keep]).astype(np.int32)
    lev = np.concatenate([k[6] for k in keep]).astype(np.int32)
    out = os.path.join(HERE, "rdoq_enc_tap.npz")
    np.savez_compressed(out, hdr=hdr, lam=lam, tables=tables, off=off, coef=coef, lev=lev)
    print(out, os.path.getsize(out), "bytes,", n, "calls kept")


if __name__ == "__main__":
    main()

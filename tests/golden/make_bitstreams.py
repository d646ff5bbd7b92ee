"""Bitstream fixtures for tests/test_ref_shim.py: synthetic clips (make_stream_golden.synthetic_clip) encoded by the
REFERENCE's own encoder (oracle/_ref/TAppEncoder) under its shipped configurations, with the picture-digest SEI
(--SEIpictureDigest=1: the MD5 of the encoder's reconstruction travels in the stream, TLibDecoder/TDecGop.cpp:344-402
checks it).  The .bin files are the encoder's output -- data, a few kilobytes each.  Needs /root/reference.

  python tests/golden/make_bitstreams.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_stream_golden import synthetic_clip  # noqa: E402

ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder")
STREAMS = [
    # name, seed, w, h, frames, bit depth, qp, cfg, extra options, motion
    ("intra_main_q32", 31, 192, 128, 2, 8, 32, "encoder_intra_main.cfg", [], False),
    ("intra_he10_q30", 32, 128, 128, 1, 10, 30, "encoder_intra_he10.cfg", [], False),
    ("lowdelay_P_main_q32", 33, 192, 128, 3, 8, 32, "encoder_lowdelay_P_main.cfg", [], True),
    ("randomaccess_main_q34", 34, 192, 128, 5, 8, 34, "encoder_randomaccess_main.cfg", [], True),
    # round 3: the default scaling lists (xDeQuant's scaling-list branch, TComTrQuant.cpp:1311-1342)
    ("lowdelay_P_scalinglist_q30", 35, 192, 128, 3, 8, 30, "encoder_lowdelay_P_main.cfg", ["--ScalingList=1"], True),
    ("intra_he10_scalinglist_q24", 36, 128, 128, 1, 10, 24, "encoder_intra_he10.cfg", ["--ScalingList=1"], False),
]


def main():
    os.makedirs(os.path.join(HERE, "streams"), exist_ok=True)
    for (name, seed, w, h, n, B, qp, cfg, extra, motion) in STREAMS:
        with tempfile.TemporaryDirectory() as d:
            yuv = os.path.join(d, "in.yuv")
            with open(yuv, "wb") as f:
                for planes in synthetic_clip(seed, w, h, n, B, False, motion):
                    for p in planes:
                        f.write(p.astype(np.uint8 if B == 8 else "<u2").tobytes())
            out = os.path.join(HERE, "streams", name + ".bin")
            cmd = [ENC, "-c", os.path.join("/root/reference/cfg", cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(n),
                   "-q", str(qp), "-b", out, "-o", os.path.join(d, "rec.yuv"), "--SEIpictureDigest=1", f"--InputBitDepth={B}",
                   f"--InternalBitDepth={B}"] + extra
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
            print(out, os.path.getsize(out), "bytes,", n, "pictures")


if __name__ == "__main__":
    main()

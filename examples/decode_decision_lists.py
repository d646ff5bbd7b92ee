"""Decode a sequence from DECISION LISTS with libhmx and write it as planar YUV.

The decision lists come from a decoder's parser -- here the fixtures the reference decoder produced under
oracle/ref_decision_tap.cpp (tests/golden/stream_*.npz: transform blocks, prediction units, levels, loop-filter
parameters per picture).  thevc_amd/decisions.py turns them into libhmx calls picture by picture; the output equals
the reference decoder's pictures (the fixture carries them, --check compares).

  python examples/decode_decision_lists.py tests/golden/stream_randomaccess_main_q34_full.npz out.yuv --check
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thevc_amd import decisions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fixture")
    ap.add_argument("out_yuv")
    ap.add_argument("--check", action="store_true", help="compare with the reference decoder's pictures in the fixture")
    a = ap.parse_args()
    pics = list(decisions.load_pictures(a.fixture))
    out = decisions.decode_sequence(pics)
    order = np.argsort([p["poc"] for p in pics])  # output order
    with open(a.out_yuv, "wb") as f:
        for i in order:
            for plane in out[i]:
                f.write(plane.astype(np.uint8 if pics[i]["B"] == 8 else "<u2").tobytes())
    kinds = {0: "B", 1: "P", 2: "I"}
    print(f"{len(pics)} picture(s) {pics[0]['w']}x{pics[0]['h']} {pics[0]['B']}-bit, decoding order "
          + " ".join(f"{kinds[p['slice_type']]}{p['poc']}" for p in pics) + f" -> {a.out_yuv}")
    if a.check:
        ok = all(np.array_equal(o[k], p["rec"][k]) for o, p in zip(out, pics) for k in range(3))
        print("identical to the reference decoder's output" if ok else "MISMATCH")
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

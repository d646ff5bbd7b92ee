"""Fixtures for tests/test_ref_enc_shim.py: synthetic clips (make_stream_golden.synthetic_clip) encoded by the REFERENCE's own,
unmodified encoder (oracle/_ref/TAppEncoder) -- the bitstreams the encoder built on libhmx (oracle/_ref/TAppEncoder_hmx) has to
reproduce byte for byte.  The option lists are OURS (enc_shim_cases.CASES, every switch on the command line: RDOQ, transform
skip, sign-bit hiding, AMP, Hadamard motion search as the shipped configurations have them; loop filters off so that the stream
is a function of the hot path alone), no configuration file of the reference is read or stored.  Needs oracle/_ref (build container).

  python tests/golden/make_enc_fixtures.py
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from enc_shim_cases import CASES, options, write_clip  # noqa: E402

ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder")


def main():
    for case in CASES:
        with tempfile.TemporaryDirectory() as d:
            yuv = write_clip(case, os.path.join(d, "in.yuv"))
            out = os.path.join(HERE, "enc", case["name"] + ".bin")
            subprocess.run([ENC] + options(case, yuv, out, os.path.join(d, "rec.yuv")), check=True, stdout=subprocess.DEVNULL)
            print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

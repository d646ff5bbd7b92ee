"""The ENCODER side of the drop-in, shown: the REFERENCE's encoder application with the bodies of its hot-path members replaced by
libhmx calls (oracle/build_ref_enc_shim.sh: xT, xIT, xTransformSkip / xITransformSkip, xQuant, xRateDistOptQuant, xDeQuant,
predIntraLumaAng / ChromaAng, xPredInterLumaBlk / ChromaBlk, filterHor/Ver Luma/Chroma, addAvg, calcHAD -- the INTEGRATION.md
section 3 bodies) encodes synthetic clips and writes the bitstream of the unmodified encoder, BYTE FOR BYTE.  Every decision of
the encoder's search goes through these members -- 35 modes x calcHAD per prediction unit incl. the 64x64 one, RDOQ with the bit
estimates of the live CABAC state (TEncSearch.cpp:1101, TEncSbac::estBit), the half- and quarter-sample planes of the motion
search -- so one differing sample or level anywhere changes a decision and the stream.  The binaries live in oracle/_ref/ (built
in the build container, they travel to the GPU box); the expected streams are tests/golden/enc/*.bin (make_enc_fixtures.py)."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(HERE, "golden"))
from enc_shim_cases import CASES, options, write_clip  # noqa: E402


def _encode(binary, case, tmp_path, tag):
    yuv = write_clip(case, str(tmp_path / "in.yuv"))
    stream = str(tmp_path / f"{tag}.bin")
    r = subprocess.run([os.path.join(REF, binary)] + options(case, yuv, stream, str(tmp_path / f"{tag}_rec.yuv")), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    return open(stream, "rb").read(), r.stderr


def test_encoder_fixtures_present():
    for case in CASES:
        assert os.path.getsize(os.path.join(HERE, "golden", "enc", case["name"] + ".bin")) > 1000


@pytest.mark.ref
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_fixtures_are_the_unmodified_encoders_streams(case, tmp_path):
    if not os.path.exists(os.path.join(REF, "TAppEncoder")):
        pytest.skip("oracle/_ref/TAppEncoder not built")
    got, _ = _encode("TAppEncoder", case, tmp_path, "ref")
    assert got == open(os.path.join(HERE, "golden", "enc", case["name"] + ".bin"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_encoder_on_libhmx(case, tmp_path):
    if not os.path.exists(os.path.join(REF, "TAppEncoder_hmx")):
        pytest.skip("oracle/_ref/TAppEncoder_hmx not built (bash oracle/build_ref_enc_shim.sh in the build container)")
    got, err = _encode("TAppEncoder_hmx", case, tmp_path, "hmx")
    want = open(os.path.join(HERE, "golden", "enc", case["name"] + ".bin"), "rb").read()
    calls = {u: int(n) for n, u in re.findall(r"libhmx shim: (\d+) calls from (\w+)", err)}
    assert got == want, (len(got), len(want), calls, err[-800:])
    # the replaced members really ran on the GPU: transforms + both quantisers, intra prediction, the Hadamard cost; and in the
    # inter case the interpolation members (fractional search) and the per-block motion compensation
    assert calls.get("TComTrQuant", 0) > 1000 and calls.get("TComPrediction", 0) > 1000 and calls.get("TComRdCost", 0) > 1000, calls
    if case["inter"]:
        assert calls.get("TComInterpolationFilter", 0) > 100, calls

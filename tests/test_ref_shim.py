"""The drop-in, shown: the REFERENCE's decoder application with the bodies of its hot-path members replaced by libhmx calls
(oracle/build_ref_shim.sh: xIT, xDeQuant, xITransformSkip, predIntraLumaAng / ChromaAng, filterHor/Ver Luma/Chroma,
addAvg -- the INTEGRATION.md section 3 bodies) decodes real streams of the reference's encoder, and the reference's own
picture-digest check (TLibDecoder/TDecGop.cpp:344-402) says (OK) for every picture; the output file equals the unmodified
decoder's byte for byte.  The binaries live in oracle/_ref/ (built in the build container, they travel to the GPU box)."""
import glob
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref")
STREAMS = sorted(glob.glob(os.path.join(HERE, "golden", "streams", "*.bin")))
N_PICTURES = {"intra_main_q32": 2, "intra_he10_q30": 1, "lowdelay_P_main_q32": 3, "randomaccess_main_q34": 5,
              "lowdelay_P_scalinglist_q30": 3, "intra_he10_scalinglist_q24": 1}


def _decode(binary, stream, out):
    r = subprocess.run([os.path.join(REF, binary), "-b", stream, "-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    pics = re.findall(r"^POC\s+\d+.*\[MD5:[0-9a-f,]+,\((OK|\*\*\*ERROR\*\*\*)\)\]", r.stdout, re.M)
    return pics, r.stderr


def test_fixture_streams_present():
    assert len(STREAMS) == len(N_PICTURES)


@pytest.mark.ref
@pytest.mark.parametrize("stream", STREAMS, ids=[os.path.basename(s)[:-4] for s in STREAMS])
def test_fixtures_decode_with_the_unmodified_reference(stream, tmp_path):
    """The fixtures are sound: the reference's decoder (compiled from /root/reference as it is) finds every digest right."""
    if not os.path.exists(os.path.join(REF, "TAppDecoder")):
        pytest.skip("oracle/_ref/TAppDecoder not built")
    pics, _ = _decode("TAppDecoder", stream, str(tmp_path / "ref.yuv"))
    assert pics == ["OK"] * N_PICTURES[os.path.basename(stream)[:-4]]


@pytest.mark.gpu
@pytest.mark.parametrize("stream", STREAMS, ids=[os.path.basename(s)[:-4] for s in STREAMS])
def test_reference_decoder_on_libhmx(stream, tmp_path):
    if not os.path.exists(os.path.join(REF, "TAppDecoder_hmx")):
        pytest.skip("oracle/_ref/TAppDecoder_hmx not built (bash oracle/build_ref_shim.sh in the build container)")
    name = os.path.basename(stream)[:-4]
    pics, err = _decode("TAppDecoder_hmx", stream, str(tmp_path / "hmx.yuv"))
    assert pics == ["OK"] * N_PICTURES[name], (pics, err[-1500:])
    calls = {u: int(n) for n, u in re.findall(r"libhmx shim: (\d+) calls from (\w+)", err)}
    # the replaced members really ran on the GPU: the transform and intra units in every stream, interpolation in the inter ones
    assert calls.get("TComTrQuant", 0) > 100 and calls.get("TComPrediction", 0) > 100, calls
    if "intra" not in name:
        assert calls.get("TComInterpolationFilter", 0) > 10, calls
    if name.startswith("randomaccess"):
        assert calls.get("TComYuv", 0) > 0, calls  # bi-prediction: addAvg
    if os.path.exists(os.path.join(REF, "TAppDecoder")):
        _decode("TAppDecoder", stream, str(tmp_path / "ref.yuv"))
        assert open(tmp_path / "hmx.yuv", "rb").read() == open(tmp_path / "ref.yuv", "rb").read()

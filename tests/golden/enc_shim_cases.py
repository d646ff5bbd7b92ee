"""The encoder runs of tests/test_ref_enc_shim.py and tests/golden/make_enc_fixtures.py: clips and option lists (our own choice of
values for the reference encoder's command-line switches; TAppEncCfg.cpp:190-330 names them)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_stream_golden import synthetic_clip  # noqa: E402

CASES = [
    # all-intra, 8 bit: 64x64 coding units down to 4x4 transform blocks, RDOQ fed by the live CABAC state, transform skip, sign hiding
    {"name": "intra_q32", "seed": 51, "w": 192, "h": 128, "frames": 1, "bits": 8, "qp": 32, "inter": False},
    # the same at 10 bit and a lower QP (more coefficients per block, the 32x32 transform's wide levels)
    {"name": "intra10_q27", "seed": 52, "w": 128, "h": 64, "frames": 1, "bits": 10, "qp": 27, "inter": False},
    # low delay, P pictures: motion search (half / quarter sample planes through the interpolation members), motion compensation,
    # the inter residual quadtree with RDOQ (root cbf branch), AMP partitions, merge
    {"name": "lowdelay_P_q30", "seed": 53, "w": 192, "h": 128, "frames": 3, "bits": 8, "qp": 30, "inter": True},
    # AdaptiveQpSelection: the quantiser's pArlDes output (hmx_arlCoeff, flat branch and RDOQ) feeds TEncSlice's statistics, which move
    # the QP of the following P slices (TEncSlice.cpp:714-722, 1387): one differing ARL coefficient changes the later pictures
    # the default scaling lists: per-position tables through xQuant / xRateDistOptQuant / xDeQuant (hmx_x*_scaled), intra and inter lists
    {"name": "intra_scalinglist_q30", "seed": 55, "w": 128, "h": 64, "frames": 1, "bits": 8, "qp": 30, "inter": False, "extra": ["--ScalingList=1"]},
    {"name": "lowdelay_P_scalinglist_q32", "seed": 56, "w": 128, "h": 128, "frames": 2, "bits": 8, "qp": 32, "inter": True, "extra": ["--ScalingList=1"]},
    {"name": "lowdelay_P_aqps_q32", "seed": 54, "w": 128, "h": 128, "frames": 3, "bits": 8, "qp": 32, "inter": True, "extra": ["--AdaptiveQpSelection=1"]},
]


def write_clip(case, path):
    with open(path, "wb") as f:
        for planes in synthetic_clip(case["seed"], case["w"], case["h"], case["frames"], case["bits"], False, case["inter"]):
            for p in planes:
                f.write(p.astype(np.uint8 if case["bits"] == 8 else "<u2").tobytes())
    return path


def options(case, yuv, stream, recon):
    o = ["-i", yuv, "-wdt", str(case["w"]), "-hgt", str(case["h"]), "-fr", "30", "-f", str(case["frames"]), "-q", str(case["qp"]),
         f"--InputBitDepth={case['bits']}", f"--InternalBitDepth={case['bits']}", "-b", stream, "-o", recon,
         "--MaxCUWidth=64", "--MaxCUHeight=64", "--MaxPartitionDepth=4", "--QuadtreeTULog2MaxSize=5", "--QuadtreeTULog2MinSize=2",
         "--QuadtreeTUMaxDepthInter=3", "--QuadtreeTUMaxDepthIntra=3", "--RDOQ=1", "--TS=1", "--TSFast=1", "--SignHideFlag=1",
         "--SAO=0", "--LoopFilterDisable=1", "--SEIpictureDigest=1", "--AMP=1", "--FEN=1", "--FDM=1", "--SearchRange=16", "--HadamardME=1",
         "--GOPSize=1"]
    if case["inter"]:
        o += ["--IntraPeriod=-1", "--Frame1=P 1 0 0.5 0 1 1 1 -1 0"]  # one P picture per GOP, one reference: the previous picture
    else:
        o += ["--IntraPeriod=1", "--DecodingRefreshType=0", "--Frame1=B 1 0 1 0 1 1 0"]
    return o + case.get("extra", [])

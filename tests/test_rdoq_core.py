"""The lane decompositions of RDOQ (thevc_amd/csrc/hmx_rdoq_core.h) emulated on the CPU lane by lane and held against the
oracle's sequential restatement, which tests/test_oracle_vs_ref.py pins to the reference's xRateDistOptQuant: (1) every
coefficient group walked for every (carry, neighbour pattern), one serial pass over the groups, a second walk of the chosen
variants, last position, sign hiding per group over arrays; (2) the form the device runs (rdoq_wave_tiles in hmx_rdoq.h):
rounds composed at run time from the variants a group can still take, levels and the search's records from a second walk, sign
hiding as a third walk with a sink -- the harness fails if a variant that was ruled out is ever needed.  The same header is
what the device routines are made of; this test needs no GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_rdoq_lane_decomposition_equals_the_oracle(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    exe = str(tmp_path / "rdoq_core_host")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "thevc_amd", "csrc"), "-I",
                           os.path.join(ROOT, "oracle"), os.path.join(HERE, "native", "rdoq_core_host.cpp"), "-o", exe, "-L",
                           os.path.join(ROOT, "oracle"), "-lhmx_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    r = subprocess.run([exe, "500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "identical to the oracle" in r.stdout

"""A short soak of the packed schedule inside the GPU suite (tools/soak.py runs it for as long as one likes): random batch
sizes, packing groups, persistent-wave counts (down to THREE waves), 4x4 wave shapes, with and without RDOQ, every picture
with its own plan, two calls per configuration -- every picture of every run against the oracle."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_packed_schedule_soak():
    import soak
    soak.soak(24, 20261004, verbose=False)

import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference oracle/_ref (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle_lib import have_ref
    skip_ref = pytest.mark.skip(reason="oracle/_ref/libhmref.so not built (reference absent)")
    for it in items:
        if "ref" in it.keywords and not have_ref():
            it.add_marker(skip_ref)


@pytest.fixture
def hmx_opts():
    """Set tuning knobs of a libhmx context for one test (hmx_set_option) and restore the defaults afterwards:
    hmx_opts(ctx, HMX_INTRA_SCHEDULE="level").  The environment is only read when a context is created."""
    applied = []

    def set_(ctx, **kw):
        for k, v in kw.items():
            ctx.set_option(k, v)
            applied.append((ctx, k))

    yield set_
    for ctx, k in applied:
        if ctx.h:
            ctx.set_option(k, None)

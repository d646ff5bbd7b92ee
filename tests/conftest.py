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

#!/usr/bin/env python3
"""tools/build_variant.py NAME [-DFLAG ...] -- an A/B build of libhmx.so with extra compiler flags: thevc_amd/libhmx_NAME.so
(objects under thevc_amd/csrc/build_NAME/).  Use it with HMX_LIB_PATH=thevc_amd/libhmx_NAME.so python bench.py ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
g.build_lib(os.path.join(g.ROOT, "thevc_amd", f"libhmx_{name}.so"), flags, os.path.join(g.CSRC, "build_" + name))

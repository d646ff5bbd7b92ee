"""thevc_amd: MI355X-native implementation of the HM (fr34k8/thevc) block hot path.

The product is libhmx.so (hand-written HIP for gfx950) behind the C-ABI of include/hmx.h;
this package holds its sources (csrc/), the C++ host mirror of the reference classes (host/) and the
ctypes/synthetic-workload plumbing used by tests/ and bench.py."""
from . import capi, workload  # noqa: F401

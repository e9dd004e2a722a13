"""cora_amd -- MI355X-native solver core for CORA (certifiably correct range-aided SLAM).

The product is cora_amd/lib/libcora_hip.so (HIP kernels + C ABI, include/cora_hip.h)
and the C++ host in cora_amd/csrc/host.  The Python modules here are plumbing for
tests/ and bench.py."""
from . import build  # noqa: F401

__all__ = ["build", "capi"]

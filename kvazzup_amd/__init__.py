"""kvazzup_amd -- MI355X-native HEVC encode/decode hot path behind uvgComm's KvazaarFilter /
OpenHEVCFilter plugin surface.  The product is the C-ABI shared library built from csrc/ (HIP
kernels + host engine); this package only loads it and mirrors the two filters for tests and the
benchmark.  There is no CPU implementation: everything fails loudly without the library/GPU."""
from ._native import load_library, library_path, build_library  # noqa: F401

__version__ = "0.1"

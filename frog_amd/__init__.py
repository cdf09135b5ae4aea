"""frog_amd -- MI355X-native groupwise-registration hot path of valette/FROG.

Native code lives in frog_amd/csrc (HIP kernels + C ABI, C++ host); this package
is the thin Python layer tests, bench.py and the multi-process launcher use.
"""
from .pairs import Pairs            # noqa: F401
from .image_group import ImageGroup  # noqa: F401

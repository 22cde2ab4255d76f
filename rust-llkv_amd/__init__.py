"""MI355X-native execution path for LLKV (scan → filter → group-by/aggregate → join).

Python here is the harness-side mirror of the reference's operator interface over the C ABI
of include/llkv_hip.h; the product is csrc/ (HIP kernels + C++ host engine).
"""
from . import abi  # noqa: F401

"""orb-slam2-dualcam_amd: MI355X-native ORB extract + Hamming match + dual-camera local BA.

Python side = thin ctypes mirror of the C ABI in include/dcs_abi.h (used by tests/ and bench.py).
The product is the HIP library built from csrc/; there is NO CPU fallback: every entry point
raises if the library is missing.
"""
from . import synth  # noqa: F401
from . import abi  # noqa: F401,E402
from .abi import ORBextractor, ORBmatcher, Optimizer, DcsError, ComputeDistinctiveDescriptors, frame_grid, ORBVocabulary, KeyFrameDatabase, isInFrustum, projection_queries  # noqa: F401,E402
from . import sharding  # noqa: F401,E402

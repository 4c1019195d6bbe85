"""loam_amd — MI355X (gfx950) implementation of DanMcGann/loam's two hot paths,
loam::extractFeatures and loam::registerFeatures, behind a C ABI (include/loamx.h).

Layout: csrc/ (HIP kernels + C ABI), capi.py (ctypes binding used by tests and bench.py),
build.py (hipcc build). The C++ drop-in headers live in include/loam/, the pybind11 module in
python/. There is no CPU fallback anywhere in this package.
"""
from . import build, capi  # noqa: F401
from .capi import (Context, FeatureExtractionParams, LidarParams, LoamxError, RegistrationParams)  # noqa: F401

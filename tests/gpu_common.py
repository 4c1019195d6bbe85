"""Shared helpers of the -m gpu parity tests."""
import contextlib

import numpy as np

from loam_amd import capi

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


@contextlib.contextmanager
def option(name, value=1, c=None):
    """a debug switch of the shared context (loamx_ctx_set_option) for the duration of a with-block"""
    c = c or ctx()
    old = c.get_option(name)
    c.set_option(name, value)
    try:
        yield
    finally:
        c.set_option(name, old)


def pose_diff(O, a, b):
    """rotation angle and translation norm of a^-1 b"""
    d = O.pose_compose(O.pose_inverse(a), b)
    return O.quat_angular_distance(d[:4], [0, 0, 0, 1.0]), float(np.linalg.norm(d[4:]))


def to_capi_fe(p):
    return capi.FeatureExtractionParams(*[getattr(p, f[0]) for f in capi.FeatureExtractionParams._fields_])


def to_capi_reg(p):
    return capi.RegistrationParams(*[getattr(p, f[0]) for f in capi.RegistrationParams._fields_])

"""TEST INFRASTRUCTURE — ctypes loader for oracle/obsnorm_ref.c (built by `make -C oracle`)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libobsnorm_ref.so")


def build():
    subprocess.run(["make", "-C", _HERE, "libobsnorm_ref.so"], check=True, stdout=subprocess.DEVNULL)
    return _SO


class NormalizerOracle:
    """State and protocol of torchrl/env/base_wrapper.py:64-96 `Normalizer` for [E][S] batches of a vec env."""

    def __init__(self, S, clip=10.0):
        self.mean, self.var = np.zeros(S), np.ones(S)
        self.count = np.array([1e-4])
        self.clip = float(clip)

    def observation(self, raw, training=True):
        """NormObs.observation (base_wrapper.py:119-122): update (training) then filt; raw [E][S] float64."""
        if not os.path.exists(_SO):
            build()
        lib = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        lib.obsnorm_ref.argtypes = [dp, C.c_long, C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_int, dp, C.c_long]
        lib.obsnorm_ref.restype = None
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        E, S = raw.shape
        out = np.empty((E, S))
        p = lambda a: a.ctypes.data_as(dp)
        lib.obsnorm_ref(p(raw), S, E, S, p(self.mean), p(self.var), p(self.count), self.clip, int(bool(training)), p(out), S)
        return out

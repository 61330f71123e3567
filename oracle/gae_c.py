"""TEST INFRASTRUCTURE — ctypes loader for oracle/gae_ref.c (built by `make -C oracle`)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgae_ref.so")


def build():
    subprocess.run(["make", "-C", _HERE, "libgae_ref.so"], check=True, stdout=subprocess.DEVNULL)
    return _SO


def gae_c(rewards, values, terminals, time_limits, last_value, gamma, tau, use_tl):
    """float64 arrays [T,E] (time_limits [T] or [T,E] or None) -> (advs, rets) [T,E]. tau=None: discount_ref
    (PPO(gae=False), replay_buffers/on_policy.py:47-71)."""
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    lib.gae_ref.argtypes = [dp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp]
    lib.gae_ref.restype = None
    r, v, t = (np.ascontiguousarray(a, dtype=np.float64) for a in (rewards, values, terminals))
    T, E = r.shape
    tl = np.ascontiguousarray(time_limits if time_limits is not None else np.zeros(T), dtype=np.float64)
    per_env = int(tl.ndim == 2 and tl.shape[1] == E and tl.size == T * E and not (E == 1 and tl.ndim == 1))
    lv = np.ascontiguousarray(last_value, dtype=np.float64).reshape(E)
    advs, rets = np.empty((T, E)), np.empty((T, E))
    p = lambda a: a.ctypes.data_as(dp)
    if tau is None:
        lib.discount_ref.argtypes = [dp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp]
        lib.discount_ref.restype = None
        lib.discount_ref(p(r), p(v), p(t), p(tl), per_env, p(lv), T, E, float(gamma), int(bool(use_tl)), p(advs), p(rets))
        return advs, rets
    lib.gae_ref(p(r), p(v), p(t), p(tl), per_env, p(lv), T, E, float(gamma), float(tau), int(bool(use_tl)), p(advs), p(rets))
    return advs, rets

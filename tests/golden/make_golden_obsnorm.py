"""Generates tests/golden/obsnorm.npz by driving the UNMODIFIED reference normaliser
(/root/reference/torchrl/env/base_wrapper.py: Normalizer / NormObs) on seeded [E][S] batches, and pins
oracle/obsnorm_ref.c against it bit for bit. Run in the build container: `python tests/golden/make_golden_obsnorm.py`.

The fixture keeps, per case, only the reference OUTPUTS (filtered observations of every step, final mean / var /
count); the raw inputs are regenerated from the seed by tests/util.py::obsnorm_inputs."""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util  # noqa: E402


def import_reference_wrappers():
    """base_wrapper.py only needs the gym wrapper base classes to exist; their behaviour is not on this path
    except ObservationWrapper.__init__ keeping `env`."""
    gym = types.ModuleType("gym")

    class Wrapper:
        def __init__(self, env):
            self.env = env

    for name in ("ObservationWrapper", "RewardWrapper", "ActionWrapper"):
        setattr(gym, name, type(name, (Wrapper,), {}))
    gym.Wrapper = Wrapper
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = type("Box", (), {})
    gym.spaces = spaces
    saved = {k: sys.modules.get(k) for k in ("gym", "gym.spaces", "torchrl", "torchrl.env", "torchrl.env.base_wrapper")}
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_base_wrapper", os.path.join(REF, "torchrl/env/base_wrapper.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return mod


def main():
    ref = import_reference_wrappers()
    from oracle.obsnorm_c import NormalizerOracle

    out = {}
    for name, case in util.OBSNORM_CASES.items():
        raws, training = util.obsnorm_inputs(case)
        E, S = case["E"], case["S"]

        class FakeEnv:
            observation_space = types.SimpleNamespace(shape=(S,))

        env = ref.NormObs(FakeEnv())  # the wrapper the vec env is wrapped in (vision4leg/get_env.py:120-126)
        orc = NormalizerOracle(S)
        for k, raw in enumerate(raws):
            env.training = bool(training[k])
            y_ref = env.observation(raw.copy())
            y_orc = orc.observation(raw, training[k])
            assert y_ref.dtype == np.float64 and y_ref.shape == (E, S)
            assert np.array_equal(y_ref, y_orc), (name, k, np.abs(y_ref - y_orc).max())
            out["%s/y%d" % (name, k)] = y_ref
        nz = env._obs_normalizer
        assert np.array_equal(nz._mean, orc.mean) and np.array_equal(nz._var, orc.var) and nz._count == orc.count[0]
        out[name + "/mean"], out[name + "/var"], out[name + "/count"] = nz._mean, nz._var, np.array([nz._count])
        print("==", name, case, "oracle == reference bit for bit; count", nz._count)
    np.savez_compressed(os.path.join(HERE, "obsnorm.npz"), **out)
    print("wrote obsnorm.npz", os.path.getsize(os.path.join(HERE, "obsnorm.npz")), "bytes")


if __name__ == "__main__":
    main()

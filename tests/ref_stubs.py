"""Stand-ins for the third-party modules the reference imports outside the hot path (gym, toolz, tensorboardX, cv2,
pybullet*) and for its simulator package (`vision4leg`), so that `/root/reference/torchrl` and the import block of
`starter/ppo_*.py` can be executed in the build container. Any attribute of a stub module is a
fresh empty class (usable as a base class or in isinstance)."""
import importlib.abc
import importlib.machinery
import sys
import types

STUB_ROOTS = {"gym", "toolz", "tensorboardX", "cv2", "pybullet", "pybullet_data", "pybullet_utils", "pybullet_envs",
              "vision4leg"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name[0].islower():  # gym.spaces, toolz.dicttoolz, ...: a stub sub-module
            import importlib
            obj = importlib.import_module(self.__name__ + "." + name)
        else:                  # gym.Wrapper, gym.spaces.Box, ...: an empty class
            obj = type(name, (), {})
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _Stub(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


def install(reference_root="/root/reference"):
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)

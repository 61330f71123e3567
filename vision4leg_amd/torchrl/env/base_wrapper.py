"""The running observation normaliser of the vectorised env, resident on the GPU.

Reference: torchrl/env/base_wrapper.py:44-122 (`update_mean_var_count`, `Normalizer`, `NormObs`) and
vision4leg/get_env.py:41-67 (`NormObsWithImg`), which run on the host per env step: np.mean / np.var over the E
envs, the Welford-style merge, the clip, and an np.hstack that copies the 16 K-float depth row. Here the raw step
goes to the device as it comes from the simulator (fp64 proprio rows + the depth stack) and one HIP launch
(`v4l_obs_norm`) produces the fp32 observation rows `RolloutActor` / `pf.explore` read, bit-identical to the
reference's numbers; the statistics live in three fp64 device arrays.

Only the arithmetic moves: these classes are not gym wrappers (gym is not a dependency of this package). A starter
keeps its env stack and calls `NormObsWithImg.observation(raw_state, image)` where the reference's wrapper would
have run; the attribute names (`_obs_normalizer`, `_mean`, `_var`, `_count`, `clip`, `should_estimate`, `training`)
are the reference's, so `RLAlgo.snapshot` (rl_algo.py:84-90) and the viewers' pickle round trip keep working through
`to_reference` / `from_reference`.
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ...engine import _ptr, _require_gpu, _stream, check


class Normalizer:
    """Device-resident `Normalizer` (base_wrapper.py:64-96). `shape`: (S,) or S."""

    def __init__(self, shape, clip=10., device=None):
        self.shape = tuple(shape) if hasattr(shape, "__len__") else (int(shape),)
        if len(self.shape) != 1:
            raise NotImplementedError("vision4leg_amd: the device normaliser handles flat [S] observations")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("vision4leg_amd: Normalizer needs a GPU device (the HIP engine has no CPU path)")
        S = self.shape[0]
        self._mean_dev = torch.zeros(S, dtype=torch.float64, device=self.device)
        self._var_dev = torch.ones(S, dtype=torch.float64, device=self.device)
        self._count_dev = torch.full((1,), 1e-4, dtype=torch.float64, device=self.device)
        self.clip = clip
        self.should_estimate = True
        self._L = _lib.lib()

    # the reference's attributes, as host copies (pickles, prints in the viewers)
    @property
    def _mean(self):
        return self._mean_dev.cpu().numpy()

    @property
    def _var(self):
        return self._var_dev.cpu().numpy()

    @property
    def _count(self):
        return float(self._count_dev.item())

    def stop_update_estimate(self):
        self.should_estimate = False

    def _run(self, raw, update, out32=None, out64=None, image=None, image_out=None):
        _require_gpu(raw, "raw observation rows")
        if raw.dtype != torch.float64 or raw.dim() != 2 or raw.stride(1) != 1 or raw.shape[1] != self.shape[0]:
            raise RuntimeError("vision4leg_amd: raw proprio rows must be float64 [E][%d] with unit column stride "
                               "(got %s %s)" % (self.shape[0], raw.dtype, tuple(raw.shape)))
        E, S = raw.shape
        for o, dt in ((out32, torch.float32), (out64, torch.float64)):
            if o is not None and (o.dtype != dt or o.dim() != 2 or o.shape[0] != E or o.shape[1] < S or o.stride(1) != 1
                                  or not o.is_cuda):
                raise RuntimeError("vision4leg_amd: normaliser output must be a cuda %s [E][>=S] tensor" % dt)
        img_args = (C.c_void_p(0), 0, 0, 0, C.c_void_p(0), 0)
        if image is not None:
            _require_gpu(image, "depth stack")
            image = image.reshape(E, -1)
            if image.dtype not in (torch.float32, torch.float64) or image.stride(1) != 1:
                raise RuntimeError("vision4leg_amd: depth stack must be float32 / float64 rows")
            if (image_out is None or image_out.dtype != torch.float32 or image_out.shape != image.shape
                    or image_out.stride(1) != 1 or not image_out.is_cuda):
                raise RuntimeError("vision4leg_amd: image_out must be a cuda float32 [E][C*H*W] view")
            img_args = (_ptr(image), int(image.dtype == torch.float64), image.stride(0), image.shape[1],
                        _ptr(image_out), image_out.stride(0))
        check(self._L.v4l_obs_norm(_ptr(raw), raw.stride(0), E, S, _ptr(self._mean_dev), _ptr(self._var_dev),
                                   _ptr(self._count_dev), float(self.clip), int(bool(update)),
                                   _ptr(out32), out32.stride(0) if out32 is not None else 0,
                                   _ptr(out64), out64.stride(0) if out64 is not None else 0, *img_args, _stream()),
              "v4l_obs_norm")

    def update_filt(self, raw, training=True, out32=None, out64=None, image=None, image_out=None):
        """`update_estimate(raw)` (when training and should_estimate) followed by `filt(raw)` in one launch — the
        body of NormObs.observation (base_wrapper.py:119-122). raw: cuda float64 [E][S]."""
        if out32 is None and out64 is None:
            out64 = torch.empty(raw.shape, dtype=torch.float64, device=raw.device)
        self._run(raw, training and self.should_estimate, out32, out64, image, image_out)
        return out64 if out64 is not None else out32

    def update_estimate(self, data):
        """base_wrapper.py:77-84 for a [E][S] batch (statistics only; the filtered rows are discarded)."""
        if self.should_estimate:
            self._run(data, True, None, torch.empty(data.shape, dtype=torch.float64, device=data.device))

    def filt(self, raw):
        """base_wrapper.py:93-96 for a [E][S] batch; float64 in, float64 out."""
        out = torch.empty(raw.shape, dtype=torch.float64, device=raw.device)
        self._run(raw, False, None, out)
        return out

    filt_torch = filt

    def inverse(self, raw):
        """base_wrapper.py:86-87 (not on the hot path; one torch expression on the device copy of the statistics)."""
        return raw * torch.sqrt(self._var_dev).to(raw.dtype) + self._mean_dev.to(raw.dtype)

    inverse_torch = inverse

    # ---- pickle round trip with the reference's class (rl_algo.py:84-90, starter/*_viewer.py) ----
    def state(self):
        return {"_mean": self._mean, "_var": self._var, "_count": self._count, "clip": self.clip,
                "should_estimate": self.should_estimate, "shape": self.shape}

    def load_state(self, st):
        self._mean_dev.copy_(torch.as_tensor(np.asarray(st["_mean"], dtype=np.float64)))
        self._var_dev.copy_(torch.as_tensor(np.asarray(st["_var"], dtype=np.float64)))
        self._count_dev.fill_(float(st["_count"]))
        self.clip = st.get("clip", self.clip)
        self.should_estimate = st.get("should_estimate", self.should_estimate)
        return self

    def to_reference(self, reference_cls):
        """-> an instance of the reference's `torchrl.env.base_wrapper.Normalizer` (pass the class) holding these
        statistics, ready for pickle.dump the way RLAlgo.snapshot writes `_obs_normalizer_{epoch}.pkl`."""
        ref = reference_cls(self.shape, self.clip)
        ref._mean, ref._var, ref._count = self._mean, self._var, self._count
        ref.should_estimate = self.should_estimate
        return ref

    @classmethod
    def from_reference(cls, ref, device=None):
        """<- an unpickled reference Normalizer (anything with _mean / _var / _count / clip)."""
        self = cls(np.shape(ref._mean), getattr(ref, "clip", 10.), device)
        return self.load_state({"_mean": ref._mean, "_var": ref._var, "_count": ref._count,
                                "should_estimate": getattr(ref, "should_estimate", True)})

    def __getstate__(self):
        return self.state()

    def __setstate__(self, st):
        self.__init__(st["shape"], st["clip"])
        self.load_state(st)


class NormObs:
    """State-only envs (base_wrapper.py:105-122): observation = filt(raw) after the training-mode update."""

    def __init__(self, state_dim, clipob=10., device=None):
        self._obs_normalizer = Normalizer((int(state_dim),), clipob, device)
        self.clipob = clipob
        self.training = True
        self.state_shape = int(state_dim)

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def observation(self, observation, out=None):
        """observation: cuda float64 [E][S] raw rows -> cuda float32 [E][S] (what torch.Tensor(ob).to(device) of
        collector/on_policy.py:93 would hold)."""
        E = observation.shape[0]
        if out is None:
            out = torch.empty(E, self.state_shape, dtype=torch.float32, device=observation.device)
        self._obs_normalizer.update_filt(observation, self.training, out32=out)
        return out


class NormObsWithImg(NormObs):
    """Proprio + depth envs (vision4leg/get_env.py:41-67): the proprio block is normalised, the depth stack passes
    through, and both land in one fp32 [E][S + C*H*W] row block — the layout `pf.explore` / `RolloutActor.step`
    take — without the host np.hstack."""

    def __init__(self, state_dim, image_elems, num_envs, clipob=10., device=None):
        super().__init__(state_dim, clipob, device)
        self.image_elems, self.num_envs = int(image_elems), int(num_envs)
        self.rows = torch.zeros(self.num_envs, self.state_shape + self.image_elems, dtype=torch.float32,
                                device=self._obs_normalizer.device)  # fixed address: a captured rollout step reads it

    def observation(self, raw_state, image, out=None):
        """raw_state: cuda float64 [E][S]; image: cuda float32 or float64 [E][C*H*W] (or [E][C][H][W]).
        Returns the fp32 observation rows (a fixed buffer unless `out` is given)."""
        rows = self.rows if out is None else out
        S = self.state_shape
        self._obs_normalizer.update_filt(raw_state, self.training, out32=rows, image=image, image_out=rows[:, S:])
        return rows

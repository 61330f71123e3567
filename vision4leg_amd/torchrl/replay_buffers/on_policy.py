"""On-policy rollout buffers (reference torchrl/replay_buffers/on_policy.py:10-92).

`OnPolicyReplayBuffer` keeps the reference's host float64 storage and minibatch iterator; its
`generalized_advantage_estimation` runs the fp64 HIP kernel (bit-identical to the reference's numpy loop).

`DeviceOnPolicyReplayBuffer` is the MI355X-native variant: observation rows are split and ingested into
HBM-resident arrays as they arrive (proprio fp32, depth stack in the contraction operand type), the per-step
scalars stay in small host arrays, and `one_iteration` yields *row indices* instead of 67 MB observation copies,
so the PPO update gathers straight from the resident arrays.
"""
import numpy as np
import torch

from .base import BaseReplayBuffer
from ... import engine


def _gae_on_device(rewards, values, terminals, time_limits, last_value, gamma, tau, use_tl, device, out=None, discount_only=False):
    """numpy [T,E,1] float64 in -> numpy [T,E,1] float64 advantages / returns via libv4l_hip's v4l_gae (discount_only:
    v4l_discount_reward; `tau` is then unused)."""
    T, E = rewards.shape[0], rewards.shape[1]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(T, -1)).to(device)
    r, v, t = up(rewards), up(values), up(terminals)
    tl = None
    if use_tl:
        tl_np = np.asarray(time_limits, dtype=np.float64).reshape(T, -1)
        tl = torch.from_numpy(np.ascontiguousarray(tl_np)).to(device)
        if tl.shape[1] == 1:
            tl = tl.reshape(T)
        elif tl.shape[1] != E:
            raise ValueError("time_limits has %d columns, expected 1 or %d" % (tl.shape[1], E))
    lv = torch.from_numpy(np.ascontiguousarray(last_value, dtype=np.float64).reshape(E)).to(device)
    if discount_only:
        return engine.discount_reward(r, v, t, tl, lv, gamma, use_tl, want32=True, out=out)
    return engine.gae(r, v, t, tl, lv, gamma, tau, use_tl, want32=True, out=out)


class OnPolicyReplayBufferBase:
    gae_device = None  # torch device used for the GAE kernel; defaults to the current cuda device

    def last_sample(self, sample_key):
        last = self._max_replay_buffer_size - 1
        return {key: getattr(self, "_" + key)[last] for key in sample_key}

    def generalized_advantage_estimation(self, last_value, gamma, tau):
        """GAE(lambda) over the stored epoch (reference on_policy.py:17-45); results land in _advs and
        _estimate_returns as float64 [T, E, 1] exactly like the reference."""
        if tau is None:  # the reference's loop fails on `gamma * None` (on_policy.py:31): PPO(gae=True) needs tau
            raise TypeError("generalized_advantage_estimation: tau is None (PPO(gae=True) needs tau; gae=False calls discount_reward)")
        self._estimate(last_value, gamma, tau, False)

    def _estimate(self, last_value, gamma, tau, discount_only):
        dev = self.gae_device or torch.device("cuda", torch.cuda.current_device())
        if not hasattr(self, "_gae_out"):
            self._gae_out = {}  # persistent device outputs: stable addresses across epochs
        advs, rets, a32, r32 = _gae_on_device(self._rewards, self._values, self._terminals,
                                              getattr(self, "_time_limits", None), last_value, gamma, tau,
                                              self.time_limit_filter, dev, self._gae_out, discount_only)
        shape = np.shape(self._rewards)
        self._advs = advs.cpu().numpy().reshape(shape)
        self._estimate_returns = rets.cpu().numpy().reshape(shape)
        self._advs32_dev, self._rets32_dev = a32.reshape(-1), r32.reshape(-1)

    def discount_reward(self, last_value, gamma):
        """Discounted rewards as return / advantage estimates (reference on_policy.py:47-71, PPO(gae=False)): the same
        fp64 HIP path on its own entry point (v4l_discount_reward, bit-identical to the reference's numpy loop)."""
        self._estimate(last_value, gamma, None, True)

    def one_iteration(self, batch_size, sample_key, shuffle):
        """Minibatches of batch_size/env_nums *time rows* (all envs of a row stay together), reference
        on_policy.py:73-92 — same np.random stream."""
        assert batch_size % self.env_nums == 0, "batch size should be dividable by env_nums"
        rows = batch_size // self.env_nums
        T = self._max_replay_buffer_size
        order = np.random.permutation(T) if shuffle else np.arange(T)
        for pos in range(0, T, rows):
            sel = order[pos:pos + rows]
            yield self._gather(sel, sample_key)

    def _gather(self, sel, sample_key):
        out = {}
        for key in sample_key:
            picked = getattr(self, "_" + key)[sel]
            out[key] = picked.reshape((len(sel) * self.env_nums,) + picked.shape[2:])
        return out


class OnPolicyReplayBuffer(OnPolicyReplayBufferBase, BaseReplayBuffer):
    pass


class DeviceOnPolicyReplayBuffer(OnPolicyReplayBufferBase, BaseReplayBuffer):
    """HBM-resident observations; attach(net) must be called (algo.PPO does) before the first add_sample."""

    def __init__(self, max_replay_buffer_size, env_nums=1, time_limit_filter=False, device=None):
        super().__init__(max_replay_buffer_size, env_nums, time_limit_filter)
        self.device = torch.device(device) if device is not None else None
        self._net = None
        self._state_dev = None
        self._image_dev = None

    def attach(self, hip_net, device):
        self._net = hip_net
        self.device = torch.device(device)
        self.gae_device = self.device
        slots = self._max_replay_buffer_size * self.env_nums
        self._state_dev, self._image_dev = hip_net.alloc_rollout(slots, self.device)
        self._acts_dev = None
        self._values32_dev = None
        self._logp_dev = None
        self._filed = False
        self._pinned = None

    def step_arrays(self, act_dim):
        """HBM arrays a RolloutActor files one env step into: (state, image, acts [slots][A], values [slots],
        logp [slots]) — `RolloutActor.attach(buf.step_arrays(A))`, then `add_sample(..., filed=True)` per step."""
        if self._net is None:
            raise RuntimeError("DeviceOnPolicyReplayBuffer.attach(net, device) has not been called")
        slots = self._max_replay_buffer_size * self.env_nums
        if self._acts_dev is None or tuple(self._acts_dev.shape) != (slots, act_dim):
            self._acts_dev = torch.zeros(slots, act_dim, dtype=torch.float32, device=self.device)
            self._values32_dev = torch.zeros(slots, dtype=torch.float32, device=self.device)
        if self._logp_dev is None:
            self._logp_dev = torch.zeros(slots, dtype=torch.float32, device=self.device)
        return self._state_dev, self._image_dev, self._acts_dev, self._values32_dev, self._logp_dev

    def add_sample(self, sample_dict, filed=False, **kwargs):
        """filed=True: a RolloutActor attached to step_arrays() has already written this step's observation rows, actions,
        values and log pi_old(a|s) at slot `_top * env_nums`; `obs` / `acts` / `values` in sample_dict are then optional
        and ignored. Otherwise the observation rows are ingested here (numpy rows are converted on the host once and
        uploaded through a pinned staging buffer)."""
        if self._net is None:
            raise RuntimeError("DeviceOnPolicyReplayBuffer.attach(net, device) has not been called")
        if filed:
            if self._logp_dev is None:
                raise RuntimeError("add_sample(filed=True) needs step_arrays() to have been handed to a RolloutActor")
            if not self._filed or self._top == 0:
                # a filed epoch starts: host arrays an earlier host-path epoch left behind (and last epoch's materialised
                # copies) must not shadow what __getattr__ serves from HBM
                for stale in ("_acts", "_values"):
                    self.__dict__.pop(stale, None)
                self._host_cache = {}
            self._filed = True
            self._filed_steps = self.__dict__.get("_filed_steps", 0) + 1
        else:
            if self._filed and self._top != 0:
                raise RuntimeError("DeviceOnPolicyReplayBuffer: filed and host steps mixed inside one epoch")
            self._filed = self._filed and self._top != 0
            obs = sample_dict["obs"]
            if isinstance(obs, np.ndarray):
                obs = self._upload(obs)
            obs = obs.reshape(self.env_nums, -1)
            self._net.ingest(obs, self._state_dev, self._image_dev, slot0=self._top * self.env_nums)
        for key, value in sample_dict.items():
            if key == "obs" or (filed and key in ("acts", "values")):
                continue
            if key == "next_obs":
                # only the epoch's LAST next_obs is ever read (last_sample at index T-1, on_rl_algo.py:24-26): that one is
                # copied (the env may reuse its buffer); the others are only referenced — copying E x 16.5 K doubles per env
                # step was the single largest host cost of the fast collector (tools/probe/collector_stages.py)
                if self._top == self._max_replay_buffer_size - 1 and not isinstance(value, torch.Tensor):
                    value = np.array(value, copy=True)
                self._last_next_obs = value
                continue
            self._store(key, value)
        self._advance()

    def _upload(self, rows):
        """float64 / float32 numpy rows -> fp32 device tensor through one of two pinned staging buffers (the copy is
        asynchronous; a buffer is reused only after the copy issued from it two steps ago has completed)."""
        if self._pinned is None or self._pinned[0].shape != rows.shape:
            self._pinned = [torch.empty(rows.shape, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._pin_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._pin_dev = [torch.empty(rows.shape, dtype=torch.float32, device=self.device) for _ in range(2)]
            self._pin_i = 0
        i = self._pin_i
        self._pin_i ^= 1
        self._pin_ev[i].synchronize()
        np.copyto(self._pinned[i].numpy(), rows, casting="same_kind")  # fp64 -> fp32 in the same pass
        self._pin_dev[i].copy_(self._pinned[i], non_blocking=True)
        self._pin_ev[i].record()
        return self._pin_dev[i]

    def generalized_advantage_estimation(self, last_value, gamma, tau):
        """As the base class; when the epoch's steps were filed by a RolloutActor the values are read where they are
        (fp32 network outputs in HBM, widened to fp64 like the reference's float64 buffer holds them) and the fp64 host
        copies `_advs` / `_estimate_returns` are produced on first access only."""
        if not self._filed:
            return super().generalized_advantage_estimation(last_value, gamma, tau)
        if tau is None:
            raise TypeError("generalized_advantage_estimation: tau is None (PPO(gae=True) needs tau; gae=False calls discount_reward)")
        self._estimate_filed(last_value, gamma, tau, False)

    def discount_reward(self, last_value, gamma):
        """As the base class; a filed epoch (PPO(gae=False) behind the fast collector) stays device-resident like the GAE path."""
        if not self._filed:
            return super().discount_reward(last_value, gamma)
        self._estimate_filed(last_value, gamma, None, True)

    def _estimate(self, last_value, gamma, tau, discount_only):
        super()._estimate(last_value, gamma, tau, discount_only)
        # host-path results are current: device-side fp64 results of an earlier filed epoch must not shadow them
        self.__dict__.pop("_advs_dev64", None)
        self.__dict__.pop("_rets_dev64", None)

    def _estimate_filed(self, last_value, gamma, tau, discount_only):
        T, E = self._max_replay_buffer_size, self.env_nums
        dev = self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(T, -1)).to(dev)
        tl = None
        if self.time_limit_filter:
            tl = up(self._time_limits)
            tl = tl.reshape(T) if tl.shape[1] == 1 and E > 1 else tl
        if isinstance(last_value, torch.Tensor):
            lv = last_value.to(dev, torch.float64).reshape(E)
        else:
            lv = torch.from_numpy(np.ascontiguousarray(last_value, dtype=np.float64).reshape(E)).to(dev)
        if not hasattr(self, "_gae_out"):
            self._gae_out = {}
        vals64 = self._values32_dev.view(T, E).double()
        if discount_only:
            advs, rets, a32, r32 = engine.discount_reward(up(self._rewards), vals64, up(self._terminals), tl, lv, gamma,
                                                          self.time_limit_filter, want32=True, out=self._gae_out)
        else:
            advs, rets, a32, r32 = engine.gae(up(self._rewards), vals64, up(self._terminals), tl, lv, gamma, tau,
                                              self.time_limit_filter, want32=True, out=self._gae_out)
        self._advs_dev64, self._rets_dev64 = advs, rets
        self.__dict__.pop("_advs", None)
        self.__dict__.pop("_estimate_returns", None)
        self._advs32_dev, self._rets32_dev = a32.reshape(-1), r32.reshape(-1)

    def __getattr__(self, name):  # lazily materialised host views of device-side results (filed epochs)
        if name == "_advs" and "_advs_dev64" in self.__dict__:
            self._advs = self._advs_dev64.cpu().numpy().reshape(self._max_replay_buffer_size, self.env_nums, 1)
            return self._advs
        if name == "_estimate_returns" and "_rets_dev64" in self.__dict__:
            self._estimate_returns = self._rets_dev64.cpu().numpy().reshape(self._max_replay_buffer_size, self.env_nums, 1)
            return self._estimate_returns
        if name in ("_acts", "_values") and self.__dict__.get("_filed"):
            # one D2H copy per (epoch position, array): entries of an older position are dropped when the next filed step
            # lands; both arrays of the CURRENT position stay (alternating _acts / _values reads do not re-copy)
            cache = self.__dict__.setdefault("_host_cache", {})
            pos = (self.__dict__.get("_top"), self.__dict__.get("_filed_steps"))
            key = (name,) + pos
            if key not in cache:
                for old in [k for k in cache if k[1:] != pos]:
                    del cache[old]
                src = self._acts_dev if name == "_acts" else self._values32_dev
                cache[key] = src.cpu().numpy().astype(np.float64).reshape(self._max_replay_buffer_size, self.env_nums, -1)
            return cache[key]
        raise AttributeError(name)

    def last_sample(self, sample_key):
        """Valid on a FULL buffer only (its one caller, process_epoch_samples, runs after the epoch's last step): `next_obs`
        is copied when step T-1 is stored and merely referenced — the env may overwrite it — for earlier steps."""
        last = self._max_replay_buffer_size - 1
        if "next_obs" in sample_key and self._top != 0:
            raise RuntimeError("DeviceOnPolicyReplayBuffer.last_sample: the epoch is not complete (%d of %d steps stored)"
                               % (self._top, self._max_replay_buffer_size))
        out = {}
        for key in sample_key:
            out[key] = self._last_next_obs if key == "next_obs" else getattr(self, "_" + key)[last]
        return out

    def device_rollout(self):
        """Device views the trainer gathers from: acts [slots][A], advs/rets/values [slots] (fp32) and, for epochs a
        RolloutActor filed, log pi_old(a|s) [slots] (None otherwise: the update evaluates the target policy)."""
        if self._filed:
            return (self._state_dev, self._image_dev, self._acts_dev, self._advs32_dev, self._rets32_dev,
                    self._values32_dev, self._logp_dev)
        slots = self._max_replay_buffer_size * self.env_nums
        acts = torch.from_numpy(np.ascontiguousarray(self._acts.reshape(slots, -1), dtype=np.float32))
        vals = torch.from_numpy(np.ascontiguousarray(self._values.reshape(slots), dtype=np.float32))
        if self._acts_dev is None or self._acts_dev.shape != acts.shape:
            self._acts_dev = torch.empty(acts.shape, dtype=torch.float32, device=self.device)
            self._values32_dev = torch.empty(vals.shape, dtype=torch.float32, device=self.device)
        self._acts_dev.copy_(acts)       # same device addresses every epoch: the update graph is keyed on them
        self._values32_dev.copy_(vals)
        return self._state_dev, self._image_dev, self._acts_dev, self._advs32_dev, self._rets32_dev, self._values32_dev, None

    def _gather(self, sel, sample_key):
        E = self.env_nums
        rowidx = (np.asarray(sel, dtype=np.int64)[:, None] * E + np.arange(E)[None, :]).reshape(-1).astype(np.int32)
        return {"rowidx": rowidx}

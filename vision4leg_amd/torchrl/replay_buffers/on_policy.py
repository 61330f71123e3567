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


def _gae_on_device(rewards, values, terminals, time_limits, last_value, gamma, tau, use_tl, device, out=None):
    """numpy [T,E,1] float64 in -> numpy [T,E,1] float64 advantages / returns via libv4l_hip's v4l_gae."""
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
    advs, rets, a32, r32 = engine.gae(r, v, t, tl, lv, gamma, tau, use_tl, want32=True, out=out)
    return advs, rets, a32, r32


class OnPolicyReplayBufferBase:
    gae_device = None  # torch device used for the GAE kernel; defaults to the current cuda device

    def last_sample(self, sample_key):
        last = self._max_replay_buffer_size - 1
        return {key: getattr(self, "_" + key)[last] for key in sample_key}

    def generalized_advantage_estimation(self, last_value, gamma, tau):
        """GAE(lambda) over the stored epoch (reference on_policy.py:17-45); results land in _advs and
        _estimate_returns as float64 [T, E, 1] exactly like the reference."""
        dev = self.gae_device or torch.device("cuda", torch.cuda.current_device())
        if not hasattr(self, "_gae_out"):
            self._gae_out = {}  # persistent device outputs: stable addresses across epochs
        advs, rets, a32, r32 = _gae_on_device(self._rewards, self._values, self._terminals,
                                              getattr(self, "_time_limits", None), last_value, gamma, tau,
                                              self.time_limit_filter, dev, self._gae_out)
        shape = np.shape(self._rewards)
        self._advs = advs.cpu().numpy().reshape(shape)
        self._estimate_returns = rets.cpu().numpy().reshape(shape)
        self._advs32_dev, self._rets32_dev = a32.reshape(-1), r32.reshape(-1)

    def discount_reward(self, last_value, gamma):
        raise NotImplementedError("vision4leg_amd: gae=False (discount_reward) is outside the HIP hot path; "
                                  "every shipped PPO config sets gae=true")

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

    def add_sample(self, sample_dict, **kwargs):
        if self._net is None:
            raise RuntimeError("DeviceOnPolicyReplayBuffer.attach(net, device) has not been called")
        obs = sample_dict["obs"]
        if isinstance(obs, np.ndarray):
            obs = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32)).to(self.device, non_blocking=True)
        obs = obs.reshape(self.env_nums, -1)
        self._net.ingest(obs, self._state_dev, self._image_dev, slot0=self._top * self.env_nums)
        for key, value in sample_dict.items():
            if key == "obs":
                continue
            if key == "next_obs":  # only the epoch's last next_obs is ever read (on_rl_algo.py:24-26)
                self._last_next_obs = np.array(value, copy=True)
                continue
            self._store(key, value)
        self._advance()

    def last_sample(self, sample_key):
        last = self._max_replay_buffer_size - 1
        out = {}
        for key in sample_key:
            out[key] = self._last_next_obs if key == "next_obs" else getattr(self, "_" + key)[last]
        return out

    def device_rollout(self):
        """Device views the trainer gathers from: acts [slots][A], advs/rets/values [slots] (fp32)."""
        slots = self._max_replay_buffer_size * self.env_nums
        acts = torch.from_numpy(np.ascontiguousarray(self._acts.reshape(slots, -1), dtype=np.float32))
        vals = torch.from_numpy(np.ascontiguousarray(self._values.reshape(slots), dtype=np.float32))
        if self._acts_dev is None or self._acts_dev.shape != acts.shape:
            self._acts_dev = torch.empty(acts.shape, dtype=torch.float32, device=self.device)
            self._values32_dev = torch.empty(vals.shape, dtype=torch.float32, device=self.device)
        self._acts_dev.copy_(acts)       # same device addresses every epoch: the update graph is keyed on them
        self._values32_dev.copy_(vals)
        return self._state_dev, self._image_dev, self._acts_dev, self._advs32_dev, self._rets32_dev, self._values32_dev

    def _gather(self, sel, sample_key):
        E = self.env_nums
        rowidx = (np.asarray(sel, dtype=np.int64)[:, None] * E + np.arange(E)[None, :]).reshape(-1).astype(np.int32)
        return {"rowidx": rowidx}

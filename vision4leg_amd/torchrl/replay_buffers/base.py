"""Host-side rollout storage with the reference's interface (torchrl/replay_buffers/base.py:20-38):
lazily allocated float64 numpy arrays [T, E, ...], one row of E env transitions per add_sample."""
import numpy as np


class BaseReplayBuffer:
    def __init__(self, max_replay_buffer_size, env_nums=1, time_limit_filter=False):
        self.env_nums = env_nums
        self._max_replay_buffer_size = max_replay_buffer_size // self.env_nums
        self._top = 0
        self._size = 0
        self.time_limit_filter = time_limit_filter

    def _store(self, key, value):
        name = "_" + key
        if not hasattr(self, name):
            # the env dimension is already part of `value`
            setattr(self, name, np.zeros((self._max_replay_buffer_size,) + np.shape(value)))
        getattr(self, name)[self._top, ...] = value

    def add_sample(self, sample_dict, **kwargs):
        for key, value in sample_dict.items():
            self._store(key, value)
        self._advance()

    def terminate_episode(self):
        pass

    def _advance(self):
        self._top = (self._top + 1) % self._max_replay_buffer_size
        if self._size < self._max_replay_buffer_size:
            self._size += 1

    def random_batch(self, batch_size, sample_key):
        assert batch_size % self.env_nums == 0, "batch size should be dividable by env_nums"
        rows = batch_size // self.env_nums
        idx = np.random.randint(0, self.num_steps_can_sample(), rows)
        out = {}
        for key in sample_key:
            picked = getattr(self, "_" + key)[idx]
            out[key] = picked.reshape((rows * self.env_nums,) + picked.shape[2:])
        return out

    def num_steps_can_sample(self):
        return self._size

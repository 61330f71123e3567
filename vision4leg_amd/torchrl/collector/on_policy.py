"""Vectorised on-policy collector with the reference's constructor and epoch protocol
(torchrl/collector/on_policy.py:84-155 on top of collector/base.py:10-52,113-127,177-288), host side only: the
simulator (`env.step / partial_reset / reset`) stays on the CPU cores exactly where the reference has it.

What changes is the per-step network work (collector/on_policy.py:90-100, 132-144). With a
`DeviceOnPolicyReplayBuffer` the step is the MI355X fast path:

  host rows [E][S+16384] float64 --(one fp64->fp32 pass into a pinned staging buffer, async H2D)--> HBM
  RolloutActor.step: shared encoder once, policy + value stacks, draw, and the step's observation rows / action /
  value / log pi_old(a|s) filed straight into the replay buffer's HBM arrays (2 launches)
  D2H of the [E][A] action only (the simulator needs it); values never leave the device.

With a host `OnPolicyReplayBuffer` it is the reference's own call sequence (`pf.explore`, `vf`, float64 host arrays).
The truncation bootstrap (`rewards += discount * vf(next_obs) * surpass_flag`, `terminals |= surpass_flag`) is kept;
the extra value forward is issued only on steps where some env actually ran past `max_episode_frames` (for steps with
plain terminations the reference multiplies it by zero).
"""
import copy
import os

import numpy as np
import torch

from ..replay_buffers import DeviceOnPolicyReplayBuffer


def _cast_threads_from_env():
    raw = os.environ.get("V4L_CAST_THREADS")
    if raw is None:
        return max(1, min(8, (os.cpu_count() or 1) // 2))
    try:
        n = int(raw)
    except ValueError:
        raise ValueError("vision4leg_amd: V4L_CAST_THREADS must be a positive integer, got %r" % raw) from None
    if n < 1:
        raise ValueError("vision4leg_amd: V4L_CAST_THREADS must be a positive integer, got %r" % raw)
    return n


class _NoGuard:
    def __enter__(self): return self
    def __exit__(self, *exc): return False


_NO_GUARD = _NoGuard()


class VecOnPolicyCollector:
    def __init__(self, vf, discount=0.99, *, env, eval_env, pf, replay_buffer, epoch_frames, train_render=False,
                 eval_episodes=1, eval_render=False, device="cpu", max_episode_frames=999):
        self.vf, self.pf = vf, pf
        self.discount = discount
        self.replay_buffer = replay_buffer
        self.env = env
        self.env.train()
        self.continuous = not hasattr(getattr(env, "action_space", None), "n")
        if not self.continuous:
            raise NotImplementedError("vision4leg_amd: the HIP policies are Gaussian (continuous actions) only")
        self.train_render = train_render
        if eval_env is None:  # collector/base.py:30-35
            eval_env = copy.deepcopy(env)
            if hasattr(env, "_obs_normalizer"):
                eval_env._obs_normalizer = env._obs_normalizer
        self.eval_env = eval_env
        self.eval_env._reward_scale = 1
        self.eval_episodes = eval_episodes
        self.eval_render = eval_render
        self.current_ob = self.env.reset()
        self.device = torch.device(device)
        self.to(self.device)
        E = self.env.env_nums
        self.epoch_frames = epoch_frames
        self.sample_epoch_frames = epoch_frames // E          # env steps per epoch (collector/base.py:180)
        self.max_episode_frames = max_episode_frames
        self.current_step = np.zeros((E, 1))
        self.train_rew = np.zeros((E, 1))
        self.train_rews = []
        self._actor = None
        self._cursor = -1
        self._pins = None
        self._split_pins = None
        self._split = None  # decided at the first fast-path step: does the actor take the observation split (bf16 depth rows)?
        # (the hand-over as a cast -> DMA pipeline — cast a row chunk, start its asynchronous copy, cast the next chunk under it,
        # kernels read HBM — was measured in round 5: 99 / 104 / 141 us per env step for 1 / 2 / 4 chunks against 87 for the
        # in-place read, every asynchronous copy costs ~15 us on this runtime; tools/probe/collector_pipe.py,
        # profiles/r5_collector_handover.txt. Not kept in the collector.)
        # threads of the per-step fp64 -> fp32 host cast (torch's intra-op pool; the env workers own the other cores). The count
        # is scoped to train_one_epoch (set on entry, put back on exit, also when a step raises): evaluation, logging and the
        # user's own torch CPU code between epochs keep the process's setting. V4L_CAST_THREADS overrides the count (1 = never
        # touch torch's setting).
        self.cast_threads = _cast_threads_from_env()
        # V4L_COLLECT_HOST_STEP=0: the per-step cast through torch's copy kernel + RolloutActor.step_host_split (the arrangement
        # before round 6; tests/test_gpu_collector.py compares the two bit for bit)
        self._host_step = os.environ.get("V4L_COLLECT_HOST_STEP", "1") != "0"
        self._in_epoch = False
        self.fast_path = isinstance(replay_buffer, DeviceOnPolicyReplayBuffer)

    # ---- reference plumbing (collector/base.py:54-58,113-115,166-174; on_policy.py:77-82) ----------------------------
    def start_episode(self):
        pass

    def finish_episode(self):
        pass

    @property
    def funcs(self):
        return {"pf": self.pf, "vf": self.vf}

    def to(self, device):
        for f in self.funcs.values():
            f.to(device)

    def terminate(self):
        self.env.close()
        self.eval_env.close()

    # ---- host -> HBM ----------------------------------------------------------------------------------------------
    def _upload(self, rows, host_only=False):
        """numpy [E][D] (float64 from the env wrappers) -> fp32 device rows via double-buffered pinned staging
        (what `torch.Tensor(self.current_ob).to(self.device)`, collector/on_policy.py:91-93, does from pageable memory).
        host_only: stop at the pinned fp32 staging buffer and return it (the fast path's rollout kernels read it in place)."""
        rows = np.asarray(rows)
        if self._pins is None or self._pins[0][0].shape != rows.shape:
            mk = lambda: (torch.empty(rows.shape, dtype=torch.float32).pin_memory(),
                          torch.empty(rows.shape, dtype=torch.float32, device=self.device), torch.cuda.Event())
            self._pins, self._pin_i = [mk(), mk()], 0
        host, dev, ev = self._pins[self._pin_i]
        self._pin_i ^= 1
        ev.synchronize()  # the copy issued from this staging buffer two uploads ago
        if rows.dtype == np.float64 and rows.flags.c_contiguous and self.cast_threads > 1:
            # the cast is the longest host stage of a step (E x 16.5 K doubles): torch's copy kernel on `cast_threads` intra-op
            # threads (40 us against 300 us for one numpy pass; a Python thread pool over numpy copies does not scale: the
            # casting copy holds the GIL). Round-to-nearest fp64 -> fp32 either way (== torch.Tensor(ob), on_policy.py:91).
            # The intra-op thread count is set once per epoch (train_one_epoch), not toggled per step.
            host.copy_(torch.from_numpy(rows))
        else:
            np.copyto(host.numpy(), rows, casting="same_kind")
        if host_only:
            return host  # (step_host returns once every block that reads these rows has written its output — action AND value
            # have arrived — and the staging is double-buffered on top of that: the buffer is free again by the step after next)
        dev.copy_(host, non_blocking=True)
        ev.record()
        return dev

    def _upload_split(self, rows):
        """The fast path's observation hand-over in the 16-bit compute modes: numpy [E][S + C*H*W] float64 rows -> a pinned fp32
        [E][S] proprio block and a pinned bfloat16 / float16 [E][C*H*W] depth block (double-buffered) that the rollout kernels read in
        place (RolloutActor.step_host_split). The depth stack crosses PCIe in the type the kernels round it to at ingest anyway:
        torch's float64 -> bfloat16 / float16 copy rounds through float32 (c10::BFloat16 / c10::Half are constructed from float),
        i.e. exactly torch.Tensor(ob) (collector/on_policy.py:91) followed by the kernels' fp32 -> operand-type cast — asserted bit
        for bit in tests/test_cpu.py::test_host_16bit_cast_is_the_two_step_rounding and tests/test_gpu_collector.py."""
        rows = np.asarray(rows)
        S = self.pf.hip.state_dim
        E, D = rows.shape
        if self._split_pins is None or self._split_pins[0][1].shape != (E, D - S):
            mk = lambda: (torch.empty(E, S, dtype=torch.float32).pin_memory() if S else None,
                          torch.empty(E, D - S, dtype=self.pf.hip.image_dtype()).pin_memory())
            self._split_pins, self._split_i = [mk(), mk()], 0
        prop, img = self._split_pins[self._split_i]
        self._split_i ^= 1
        src = torch.from_numpy(rows)
        if S:
            prop.copy_(src[:, :S])
        img.copy_(src[:, S:])
        return prop, img

    def _ensure_actor(self):
        if self._actor is None:
            from ..policies import RolloutActor
            buf = self.replay_buffer
            if buf._net is None:  # PPO.__init__ normally attaches it
                buf.attach(self.pf.hip, self.device)
            self._actor = RolloutActor(self.pf, self.vf, self.env.env_nums)
            self._actor.attach(buf.step_arrays(self.pf.hip.out_dim))
        return self._actor

    # ---- one vectorised env step -----------------------------------------------------------------------------------
    def take_actions(self):
        # (train_one_epoch holds the device guard for its whole loop: entering it per step costs ~3 us of interpreter time)
        with (_NO_GUARD if self._in_epoch else torch.cuda.device(self.device)):
            if self.fast_path:
                actor = self._ensure_actor()
                top = self.replay_buffer._top
                if top != self._cursor:  # the device-side step cursor advances by one per step; re-aim it when the buffer wrapped
                    actor.seek(top)
                self._cursor = top + 1
                # observation rows and action cross PCIe inside the two rollout launches themselves: the kernels read the
                # pinned staging buffer in place and write the [E][A] action into pinned host memory — no copy launches
                if self._split is None:
                    self._split = os.environ.get("V4L_COLLECT_SPLIT", "1") != "0" and actor.split_supported()
                ob = self.current_ob
                if (self._split and self._host_step and isinstance(ob, np.ndarray) and ob.dtype == np.float64
                        and ob.flags.c_contiguous):
                    # round 6: cast + launches + completion as ONE library call on the library's own thread pool
                    acts = np.array(actor.step_host_rows(ob, threads=self.cast_threads), dtype=np.float32, copy=True)
                elif self._split:  # 16-bit compute: the depth stack goes over in the operand type (half the PCIe bytes, same numbers)
                    acts = np.array(actor.step_host_split(*self._upload_split(self.current_ob)), dtype=np.float32, copy=True)
                else:
                    acts = np.array(actor.step_host(self._upload(self.current_ob, host_only=True)), dtype=np.float32, copy=True)
                values = None
            else:
                ob_tensor = self._upload(self.current_ob)
                acts = self.pf.explore(ob_tensor)["action"].detach().cpu().numpy()
                values = self.vf(ob_tensor).detach().cpu().numpy()
        # (ndarray methods below instead of the np.any / np.sum dispatchers: six reductions over [E][1] arrays per step, ~2 us each
        # of pure call overhead through the wrappers — tools/probe/host_step.py)
        if not np.isfinite(acts).all():  # collector/on_policy.py:102-107 ("NaN detected. BOOM")
            raise FloatingPointError("vision4leg_amd: non-finite action from the policy; observation rows "
                                     "finite: %s" % bool(np.isfinite(np.asarray(self.current_ob)).all()))
        next_obs, rewards, dones, infos = self.env.step(acts)
        if self.train_render:
            self.env.render()
        self.current_step += 1
        sample = {"obs": self.current_ob, "next_obs": next_obs, "acts": acts, "values": values, "rewards": rewards,
                  "terminals": dones,
                  "time_limits": infos["time_limit"][:, np.newaxis] if "time_limit" in infos else [False]}
        self.train_rew += rewards
        dones = np.asarray(dones)
        any_done = bool(dones.any())
        if any_done:
            self.train_rews += list(self.train_rew[dones])
            self.train_rew[dones] = 0
        surpass = self.current_step >= self.max_episode_frames
        any_surpass = bool(surpass.any())
        if any_done or any_surpass:
            if any_surpass:  # bootstrap the truncated envs with V(next_obs) (collector/on_policy.py:132-144)
                with torch.cuda.device(self.device):
                    last_value = self.vf(self._upload(next_obs)).detach().cpu().numpy()
                sample["rewards"] = rewards + self.discount * last_value * surpass
            ended = dones | surpass
            sample["terminals"] = ended
            next_obs = self.env.partial_reset(np.squeeze(ended, axis=-1))
            self.current_step[ended] = 0
            self.train_rew[ended] = 0
        if self.fast_path:
            del sample["obs"], sample["acts"], sample["values"]
            self.replay_buffer.add_sample(sample, filed=True)
        else:
            self.replay_buffer.add_sample(sample)
        self.current_ob = next_obs
        return rewards.sum() if isinstance(rewards, np.ndarray) else np.sum(rewards)

    def train_one_epoch(self):
        self.train_rews = []
        self.train_epoch_reward = 0
        self.env.train()
        before = torch.get_num_threads()
        scoped = self.fast_path and self.cast_threads > 1 and before != self.cast_threads
        if scoped:
            torch.set_num_threads(self.cast_threads)
        actor = None
        try:
            with torch.cuda.device(self.device):
                if self.fast_path:
                    # nothing steps an optimiser between two env steps of this loop: the "did the parameters change?" question
                    # (version counters of every parameter tensor, ~12 us per step for the two nets) is asked once, here
                    actor = self._ensure_actor()
                    actor.freeze_params(True)
                self._in_epoch = True
                for _ in range(self.sample_epoch_frames):
                    self.train_epoch_reward += self.take_actions()
        finally:
            self._in_epoch = False
            if actor is not None:
                actor.freeze_params(False)
            if scoped:
                torch.set_num_threads(before)
        if self.fast_path and self._actor is not None:
            with torch.cuda.device(self.device):
                self._actor.check()  # a lost device-side hand-over on the value side must not reach the update silently
        return {"train_rewards": self.train_rews, "train_epoch_reward": self.train_epoch_reward}

    def eval_one_epoch(self):
        """collector/base.py:237-288: `eval_episodes` rounds over the vectorised eval env with the policy mean."""
        if hasattr(self.env, "_obs_normalizer"):
            self.eval_env._obs_normalizer = copy.deepcopy(self.env._obs_normalizer)
        self.eval_env.eval()
        E = self.eval_env.env_nums
        eval_rews, traj_lens = [], []
        for _ in range(self.eval_episodes):
            epi_done = np.zeros((E, 1), dtype=bool)
            eval_obs = self.eval_env.reset()
            rews = np.zeros((E, 1))
            traj_len = np.zeros((E, 1))
            while not np.all(epi_done):
                with torch.cuda.device(self.device):
                    act = self.pf.eval_act(self._upload(eval_obs))
                if not np.isfinite(act).all():
                    raise FloatingPointError("vision4leg_amd: non-finite action from the policy (eval)")
                eval_obs, r, done, _ = self.eval_env.step(act)
                rews = rews + (1 - epi_done) * r
                traj_len = traj_len + (1 - epi_done)
                epi_done = epi_done | done
                if np.any(done):
                    eval_obs = self.eval_env.partial_reset(np.squeeze(done, axis=-1))
                if self.eval_render:
                    self.eval_env.render()
            eval_rews += list(rews)
            traj_lens += list(traj_len)
        return {"eval_rewards": eval_rews, "eval_traj_length": np.mean(traj_lens)}

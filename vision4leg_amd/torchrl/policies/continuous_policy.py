"""Gaussian policies of the PPO hot path with the reference's names and method protocol
(torchrl/policies/continuous_policy.py: GaussianContPolicyBase 77-146, ...BasicBias 239-254,
...NatureEncoderProj 257-272, ...ImpalaEncoderProj 275-290, ...Transformer 461-475, ...LocoTransformer 478-492).

A policy is a top-level HIP net plus the state-independent `logstd` parameter. `forward/explore/eval_act/update`
evaluate the trunk and the Gaussian head (clamp, exp, entropy, log-prob) with libv4l_hip.so kernels; only the
random draw of `explore` uses torch's generator so that seeded rollouts consume the RNG stream exactly like
the reference's `Normal(mean, std).sample()`.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from .. import networks

LOG_SIG_MAX = 2
LOG_SIG_MIN = -5

__all__ = ["LOG_SIG_MAX", "LOG_SIG_MIN", "GaussianContPolicyBase", "GaussianContPolicyBasicBias",
           "GaussianContPolicyImpalaEncoderProj", "GaussianContPolicyLocoTransformer",
           "GaussianContPolicyNatureEncoderProj", "GaussianContPolicyTransformer", "RolloutActor"]


class RolloutActor:
    """MI355X-native fast path for the collector's per-step network calls (reference
    torchrl/collector/on_policy.py:90-100 does `pf.explore(ob)` then `vf(ob)`, i.e. two full passes through the
    encoder both nets share): one captured launch sequence per env step that runs the shared encoder once, samples
    the action and reads the value. `step(ob)` returns {"action","mean","std","ent","value"}; with
    `attach(replay_buffer)` of a DeviceOnPolicyReplayBuffer it also files observation/action/value into HBM."""

    def __init__(self, pf, vf, env_nums, graph=False):
        from ...engine import HipActor
        self._actor = HipActor(pf.hip, vf.hip, env_nums, graph=graph)

    def attach(self, rollout):
        self._actor.attach(rollout)

    def seek(self, t):
        self._actor.seek(t)

    def draw_noise(self, n_steps):
        """One generator call for the exploration noise of the next n_steps steps (see HipActor.draw_noise)."""
        self._actor.draw_noise(n_steps)

    def check(self):
        """Raise if a device-side hand-over of the rollout step timed out since the last call (see HipActor.check)."""
        self._actor.check()

    def step(self, ob, deterministic=False):
        return self._actor.step(ob, deterministic)

    def step_host(self, ob_pinned, deterministic=False):
        """One env step straight from / to pinned host memory (see HipActor.step_host): -> numpy action [E][A]."""
        return self._actor.step_host(ob_pinned, deterministic)

    def split_supported(self):
        return self._actor.split_supported()

    def step_host_split(self, prop_pinned, img16_pinned, deterministic=False, on_device=False):
        """step_host with the depth stack handed over in bfloat16 (see HipActor.step_host_split): -> numpy action [E][A]."""
        return self._actor.step_host_split(prop_pinned, img16_pinned, deterministic, on_device=on_device)

    def split_device_buffers(self):
        return self._actor.split_device_buffers()

    def freeze_params(self, on):
        """Skip the per-step "did the parameters change?" check between freeze_params(True) and freeze_params(False) (see HipActor)."""
        self._actor.freeze_params(on)

    def step_host_rows(self, rows, deterministic=False, threads=8):
        """One env step from the env wrappers' float64 rows as ONE library call (see HipActor.step_host_rows): -> numpy action [E][A]."""
        return self._actor.step_host_rows(rows, deterministic, threads)

    def eval_act(self, x):
        """`pf.eval_act(x)` (policies/continuous_policy.py:78-83) on the fused step: the policy mean as a numpy array,
        no draw. With env_nums = 1 this is the batch-1 deployment call — the role the reference's TensorRT engine
        plays on the robot (a1_hardware/convert_tensor_rt/convert_locotransformer_trt.py:67-88): one or two launches,
        operands in the net's compute type."""
        out = self._actor.step(x.reshape(self._actor.E, -1), deterministic=True)
        return out["action"].squeeze(0).cpu().numpy()


class GaussianContPolicyBase:
    """Mixin: needs `self.hip` (HipNet), `self.logstd`, `self.tanh_action`."""

    def _init_policy(self, output_shape, tanh_action, log_init):
        self.continuous = True
        self.logstd = nn.Parameter(torch.ones(output_shape) * np.log(log_init))
        self.tanh_action = bool(tanh_action)  # TanhNormal head (policies/distribution.py:5-80); read by _net_cfg -> v4l_net_cfg

    def _gaussian(self, x, actions=None):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        mean, std, log_std, ent, logp = self.hip.gaussian(x2, self.logstd.data, actions, tanh_action=self.tanh_action)
        A = mean.shape[-1]
        mean, std = mean.view(*lead, A), std.view(*lead, A)
        ent = ent.view(*lead, 1)
        if logp is not None:
            logp = logp.view(*lead, 1)
        return mean, std, log_std, ent, logp

    def forward(self, x):
        mean, std, log_std, _, _ = self._gaussian(x)
        return mean, std, log_std

    def eval_act(self, x):
        mean, _, _ = self.forward(x)
        if self.tanh_action:  # continuous_policy.py:64-69
            mean = torch.tanh(mean)
        return mean.squeeze(0).cpu().numpy()

    def explore(self, x, return_log_probs=False, return_pre_tanh=False):
        mean, std, log_std, ent, _ = self._gaussian(x)
        # == Normal(mean, std).sample(): torch.normal(mean, std) is randn * std + mean on the same generator;
        # spelled out it skips normal()'s `std >= 0` validation (a reduction + a device->host sync per step)
        z = torch.addcmul(mean, std, torch.randn_like(mean))
        action = torch.tanh(z) if self.tanh_action else z  # TanhNormal.rsample (distribution.py:61-80)
        dic = {"mean": mean, "log_std": log_std, "std": std, "ent": ent}
        if self.tanh_action and (return_log_probs or return_pre_tanh):
            dic["pre_tanh"] = z.squeeze(0)
        if return_log_probs:
            A = mean.shape[-1]
            n = mean.numel() // A
            padded = torch.zeros(n, 16, dtype=torch.float32, device=mean.device)  # V4L_OUT_LD rows
            padded[:, :A] = mean.reshape(n, A)
            # tanh: dis.log_prob(action, pre_tanh_value=z) (continuous_policy.py:99-106) — the draw itself, not atanh(action)
            *_, logp = self.hip.gauss_head(padded, self.logstd.data, n, action, tanh_action=self.tanh_action,
                                           pre_tanh=z if self.tanh_action else None)
            dic["log_prob"] = logp.view(*mean.shape[:-1], 1)
        dic["action"] = action.squeeze(0)
        return dic

    def update(self, obs, actions):
        mean, std, log_std, ent, logp = self._gaussian(obs, actions)
        return {"mean": mean, "dis": Normal(mean, std), "log_std": log_std, "std": std, "log_prob": logp, "ent": ent}


class GaussianContPolicyBasicBias(networks.Net, GaussianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self._init_policy(output_shape, tanh_action, log_init)

    forward = GaussianContPolicyBase.forward


class GaussianContPolicyImpalaEncoderProj(networks.ImpalaEncoderProjNet, GaussianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self._init_policy(output_shape, tanh_action, log_init)

    forward = GaussianContPolicyBase.forward


class GaussianContPolicyLocoTransformer(networks.LocoTransformer, GaussianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self._init_policy(output_shape, tanh_action, log_init)

    forward = GaussianContPolicyBase.forward


class GaussianContPolicyNatureEncoderProj(networks.NatureEncoderProjNet, GaussianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self._init_policy(output_shape, tanh_action, log_init)

    forward = GaussianContPolicyBase.forward


class GaussianContPolicyTransformer(networks.Transformer, GaussianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self._init_policy(output_shape, tanh_action, log_init)

    forward = GaussianContPolicyBase.forward

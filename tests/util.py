"""Shared case definitions for the parity tests and the golden generator (tests/golden/make_golden.py).
Everything is regenerated from seeds so fixtures only carry reference *outputs*."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

STAT_KEYS = [
    "advs/mean", "advs/std", "advs/max", "advs/min", "Training/vf_loss", "grad_norm/vf",
    "Training/policy_loss", "logprob/mean", "logprob/std", "logprob/max", "logprob/min",
    "log_std/mean", "log_std/std", "log_std/max", "log_std/min", "ratio/max", "ratio/min",
    "grad_norm/pf",
]

# shipped hyper-parameters (config/rl/static/locotransformer/thin-goal.json:54-103); S=84 is what the configs
# produce (no_displacement), S=93 the paper/benchmark setting
CASES = {
    "loco_s93": dict(kind="loco", S=93, A=6, seed=0, B=64, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "loco_s84": dict(kind="loco", S=84, A=6, seed=1, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    # S = 90: the third proprio width the reference's env configurations produce (S in {84, 90, 93})
    "loco_s90": dict(kind="loco", S=90, A=6, seed=11, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "cnn_s93": dict(kind="cnn", S=93, A=6, seed=2, B=64, enc=[256, 256], head=[256, 256], visual_dim=256),
    "mlp_s93": dict(kind="mlp", S=93, A=6, seed=3, B=128, enc=[256, 256], head=[256, 256]),
    # a NON-shipped geometry (one layer, ff = 128, 128-wide MLPs, odd batch): runs on the general layer-by-layer kernels
    # (gemm_nt / attn / ln / gemm_tn), not on the fused ones that are specialised for the shipped shapes
    "loco_gen": dict(kind="loco", S=45, A=4, seed=4, B=22, enc=[128, 128], head=[128, 128], layers=1, ff=128),
    # mixed geometry (round 5): shipped layers and heads but a 128-wide proprio MLP -> layers and heads on the wave-per-sample
    # kernels in both passes, the backward launch ending with the layer-0 input gradient (its encoder-side tail is layer by layer)
    "loco_mix": dict(kind="loco", S=45, A=4, seed=20, B=22, enc=[128, 128], head=[256, 256], layers=2, ff=256),
    # vision-only variants (starter/ppo_locotransformer_vision_only.py, ppo_nature_cnn_vision_only.py
    # with the config/mpc_vision_only/{locotransformer,baseline}/thin-goal.json hyper-parameters): the observation row is the depth stack alone (S = 0)
    "loco_vis": dict(kind="loco_vis", S=0, A=6, seed=5, B=32, enc=[], head=[256, 256], layers=2, ff=256),
    "cnn_vis": dict(kind="cnn_vis", S=0, A=6, seed=6, B=32, enc=[], head=[256, 256]),
    # max_pool=True (nets.py:1022-1030, 884-889; no shipped config sets it): the depth tokens pooled by max — inside the
    # wave-per-sample kernels (round 5; arg-max mask rebuilt from the stack's output rows), pool_fwd / pool_bwd under
    # V4L_NO_WPS_LAYERS
    "loco_max": dict(kind="loco_max", S=84, A=6, seed=12, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "loco_vis_max": dict(kind="loco_vis_max", S=0, A=6, seed=13, B=32, enc=[], head=[256, 256], layers=2, ff=256),
    # token_norm=True (nets.py:815-818, 879-880, 1007-1008; no shipped config sets it): token_ln over every token in front of the
    # transformer layers (and the state_token_ln parameters nobody uses) — round 5: fused rollout step; update with the two layers
    # on the wave-per-sample kernels around token_ln's own launches
    "loco_tn": dict(kind="loco_tn", S=84, A=6, seed=16, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "loco_vis_tn": dict(kind="loco_vis_tn", S=0, A=6, seed=17, B=32, enc=[], head=[256, 256], layers=2, ff=256, param_tol_f32=4e-5),
    # use_pytorch_encoder=True (nets.py:955-963; no shipped config sets it): the layers are an nn.TransformerEncoder — clones of
    # one layer, so they start identical — with a final LayerNorm; round 5: as token_norm (final norm + pooling + heads as launches)
    "loco_pe": dict(kind="loco_pe", S=84, A=6, seed=18, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "loco_vis_pe": dict(kind="loco_vis_pe", S=0, A=6, seed=19, B=32, enc=[], head=[256, 256], layers=2, ff=256, param_tol_f32=4e-5),
    # tanh_action=True (TanhNormal head, policies/distribution.py:5-80; no shipped config sets it): the update's log-probs go
    # through atanh(stored action) with the -log(1 - a^2 + 1e-6) correction; the rollout step on the fused kernels (round 5)
    "mlp_tanh": dict(kind="mlp_tanh", S=93, A=6, seed=14, B=64, enc=[256, 256], head=[256, 256]),
    "loco_tanh": dict(kind="loco_tanh", S=84, A=6, seed=15, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    # the minibatch bench.py times (BASELINE configs[2] / configs[1], B = 1024): 4 samples per persistent block in the fused
    # conv backward (register-resident dW carried across samples), 256 layer blocks of 4 samples
    "loco_b1024": dict(kind="loco", S=93, A=6, seed=7, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    "cnn_b1024": dict(kind="cnn", S=93, A=6, seed=8, B=1024, enc=[256, 256], head=[256, 256], visual_dim=256),
    # ragged: 300 = 256 + 44 -> persistent conv blocks own 1 or 2 samples, 75 layer blocks of 4, MLP row tiles with a tail
    "loco_rag": dict(kind="loco", S=93, A=6, seed=9, B=300, enc=[256, 256], head=[256, 256], layers=2, ff=256),
    # PPO(clipped_value_loss=True): the clipped critic objective of ppo.py:105-112 (off in the shipped configs)
    "loco_clipvf": dict(kind="loco", S=84, A=6, seed=10, B=32, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                        clipped_value_loss=True),
}

GAE_CASES = {
    "t64_e8_tl_bcast": dict(T=64, E=8, seed=10, gamma=0.99, tau=0.95, tl_filter=True, tl_per_env=False),
    "t128_e32_tl_env": dict(T=128, E=32, seed=11, gamma=0.99, tau=0.95, tl_filter=True, tl_per_env=True),
    "t100_e1_nofilter": dict(T=100, E=1, seed=12, gamma=0.99, tau=0.95, tl_filter=False, tl_per_env=False),
    "t7_e3_edge": dict(T=7, E=3, seed=13, gamma=0.9, tau=1.0, tl_filter=True, tl_per_env=True),
}


# running observation normaliser: K vec-env steps of [E][S] raw proprio rows; `eval_from`: the
# step from which the wrapper is in eval mode (statistics frozen, base_wrapper.py:119-122)
OBSNORM_CASES = {
    "e32_s93": dict(E=32, S=93, K=6, seed=20, eval_from=5),
    "e16_s84": dict(E=16, S=84, K=5, seed=21, eval_from=99),
    "e1_s93": dict(E=1, S=93, K=4, seed=22, eval_from=3),    # ppo_state: one env, batch variance 0
    "e7_s5_wild": dict(E=7, S=5, K=4, seed=23, eval_from=3, scale=1e3),  # outliers in an eval step: the clip at +-10 is hit
}


def obsnorm_inputs(case):
    """-> (list of K float64 [E][S] raw batches, list of K training flags). Per-dimension offsets and scales, like
    joint angles vs velocities vs foot contacts in the real observation."""
    rs = np.random.RandomState(case["seed"])
    E, S, K = case["E"], case["S"], case["K"]
    sc = case.get("scale", 1.0)
    off, amp = rs.randn(S) * 3.0 * sc, np.exp(rs.randn(S)) * sc
    raws = [off + amp * rs.randn(E, S) for _ in range(K)]
    if sc > 1:
        raws[-1][0, 0] = 1e9  # clipped to +10
        raws[-1][1, 0] = -1e9
    return raws, [k < case["eval_from"] for k in range(K)]


from vision4leg_amd.recipes import build_nets, make_batch, obs_dim, obs_rows, share_encoder  # noqa: E402,F401
# (the starters' wiring and the synthetic-batch generators live in the package — vision4leg_amd/recipes.py — so that
# bench.py and __graft_entry__ do not import the test tree; the tests and make_golden.py use the same functions)


def small_param_names(pf_sd, vf_sd, limit=4096):
    return {(tag, k) for tag, sd in (("pf", pf_sd), ("vf", vf_sd)) for k, v in sd.items() if v.numel() <= limit}


def make_gae_inputs(g):
    rs = np.random.RandomState(g["seed"])
    T, E = g["T"], g["E"]
    tl = (rs.rand(T, E, 1) < 0.05).astype(np.float64) if g["tl_per_env"] else (rs.rand(T, 1) < 0.05).astype(np.float64)
    return {
        "rewards": rs.randn(T, E, 1),
        "values": rs.randn(T, E, 1).astype(np.float32).astype(np.float64),  # network outputs are fp32
        "terminals": (rs.rand(T, E, 1) < 0.05).astype(np.float64),
        "time_limits": tl,
        "last_value": rs.randn(E, 1).astype(np.float32).astype(np.float64),
    }


# measured parity errors, keyed "test/case/mode/what"; tests/conftest.py dumps them to gpurun_out/parity.json at the end
# of a GPU session (copied to profiles/parity_rNN.json for the record)
PARITY = {}


def record(key, value):
    PARITY[key] = float(value)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


class FakeVecEnv:
    """Deterministic stand-in for the reference's vectorised env (torchrl/env/vecenv.py protocol: env_nums, train / eval,
    reset, step -> (obs [E][D] float64, rewards [E][1], dones [E][1] bool, infos), partial_reset(mask), close). Rewards and
    the proprio part of the next observation depend on the actions, so a collector that feeds different actions diverges."""

    class _Space:
        def __init__(self, shape):
            self.shape = shape

    def __init__(self, E, S, A, img=4 * 64 * 64, seed=0, p_done=0.1, time_limit_key=False):
        self.env_nums, self.S, self.A, self.img = E, S, A, img
        self.rs = np.random.RandomState(seed)
        self.action_space = self._Space((A,))
        self.observation_space = self._Space((S,))
        self.image_channels = 4
        self.p_done, self.time_limit_key = p_done, time_limit_key
        self.training, self.closed, self._reward_scale = True, False, 1
        self.log = []

    def _rows(self, n):
        cols = [np.clip(self.rs.randn(n, self.S), -10, 10)]
        if self.img:
            cols.append(np.clip(self.rs.randn(n, self.img), -2.5, 2.8))
        return np.concatenate(cols, axis=1)

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def reset(self):
        self.ob = self._rows(self.env_nums)
        return self.ob.copy()

    def step(self, acts):
        acts = np.asarray(acts, dtype=np.float64).reshape(self.env_nums, self.A)
        self.log.append(acts.copy())
        self.ob = self._rows(self.env_nums)
        self.ob[:, :min(self.S, self.A)] += 0.5 * acts[:, :min(self.S, self.A)]
        rewards = acts.sum(axis=1, keepdims=True) + self.rs.randn(self.env_nums, 1)
        dones = self.rs.rand(self.env_nums, 1) < self.p_done
        infos = {}
        if self.time_limit_key:
            infos["time_limit"] = self.rs.rand(self.env_nums) < 0.05
        return self.ob.copy(), rewards, dones, infos

    def partial_reset(self, mask):
        mask = np.asarray(mask, dtype=bool).reshape(-1)
        self.ob[mask] = self._rows(int(mask.sum()))
        return self.ob.copy()

    def close(self):
        self.closed = True

"""How the reference's starters wire the hot-path nets, as functions (starter/ppo_{locotransformer,nature_cnn,state}.py:76-100,
starter/ppo_{locotransformer,nature_cnn}_vision_only.py:77-97), plus the synthetic observation / minibatch generators of
BASELINE.md section 3. Used by bench.py, __graft_entry__.smoke() and the tests (tests/util.py re-exports them), so that none
of the product-side entry points has to import the test tree.

`networks` / `policies` are passed in: either vision4leg_amd.torchrl's modules or — in tests/golden/make_golden.py — the
unmodified reference's, which is what pins the seeded construction to the reference bit for bit.
"""
import numpy as np

from ._lib import STAT_KEYS  # noqa: F401  (the 18 logger keys, torchrl/algo/on_policy/ppo.py:77-92,122-123,142-145)

IMG_ELEMS = 4 * 64 * 64  # depth stack of one observation row (vision4leg/envs/locomotion_gym_env_with_rich_information.py:122-131)


def obs_dim(case):
    """Columns of one observation row [S proprio | 4*64*64 depth] (env_utils.py:27-51); the state-only net has no image."""
    return case["S"] + (0 if case["kind"].startswith("mlp") else IMG_ELEMS)


def build_nets(networks, policies, case):
    """pf / vf exactly as the starters wire them (shared encoder / base). case: kind in {loco, cnn, mlp, loco_vis, cnn_vis},
    S, A, enc (encoder hidden_shapes), head (append_hidden_shapes) and, per kind, layers / ff / visual_dim."""
    S, A, kind = case["S"], case["A"], case["kind"]
    net = {"append_hidden_shapes": list(case["head"]), "base_type": networks.MLPBase}
    pol = {}
    if kind.endswith("_tanh"):  # the same nets with a TanhNormal policy head (tanh_action=True, continuous_policy.py:85-146)
        pol["tanh_action"] = True
        kind = kind[:-5]
    if kind in ("loco_max", "loco_vis_max"):  # the same nets with max_pool=True (nets.py:1022-1030, 884-889)
        net["max_pool"] = True
        kind = kind[:-4]
    if kind in ("loco_pe", "loco_vis_pe"):  # use_pytorch_encoder=True: nn.TransformerEncoder (cloned layers) + final norm (nets.py:955-963)
        net["use_pytorch_encoder"] = True
        kind = kind[:-3]
    if kind in ("loco_tn", "loco_vis_tn"):  # the same nets with token_norm=True (nets.py:815-818, 879-880, 1007-1008)
        net["token_norm"] = True
        kind = kind[:-3]
    if kind == "loco":
        net["transformer_params"] = [[1, case["ff"]] for _ in range(case["layers"])]
        encoder = networks.LocoTransformerEncoder(in_channels=4, state_input_dim=S, hidden_shapes=list(case["enc"]),
                                                  visual_dim=256)
        pf = policies.GaussianContPolicyLocoTransformer(encoder=encoder, state_input_shape=S,
                                                        visual_input_shape=(4, 64, 64), output_shape=A, **net, **pol)
        vf = networks.LocoTransformer(encoder=encoder, state_input_shape=S, visual_input_shape=(4, 64, 64),
                                      output_shape=1, **net)
    elif kind == "cnn":
        encoder = networks.NatureFuseEncoder(in_channels=4, state_input_dim=S, hidden_shapes=list(case["enc"]),
                                             visual_dim=case["visual_dim"])
        pf = policies.GaussianContPolicyImpalaEncoderProj(encoder=encoder, state_input_shape=S,
                                                          visual_input_shape=(4, 64, 64), output_shape=A, **net)
        vf = networks.ImpalaEncoderProjNet(encoder=encoder, state_input_shape=S, visual_input_shape=(4, 64, 64),
                                           output_shape=1, **net)
    elif kind == "loco_vis":
        net["transformer_params"] = [[1, case["ff"]] for _ in range(case["layers"])]
        encoder = networks.TransformerEncoder(in_channels=4)
        pf = policies.GaussianContPolicyTransformer(encoder=encoder, visual_input_shape=(4, 64, 64), output_shape=A, **net)
        vf = networks.Transformer(encoder=encoder, visual_input_shape=(4, 64, 64), output_shape=1, **net)
    elif kind == "cnn_vis":
        encoder = networks.NatureEncoder(in_channels=4)
        pf = policies.GaussianContPolicyNatureEncoderProj(encoder=encoder, visual_input_shape=(4, 64, 64),
                                                          output_shape=A, **net)
        vf = networks.NatureEncoderProjNet(encoder=encoder, visual_input_shape=(4, 64, 64), output_shape=1, **net)
    else:
        net["hidden_shapes"] = list(case["enc"])
        pf = policies.GaussianContPolicyBasicBias(input_shape=S, output_shape=A, **net, **pol)
        vf = networks.Net(input_shape=(S,), output_shape=1, **net)
        vf.base = pf.base
    return pf, vf


def share_encoder(pf_params, vf_params, kind):
    """Make vf's name->tensor dict reference pf's tensor objects for the shared sub-module (encoder.* / base.*), the way the
    starters hand ONE encoder module to both nets (starter/ppo_locotransformer.py:79-100, ppo_state.py:104)."""
    pre = "base." if kind.startswith("mlp") else "encoder."
    for k in vf_params:
        if k.startswith(pre):
            vf_params[k] = pf_params[k]
    return vf_params


def obs_rows(rs, n, case):
    """n float64 observation rows in BASELINE.md section 3's distributions: proprio ~ clip(N(0,1), +-10) (the normaliser's
    clip, torchrl/env/base_wrapper.py:91-94), depth ~ clip(N(0,1), -2.5, 2.8) (the range of the depth normalisation)."""
    cols = [np.clip(rs.randn(n, case["S"]), -10, 10)]
    if not case["kind"].startswith("mlp"):
        cols.append(np.clip(rs.randn(n, IMG_ELEMS), -2.5, 2.8))
    return np.concatenate(cols, axis=1)


def make_batch(case, update=0, B=None):
    """A seeded minibatch in the distributions of BASELINE.md section 3."""
    B = case["B"] if B is None else B
    rs = np.random.RandomState(1000 * case["seed"] + 17 + update)
    return {
        "obs": obs_rows(rs, B, case),
        "acts": 0.1 * rs.randn(B, case["A"]),
        "advs": rs.randn(B, 1),
        "estimate_returns": rs.randn(B, 1),
        "values": rs.randn(B, 1),
    }


class ZeroCostVecEnv:
    """A vectorised env that costs (almost) nothing: the reference's vec-env protocol (torchrl/env/vecenv.py: env_nums,
    train / eval, reset, step -> (obs [E][D] float64, rewards [E][1], dones [E][1] bool, infos), partial_reset(mask), close)
    handing out pre-generated float64 observation rows from a small pool. bench.py drives the product's collector over it so
    that everything BUT the simulator is timed: fp64 -> fp32 cast, pinned upload, the rollout launches, the action's D2H."""

    class _Space:
        def __init__(self, shape):
            self.shape = shape

    def __init__(self, E, case, pool=8, seed=0, p_done=0.01):
        self.env_nums, self.case = E, case
        rs = np.random.RandomState(seed)
        self._pool = [obs_rows(rs, E, case) for _ in range(pool)]
        self._rew = [rs.randn(E, 1) for _ in range(pool)]
        self._done = [rs.rand(E, 1) < p_done for _ in range(pool)]
        self._i = 0
        self.action_space = self._Space((case["A"],))
        self.observation_space = self._Space((case["S"],))
        self.image_channels = 4
        self._reward_scale = 1
        self.training = True

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def reset(self):
        return self._pool[0]

    def step(self, acts):
        self._i = (self._i + 1) % len(self._pool)
        return self._pool[self._i], self._rew[self._i], self._done[self._i], {}

    def partial_reset(self, mask):
        return self._pool[self._i]

    def close(self):
        pass

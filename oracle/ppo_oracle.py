"""TEST INFRASTRUCTURE — CPU oracle for the vision4leg PPO hot path. NOT part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(vision4leg_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED. The reference ships no tests or golden vectors for this path, so the pin
is the reference itself executed in the build container: tests/golden/make_golden.py imports the unmodified
reference from /root/reference, checks this restatement against it (forward, backward, PPO.update, GAE) and
writes the fixtures under tests/golden/ that the test-suite replays where /root/reference is absent.

A plain-PyTorch (CPU, fp32) restatement of the reference's arithmetic, which itself lives in PyTorch
(nn.Conv2d / nn.Linear / nn.TransformerEncoderLayer / optim.Adam / clip_grad_norm_ / distributions.Normal,
unpinned `torch` dependency of /root/reference/setup.py:244-249; restated against torch 2.10 semantics).
Every function cites the reference lines it follows (paths relative to /root/reference).

Three numeric flavours:
  mode="f32"  : what the reference computes.
  mode="bf16" : same graph, but every contraction (Linear / Conv2d and the two attention products Q K^T and
                P V, forward and both backward products of each) rounds its two operands to bfloat16
                (round-to-nearest-even) and accumulates in fp32 — the rounding points of the MI355X bf16 MFMA
                path (BASELINE.json north_star: "the transformer QK^T/softmax/V ... use MFMA bf16 tiles").
                Softmax, the 1/sqrt(d) scale, LayerNorm, residuals, losses and Adam stay fp32 in every flavour.
  mode="f16"  : the same rounding points with IEEE half operands (v_mfma_f32_16x16x32_f16: bf16's rate and bytes,
                11 significand bits instead of 8, 5 exponent bits instead of 8). The backward operands need the
                exponent range back: the loss is multiplied by grad_scale(B) — a power of two, so exact in fp32 —
                before the backward and the parameter gradients divided by it afterwards (PPOOracle._grads), which
                is where the HIP path scales (loss-gradient rows) and unscales (weight-grad reduction).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_SIG_MAX, LOG_SIG_MIN = 2.0, -5.0  # torchrl/policies/continuous_policy.py:8-9


def rbf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rf16(x):
    """IEEE half, round-to-nearest-even, subnormals kept, overflow -> inf (what v_cvt_f16_f32 does)."""
    return x.to(torch.float16).to(torch.float32)


ROUND = {"bf16": rbf16, "f16": rf16}
F16_SCALE_LOG2 = 4  # f16 flavour: loss-gradient rows carry 2^F16_SCALE_LOG2 * (rows of the minibatch rounded up to a power of two)


def grad_scale(mode, n):
    """The power of two the f16 flavour multiplies the loss (= every loss-gradient row) by: it takes the mean's 1/n out and puts
    the rows' bulk well inside half's normal range (1 for the other flavours; mirrors v4l_net_grad_scale of the HIP library)."""
    if mode != "f16":
        return 1.0
    return float(2 ** (F16_SCALE_LOG2 + max(0, int(n) - 1).bit_length()))


def probe_scale(mode, amax):
    """f16 flavour, gradient rows that are NOT a mean loss's (the tests' probe loss sum(out * w), w ~ N(0, 1)): the power of two
    that puts the largest |d(out)| element into [2^12, 2^13) — mirrors HipNet.f16_scale_for / v4l_net_set_grad_scale."""
    if mode != "f16" or not amax > 0:
        return 1.0
    return float(2.0 ** min(30, max(-14, math.floor(math.log2(8192.0 / amax)))))


class _LinearR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, rnd):
        xr, wr = rnd(x), rnd(w)
        ctx.save_for_backward(xr, wr)
        ctx.rnd = rnd
        return xr @ wr.t() + b

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = ctx.rnd(dy)
        dx = dyr @ wr
        dw = dyr.reshape(-1, dyr.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0)
        return dx, dw, db, None


class _ConvR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, rnd):
        xr, wr = rnd(x), rnd(w)
        ctx.save_for_backward(xr, wr)
        ctx.stride, ctx.rnd = stride, rnd
        return F.conv2d(xr, wr, b, stride=stride)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = ctx.rnd(dy)
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, stride=ctx.stride)
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, stride=ctx.stride)
        db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None


class _MatmulR(torch.autograd.Function):
    """a @ b with both operands rounded to the 16-bit operand type, fp32 accumulate; the backward products round theirs the same way."""

    @staticmethod
    def forward(ctx, a, b, rnd):
        ar, br = rnd(a), rnd(b)
        ctx.save_for_backward(ar, br)
        ctx.rnd = rnd
        return ar @ br

    @staticmethod
    def backward(ctx, dc):
        ar, br = ctx.saved_tensors
        dcr = ctx.rnd(dc)
        return dcr @ br.transpose(-1, -2), ar.transpose(-1, -2) @ dcr, None


# Contraction groups of the LocoTransformer / NatureCNN nets. `mode` is "f32", "bf16" or "f16" for every contraction, or a mapping
# group -> "f32" / "bf16" / "f16" (missing groups: "f32") — tools/bf16_attribution.py rounds ONE group at a time to attribute the
# bf16 flavour's distance to the fp32 reference.
GROUPS = ("conv", "upconv", "proprio", "projector", "in_proj", "attn", "out_proj", "ffn", "heads")


def _m(mode, group):
    return mode if isinstance(mode, str) else mode.get(group, "f32")


def matmul(a, b, mode, group="attn"):
    m = _m(mode, group)
    return a @ b if m == "f32" else _MatmulR.apply(a, b, ROUND[m])


def linear(x, w, b, mode, group="heads"):
    m = _m(mode, group)
    return F.linear(x, w, b) if m == "f32" else _LinearR.apply(x, w, b, ROUND[m])


def conv2d(x, w, b, stride, mode, group="conv"):
    m = _m(mode, group)
    return F.conv2d(x, w, b, stride=stride) if m == "f32" else _ConvR.apply(x, w, b, stride, ROUND[m])


# ------------------------------------------------------------------------------------------ building blocks
def mlp(p, prefix, x, n, mode, group="proprio"):
    """MLPBase: Linear+ReLU per hidden layer, last activation is ReLU too (torchrl/networks/base.py:8-44)."""
    for i in range(n):
        x = torch.relu(linear(x, p["%s.%d.weight" % (prefix, 2 * i)], p["%s.%d.bias" % (prefix, 2 * i)], mode, group))
    return x


def head(p, prefix, x, n_hidden, mode):
    """append fcs: (Linear+ReLU)*n_hidden then Linear (nets.py:35-50, 224-243, 973-992)."""
    x = mlp(p, prefix, x, n_hidden, mode, "heads")
    return linear(x, p["%s.%d.weight" % (prefix, 2 * n_hidden)], p["%s.%d.bias" % (prefix, 2 * n_hidden)], mode, "heads")


def nature_cnn(p, prefix, img, mode):
    """NatureEncoder: conv 8/4, 4/2, 3/1 with ReLU (base.py:317-324). img [B,4,64,64] -> [B,64,4,4]."""
    x = img
    for i, s in zip((0, 2, 4), (4, 2, 1)):
        x = torch.relu(conv2d(x, p["%s.layers.%d.weight" % (prefix, i)], p["%s.layers.%d.bias" % (prefix, i)], s, mode))
    return x


def transformer_layer(p, prefix, x, mode):
    """nn.TransformerEncoderLayer(64, nhead=1, ff, dropout=0): post-norm, ReLU FFN, LN eps 1e-5
    (built at nets.py:948-955, applied :1009-1011). x is batch-first [B,17,64] here; the reference runs the
    same arithmetic sequence-first."""
    d = x.shape[-1]
    qkv = linear(x, p[prefix + ".self_attn.in_proj_weight"], p[prefix + ".self_attn.in_proj_bias"], mode, "in_proj")
    q, k, v = qkv.split(d, dim=-1)
    scores = matmul(q, k.transpose(-1, -2), mode) * (1.0 / math.sqrt(d))
    ctx = matmul(torch.softmax(scores, dim=-1), v, mode)
    a = linear(ctx, p[prefix + ".self_attn.out_proj.weight"], p[prefix + ".self_attn.out_proj.bias"], mode, "out_proj")
    x = F.layer_norm(x + a, (d,), p[prefix + ".norm1.weight"], p[prefix + ".norm1.bias"], 1e-5)
    f = torch.relu(linear(x, p[prefix + ".linear1.weight"], p[prefix + ".linear1.bias"], mode, "ffn"))
    f = linear(f, p[prefix + ".linear2.weight"], p[prefix + ".linear2.bias"], mode, "ffn")
    return F.layer_norm(x + f, (d,), p[prefix + ".norm2.weight"], p[prefix + ".norm2.bias"], 1e-5)


def _count(p, fmt):
    n = 0
    while (fmt % (2 * n)) in p:
        n += 1
    return n


# ------------------------------------------------------------------------------------------ the three nets
def split_obs(x, S):
    """state = x[..., :S]; img = x[..., S:].view(B,4,64,64) (nets.py:997-1000, 249-252)."""
    return x[..., :S], x[..., S:].reshape(-1, 4, 64, 64)


def _layers(p, tok, mode, taps=None):
    """The transformer stack: visual_append_layers.N (a ModuleList) or, with use_pytorch_encoder=True, an nn.TransformerEncoder
    — visual_trans_encoder.layers.N followed by its final LayerNorm visual_trans_encoder.norm (nets.py:955-963, 1009-1013)."""
    pe = "visual_trans_encoder.layers.0.norm1.weight" in p
    fmt = "visual_trans_encoder.layers.%d" if pe else "visual_append_layers.%d"
    l = 0
    while ((fmt + ".norm1.weight") % l) in p:
        tok = transformer_layer(p, fmt % l, tok, mode)
        if taps is not None:
            taps["x%d" % (l + 1)] = tok
        l += 1
    if pe:
        tok = F.layer_norm(tok, (tok.shape[-1],), p["visual_trans_encoder.norm.weight"], p["visual_trans_encoder.norm.bias"], 1e-5)
    return tok


def loco_forward(p, x, S, mode="f32", taps=None, max_pool=False):
    """LocoTransformer.forward + LocoTransformerEncoder.forward (nets.py:996-1038, base.py:550-626), depth-only.
    max_pool: nets.py:1022-1023 — the depth tokens are pooled by `.max(dim=0)[0]` instead of the mean."""
    state, img = split_obs(x, S)
    B = state.shape[0]
    c3 = nature_cnn(p, "encoder.depth_visual_base", img, mode)                                  # base.py:578
    up = conv2d(c3, p["encoder.depth_up_conv.weight"], p["encoder.depth_up_conv.bias"], 1, mode, "upconv")   # base.py:581
    depth_tok = up.reshape(B, 64, 16).permute(0, 2, 1)                                          # base.py:602-608
    ne = _count(p, "encoder.base.seq_fcs.%d.weight")
    h = mlp(p, "encoder.base.seq_fcs", state, ne, mode)                                         # base.py:611
    st = torch.relu(linear(h, p["encoder.state_projector.projection.0.weight"],
                           p["encoder.state_projector.projection.0.bias"], mode, "proprio"))    # base.py:613
    tok = torch.cat([st.unsqueeze(1), depth_tok], dim=1)                                        # base.py:617-622
    if "token_ln.weight" in p:                                                                  # token_norm=True: nets.py:1007-1008
        tok = F.layer_norm(tok, (tok.shape[-1],), p["token_ln.weight"], p["token_ln.bias"], 1e-5)
    if taps is not None:
        taps["c3"] = c3; taps["x0"] = tok
    tok = _layers(p, tok, mode, taps)                                                           # nets.py:1009-1013
    depth = tok[:, 1:17].max(dim=1)[0] if max_pool else tok[:, 1:17].mean(dim=1)
    pooled = torch.cat([tok[:, 0], depth], dim=-1)                                              # nets.py:1015-1034
    nh = _count(p, "visual_seq_append_fcs.%d.weight") - 1
    return head(p, "visual_seq_append_fcs", pooled, nh, mode)                                   # nets.py:1036


def cnn_forward(p, x, S, mode="f32", taps=None):
    """ImpalaEncoderProjNet.forward + NatureFuseEncoder.forward (nets.py:247-262, base.py:371-385)."""
    state, img = split_obs(x, S)
    c3 = nature_cnn(p, "encoder.visual_base", img, mode)
    vis = torch.relu(linear(c3.flatten(1), p["encoder.visual_projector.projection.0.weight"],
                            p["encoder.visual_projector.projection.0.bias"], mode, "projector"))
    ne = _count(p, "encoder.base.seq_fcs.%d.weight")
    h = mlp(p, "encoder.base.seq_fcs", state, ne, mode)
    nh = _count(p, "seq_append_fcs.%d.weight") - 1
    return head(p, "seq_append_fcs", torch.cat([vis, h], dim=-1), nh, mode)


def mlp_forward(p, x, S=None, mode="f32", taps=None):
    """Net.forward with an MLPBase trunk (nets.py:52-55)."""
    ne = _count(p, "base.seq_fcs.%d.weight")
    nh = _count(p, "seq_append_fcs.%d.weight") - 1
    return head(p, "seq_append_fcs", mlp(p, "base.seq_fcs", x, ne, mode), nh, mode)


def loco_vis_forward(p, x, S=0, mode="f32", taps=None, max_pool=False):
    """Transformer.forward + TransformerEncoder.forward, depth only (nets.py:868-906, base.py:430-494): the observation
    row is the depth stack; 16 patch tokens, mean over all of them (out[0:1+16] of a 16-token sequence), head."""
    img = x.reshape(-1, 4, 64, 64)                                                              # nets.py:869-871
    B = img.shape[0]
    c3 = nature_cnn(p, "encoder.depth_visual_base", img, mode)                                  # base.py:449
    up = conv2d(c3, p["encoder.depth_up_conv.weight"], p["encoder.depth_up_conv.bias"], 1, mode, "upconv")   # base.py:452
    tok = up.reshape(B, 64, 16).permute(0, 2, 1)                                                # base.py:474-481
    if "token_ln.weight" in p:                                                                  # token_norm=True: nets.py:879-880
        tok = F.layer_norm(tok, (tok.shape[-1],), p["token_ln.weight"], p["token_ln.bias"], 1e-5)
    if taps is not None:
        taps["c3"] = c3; taps["x0"] = tok
    tok = _layers(p, tok, mode)                                                                 # nets.py:881-885
    pooled = tok[:, 0:17].max(dim=1)[0] if max_pool else tok[:, 0:17].mean(dim=1)               # nets.py:886-889
    nh = _count(p, "visual_seq_append_fcs.%d.weight") - 1
    return head(p, "visual_seq_append_fcs", pooled, nh, mode)                                   # nets.py:904


def cnn_vis_forward(p, x, S=0, mode="f32", taps=None):
    """NatureEncoderProjNet.forward + NatureEncoder.forward with Flatten (nets.py:176-191, base.py:333-342)."""
    img = x.reshape(-1, 4, 64, 64)
    c3 = nature_cnn(p, "encoder", img, mode)
    nh = _count(p, "seq_append_fcs.%d.weight") - 1
    return head(p, "seq_append_fcs", c3.flatten(1), nh, mode)


def loco_max_forward(p, x, S, mode="f32", taps=None):
    return loco_forward(p, x, S, mode, taps, max_pool=True)


def loco_vis_max_forward(p, x, S=0, mode="f32", taps=None):
    return loco_vis_forward(p, x, S, mode, taps, max_pool=True)


FORWARDS = {"loco": loco_forward, "cnn": cnn_forward, "mlp": mlp_forward, "loco_vis": loco_vis_forward,
            "cnn_vis": cnn_vis_forward, "loco_max": loco_max_forward, "loco_vis_max": loco_vis_max_forward,
            # tanh_action=True policies: the same nets, a TanhNormal head (PPOOracle reads the suffix)
            "mlp_tanh": mlp_forward, "loco_tanh": loco_forward,
            # token_norm=True: the forwards see token_ln.* among the parameters
            "loco_tn": loco_forward, "loco_vis_tn": loco_vis_forward,
            # use_pytorch_encoder=True: the forwards see visual_trans_encoder.* among the parameters
            "loco_pe": loco_forward, "loco_vis_pe": loco_vis_forward}


# ------------------------------------------------------------------------------------------ Gaussian head
def gaussian(mean, logstd_param):
    """forward of the policy classes: clamp, exp, broadcast (continuous_policy.py:486-492)."""
    log_std = torch.clamp(logstd_param, LOG_SIG_MIN, LOG_SIG_MAX)
    std = torch.exp(log_std).unsqueeze(0).expand_as(mean)
    return mean, std, log_std


def log_prob_entropy(mean, std, actions, tanh_action=False):
    """Normal(mean,std).log_prob(a).sum(-1,keepdim) and .entropy().sum(-1,keepdim) (continuous_policy.py:127-146;
    torch/distributions/normal.py). tanh_action: TanhNormal.log_prob of the stored post-tanh actions
    (policies/distribution.py:38-51): Normal.log_prob(log((1 + a) / (1 - a)) / 2) - log(1 - a * a + 1e-6); the entropy is
    the Normal's (distribution.py:82-83)."""
    var = std ** 2
    corr = 0.0
    if tanh_action:
        corr = torch.log(1 - actions * actions + 1e-6)
        actions = torch.log((1 + actions) / (1 - actions)) / 2
    lp = -((actions - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi)) - corr
    ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)
    return lp.sum(-1, keepdim=True), ent.sum(-1, keepdim=True)


# ------------------------------------------------------------------------------------------ GAE
def gae(rewards, values, terminals, time_limits, last_value, gamma, tau, time_limit_filter):
    """generalized_advantage_estimation (torchrl/replay_buffers/on_policy.py:17-45): float64 numpy, same
    expression order as the reference. Arrays [T,E,1]; time_limits [T,1] or [T,E,1]; last_value [E,1]."""
    rewards, values, terminals = (np.asarray(a, dtype=np.float64) for a in (rewards, values, terminals))
    T = len(rewards)
    vals = np.concatenate([values, np.array([last_value], dtype=np.float64)], 0)
    A = 0
    advs, rets = [None] * T, [None] * T
    for t in reversed(range(T)):
        delta = rewards[t] + (1 - terminals[t]) * gamma * vals[t + 1] - vals[t]
        A = delta + (1 - terminals[t]) * gamma * tau * A
        if time_limit_filter:
            A = A * (1 - np.asarray(time_limits[t], dtype=np.float64))
        advs[t] = A
        rets[t] = A + vals[t]
    return np.array(advs), np.array(rets)


def discount_reward(rewards, values, terminals, time_limits, last_value, gamma, time_limit_filter):
    """discount_reward (torchrl/replay_buffers/on_policy.py:47-71; PPO(gae=False)): float64 numpy, the reference's
    expression order. Shapes as gae()."""
    rewards, values, terminals = (np.asarray(a, dtype=np.float64) for a in (rewards, values, terminals))
    T = len(rewards)
    R = np.asarray(last_value, dtype=np.float64)
    advs, rets = [None] * T, [None] * T
    for t in reversed(range(T)):
        if time_limit_filter:
            tl = np.asarray(time_limits[t], dtype=np.float64)
            R = (rewards[t] + (1 - terminals[t]) * gamma * R * (1 - tl)) + tl * values[t]
        else:
            R = rewards[t] + (1 - terminals[t]) * gamma * R
        advs[t] = R - values[t]
        rets[t] = R
    return np.array(advs), np.array(rets)


# ------------------------------------------------------------------------------------------ optimiser pieces
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (ppo.py:73-74,118-119): L2 over all grads, scale by
    min(1, max_norm/(norm+1e-6)); returns the pre-clip norm."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(params, grads, state, lr, step, betas=(0.9, 0.999), eps=1e-5):
    """torch.optim.Adam, no weight decay, eps added after sqrt(v_hat) (a2c.py:30-40;
    torch/optim/adam.py::_single_tensor_adam operation order). state: list of (m, v) per param, updated in place."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    with torch.no_grad():
        for p, g, (m, v) in zip(params, grads, state):
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(eps)
            p.addcdiv_(m, denom, value=-step_size)


class PPOOracle:
    """PPO.update (ppo.py:42-153) over name->tensor parameter dicts.

    pf / vf dicts may hold the *same tensor objects* for shared encoder entries (the reference passes one
    encoder module to both nets, starter/ppo_locotransformer.py:79-100): each optimiser then keeps its own
    moments for them (a2c.py:30-40) and the critic step is visible to the actor forward (ppo.py:150-151)."""

    def __init__(self, kind, pf, vf, target_pf, S, mode="f32", clip_para=0.2, entropy_coeff=0.005, max_norm=0.5,
                 clipped_value_loss=False):
        self.fwd = FORWARDS[kind]
        self.tanh_action = kind.endswith("_tanh")  # policies built with tanh_action=True (TanhNormal head)
        self.pf, self.vf, self.tpf = pf, vf, target_pf
        self.S, self.mode = S, mode
        self.clip_para, self.entropy_coeff, self.max_norm = clip_para, entropy_coeff, max_norm
        self.clipped_value_loss = clipped_value_loss
        self.pf_keys = [k for k in pf]
        self.vf_keys = [k for k in vf]
        self.pf_state = [(torch.zeros_like(pf[k]), torch.zeros_like(pf[k])) for k in self.pf_keys]
        self.vf_state = [(torch.zeros_like(vf[k]), torch.zeros_like(vf[k])) for k in self.vf_keys]
        self.step = 0
        self.last_grads = {}

    def sync_target(self):
        """copy_model_params_from_to(pf, target_pf) (algo/utils.py:23-25, ppo.py:34)"""
        with torch.no_grad():
            for k in self.pf_keys:
                self.tpf[k].copy_(self.pf[k])

    def _grads(self, loss, pdict, keys, n):
        for k in keys:
            pdict[k].requires_grad_(True)
        sc = grad_scale(self.mode if isinstance(self.mode, str) else "f32", n)  # f16: scaled backward (exact power of two)
        gs = torch.autograd.grad(loss * sc if sc != 1.0 else loss, [pdict[k] for k in keys], allow_unused=True)
        for k in keys:
            pdict[k].requires_grad_(False)
        # (log sigma's gradient never passes a contraction: the HIP loss kernel writes it unscaled; x * sc / sc is exact anyway)
        return [(g / sc if sc != 1.0 else g) if g is not None else torch.zeros_like(pdict[k]) for g, k in zip(gs, keys)]

    def update(self, obs, acts, advs, est_rets, old_values, lr_pf, lr_vf):
        """obs [B,D], acts [B,A], advs/est_rets/old_values [B,1] (fp32 tensors). Returns the 18-key info dict."""
        info = {}
        self.step += 1
        info["advs/mean"] = advs.mean().item()
        info["advs/std"] = advs.std().item()
        info["advs/max"] = advs.max().item()
        info["advs/min"] = advs.min().item()
        advs = (advs - advs.mean()) / (advs.std() + 1e-5)                                    # ppo.py:148
        # ---- critic (ppo.py:94-123)
        for k in self.vf_keys:
            self.vf[k].requires_grad_(True)
        values = self.fwd(self.vf, obs, self.S, self.mode)
        if self.clipped_value_loss:
            vc = old_values + (values - old_values).clamp(-self.clip_para, self.clip_para)
            vf_loss = 0.5 * torch.max((values - est_rets).pow(2), (vc - est_rets).pow(2)).mean()
        else:
            vf_loss = F.mse_loss(values, est_rets)
        g = self._grads(vf_loss, self.vf, self.vf_keys, obs.shape[0])
        self.last_grads["vf"] = dict(zip(self.vf_keys, g))
        g, gn = clip_grad_norm(g, self.max_norm)
        adam_step([self.vf[k] for k in self.vf_keys], g, self.vf_state, lr_vf, self.step)
        info["Training/vf_loss"] = vf_loss.item()
        info["grad_norm/vf"] = gn.item()
        # ---- actor (ppo.py:42-92)
        for k in self.pf_keys:
            self.pf[k].requires_grad_(True)
        mean = self.fwd({k: v for k, v in self.pf.items() if k != "logstd"}, obs, self.S, self.mode)
        mean, std, log_std = gaussian(mean, self.pf["logstd"])
        log_probs, ent = log_prob_entropy(mean, std, acts, self.tanh_action)
        with torch.no_grad():
            tmean = self.fwd({k: v for k, v in self.tpf.items() if k != "logstd"}, obs, self.S, self.mode)
            tmean, tstd, _ = gaussian(tmean, self.tpf["logstd"])
            target_log_probs, _ = log_prob_entropy(tmean, tstd, acts, self.tanh_action)
        ratio = torch.exp(log_probs - target_log_probs)
        s1 = ratio * advs
        s2 = torch.clamp(ratio, 1.0 - self.clip_para, 1.0 + self.clip_para) * advs
        policy_loss = -torch.mean(torch.min(s2, s1)) - self.entropy_coeff * ent.mean()
        g = self._grads(policy_loss, self.pf, self.pf_keys, obs.shape[0])
        self.last_grads["pf"] = dict(zip(self.pf_keys, g))
        g, gn = clip_grad_norm(g, self.max_norm)
        adam_step([self.pf[k] for k in self.pf_keys], g, self.pf_state, lr_pf, self.step)
        info["Training/policy_loss"] = policy_loss.item()
        lp = log_probs.detach()
        info["logprob/mean"], info["logprob/std"] = lp.mean().item(), lp.std().item()
        info["logprob/max"], info["logprob/min"] = lp.max().item(), lp.min().item()
        ls = log_std.detach()
        info["log_std/mean"], info["log_std/std"] = ls.mean().item(), ls.std().item()
        info["log_std/max"], info["log_std/min"] = ls.max().item(), ls.min().item()
        info["ratio/max"], info["ratio/min"] = ratio.max().item(), ratio.min().item()
        info["grad_norm/pf"] = gn.item()
        return info


def synthetic_rollout(T, E, S, A, seed=0, with_images=True):
    """Synthetic epoch in the distributions of BASELINE.md §3 (numpy RandomState(seed))."""
    rs = np.random.RandomState(seed)
    state = np.clip(rs.randn(T, E, S), -10, 10)
    parts = [state]
    if with_images:
        parts.append(np.clip(rs.randn(T, E, 4 * 64 * 64), -2.5, 2.8))
    obs = np.concatenate(parts, axis=-1)
    return {
        "obs": obs,
        "acts": 0.1 * rs.randn(T, E, A),
        "values": rs.randn(T, E, 1),
        "rewards": rs.randn(T, E, 1),
        "terminals": (rs.rand(T, E, 1) < 0.01).astype(np.float64),
        "time_limits": (rs.rand(T, E, 1) < 0.002).astype(np.float64),
        "last_value": rs.randn(E, 1),
    }

"""Top-level networks of the PPO hot path with the reference's names and constructor signatures
(torchrl/networks/nets.py: Net 16-55, NatureEncoderProjNet 133-191, ImpalaEncoderProjNet 194-262, Transformer
784-906, LocoTransformer 909-1038).

Each module owns its parameters as ordinary nn.Parameters (reference state_dict keys, reference seeded
initialisation) and evaluates `forward` by enqueuing the hand-written gfx950 kernels of libv4l_hip.so on the
current HIP stream. There is no autograd graph and no torch arithmetic: gradients are produced by the
library's own backward pass inside algo.PPO.
"""
import copy

import torch
import torch.nn as nn

from . import init
from ... import _lib
from ...engine import HipNet, default_compute


def _fill_hidden(dst, values, what):
    values = [int(v) for v in values]
    if len(values) > _lib.V4L_MAX_HIDDEN:
        raise NotImplementedError("vision4leg_amd: at most %d %s layers are supported on the HIP engine (got %d)"
                                  % (_lib.V4L_MAX_HIDDEN, what, len(values)))
    for i, v in enumerate(values):
        dst[i] = v
    return len(values)


def _check_relu(module, what):
    if module.activation_func is not nn.ReLU or getattr(module, "add_ln", False):
        raise NotImplementedError("vision4leg_amd: %s: only ReLU without LayerNorm runs on the HIP engine" % what)


class _HipNetMixin:
    """forward(x) on the HIP engine for x = [..., S + C*H*W] float32 GPU rows (reference nets.py:996-1000)."""

    _hip = None

    def _net_cfg(self):  # pragma: no cover - overridden
        raise NotImplementedError

    @property
    def hip(self):
        if self.__dict__.get("_hip") is None:
            cfg = self._net_cfg()
            cfg.tanh_action = int(bool(getattr(self, "tanh_action", False)))  # Gaussian policies (continuous_policy.py)
            self.__dict__["_hip"] = HipNet(self, cfg)
        return self.__dict__["_hip"]

    def mark_params_changed(self):
        """Tell the engine the parameters were rewritten through `.data` (invisible to torch's version counters, e.g.
        `soft_update_from_to`, `dist.broadcast(p.data)`): the operand-type weight copies are rebuilt before the next
        launch. Plain module calls repack anyway; a RolloutActor between two steps needs this call."""
        if self.__dict__.get("_hip") is not None:
            self.__dict__["_hip"].mark_dirty()

    def __deepcopy__(self, memo):
        # copy.deepcopy(pf) -> target_pf (reference ppo.py:21): clone parameters, never the engine handle
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_hip" else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_hip"] = None
        return d

    def _head_out(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        net = self.hip
        if net.out_dim == 1:
            return net.value(x2).view(*lead, 1)
        st, im, n = net.stage(x2)
        out = net.forward(st, im, n)
        # un-pad the head output (a strided device copy; policies use the fused gauss_head kernel instead)
        return out[:, :net.out_dim].clone().view(*lead, net.out_dim)

    def forward(self, x):
        return self._head_out(x)


def _make_head(in_dim, append_hidden_shapes, output_shape, activation_func, add_ln, hidden_init, last_init):
    mods = []
    for h in append_hidden_shapes:
        fc = nn.Linear(in_dim, h)
        hidden_init(fc)
        mods.append(fc)
        mods.append(activation_func())
        if add_ln:
            mods.append(nn.LayerNorm(h))
        in_dim = h
    last = nn.Linear(in_dim, output_shape)
    last_init(last)
    mods.append(last)
    return mods


class Net(_HipNetMixin, nn.Module):
    """MLP trunk + head: `base.seq_fcs.*`, `seq_append_fcs.*` (reference nets.py:16-55; ppo_state.py)."""

    def __init__(self, output_shape, base_type, append_hidden_shapes=[], append_hidden_init_func=init.basic_init,
                 net_last_init_func=init.uniform_init, activation_func=nn.ReLU, add_ln=False, **kwargs):
        super().__init__()
        self.base = base_type(activation_func=activation_func, add_ln=add_ln, **kwargs)
        self.add_ln = add_ln
        self.activation_func = activation_func
        self.output_dim = int(output_shape)
        self.append_hidden_shapes = [int(h) for h in append_hidden_shapes]
        self.append_fcs = _make_head(self.base.output_shape, self.append_hidden_shapes, output_shape, activation_func,
                                     add_ln, append_hidden_init_func, net_last_init_func)
        self.seq_append_fcs = nn.Sequential(*self.append_fcs)

    def _net_cfg(self):
        _check_relu(self, "Net")
        if not self.base.hip_supported():
            raise NotImplementedError("vision4leg_amd: Net.base must be an MLPBase with ReLU activations")
        c = _lib.NetCfg()
        c.kind = _lib.V4L_NET_MLP
        c.compute = default_compute()
        c.state_dim = self.base.input_dim
        c.out_dim = self.output_dim
        c.n_enc_hidden = _fill_hidden(c.enc_hidden, self.base.hidden_shapes, "base hidden")
        c.n_head_hidden = _fill_hidden(c.head_hidden, self.append_hidden_shapes, "append hidden")
        c.has_logstd = int(hasattr(self, "logstd"))
        return c


class ImpalaEncoderProjNet(_HipNetMixin, nn.Module):
    """NatureFuseEncoder + head over [visual_out | state_out]: `encoder.*`, `seq_append_fcs.*`
    (reference nets.py:194-262; ppo_nature_cnn.py)."""

    def __init__(self, encoder, output_shape, state_input_shape, visual_input_shape, append_hidden_shapes=[],
                 append_hidden_init_func=init.basic_init, net_last_init_func=init.uniform_init, activation_func=nn.ReLU,
                 add_ln=False, detach=False, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.add_ln = add_ln
        self.detach = detach
        self.state_input_shape = state_input_shape
        self.visual_input_shape = visual_input_shape
        self.activation_func = activation_func
        self.output_dim = int(output_shape)
        self.append_hidden_shapes = [int(h) for h in append_hidden_shapes]
        in_dim = self.encoder.base.output_shape + self.encoder.visual_dim
        self.append_fcs = _make_head(in_dim, self.append_hidden_shapes, output_shape, activation_func, add_ln,
                                     append_hidden_init_func, net_last_init_func)
        self.seq_append_fcs = nn.Sequential(*self.append_fcs)
        self.normalizer = None

    def _net_cfg(self):
        _check_relu(self, "ImpalaEncoderProjNet")
        enc = self.encoder
        if self.detach or not enc.base.hip_supported():
            raise NotImplementedError("vision4leg_amd: ImpalaEncoderProjNet needs detach=False and a ReLU MLPBase")
        c = _lib.NetCfg()
        c.kind = _lib.V4L_NET_CNN
        c.compute = default_compute()
        c.state_dim = int(self.state_input_shape)
        c.out_dim = self.output_dim
        c.in_channels = int(self.visual_input_shape[0])
        c.img_hw = int(self.visual_input_shape[1])
        c.visual_dim = int(enc.visual_dim)
        c.n_enc_hidden = _fill_hidden(c.enc_hidden, enc.base.hidden_shapes, "encoder hidden")
        c.n_head_hidden = _fill_hidden(c.head_hidden, self.append_hidden_shapes, "append hidden")
        c.has_logstd = int(hasattr(self, "logstd"))
        return c


class LocoTransformer(_HipNetMixin, nn.Module):
    """Cross-modal transformer over [proprio token | 16 depth tokens], pooled, then the head:
    `encoder.*`, `visual_append_layers.*`, `visual_seq_append_fcs.*` (reference nets.py:909-1038)."""

    def __init__(self, encoder, output_shape, state_input_shape, visual_input_shape, transformer_params=[],
                 append_hidden_shapes=[], append_hidden_init_func=init.basic_init,
                 net_last_init_func=init.uniform_init, activation_func=nn.ReLU, add_ln=False, detach=False,
                 state_detach=False, max_pool=False, token_norm=False, use_pytorch_encoder=False, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.add_ln = add_ln
        self.detach = detach
        self.state_detach = state_detach
        self.state_input_shape = state_input_shape
        self.visual_input_shape = visual_input_shape
        self.activation_func = activation_func
        self.max_pool = max_pool
        self.token_norm = token_norm
        self.use_pytorch_encoder = use_pytorch_encoder
        self.output_dim = int(output_shape)
        self.transformer_params = [(int(h), int(f)) for h, f in transformer_params]
        self.append_hidden_shapes = [int(h) for h in append_hidden_shapes]
        d = self.encoder.visual_dim
        if token_norm:
            self.token_ln = nn.LayerNorm(d)
            self.state_token_ln = nn.LayerNorm(d)
        if use_pytorch_encoder:
            layer = nn.TransformerEncoderLayer(d, self.transformer_params[0][0], self.transformer_params[0][1], dropout=0)
            self.visual_trans_encoder = nn.TransformerEncoder(layer, len(self.transformer_params), nn.LayerNorm(d),
                                                              enable_nested_tensor=False)
        else:
            self.visual_append_layers = nn.ModuleList(
                [nn.TransformerEncoderLayer(d, n_head, ff, dropout=0) for n_head, ff in self.transformer_params])
        self.per_modal_tokens = self.encoder.per_modal_tokens
        self.second = self.encoder.in_channels not in (4, 12)
        in_dim = 2 * d + (d if self.second else 0)
        self.visual_append_fcs = _make_head(in_dim, self.append_hidden_shapes, output_shape, activation_func, add_ln,
                                            append_hidden_init_func, net_last_init_func)
        self.visual_seq_append_fcs = nn.Sequential(*self.visual_append_fcs)
        self.normalizer = None

    def _net_cfg(self):
        _check_relu(self, "LocoTransformer")
        enc = self.encoder
        bad = []
        if enc.in_channels != 4: bad.append("in_channels=%d (depth-only 4 supported)" % enc.in_channels)
        if enc.two_by_two: bad.append("two_by_two")
        if self.detach or self.state_detach: bad.append("detach")
        if any(h != 1 for h, _ in self.transformer_params): bad.append("n_head != 1")
        if len({f for _, f in self.transformer_params}) != 1: bad.append("per-layer dim_feedforward differs")
        if not enc.base.hip_supported(): bad.append("encoder.base is not a ReLU MLPBase")
        if bad:
            raise NotImplementedError("vision4leg_amd: LocoTransformer option(s) not on the HIP engine: " + ", ".join(bad))
        c = _lib.NetCfg()
        c.kind = _lib.V4L_NET_LOCO
        c.compute = default_compute()
        c.state_dim = int(self.state_input_shape)
        c.out_dim = self.output_dim
        c.in_channels = int(self.visual_input_shape[0])
        c.img_hw = int(self.visual_input_shape[1])
        c.token_dim = int(enc.token_dim)
        c.n_layers = len(self.transformer_params)
        c.ff_dim = self.transformer_params[0][1]
        c.n_enc_hidden = _fill_hidden(c.enc_hidden, enc.base.hidden_shapes, "encoder hidden")
        c.n_head_hidden = _fill_hidden(c.head_hidden, self.append_hidden_shapes, "append hidden")
        c.max_pool = int(bool(self.max_pool))  # max instead of mean pooling: layer-by-layer kernels (nets.py:1022-1030, 884-889)
        c.token_norm = int(bool(self.token_norm))  # token_ln over every token in front of the layers (nets.py:879-880, 1007-1008)
        c.pytorch_encoder = int(bool(self.use_pytorch_encoder))  # nn.TransformerEncoder + final norm (nets.py:955-963)
        c.has_logstd = int(hasattr(self, "logstd"))
        return c


class NatureEncoderProjNet(_HipNetMixin, nn.Module):
    """Vision-only NatureCNN net: NatureEncoder(flatten) -> 1024 -> head; the observation row is the depth stack alone:
    `encoder.layers.*`, `seq_append_fcs.*` (reference nets.py:133-191; ppo_nature_cnn_vision_only.py)."""

    def __init__(self, encoder, output_shape, visual_input_shape, append_hidden_shapes=[],
                 append_hidden_init_func=init.basic_init, net_last_init_func=init.uniform_init, activation_func=nn.ReLU,
                 add_ln=False, detach=False, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.add_ln = add_ln
        self.detach = detach
        self.visual_input_shape = visual_input_shape
        self.activation_func = activation_func
        self.output_dim = int(output_shape)
        self.append_hidden_shapes = [int(h) for h in append_hidden_shapes]
        self.append_fcs = _make_head(self.encoder.output_dim, self.append_hidden_shapes, output_shape, activation_func,
                                     add_ln, append_hidden_init_func, net_last_init_func)
        self.seq_append_fcs = nn.Sequential(*self.append_fcs)
        self.normalizer = None

    def _net_cfg(self):
        _check_relu(self, "NatureEncoderProjNet")
        enc = self.encoder
        if self.detach or getattr(enc, "groups", 1) != 1 or enc.output_dim != 1024 or len(enc.layers) != 7:
            raise NotImplementedError("vision4leg_amd: NatureEncoderProjNet needs detach=False and a flattening "
                                      "NatureEncoder with groups=1")
        c = _lib.NetCfg()
        c.kind = _lib.V4L_NET_CNN_VIS
        c.compute = default_compute()
        c.state_dim = 0
        c.out_dim = self.output_dim
        c.in_channels = int(self.visual_input_shape[0])
        c.img_hw = int(self.visual_input_shape[1])
        c.n_enc_hidden = 0
        c.n_head_hidden = _fill_hidden(c.head_hidden, self.append_hidden_shapes, "append hidden")
        c.has_logstd = int(hasattr(self, "logstd"))
        return c


class Transformer(_HipNetMixin, nn.Module):
    """Vision-only transformer: the 16 depth tokens through the encoder layers, mean over all tokens, then the head:
    `encoder.*`, `visual_append_layers.*`, `visual_seq_append_fcs.*` (reference nets.py:784-906;
    ppo_locotransformer_vision_only.py)."""

    def __init__(self, encoder, output_shape, visual_input_shape, transformer_params=[], append_hidden_shapes=[],
                 append_hidden_init_func=init.basic_init, net_last_init_func=init.uniform_init, activation_func=nn.ReLU,
                 add_ln=False, detach=False, state_detach=False, max_pool=False, token_norm=False,
                 use_pytorch_encoder=False, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.add_ln = add_ln
        self.detach = detach
        self.state_detach = state_detach
        self.visual_input_shape = visual_input_shape
        self.activation_func = activation_func
        self.max_pool = max_pool
        self.token_norm = token_norm
        self.use_pytorch_encoder = use_pytorch_encoder
        self.output_dim = int(output_shape)
        self.transformer_params = [(int(h), int(f)) for h, f in transformer_params]
        self.append_hidden_shapes = [int(h) for h in append_hidden_shapes]
        d = self.encoder.visual_dim
        if token_norm:
            self.token_ln = nn.LayerNorm(d)
            self.state_token_ln = nn.LayerNorm(d)
        if use_pytorch_encoder:
            layer = nn.TransformerEncoderLayer(d, self.transformer_params[0][0], self.transformer_params[0][1], dropout=0)
            self.visual_trans_encoder = nn.TransformerEncoder(layer, len(self.transformer_params), nn.LayerNorm(d),
                                                              enable_nested_tensor=False)
        else:
            self.visual_append_layers = nn.ModuleList(
                [nn.TransformerEncoderLayer(d, n_head, ff, dropout=0) for n_head, ff in self.transformer_params])
        self.per_modal_tokens = self.encoder.per_modal_tokens
        self.second = self.encoder.in_channels not in (4, 12)
        in_dim = d + (d if self.second else 0)
        self.visual_append_fcs = _make_head(in_dim, self.append_hidden_shapes, output_shape, activation_func, add_ln,
                                            append_hidden_init_func, net_last_init_func)
        self.visual_seq_append_fcs = nn.Sequential(*self.visual_append_fcs)
        self.normalizer = None

    def _net_cfg(self):
        _check_relu(self, "Transformer")
        enc = self.encoder
        bad = []
        if enc.in_channels != 4: bad.append("in_channels=%d (depth-only 4 supported)" % enc.in_channels)
        if enc.two_by_two: bad.append("two_by_two")
        if self.detach or self.state_detach: bad.append("detach")
        if any(h != 1 for h, _ in self.transformer_params): bad.append("n_head != 1")
        if len({f for _, f in self.transformer_params}) != 1: bad.append("per-layer dim_feedforward differs")
        if bad:
            raise NotImplementedError("vision4leg_amd: Transformer option(s) not on the HIP engine: " + ", ".join(bad))
        c = _lib.NetCfg()
        c.kind = _lib.V4L_NET_LOCO_VIS
        c.compute = default_compute()
        c.state_dim = 0
        c.out_dim = self.output_dim
        c.in_channels = int(self.visual_input_shape[0])
        c.img_hw = int(self.visual_input_shape[1])
        c.token_dim = int(enc.token_dim)
        c.n_layers = len(self.transformer_params)
        c.ff_dim = self.transformer_params[0][1]
        c.n_enc_hidden = 0
        c.n_head_hidden = _fill_hidden(c.head_hidden, self.append_hidden_shapes, "append hidden")
        c.max_pool = int(bool(self.max_pool))  # max instead of mean pooling: layer-by-layer kernels (nets.py:1022-1030, 884-889)
        c.token_norm = int(bool(self.token_norm))  # token_ln over every token in front of the layers (nets.py:879-880, 1007-1008)
        c.pytorch_encoder = int(bool(self.use_pytorch_encoder))  # nn.TransformerEncoder + final norm (nets.py:955-963)
        c.has_logstd = int(hasattr(self, "logstd"))
        return c

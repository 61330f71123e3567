"""Encoder building blocks with the reference's class names, constructor signatures, state_dict keys and
seeded initial values (torchrl/networks/base.py). They are *parameter containers*: torch.nn modules are used
for what they are good at here — owning device memory, naming parameters, (de)serialising checkpoints — while
the arithmetic of a forward/backward pass is enqueued on the MI355X by the top-level nets in nets.py through
libv4l_hip.so. Calling a building block on its own is not part of the hot path and raises.
"""
import numpy as np
import torch
import torch.nn as nn

from . import init

_STANDALONE_MSG = ("vision4leg_amd: %s is a parameter container; it runs on the HIP engine only as part of a "
                   "top-level net (networks.Net / ImpalaEncoderProjNet / LocoTransformer or their policies)")


class _Container(nn.Module):
    def forward(self, *args, **kwargs):
        raise NotImplementedError(_STANDALONE_MSG % type(self).__name__)


def weight_init(m):
    """Linear: orthogonal, zero bias. Conv: delta-orthogonal (reference base.py:192-206)."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        if hasattr(m.bias, "data"):
            m.bias.data.fill_(0.0)
    elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        assert m.weight.size(2) == m.weight.size(3)
        m.weight.data.fill_(0.0)
        if hasattr(m.bias, "data"):
            m.bias.data.fill_(0.0)
        mid = m.weight.size(2) // 2
        nn.init.orthogonal_(m.weight.data[:, :, mid, mid], nn.init.calculate_gain("relu"))


def orthogonal_init(module, gain=nn.init.calculate_gain("relu")):
    """reference base.py:297-301"""
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        nn.init.orthogonal_(module.weight.data, gain)
        nn.init.constant_(module.bias.data, 0)
    return module


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class MLPBase(_Container):
    """Linear+ReLU stack; `seq_fcs.{0,2,...}` (reference base.py:8-44). Only ReLU / no LayerNorm runs on HIP."""

    def __init__(self, input_shape, hidden_shapes, activation_func=nn.ReLU, init_func=init.basic_init,
                 add_ln=False, last_activation_func=None):
        super().__init__()
        self.activation_func = activation_func
        self.add_ln = add_ln
        self.last_activation_func = activation_func if last_activation_func is None else last_activation_func
        width = int(np.prod(input_shape))
        self.input_dim = width
        self.hidden_shapes = [int(h) for h in hidden_shapes]
        self.output_shape = width
        mods = []
        for h in self.hidden_shapes:
            fc = nn.Linear(width, h)
            init_func(fc)
            mods.append(fc)
            mods.append(activation_func())
            if add_ln:
                mods.append(nn.LayerNorm(h))
            width = h
            self.output_shape = h
        mods.pop(-1)
        mods.append(self.last_activation_func())
        self.fcs = mods
        self.seq_fcs = nn.Sequential(*mods)

    def hip_supported(self):
        return (self.activation_func is nn.ReLU and self.last_activation_func is nn.ReLU and not self.add_ln
                and len(self.hidden_shapes) >= 1)


class NatureEncoder(_Container):
    """Conv 8x8/4 -> 4x4/2 -> 3x3/1, ReLU each; `layers.{0,2,4}` (reference base.py:304-342)."""

    def __init__(self, in_channels, groups=1, flatten=True, **kwargs):
        super().__init__()
        self.groups = groups
        self.in_channels = in_channels
        mods = [
            nn.Conv2d(in_channels, 32 * groups, kernel_size=8, stride=4), nn.ReLU(),
            nn.Conv2d(32 * groups, 64 * groups, kernel_size=4, stride=2), nn.ReLU(),
            nn.Conv2d(64 * groups, 64 * groups, kernel_size=3, stride=1), nn.ReLU(),
        ]
        if flatten:
            mods.append(Flatten())
        self.layers = nn.Sequential(*mods)
        self.output_dim = 1024 * groups
        self.apply(orthogonal_init)


class RLProjection(_Container):
    """Linear (+ReLU); `projection.0` (reference base.py:209-230)."""

    def __init__(self, in_dim, out_dim, proj=True):
        super().__init__()
        self.out_dim = out_dim
        self.proj = proj
        mods = [nn.Linear(in_dim, out_dim)]
        if proj:
            mods.append(nn.ReLU())
        self.projection = nn.Sequential(*mods)
        self.output_dim = out_dim
        self.apply(weight_init)


class NatureFuseEncoder(_Container):
    """NatureCNN -> 1024 -> visual_dim projection, next to the proprio MLP (reference base.py:345-385)."""

    def __init__(self, in_channels, state_input_dim, visual_dim, hidden_shapes, proj=True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.state_input_dim = state_input_dim
        self.visual_base = NatureEncoder(in_channels)
        self.visual_dim = visual_dim
        self.visual_projector = RLProjection(in_dim=self.visual_base.output_dim, out_dim=visual_dim)
        self.base = MLPBase(input_shape=state_input_dim, hidden_shapes=hidden_shapes, **kwargs)


class LocoTransformerEncoder(_Container):
    """Depth NatureCNN (un-flattened) + 1x1 up-conv -> 16 depth tokens, proprio MLP + projector -> 1 token
    (reference base.py:497-626). Depth-only (in_channels == 4) is what the shipped configs use and what the
    HIP engine implements; the RGB branches are constructed for checkpoint compatibility only."""

    def __init__(self, in_channels, state_input_dim, hidden_shapes, token_dim=64, two_by_two=False, visual_dim=None,
                 proj=True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.state_input_dim = state_input_dim
        self.token_dim = token_dim
        self.two_by_two = two_by_two
        if in_channels in (12, 16):
            self.rgb_visual_base = NatureEncoder(12, flatten=False)
            self.rgb_up_conv = nn.Conv2d(64, token_dim, 2, stride=2) if two_by_two else nn.Conv2d(64, token_dim, 1)
        if in_channels in (4, 16):
            self.depth_visual_base = NatureEncoder(4, flatten=False)
            self.depth_up_conv = nn.Conv2d(64, token_dim, 2, stride=2) if two_by_two else nn.Conv2d(64, token_dim, 1)
        self.base = MLPBase(input_shape=state_input_dim, hidden_shapes=hidden_shapes, **kwargs)
        self.state_projector = RLProjection(in_dim=self.base.output_shape, out_dim=token_dim)
        self.visual_dim = token_dim
        self.per_modal_tokens = 4 if two_by_two else 16
        self.flatten_layer = Flatten()


class TransformerEncoder(_Container):
    """Vision-only token encoder: depth NatureCNN (un-flattened) + 1x1 up-conv -> 16 tokens, no proprio branch
    (reference base.py:388-494; starter/ppo_locotransformer_vision_only.py:77-80). Depth-only (in_channels == 4)
    runs on the HIP engine; the RGB branches are constructed for checkpoint compatibility only."""

    def __init__(self, in_channels, token_dim=64, two_by_two=False, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.token_dim = token_dim
        self.two_by_two = two_by_two
        if in_channels in (12, 16):
            self.rgb_visual_base = NatureEncoder(12, flatten=False)
            self.rgb_up_conv = nn.Conv2d(64, token_dim, 2, stride=2) if two_by_two else nn.Conv2d(64, token_dim, 1)
        if in_channels in (4, 16):
            self.depth_visual_base = NatureEncoder(4, flatten=False)
            self.depth_up_conv = nn.Conv2d(64, token_dim, 2, stride=2) if two_by_two else nn.Conv2d(64, token_dim, 1)
        self.visual_dim = token_dim
        self.per_modal_tokens = 4 if two_by_two else 16
        self.flatten_layer = Flatten()

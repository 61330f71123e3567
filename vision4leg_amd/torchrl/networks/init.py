"""Parameter initialisers with the reference's names and RNG consumption (torchrl/networks/init.py:5-47),
so that a seeded construction yields bit-identical initial parameters."""
import math

import torch.nn as nn


def _fanin_init(tensor, alpha=0):
    # the reference takes size[0] (= out_features of an nn.Linear weight) as "fan in" (init.py:7-8)
    dims = tuple(tensor.size())
    if len(dims) < 2:
        raise Exception("Shape must be have dimension at least 2.")
    fan_in = dims[0] if len(dims) == 2 else math.prod(dims[1:])
    bound = math.sqrt(1.0 / ((1 + alpha * alpha) * fan_in))
    return tensor.data.uniform_(-bound, bound)


def _uniform_init(tensor, param=3e-3):
    return tensor.data.uniform_(-param, param)


def _constant_bias_init(tensor, constant=0.1):
    tensor.data.fill_(constant)


def layer_init(layer, weight_init=_fanin_init, bias_init=_constant_bias_init):
    weight_init(layer.weight)
    bias_init(layer.bias)


def basic_init(layer):
    layer_init(layer, _fanin_init, _constant_bias_init)


def uniform_init(layer):
    layer_init(layer, _uniform_init, _uniform_init)


def _orthogonal_init(tensor, gain=math.sqrt(2)):
    nn.init.orthogonal_(tensor, gain=gain)


def orthogonal_init(layer, scale=math.sqrt(2), constant=0):
    layer_init(layer, lambda w: _orthogonal_init(w, gain=scale), lambda b: _constant_bias_init(b, 0))

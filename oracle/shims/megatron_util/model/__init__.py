"""Oracle shim (TEST INFRASTRUCTURE ONLY): megatron_util.model subset (see package docstring)."""
import enum

import torch
import torch.nn.functional as F
from torch import nn

from . import fused_softmax  # noqa: F401


class AttnMaskType(enum.Enum):
    padding = 1
    causal = 2


class LayerNorm(nn.Module):
    """MixedFusedLayerNorm: statistics and affine in fp32, output in the input dtype."""

    def __init__(self, normalized_shape, eps=1e-5, no_persist_layer_norm=True, sequence_parallel=False):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(*self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(*self.normalized_shape))

    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape, self.weight.float(), self.bias.float(),
                            self.eps).to(x.dtype)


def bias_gelu_impl(x, bias):
    """tanh-approximate GELU of (x + bias); the real one is a jit-fused kernel that evaluates
    in fp32 registers and rounds once."""
    y = (x + bias).float()
    return (y * 0.5 * (1.0 + torch.tanh(0.79788456 * y * (1.0 + 0.044715 * y * y)))).to(x.dtype)


class Float16Module(nn.Module):  # pragma: no cover - config.fp16/bf16 are False in the shipped JSONs
    def __init__(self, module, config):
        super().__init__()
        self.module = module.bfloat16() if getattr(config, "bf16", False) else module.half()

    def forward(self, *a, **k):
        return self.module(*a, **k)

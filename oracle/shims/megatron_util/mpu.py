"""Oracle shim (TEST INFRASTRUCTURE ONLY): megatron_util.mpu at TP=1 (see package docstring)."""
import contextlib
import types

import torch
import torch.nn.functional as F
from torch import nn


def divide(a, b):
    assert a % b == 0, (a, b)
    return a // b


utils = types.SimpleNamespace(divide=divide)


def get_tensor_model_parallel_world_size():
    return 1


def get_tensor_model_parallel_rank():
    return 0


def set_defaults_if_not_set_tensor_model_parallel_attributes(p):
    for k, v in (("tensor_model_parallel", False), ("partition_dim", -1), ("partition_stride", 1)):
        if not hasattr(p, k):
            setattr(p, k, v)


def make_viewless_tensor(inp, requires_grad, keep_graph):
    return inp


def gather_from_tensor_model_parallel_region(x):
    return x


def scatter_to_sequence_parallel_region(x):
    return x


def split_tensor_along_last_dim(t, n, contiguous_split_chunks=False):
    return torch.split(t, divide(t.size(-1), n), dim=-1)


class _Tracker:
    def fork(self, name=None):
        return contextlib.nullcontext()


def get_cuda_rng_tracker():
    return _Tracker()


class ColumnParallelLinear(nn.Module):
    """Y = XW^T (+b).  Returns (Y, bias-or-None); with skip_bias_add the bias is NOT added
    and is handed back for a fused consumer (Megatron-LM layers.py semantics)."""

    def __init__(self, input_size, output_size, bias=True, gather_output=True, init_method=None,
                 stride=1, keep_master_weight_for_test=False, skip_bias_add=False, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        if init_method is not None:
            init_method(self.weight)
        self.bias = nn.Parameter(torch.zeros(output_size)) if bias else None
        self.skip_bias_add = skip_bias_add

    def forward(self, x):
        b = None if self.skip_bias_add else self.bias
        return F.linear(x, self.weight, b), (self.bias if self.skip_bias_add else None)


class RowParallelLinear(nn.Module):
    def __init__(self, input_size, output_size, bias=True, input_is_parallel=False, init_method=None,
                 stride=1, keep_master_weight_for_test=False, skip_bias_add=False, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        if init_method is not None:
            init_method(self.weight)
        self.bias = nn.Parameter(torch.zeros(output_size)) if bias else None
        self.skip_bias_add = skip_bias_add

    def forward(self, x):
        y = F.linear(x, self.weight)
        if self.skip_bias_add:
            return y, self.bias
        return (y + self.bias if self.bias is not None else y), None


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, init_method=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        if init_method is not None:
            init_method(self.weight)

    def forward(self, ids):
        return F.embedding(ids, self.weight)


class LinearWithGradAccumulationAndAsyncCommunication:
    @staticmethod
    def apply(inp, weight, bias, gradient_accumulation_fusion, async_grad_allreduce, sequence_parallel):
        return F.linear(inp, weight, bias)


def vocab_parallel_cross_entropy(logits, target):
    """Per-token CE = log(sum exp(l - max)) - (l[target] - max); TP all-reduces are identities."""
    m = logits.max(dim=-1, keepdim=True)[0]
    z = logits - m
    tgt = z.gather(-1, target.unsqueeze(-1)).squeeze(-1)
    return torch.log(z.exp().sum(-1)) - tgt

"""Oracle shim (TEST INFRASTRUCTURE ONLY): timm.models.layers subset.
Semantics follow timm 0.6.7's public behaviour (restated, not copied)."""
import collections.abc
import torch


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return (x, x)


def drop_path(x, drop_prob: float = 0.0, training: bool = False, scale_by_keep: bool = True):
    # stochastic depth; identity at rate 0 / eval (configs/models/clip-b16.json:10 sets 0)
    if not drop_prob or not training:
        return x
    keep = 1.0 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0.0 and scale_by_keep:
        mask.div_(keep)
    return x * mask


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    # init only -- the oracle overwrites every weight from a seeded generator afterwards
    with torch.no_grad():
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

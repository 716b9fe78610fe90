"""Oracle shim (TEST INFRASTRUCTURE ONLY): FusedScaleMaskSoftmax, torch path only (the 1.3B/2.7B
JSONs set masked_softmax_fusion:false, configs/models/config_gpt3_1.3B.json:16)."""
import torch
from torch import nn


class FusedScaleMaskSoftmax(nn.Module):
    def __init__(self, input_in_fp16, input_in_bf16, attn_mask_type, scaled_masked_softmax_fusion,
                 mask_func, softmax_in_fp32, scale):
        super().__init__()
        self.input_in_float16 = input_in_fp16 or input_in_bf16
        self.input_in_bf16 = input_in_bf16
        self.mask_func = mask_func
        self.softmax_in_fp32 = softmax_in_fp32
        self.scale = scale
        assert self.scale is None or softmax_in_fp32

    def forward(self, x, mask):
        dt = x.dtype
        up = self.input_in_float16 and self.softmax_in_fp32
        if up:
            x = x.float()
        if self.scale is not None:
            x = x * self.scale
        if mask is not None:
            x = self.mask_func(x, mask)
        p = torch.softmax(x, dim=-1)
        if up:
            p = p.to(dt)
        return p

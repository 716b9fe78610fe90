"""Device-side video input transforms -- the per-pixel half of the reference's loader pipeline
(dataset/__init__.py:60-85: video_pretrain_transform / video_train_transform / video_test_transform), run on the GPU
on the decoder's uint8 clip [T, H, W, 3] (read_frames_decord, dataset/video_utils/utils.py:97-117):

    train:  RandomResizedCrop(res, scale=(0.5, 1.0), "bicubic") -> RandomHorizontalFlip -> ClipToTensor -> Normalize
    test:   Resize((res, res)) ("nearest", the class default)   ->                         ClipToTensor -> Normalize

One kernel (mpv_video_resized_crop_normalize) does crop + torch-semantics interpolate + .long() + flip + /255 +
normalise + bf16 and writes straight into the [B, 3, T, res, res] batch the model consumes.  The random draws are the
reference's (python `random`, same call order: crop box, then flip), so a seeded run picks the same boxes.
TemporalConsistentRandomAugment (dataset/video_utils/randaugment_video.py) is cv2-based; opencv is not in this image,
so it cannot be pinned against the reference and is not implemented (rand_augment=True raises).
"""
from __future__ import annotations

import ctypes as C
import math
import random
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from .ops import _stream, check

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}


def resized_crop_normalize(clip: torch.Tensor, box: Tuple[int, int, int, int], size: Tuple[int, int], mode: str, flip: bool,
                           mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD, out: Optional[torch.Tensor] = None):
    """clip uint8 [T,H,W,3] on the GPU -> bf16 [3,T,oh,ow] (`out` may be a slot of a [B,3,T,oh,ow] batch)."""
    if not clip.is_cuda:
        raise _lib.MpvError("mpv ops run on the GPU only (no CPU fallback): got a CPU tensor")
    assert clip.dtype == torch.uint8 and clip.dim() == 4 and clip.shape[-1] == 3 and clip.is_contiguous()
    T, H, W, _ = clip.shape
    oh, ow = size
    if out is None:
        out = torch.empty((3, T, oh, ow), dtype=torch.bfloat16, device=clip.device)
    assert out.shape == (3, T, oh, ow) and out.stride(3) == 1 and out.stride(2) == ow
    i, j, h, w = box
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(_lib.lib().mpv_video_resized_crop_normalize(clip.data_ptr(), T, H, W, i, j, h, w, oh, ow, _MODES[mode], int(flip), m3, s3,
                                                      out.data_ptr(), out.stride(0), out.stride(1), _stream()),
          "mpv_video_resized_crop_normalize")
    return out


class VideoInputTransform:
    def __init__(self, image_res: int, train: bool = True, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation: Optional[str] = None,
                 mean=CLIP_MEAN, std=CLIP_STD, rand_augment: bool = False):
        if rand_augment:
            raise NotImplementedError("TemporalConsistentRandomAugment is cv2-based (not installed here): unpinned, not implemented")
        self.size, self.train, self.scale, self.ratio = (image_res, image_res), train, scale, ratio
        self.interpolation = interpolation or ("bicubic" if train else "nearest")
        self.mean, self.std = tuple(mean), tuple(std)

    def get_params(self, height: int, width: int):
        """video_transforms.RandomResizedCrop.get_params (:1078-1117), same draws in the same order."""
        area = height * width
        for _ in range(10):
            target_area = random.uniform(*self.scale) * area
            log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
            aspect_ratio = math.exp(random.uniform(*log_ratio))
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                return random.randint(0, height - h), random.randint(0, width - w), h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(self.ratio):
            w = width
            h = int(round(w / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            h = height
            w = int(round(h * max(self.ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, clip: torch.Tensor, out: Optional[torch.Tensor] = None):
        T, H, W, _ = clip.shape
        if self.train:
            box = self.get_params(H, W)
            flip = random.random() < 0.5                                      # RandomHorizontalFlip (:932)
        else:
            box, flip = (0, 0, H, W), False
        return resized_crop_normalize(clip, box, self.size, self.interpolation, flip, self.mean, self.std, out=out)

    def batch(self, clips: Sequence[torch.Tensor]) -> torch.Tensor:
        """[B, 3, T, res, res] bf16, each clip transformed straight into its slot."""
        T = clips[0].shape[0]
        out = torch.empty((len(clips), 3, T, *self.size), dtype=torch.bfloat16, device=clips[0].device)
        for b, c in enumerate(clips):
            self(c, out=out[b])
        return out

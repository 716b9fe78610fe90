"""Device-side video input transforms -- the per-pixel half of the reference's loader pipeline
(dataset/__init__.py:60-85: video_pretrain_transform / video_train_transform / video_test_transform), run on the GPU
on the decoder's uint8 clip [T, H, W, 3] (read_frames_decord, dataset/video_utils/utils.py:97-117):

    train:  RandomResizedCrop(res, scale=(0.5, 1.0), "bicubic") -> RandomHorizontalFlip -> ClipToTensor -> Normalize
    test:   Resize((res, res)) ("nearest", the class default)   ->                         ClipToTensor -> Normalize

One kernel (mpv_video_resized_crop_normalize) does crop + torch-semantics interpolate + .long() + flip + /255 +
normalise + bf16 and writes straight into the [B, 3, T, res, res] batch the model consumes.  The random draws are the
reference's (python `random`, same call order: crop box, then flip), so a seeded run picks the same boxes.
With rand_augment=True the clip-consistent RandAugment of the training recipes sits between the flip and the normalisation, as in
dataset/__init__.py:60-78:  ... -> RandomHorizontalFlip -> TemporalConsistentRandomAugment(N = 2, M = 5, nine ops) -> ClipToTensor ->
Normalize.  Its draws are the reference's (numpy's global RNG: np.random.choice without replacement once per CLIP, then the apply
mask), its ops run as HIP kernels on the uint8 clip (csrc/augment.hip), bit-exact against oracle/augment.py.  The reference ops
are cv2-based and opencv is not in this image: what is pinned against the reference module and what is restated from opencv's
published algorithm is spelled out in oracle/augment.py.
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


# ---- TemporalConsistentRandomAugment (dataset/video_utils/randaugment_video.py:213-361) ------------------------------------------
AUG_MAX_LEVEL, AUG_TRANSLATE_CONST, AUG_REPLACE_VALUE = 10, 10, (128, 128, 128)      # :300-302
PRETRAIN_AUGS = ("Identity", "Contrast", "Brightness", "Sharpness", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate")   # dataset/__init__.py:65-66


def _invert_affine(m):
    """the inversion cv::warpAffine applies to a forward 2 x 3 matrix, in double (host; 6 numbers)"""
    m00, m01, m02, m10, m11, m12 = (float(v) for v in m)
    d = m00 * m11 - m01 * m10
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m11 * d, m00 * d
    m00, m01, m10, m11 = a11, m01 * -d, m10 * -d, a22
    b1 = -m00 * m02 - m01 * m12
    b2 = -m10 * m02 - m11 * m12
    return (m00, m01, b1, m10, m11, b2)


def _f32(x):
    return C.c_float(x).value      # np.float32([...]) in the reference: the matrix entries are rounded to float32 first


class TemporalConsistentRandomAugment:
    """The reference class on device clips: uint8 [T, H, W, 3] in, uint8 [T, H, W, 3] out (the reference returns .float() of the same
    values).  One set of ops per clip, every frame gets the same ones (`num_frames * [self.get_random_ops()]`, :346)."""

    def __init__(self, N=2, M=10, p=0.0, augs=()):
        self.N, self.M, self.p = N, M, p
        self.augs = list(augs) if augs else list(PRETRAIN_AUGS)
        unknown = [a for a in self.augs if a not in PRETRAIN_AUGS]
        if unknown:
            raise NotImplementedError(f"RandAugment ops {unknown}: only the nine ops of the shipped recipes are built ({', '.join(PRETRAIN_AUGS)})")

    def get_random_ops(self):
        import numpy as np
        return [(str(op), self.M) for op in np.random.choice(self.augs, self.N, replace=False)]      # :334-337

    def __call__(self, frames: torch.Tensor) -> torch.Tensor:
        import numpy as np
        if not frames.is_cuda:
            raise _lib.MpvError("mpv ops run on the GPU only (no CPU fallback): got a CPU tensor")
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3 and frames.is_contiguous()
        ops = self.get_random_ops()
        apply_or_not = np.random.random(size=self.N) > self.p                                        # :347
        for i, (name, level) in enumerate(ops):
            if apply_or_not[i]:
                frames = self.apply(frames, name, level)
        return frames

    def apply(self, frames: torch.Tensor, name: str, level) -> torch.Tensor:
        T, H, W, _ = frames.shape
        lib, st = _lib.lib(), _stream()
        if name == "Identity":
            return frames
        if name in ("Contrast", "Brightness", "Sharpness"):
            factor = (level / AUG_MAX_LEVEL) * 1.8 + 0.1                                             # enhance_level_to_args :219-222
            if name == "Sharpness":
                if factor == 1.0:
                    return frames                                                                    # :156-157
                out = torch.empty_like(frames)
                check(lib.mpv_video_aug_sharpness(frames.data_ptr(), out.data_ptr(), T, H, W, factor, st), "mpv_video_aug_sharpness")
                return out
            ws = torch.empty(3 * T, dtype=torch.int64, device=frames.device) if name == "Contrast" else None
            out = frames.clone()
            check(lib.mpv_video_aug_pointwise(out.data_ptr(), T, H, W, 0 if name == "Contrast" else 1, factor, ws.data_ptr() if ws is not None else None, st),
                  "mpv_video_aug_pointwise")
            return out
        if name in ("ShearX", "ShearY"):
            f = _f32((level / AUG_MAX_LEVEL) * 0.3)                                                  # :225-231
            m = (1.0, f, 0.0, 0.0, 1.0, 0.0) if name == "ShearX" else (1.0, 0.0, 0.0, f, 1.0, 0.0)   # :165, :200
        elif name in ("TranslateX", "TranslateY"):
            o = _f32(-((level / AUG_MAX_LEVEL) * float(AUG_TRANSLATE_CONST)))                        # :234-240
            m = (1.0, 0.0, o, 0.0, 1.0, 0.0) if name == "TranslateX" else (1.0, 0.0, 0.0, 0.0, 1.0, o)   # :175, :185
        elif name == "Rotate":
            a = ((level / AUG_MAX_LEVEL) * 30) * math.pi / 180.0                                     # :269-275; cv2.getRotationMatrix2D
            alpha, beta, cx, cy = math.cos(a), math.sin(a), W / 2, H / 2
            m = (alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy)
        else:
            raise KeyError(name)
        out = torch.empty_like(frames)
        check(lib.mpv_video_aug_warp_affine(frames.data_ptr(), out.data_ptr(), T, H, W, (C.c_double * 6)(*_invert_affine(m)),
                                            (C.c_uint8 * 3)(*AUG_REPLACE_VALUE), st), "mpv_video_aug_warp_affine")
        return out


def resized_crop_u8(clip: torch.Tensor, box, size, mode: str, flip: bool) -> torch.Tensor:
    """clip uint8 [T,H,W,3] -> uint8 [T,oh,ow,3]: crop + interpolate + .long() + flip, as the uint8 view the augmentation takes"""
    if not clip.is_cuda:
        raise _lib.MpvError("mpv ops run on the GPU only (no CPU fallback): got a CPU tensor")
    assert clip.dtype == torch.uint8 and clip.dim() == 4 and clip.shape[-1] == 3 and clip.is_contiguous()
    T, H, W, _ = clip.shape
    oh, ow = size
    out = torch.empty((T, oh, ow, 3), dtype=torch.uint8, device=clip.device)
    i, j, h, w = box
    check(_lib.lib().mpv_video_resized_crop_u8(clip.data_ptr(), T, H, W, i, j, h, w, oh, ow, _MODES[mode], int(flip), out.data_ptr(), _stream()),
          "mpv_video_resized_crop_u8")
    return out


def u8_normalize(frames: torch.Tensor, mean=CLIP_MEAN, std=CLIP_STD, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [T,H,W,3] -> bf16 [3,T,H,W]: ClipToTensor (/255) + Normalize"""
    T, H, W, _ = frames.shape
    if out is None:
        out = torch.empty((3, T, H, W), dtype=torch.bfloat16, device=frames.device)
    assert out.shape == (3, T, H, W) and out.stride(3) == 1 and out.stride(2) == W
    check(_lib.lib().mpv_video_u8_normalize(frames.data_ptr(), T, H, W, (C.c_float * 3)(*mean), (C.c_float * 3)(*std), out.data_ptr(), out.stride(0),
                                            out.stride(1), _stream()), "mpv_video_u8_normalize")
    return out


class VideoInputTransform:
    def __init__(self, image_res: int, train: bool = True, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation: Optional[str] = None,
                 mean=CLIP_MEAN, std=CLIP_STD, rand_augment: bool = False):
        # dataset/__init__.py:65-66, 75-76: TemporalConsistentRandomAugment(N = 2, M = 5, augs = the nine ops) in the two training recipes
        self.augment = TemporalConsistentRandomAugment(N=2, M=5, augs=PRETRAIN_AUGS) if (rand_augment and train) else None
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
        if self.augment is not None:
            frames = resized_crop_u8(clip, box, self.size, self.interpolation, flip)
            return u8_normalize(self.augment(frames), self.mean, self.std, out=out)
        return resized_crop_normalize(clip, box, self.size, self.interpolation, flip, self.mean, self.std, out=out)

    def batch(self, clips: Sequence[torch.Tensor]) -> torch.Tensor:
        """[B, 3, T, res, res] bf16, each clip transformed straight into its slot."""
        T = clips[0].shape[0]
        out = torch.empty((len(clips), 3, T, *self.size), dtype=torch.bfloat16, device=clips[0].device)
        for b, c in enumerate(clips):
            self(c, out=out[b])
        return out

"""TEST INFRASTRUCTURE ONLY -- oracle for the device-side video input transform.

(i) import_video_transforms(): the reference's own dataset/video_utils modules, loaded as a synthetic package (the
    dataset package __init__ pulls in the whole training stack) with import stubs for torchvision / cv2 (oracle/shims_video);
    the tensor code paths used here are pure torch.  This container only.
(ii) restate_video_transform(): the same chain restated with plain torch (travels to the GPU box):
    crop -> F.interpolate -> .long() -> flip(W) -> /255, C,T,H,W -> (x - mean) / std
    (dataset/video_utils/functional.py:12-29, 51-72, 95-112, 125-136; video_transforms.py:933-936; volume_transforms.py:40-42).
Parity pinned against (i) through tests/golden/video_tiny.pt; TemporalConsistentRandomAugment is cv2-based (opencv not
installed, version not pinned by the reference): parity unpinned, not restated.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch
import torch.nn.functional as F

REF_ROOT = os.environ.get("MPLUG_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims_video")   # kept apart from oracle/shims: a visible `torchvision` stub confuses transformers


def import_video_transforms():
    if "refvid" not in sys.modules:
        pkg = types.ModuleType("refvid")
        pkg.__path__ = [os.path.join(REF_ROOT, "dataset", "video_utils")]
        sys.modules["refvid"] = pkg
    sys.path.insert(0, _SHIMS)
    try:
        mods = importlib.import_module("refvid.video_transforms"), importlib.import_module("refvid.volume_transforms")
    finally:
        sys.path.remove(_SHIMS)
        for name in [n for n in sys.modules if n == "cv2" or n == "torchvision" or n.startswith("torchvision.")]:
            if str(getattr(sys.modules[name], "__file__", "") or "").startswith(_SHIMS):
                del sys.modules[name]          # the stubs must not leak to other importers (transformers probes torchvision)
    return mods


def restate_video_transform(clip_u8, box, size, mode, flip, mean, std):
    """clip uint8 [T,H,W,3] (CPU) -> float32 [3,T,oh,ow]."""
    i, j, h, w = box
    c = clip_u8[:, i:i + h, j:j + w, :].permute(0, 3, 1, 2).float()
    c = F.interpolate(c, size=(size[0], size[1]), mode=mode)
    c = c.permute(0, 2, 3, 1).long()
    if flip:
        c = c.flip(-2)
    c = c.permute(3, 0, 1, 2) / 255.
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None, None]
    return (c - m) / s

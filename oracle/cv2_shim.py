"""TEST INFRASTRUCTURE: a stand-in `cv2` module, just enough for `import cv2` in the reference's
dataset/video_utils/randaugment_video.py to succeed in an image without opencv.  The functions the nine pre-train ops reach
(warpAffine, getRotationMatrix2D, filter2D) are routed to the restatements in oracle/augment.py -- see the header there for what
that does and does not pin.  Everything else raises."""
import sys
import types

import numpy as np

from . import augment as A

INTER_LINEAR = 1
INTER_NEAREST = 0


def warpAffine(img, M, dsize, borderValue=(0, 0, 0), flags=INTER_LINEAR, **kw):
    assert flags == INTER_LINEAR and tuple(dsize) == (img.shape[1], img.shape[0]) and not kw
    return A.warp_affine_linear(img, M, borderValue)


def getRotationMatrix2D(center, angle, scale):
    return A.get_rotation_matrix_2d(center, angle, scale)


def filter2D(img, ddepth, kernel):
    assert ddepth == -1 and kernel.shape == (3, 3)
    return A.filter2d_3x3(img, kernel)


def _missing(name):
    def f(*a, **k):
        raise NotImplementedError(f"cv2.{name}: not part of the oracle's cv2 shim")
    return f


def install():
    """put the shim at sys.modules['cv2'] (no-op when a real cv2 is importable)"""
    try:
        import cv2  # noqa: F401
        return False
    except ImportError:
        pass
    m = types.ModuleType("cv2")
    m.INTER_LINEAR, m.INTER_NEAREST = INTER_LINEAR, INTER_NEAREST
    m.warpAffine, m.getRotationMatrix2D, m.filter2D = warpAffine, getRotationMatrix2D, filter2D
    for n in ("calcHist", "split", "merge", "flip", "LUT", "cvtColor", "resize"):
        setattr(m, n, _missing(n))
    sys.modules["cv2"] = m
    return True

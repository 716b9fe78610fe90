"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's TemporalConsistentRandomAugment
(dataset/video_utils/randaugment_video.py) -- the clip-consistent RandAugment of the pre-train / fine-tune loaders
(dataset/__init__.py:60-78: N = 2, M = 5, nine ops).  Only tests/, smoke() and bench.py's cpu_baseline may import this.

What is pinned and what is not.  The reference module imports cv2, and opencv is not in this image:
  * the draw logic (get_random_ops / __call__: np.random.choice without replacement, one draw per CLIP, p), the level -> argument
    maps and every op that is plain numpy in the reference (Identity, Contrast, Brightness, Solarize, Posterize, Color) are PINNED:
    tests/test_host_cpu.py imports the reference module itself with a `cv2` shim (oracle/cv2_shim.py) and compares;
  * the ops that call into opencv -- cv2.warpAffine (ShearX/Y, TranslateX/Y, Rotate), cv2.getRotationMatrix2D and
    cv2.filter2D (Sharpness) -- are restated from opencv's published algorithm (modules/imgproc/src/imgwarp.cpp:
    WarpAffineInvoker + remapBilinear with the 5-bit fixed-point bilinear table; filter.cpp: float accumulation in tap order,
    BORDER_REFLECT_101, round-half-even) and are "PARITY UNPINNED" at that boundary: the shim routes the reference's cv2 calls to
    these very functions, so the comparison pins the reference's own code around them, not opencv's arithmetic.
"""
from __future__ import annotations

import math

import numpy as np

MAX_LEVEL = 10                      # randaugment_video.py:301
TRANSLATE_CONST = 10                # :300
REPLACE_VALUE = (128, 128, 128)     # :302
DEFAULT_AUGS = ["Identity", "Contrast", "Brightness", "Sharpness", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate"]   # dataset/__init__.py:65-66


# ----------------------------------------------------------------------------------------------- opencv restatements (unpinned)
def cv_round(v):
    """cvRound / saturate_cast<int>(double): round half to even"""
    return np.rint(v).astype(np.int64)


def get_rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D (imgwarp.cpp): angle in degrees, counter-clockwise, double precision"""
    a = angle * math.pi / 180.0
    alpha, beta = scale * math.cos(a), scale * math.sin(a)
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def invert_affine(M):
    """the inversion cv::warpAffine applies to the forward matrix (no WARP_INVERSE_MAP), in double"""
    M = np.asarray(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def warp_affine_linear(img, M, fill):
    """cv2.warpAffine(img, M, (W, H), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=fill) on uint8 [H, W, C]:
    source coordinates in 10-bit fixed point (round_delta 16), 5-bit interpolation fractions, integer bilinear weights
    (32 - fx)(32 - fy) * 32 ... summing to 2^15, result (sum + 2^14) >> 15; a neighbour outside the image is the border value."""
    H, W, C = img.shape
    Mi = invert_affine(M)
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    round_delta = AB_SCALE // 32 // 2
    xs = np.arange(W, dtype=np.float64)
    adelta = cv_round(Mi[0, 0] * xs * AB_SCALE)
    bdelta = cv_round(Mi[1, 0] * xs * AB_SCALE)
    ys = np.arange(H, dtype=np.float64)
    X0 = cv_round((Mi[0, 1] * ys + Mi[0, 2]) * AB_SCALE) + round_delta
    Y0 = cv_round((Mi[1, 1] * ys + Mi[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy = np.clip(X >> INTER_BITS, -32768, 32767), np.clip(Y >> INTER_BITS, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    fillv = np.asarray(fill, dtype=np.int64).reshape(1, 1, -1)[..., :C]
    src = img.astype(np.int64)

    def tap(dy, dx):
        yy, xx = sy + dy, sx + dx
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(inside[..., None], v, fillv)
    acc = tap(0, 0) * w[0][..., None] + tap(0, 1) * w[1][..., None] + tap(1, 0) * w[2][..., None] + tap(1, 1) * w[3][..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def filter2d_3x3(img, kernel):
    """cv2.filter2D(img, -1, kernel) for a 3 x 3 float32 kernel on uint8 [H, W, C]: BORDER_REFLECT_101, float32 accumulation in
    row-major tap order, saturate_cast<uchar> = round half to even"""
    H, W, C = img.shape
    k = np.asarray(kernel, dtype=np.float32)
    p = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="reflect").astype(np.float32)
    acc = np.zeros((H, W, C), dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            acc = (acc + k[dy, dx] * p[dy:dy + H, dx:dx + W]).astype(np.float32)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------------------------- the reference's ops (:7-212)
def identity_func(img):
    return img


def contrast_func(img, factor):      # :120-130
    mean = np.sum(np.mean(img, axis=(0, 1)) * np.array([0.114, 0.587, 0.299]))
    table = np.array([(el - mean) * factor + mean for el in range(256)]).clip(0, 255).astype(np.uint8)
    return table[img]


def brightness_func(img, factor):    # :133-139
    table = (np.arange(256, dtype=np.float32) * factor).clip(0, 255).astype(np.uint8)
    return table[img]


def sharpness_func(img, factor):     # :142-160
    kernel = np.ones((3, 3), dtype=np.float32)
    kernel[1][1] = 5
    kernel /= 13
    degenerate = filter2d_3x3(img, kernel)
    if factor == 0.0:
        return degenerate
    if factor == 1.0:
        return img
    out = img.astype(np.float32)
    degenerate = degenerate.astype(np.float32)[1:-1, 1:-1, :]
    out[1:-1, 1:-1, :] = degenerate + factor * (out[1:-1, 1:-1, :] - degenerate)
    return out.astype(np.uint8)


def shear_x_func(img, factor, fill=(0, 0, 0)):       # :163-167
    return warp_affine_linear(img, np.float32([[1, factor, 0], [0, 1, 0]]), fill)


def shear_y_func(img, factor, fill=(0, 0, 0)):       # :198-202
    return warp_affine_linear(img, np.float32([[1, 0, 0], [factor, 1, 0]]), fill)


def translate_x_func(img, offset, fill=(0, 0, 0)):   # :170-177
    return warp_affine_linear(img, np.float32([[1, 0, -offset], [0, 1, 0]]), fill)


def translate_y_func(img, offset, fill=(0, 0, 0)):   # :180-187
    return warp_affine_linear(img, np.float32([[1, 0, 0], [0, 1, -offset]]), fill)


def rotate_func(img, degree, fill=(0, 0, 0)):        # :67-75
    H, W = img.shape[0], img.shape[1]
    return warp_affine_linear(img, get_rotation_matrix_2d((W / 2, H / 2), degree, 1), fill)


FUNC = {"Identity": identity_func, "Contrast": contrast_func, "Brightness": brightness_func, "Sharpness": sharpness_func,
        "ShearX": shear_x_func, "ShearY": shear_y_func, "TranslateX": translate_x_func, "TranslateY": translate_y_func, "Rotate": rotate_func}


def level_to_args(name, level):      # :219-298
    if name == "Identity":
        return ()
    if name in ("Contrast", "Brightness", "Sharpness", "Color"):
        return ((level / MAX_LEVEL) * 1.8 + 0.1,)
    if name in ("ShearX", "ShearY"):
        return ((level / MAX_LEVEL) * 0.3, REPLACE_VALUE)
    if name in ("TranslateX", "TranslateY"):
        return ((level / MAX_LEVEL) * float(TRANSLATE_CONST), REPLACE_VALUE)
    if name == "Rotate":
        return ((level / MAX_LEVEL) * 30, REPLACE_VALUE)
    raise KeyError(name)


class TemporalConsistentRandomAugment:
    """:323-361 -- one set of ops per CLIP (drawn once, applied to every frame), numpy's global RNG, the same calls in the same order"""

    def __init__(self, N=2, M=10, p=0.0, augs=()):
        self.N, self.M, self.p = N, M, p
        self.augs = list(augs) if augs else list(DEFAULT_AUGS)

    def draw(self):
        ops = [(str(op), self.M) for op in np.random.choice(self.augs, self.N, replace=False)]
        apply_or_not = np.random.random(size=self.N) > self.p
        return ops, apply_or_not

    def __call__(self, frames: np.ndarray) -> np.ndarray:
        """uint8 [T, H, W, 3] -> float32 [T, H, W, 3] (the reference returns .float())"""
        assert frames.shape[-1] == 3
        ops, apply_or_not = self.draw()
        out = []
        for img in frames.astype(np.uint8):
            for i, (name, level) in enumerate(ops):
                if apply_or_not[i]:
                    img = FUNC[name](img, *level_to_args(name, level))
            out.append(img)
        return np.stack(out).astype(np.float32)

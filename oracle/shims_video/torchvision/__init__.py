"""TEST INFRASTRUCTURE ONLY -- import stub: the reference's dataset/video_utils modules import torchvision at module
scope; the tensor code paths the oracle exercises (crop / F.interpolate resize / flip / ClipToTensor / Normalize)
never call it.  torchvision is not installed in this image."""
from . import transforms  # noqa: F401

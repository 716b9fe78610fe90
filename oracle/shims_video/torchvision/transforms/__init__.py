from . import functional  # noqa: F401


class _Unavailable:
    def __init__(self, *a, **k):
        raise NotImplementedError("torchvision is not installed; only the tensor video transforms are exercised")


Compose = Resize = ToTensor = Normalize = RandomResizedCrop = RandomHorizontalFlip = _Unavailable

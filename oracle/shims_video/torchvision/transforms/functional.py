def __getattr__(name):
    raise NotImplementedError(f"torchvision.transforms.functional.{name}: torchvision is not installed")

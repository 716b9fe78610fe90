import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import youku_mplug_amd  # noqa: F401  (raises loudly if libmpv_hip.so is missing)
    from youku_mplug_amd import _lib
    _lib.check(_lib.lib().mpv_check_device(), "mpv_check_device")
    return torch.device("cuda:0")

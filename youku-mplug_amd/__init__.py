"""youku-mplug_amd: MI355X (gfx950) native hot path of mPLUG-Video pre-training.

Host side mirrors the reference's module tree (state-dict compatible) and engine surface and
calls hand-written HIP kernels through the C ABI in include/mpv.h (libmpv_hip.so).
Import as `import youku_mplug_amd` (alias module at the repo root).
"""
from . import _lib, ops  # noqa: F401

__all__ = ["_lib", "ops"]
__version__ = "0.1.0"

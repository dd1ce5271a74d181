"""MI355X-native multi-pass GAN hot path (gfx950 HIP kernels behind a C ABI).

Import with ``importlib.import_module("multi-pass-gan_amd")`` or through the
``mpgan_amd`` alias module at the repository root.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]

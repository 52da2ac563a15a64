"""MI355X-native implementation of MO-VAE's per-step training hot path.

The directory is named `mo-vae_amd` (not an importable identifier); import it as `movae_amd`
through the loader module at the repository root (movae_amd.py).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def load_library():
    """dlopen libmovae_hip.so (raises if absent: there is no CPU fallback)."""
    return _lib.load()

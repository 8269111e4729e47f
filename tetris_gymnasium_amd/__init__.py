"""tetris_gymnasium_amd — MI355X-native batched Tetris step engine behind the Tetris-Gymnasium API surfaces.

Python host code calls hand-written HIP (gfx950) through a thin ctypes C ABI.  No PyTorch/Triton/JAX.
"""
from . import _native  # noqa: F401
from .engine import DeviceArray, Engine, EngineError  # noqa: F401

__all__ = ["DeviceArray", "Engine", "EngineError"]

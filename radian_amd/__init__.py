"""radian_amd -- MI355X (gfx950) backend for RADIAN's inference + decode hot path.

Python host code calling hand-written HIP kernels through a ctypes C ABI (libradian_hip.so,
declared in include/radian_hip.h).  There is no CPU fallback: without the built library or without a
GPU every compute entry point raises.
"""
from .backend import Backend, RadianHipError, lib_path  # noqa: F401

__all__ = ["Backend", "RadianHipError", "lib_path"]

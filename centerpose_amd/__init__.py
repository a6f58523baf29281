"""centerpose_amd — MI355X-native (gfx950) inference hot path for CenterPose.

The compute lives in ``csrc/`` (hand-written HIP kernels behind a C ABI declared in
``include/centerpose_hip.h``); ``lib/`` mirrors the reference's Python entry points
(``lib.opts``, ``lib.models.model``, ``lib.detectors``) so ``demo.py`` stays drop-in.
Importing this package does not load the HIP library; ``centerpose_amd.hip`` does, and it
raises if the library is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"

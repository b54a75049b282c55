"""interactive_deep_colorization_amd -- MI355X-native Local-Hints colorization forward pass.

One hot path, built from scratch for gfx950: ``ColorizeImageTorch.net_forward``
(``data/colorize_image.py:249-268`` of junyanz/interactive-deep-colorization) ->
``SIGGRAPHGenerator.forward`` (``models/pytorch/model.py:134-175``), as hand-written HIP
kernels behind a C ABI (``include/ideepcolor.h``), driven from Python through ctypes.

Importing the package does not load the shared library; constructing a model does, and fails
loudly if the library is not built or no gfx950 device is present (there is no CPU fallback).
"""
__all__ = ["api", "colorize_image", "colorspace", "engine", "sharded", "workloads"]
__version__ = "0.1.0"

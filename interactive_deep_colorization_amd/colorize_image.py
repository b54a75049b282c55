"""Import shim so reference callers change one line:

    from data import colorize_image as CI                                  # reference
    from interactive_deep_colorization_amd import colorize_image as CI      # this package

The classes live in :mod:`interactive_deep_colorization_amd.api`.
"""
from .api import (ColorizeImageBase, ColorizeImageCaffe, ColorizeImageTorch,  # noqa: F401
                  ColorizeImageTorchDist, create_temp_directory, lab2rgb_transpose,
                  put_point, rgb2lab_transpose)

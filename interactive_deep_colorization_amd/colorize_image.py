"""Import shim so reference callers change one line:

    from data import colorize_image as CI                                  # reference (ideepcolor.py:8, both notebooks)
    from interactive_deep_colorization_amd import colorize_image as CI      # this package

Every class the reference callers construct resolves here -- ``ColorizeImageCaffe`` / ``ColorizeImageCaffeDist``
(``ideepcolor.py:62-66``, the default ``--backend caffe``), ``ColorizeImageTorch`` / ``ColorizeImageTorchDist``
(``ideepcolor.py:68-72``), ``ColorizeImageCaffeGlobDist`` (``DemoGlobalHistogramTransfer.ipynb``) -- with the reference's
constructor / ``prep_net`` argument names and defaults (``tests/test_round4_cpu.py`` checks the list against
``data/colorize_image.py``).  The classes live in :mod:`interactive_deep_colorization_amd.api`.
"""
from .api import *  # noqa: F401,F403
from .api import __all__  # noqa: F401

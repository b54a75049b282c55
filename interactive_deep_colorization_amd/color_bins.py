"""The ab quantisation tables the Caffe wrappers load (``data/colorize_image.py:398-399,486-489``:
``data/color_bins/pts_in_hull.npy``, ``pts_grid.npy``, ``in_hull.npy``).

The reference ships them as data files; here they are rebuilt from their definition so that the package needs no
side files:

* ``pts_grid``   (529, 2) int64: the 23 x 23 grid a, b in {-110, -100, ..., 110}, **a-major** (b varies fastest);
* ``in_hull``    (529,) bool   : which grid points lie inside the sRGB gamut hull (313 of them).  This mask is the
  reference's data table, carried here as its 529 bits (``oracle/make_color_bins.py`` regenerates the string from the
  reference file and ``tests/test_api_host_cpu.py`` compares all three tables with the reference's files when the
  reference checkout is present);
* ``pts_in_hull``(313, 2) int64: ``pts_grid[in_hull]`` -- the 313 bin centres of the classification head.

``load(dir)`` reads the three ``.npy`` files from a directory instead (e.g. a reference checkout's
``data/color_bins``) and fails loudly when one is missing.
"""
import os

import numpy as np

_IN_HULL_BITS = ("0000000000000003e0003fc001ff800fff003fff00fffe03fffc07fff81ffff07fffe1ffff83ffff0ffffe3ffffc7ffff1ffffe3ffff"
                 "c7ffff0ffffe07fe0000000000")


def pts_grid():
    axis = np.arange(-110, 120, 10)
    return np.array(np.meshgrid(axis, axis, indexing='ij')).reshape((2, 529)).T.astype(np.int64)


def in_hull():
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(_IN_HULL_BITS), dtype=np.uint8))[:529]
    return bits.astype(bool)


def pts_in_hull():
    return pts_grid()[in_hull()]


def load(directory=None):
    """(pts_in_hull, pts_grid, in_hull): built in (``directory=None``) or read from ``directory``'s .npy files."""
    if directory is None:
        return pts_in_hull(), pts_grid(), in_hull()
    out = []
    for name in ('pts_in_hull.npy', 'pts_grid.npy', 'in_hull.npy'):
        p = os.path.join(directory, name)
        if not os.path.exists(p):
            raise FileNotFoundError('colour bin table %s not found' % p)
        out.append(np.load(p))
    return tuple(out)

"""Round-3 CPU tests: lifetime of pinned buffers, wrapper methods the reference's Caffe distribution class has, and the
bench line's helper functions.  No GPU, no compute calls into the library."""
import ctypes
import gc
import os
import sys

import numpy as np

from interactive_deep_colorization_amd import api, engine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeLib(object):
    """idc_alloc_host / idc_free_host over ctypes buffers: the lifetime logic is host Python."""

    def __init__(self):
        self.live, self.freed = {}, []

    def idc_alloc_host(self, n):
        b = (ctypes.c_char * n)()
        a = ctypes.addressof(b)
        self.live[a] = b
        return a

    def idc_free_host(self, p):
        self.freed.append(p.value)
        self.live.pop(p.value, None)
        return 0


def test_pinned_arrays_own_their_memory():
    """engine.pinned_empty: the pinned allocation lives as long as ANY view of it, not as long as the engine
    (arrays that outlived close() were a silent use-after-free)."""
    lib = _FakeLib()
    raw = np.asarray(engine._PinnedBuffer(lib, 64))
    a = raw[:64].view(np.float32).reshape(4, 4)
    del raw
    a[:] = 3.0
    gc.collect()
    assert lib.freed == [] and float(a.sum()) == 48.0
    b = a[1:]
    del a
    gc.collect()
    assert lib.freed == []                       # a slice still holds the owner
    del b
    gc.collect()
    assert len(lib.freed) == 1 and not lib.live


def test_caffe_dist_class_has_the_plot_methods():
    """data/colorize_image.py:549-561: plot_dist_grid / plot_dist_entropy exist on both distribution classes with the
    reference's signatures."""
    import inspect
    for cls in (api.ColorizeImageCaffeDist, api.ColorizeImageTorchDist):
        assert list(inspect.signature(cls.plot_dist_grid).parameters) == ["self", "h", "w"]
        assert list(inspect.signature(cls.plot_dist_entropy).parameters) == ["self"]
    # own definitions in the Caffe class (not only aliases of the torch twin's)
    assert "plot_dist_grid" in api.ColorizeImageCaffeDist.__dict__ and "plot_dist_entropy" in api.ColorizeImageCaffeDist.__dict__


def test_plot_dist_grid_draws_the_23x23_slice():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    m = api.ColorizeImageCaffeDist.__new__(api.ColorizeImageCaffeDist)
    grid = np.random.RandomState(0).rand(23, 23, 8, 8)
    m.__dict__["_lazy_dist_ab_grid"] = grid
    m._dist_on_device = False
    m.dist_entropy = -np.ones((8, 8))
    m.plot_dist_grid(2, 3)
    img = plt.gca().images[0] if plt.gca().images else plt.gcf().axes[0].images[0]
    np.testing.assert_array_equal(np.asarray(img.get_array()), grid[:, :, 2, 3])
    m.plot_dist_entropy()
    plt.close("all")


def test_bench_spread_helper():
    sys.path.insert(0, REPO)
    import bench
    s = bench._spread([100.0, 90.0, 110.0], 20)
    assert s["min"] == 90.0 and s["median"] == 100.0 and s["max"] == 110.0 and s["values"][0] == 100.0 and s["n"] == 3

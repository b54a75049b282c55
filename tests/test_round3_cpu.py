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


def test_emulation_fp32_mode_is_the_oracle_and_winograd_forms_are_exact_in_fp32():
    """oracle/emulate.py (the CPU restatement of the engine's reduced-precision arithmetic behind profiles/parity_r03.json):
    with every rounding switched off it is the oracle; its Winograd F(2x2,3x3) / F(2,3) forms equal the direct conv up to
    fp32 rounding (dilation 1 and 2); the all-bf16 mode sits at the distance the GPU's bf16 path is measured at."""
    import torch
    from interactive_deep_colorization_amd import workloads
    from oracle import emulate, siggraph_torch, weights
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 8, 12, 16), torch.randn(5, 8, 3, 3), torch.randn(5)
    for d in (1, 2):
        ref = torch.nn.functional.conv2d(x, w, b, padding=d, dilation=d)
        for mode in ("wino2d_fp32", "wino1d_fp32"):
            assert float((emulate._conv3(x, w, b, d, mode) - ref).abs().max()) < 5e-5, (d, mode)
        q = emulate._conv3(emulate.q(x), w, b, d, "wino2d")                  # bf16 operands: close to the bf16 direct conv, not equal
        assert 1e-4 < float((q - ref).abs().max()) < 0.5
    sd = weights.make_state_dict(1, "torch")
    L, ab, m = workloads.random_batch(2, 32, seed=3)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.5)
    assert np.abs(emulate.forward(sd, L, ab, m, 0.5, default="fp32") - ref).max() < 1e-3
    wino = emulate.forward(sd, L, ab, m, 0.5, default="fp32", modes={n: "wino2d_fp32" for n in emulate.WINO_ELIGIBLE})
    assert np.abs(wino - ref).max() < 1e-3
    bf = emulate.error_stats(emulate.forward(sd, L, ab, m, 0.5, default="bf16"), ref)
    assert 1e-3 < bf["max_abs"] < 0.6 and bf["mean_abs"] < 0.06          # the bf16 bound of the GPU tests
    assert set(emulate.GROUPS["encoder"] + emulate.GROUPS["trunk"] + emulate.GROUPS["decoder"]) == set(emulate.LAYER_NAMES)


def test_parity_r03_record_is_consistent():
    """profiles/parity_r03.json: the all-bf16 row reproduces what the GPU measured (profiles/parity_r02.json) and no single
    stored-tensor group carries the error on both weight styles."""
    import json
    with open(os.path.join(REPO, "profiles", "parity_r03.json")) as f:
        d = json.load(f)["weights"]
    assert 10.0 < d["he"]["all_bf16"]["max_abs"] < 16.0 and 0.10 < d["torch"]["all_bf16"]["max_abs"] < 0.18
    for style in ("he", "torch"):
        base = d[style]["all_bf16"]["mean_abs"]
        assert all(d[style][g]["mean_abs"] > 0.4 * base for g in ("encoder_fp32", "trunk_fp32"))
    assert d["he"]["decoder_fp32"]["mean_abs"] > 0.9 * d["he"]["all_bf16"]["mean_abs"]          # he: not the decoder
    assert d["torch"]["decoder_fp32"]["mean_abs"] < 0.5 * d["torch"]["all_bf16"]["mean_abs"]    # torch-init: the decoder


def test_result_pool_recycles_only_dead_buffers():
    """engine._PinnedPool: results of the blocking calls come from recycled pinned buffers.  A buffer goes back on the free
    list only when no array views it any more; it is never handed out twice at a time; the retained bytes are bounded."""
    lib = _FakeLib()
    pool = engine._PinnedPool(lib)
    a = pool.take((2, 8), np.float32)
    b = pool.take((2, 8), np.float32)
    assert a.ctypes.data != b.ctypes.data and len(lib.live) == 2 and pool.retained == 0
    a[:] = 1.0
    addr_a = a.ctypes.data
    view = a[0]                                  # a view keeps the buffer out of the pool
    del a
    gc.collect()
    assert pool.retained == 0 and float(view[0]) == 1.0
    del view
    gc.collect()
    assert pool.retained == 64 and not lib.freed          # recycled, not returned to the driver
    c = pool.take((16,), np.float32)             # same byte size -> the recycled buffer
    assert c.ctypes.data == addr_a and pool.retained == 0
    assert pool.take((0, 4), np.float32).shape == (0, 4)  # empty and oversized requests are plain numpy
    big = pool.take((engine._PinnedPool.MAX_ONE // 4 + 1,), np.float32)
    assert len(lib.live) == 2 and big.nbytes > engine._PinnedPool.MAX_ONE
    # beyond MAX_TOTAL retained bytes a dead buffer is freed instead of kept
    pool.MAX_TOTAL = 64
    del b, c
    gc.collect()
    assert pool.retained == 64 and len(lib.freed) == 1
    pool.drain()
    assert pool.retained == 0 and len(lib.freed) == 2 and not lib.live

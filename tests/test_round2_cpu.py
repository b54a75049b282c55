"""Round-2 CPU tests: colour-bin tables, the display/full-res restatements (pinned against scipy where scipy is the
reference's own dependency), wrapper bookkeeping added in round 2, the .pth reader, the C ABI surface."""
import collections
import os
import re

import numpy as np
import pytest

from interactive_deep_colorization_amd import _native as N
from interactive_deep_colorization_amd import api, color_bins
from oracle import display

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BINS = "/root/reference/data/color_bins"


def test_color_bin_tables_definition():
    g, h, p = color_bins.pts_grid(), color_bins.in_hull(), color_bins.pts_in_hull()
    assert g.shape == (529, 2) and h.shape == (529,) and h.sum() == 313 and p.shape == (313, 2)
    assert tuple(g[0]) == (-110, -110) and tuple(g[1]) == (-110, -100) and tuple(g[-1]) == (110, 110)   # a-major
    assert np.array_equal(g[h], p)
    # the torch class uses the TRANSPOSED (a fastest) grid (colorize_image.py:213,283): not the same order
    assert not np.array_equal(api._grid_529(), g) and np.array_equal(np.sort(api._grid_529(), axis=0), np.sort(g, axis=0))


@pytest.mark.skipif(not os.path.isdir(REF_BINS), reason="reference checkout not present (authoring container only)")
def test_color_bin_tables_equal_the_reference_files():
    assert np.array_equal(color_bins.pts_in_hull(), np.load(os.path.join(REF_BINS, "pts_in_hull.npy")))
    assert np.array_equal(color_bins.pts_grid(), np.load(os.path.join(REF_BINS, "pts_grid.npy")))
    assert np.array_equal(color_bins.in_hull(), np.load(os.path.join(REF_BINS, "in_hull.npy")))
    got = color_bins.load(REF_BINS)
    assert np.array_equal(got[0], color_bins.pts_in_hull())


def test_color_bins_directory_must_be_complete(tmp_path):
    np.save(str(tmp_path / "pts_in_hull.npy"), color_bins.pts_in_hull())
    with pytest.raises(FileNotFoundError):                          # fail at construction, not at first use
        api.ColorizeImageCaffeDist(Xd=16, color_bins_dir=str(tmp_path))
    m = api.ColorizeImageCaffeDist(Xd=16)
    assert m.pts_in_hull.shape == (313, 2) and m.in_hull.sum() == 313 and m.pts_grid.shape == (529, 2)
    assert hasattr(m, "plot_dist_grid") and hasattr(m, "plot_dist_entropy")


def test_zoom_restatements_are_scipy(tmp_path):
    """The full-resolution getters use scipy.ndimage.zoom (colorize_image.py:128,135,151-157); scipy is installed, so the
    restatement the GPU kernel is checked against is pinned to it here."""
    from scipy.ndimage import zoom
    rs = np.random.RandomState(1)
    x = rs.uniform(-110, 110, (2, 48, 40))
    for oh, ow in ((48, 40), (101, 77), (96, 80), (300, 333), (49, 41)):
        np.testing.assert_allclose(display.zoom_linear(x, oh, ow), zoom(x, (1, 1. * oh / 48, 1. * ow / 40), order=1), atol=1e-11)
        assert np.array_equal(display.zoom_nearest(x, oh, ow), zoom(x, (1, 1. * oh / 48, 1. * ow / 40), order=0))


def test_cubic_restatement_properties():
    """cv2 INTER_CUBIC restatement (unpinned against cv2 itself): identity at scale 1, constants preserved (the four Keys
    coefficients sum to 1; A = -0.75 does not reproduce ramps, unlike A = -0.5), mirror symmetry, and the A = -0.75
    kernel values at the half-pixel phase."""
    rs = np.random.RandomState(2)
    x = rs.uniform(-100, 100, (24, 32))
    assert np.array_equal(display.resize_cubic_cv2(x, 24, 32), x)
    np.testing.assert_allclose(display.resize_cubic_cv2(np.full((24, 32), 7.25), 50, 61), 7.25, atol=1e-5)
    np.testing.assert_allclose(display.resize_cubic_cv2(x[:, ::-1], 40, 50)[:, ::-1], display.resize_cubic_cv2(x, 40, 50), atol=1e-3)   # float32 coefficients: c3 = 1 - c0 - c1 - c2 is not the mirror of c0 to the last bit
    c = display._cubic_coeffs(np.array([0.5], np.float32))[0]
    np.testing.assert_allclose(c, [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)


def test_cubic_restatement_against_an_independent_library():
    """Round 6: cv2 itself cannot be installed here, but torch ships an independent implementation of the SAME published algorithm -- bicubic interpolation
    with the Keys kernel at A = -0.75, half-pixel centres (align_corners=False) and indices clamped to the image, which is OpenCV's INTER_CUBIC convention
    (ui/gui_draw.py:281 calls cv2.resize(..., interpolation=cv2.INTER_CUBIC)).  oracle/display.py must agree with it: to rounding where the sampling
    phases are exact in float32 (2x), and to the precision of OpenCV's FLOAT32 coordinate / coefficient arithmetic -- which the restatement follows and
    torch's float64 path does not -- everywhere else (measured 2e-5 of the data range)."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(4)
    for (H, W, oh, ow) in ((64, 64, 200, 150), (256, 256, 500, 333), (48, 40, 96, 80), (100, 120, 70, 90), (256, 256, 531, 400)):
        x = rs.uniform(-100, 100, (H, W))
        ours = display.resize_cubic_cv2(x, oh, ow)
        lib = F.interpolate(torch.from_numpy(x)[None, None].double(), size=(oh, ow), mode="bicubic", align_corners=False)[0, 0].numpy()
        tol = 1e-11 if (oh, ow) == (2 * H, 2 * W) else 5e-5 * 200
        assert np.abs(ours - lib).max() <= tol, ((H, W, oh, ow), float(np.abs(ours - lib).max()))


def test_global_stats_saturation_against_an_independent_library():
    """Round 6: the saturation half of models/global_model/global_stats.prototxt (caffe_files/caffe_traininglayers.py:78 calls skimage.color.rgb2hsv, not
    installable here) held against matplotlib's rgb_to_hsv -- an independent implementation of the same HSV definition S = (max - min) / max, 0 for black --
    on random, black, white and grey pixels: the checker's global mean saturation must equal the library's to rounding."""
    mc = pytest.importorskip("matplotlib.colors")
    from interactive_deep_colorization_amd import color_bins
    from oracle import colorspace
    rs = np.random.RandomState(0)
    rgb = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    rgb[:8] = 0; rgb[8:16] = 255
    rgb[16:20, :, 1] = rgb[16:20, :, 0]; rgb[16:20, :, 2] = rgb[16:20, :, 0]
    _, s_avg = colorspace.global_stats(rgb, color_bins.pts_in_hull())
    lib = float(mc.rgb_to_hsv(rgb.astype(np.float64) / 255.0)[..., 1].mean())
    assert abs(s_avg - lib) <= 1e-12, (s_avg, lib)


class _Stub(object):
    closed = False

    def close(self):
        self.closed = True


def test_new_engine_resets_every_device_side_flag():
    m = api.ColorizeImageTorchDist(Xd=16)
    m._l_resident = True; m._hints_on_device = True; m._dist_on_device = True; m.dist_ab_set = True
    old = _Stub(); m.net = old
    new = _Stub()
    m._new_engine(new)
    assert m.net is new and old.closed and m.net_set
    assert not m._l_resident and not m._hints_on_device and not m._dist_on_device and not m.dist_ab_set
    assert m._dev_out_valid is False and not m._out_pending       # (round 5: the token became a flag + the lazily fetched attributes)


def test_read_state_dict_drops_metadata(tmp_path):
    import torch
    od = collections.OrderedDict([("model1.0.weight", torch.zeros(2, 2)), ("model1.4.num_batches_tracked", torch.tensor(3))])
    od._metadata = collections.OrderedDict([("", {"version": 1}), ("model1", {"version": 1})])
    p = str(tmp_path / "w.pth")
    torch.save(od, p)
    sd = api.read_state_dict(p)
    assert not hasattr(sd, "_metadata") and list(sd.keys()) == list(od.keys())
    np.savez(str(tmp_path / "w.npz"), **{"model1.0.weight": np.ones((2, 2), np.float32)})
    assert api.read_state_dict(str(tmp_path / "w.npz"))["model1.0.weight"].shape == (2, 2)


def test_header_and_binding_agree():
    """Every idc_* function the public header declares is bound (with a prototype) by _native.py and exported by the
    built library -- the C ABI is the product boundary (include/ideepcolor.h)."""
    text = open(os.path.join(REPO, "include", "ideepcolor.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(idc_[a-z0-9_]+)\s*\(", text))
    assert declared == set(N.EXPORTED_SYMBOLS), (declared ^ set(N.EXPORTED_SYMBOLS))
    lib = N.load()
    for s in declared:
        assert hasattr(lib, s) and getattr(lib, s).argtypes is not None, s

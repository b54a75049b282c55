"""Host colour maths (skimage restatement: PARITY UNPINNED against the library -- skimage is not installable here -- pinned
against an independent 50-digit evaluation of the published formulas with skimage's constants, tests/golden/colorspace_pairs.npz,
written by oracle/make_golden_colorspace.py; plus the textbook known answers)."""
import os

import numpy as np
import pytest

from interactive_deep_colorization_amd import colorspace as prod
from oracle import colorspace as ora

# CIE Lab (D65, 2 deg) of the sRGB primaries / white / black -- published values
KNOWN = {
    (255, 255, 255): (100.0, 0.0053, -0.0104),
    (0, 0, 0): (0.0, 0.0, 0.0),
    (255, 0, 0): (53.2408, 80.0925, 67.2032),
    (0, 255, 0): (87.7347, -86.1827, 83.1793),
    (0, 0, 255): (32.2970, 79.1875, -107.8602),
}


def test_known_answers():
    for rgb, lab in KNOWN.items():
        px = np.array(rgb, np.uint8).reshape(1, 1, 3)
        for impl in (prod, ora):
            got = impl.rgb2lab(px)[0, 0]
            np.testing.assert_allclose(got, lab, atol=0.02, err_msg="%s %s" % (impl.__name__, rgb))


@pytest.mark.parametrize("impl", [prod, ora], ids=["product", "oracle"])
def test_against_the_independent_50_digit_evaluation(impl):
    """1200 sRGB -> Lab and 1200 Lab -> sRGB pairs (uint8 and float inputs, both sides of every branch knee,
    out-of-gamut and negative-z Lab, the grey axis), each restatement at 1e-9 -- the tolerance the device kernel
    (`lab_post_kernel`, tests/test_net_gpu.py) is then held to against oracle/colorspace.py."""
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colorspace_pairs.npz")) as z:
        g = {k: z[k] for k in z.files}
    assert len(g["rgb_u8"]) + len(g["rgb_f64"]) >= 1000 and len(g["lab_in"]) >= 1000
    np.testing.assert_allclose(impl.rgb2lab(g["rgb_u8"].reshape(-1, 1, 3))[:, 0], g["lab_of_rgb_u8"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(impl.rgb2lab(g["rgb_f64"].reshape(-1, 1, 3))[:, 0], g["lab_of_rgb_f64"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(impl.lab2rgb(g["lab_in"].reshape(-1, 1, 3))[:, 0], g["rgb_of_lab"], rtol=0, atol=1e-9)
    # and the uint8 rendering of data/colorize_image.py:27 (truncating cast): identical wherever the exact value is not within
    # 1e-9 of an integer step
    lab = g["lab_in"].T.reshape(3, 1, -1)
    u8 = impl.lab2rgb_transpose(lab[[0]], lab[1:])[0]
    exact = g["rgb_of_lab"] * 255
    safe = np.abs(exact - np.round(exact)) > 1e-6
    assert np.array_equal(u8[safe], np.floor(exact).astype(np.uint8)[safe]) and safe.mean() > 0.7     # (clipped channels sit exactly on 0 / 255)


def test_product_matches_oracle_and_roundtrips():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (12, 9, 3)).astype(np.uint8)
    lab_p, lab_o = prod.rgb2lab(img), ora.rgb2lab(img)
    np.testing.assert_allclose(lab_p, lab_o, atol=1e-10)
    back = prod.lab2rgb(lab_p)
    np.testing.assert_allclose(back, ora.lab2rgb(lab_o), atol=1e-10)
    assert np.abs(back * 255 - img).max() < 1e-6
    # out-of-gamut Lab is clipped, negative z is zeroed (skimage behaviour)
    weird = np.array([[[50.0, 120.0, -120.0], [5.0, 0.0, 100.0], [99.0, -110.0, 110.0]]])
    np.testing.assert_allclose(prod.lab2rgb(weird), ora.lab2rgb(weird), atol=1e-12)
    assert prod.lab2rgb(weird).min() >= 0 and prod.lab2rgb(weird).max() <= 1
    # uint8 rendering (truncating cast, as data/colorize_image.py:27): generic Lab values, not an exact
    # round trip of integers (those sit on the truncation edge and flip with 1e-13 of rounding)
    l = np.concatenate((rs.uniform(0, 100, (1, 12, 9)), rs.uniform(-90, 90, (2, 12, 9))), axis=0)
    a8, b8 = prod.lab2rgb_transpose(l[[0]], l[1:]), ora.lab2rgb_transpose(l[[0]], l[1:])
    assert a8.dtype == np.uint8 and a8.shape == (12, 9, 3)
    assert np.abs(a8.astype(int) - b8.astype(int)).max() <= 1 and (a8 == b8).mean() > 0.99


def test_resize_bilinear():
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    assert np.array_equal(prod.resize_bilinear_u8(img, 4, 6), img)              # identity
    up = prod.resize_bilinear_u8(img[:, :, 0], 8, 12)
    assert up.shape == (8, 12) and up[0, 0] == img[0, 0, 0] and up[-1, -1] == img[-1, -1, 0]
    const = np.full((5, 7, 3), 77, np.uint8)
    assert (prod.resize_bilinear_u8(const, 256, 256) == 77).all()

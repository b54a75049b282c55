"""GPU parity of the click-session entry points against oracle/session.py:
device-side hint rasterisation, forward from resident planes, the resident distribution and colour suggestions.

Bars: rasterised ab-valued hints and masks bit-exact; RGB-valued hints 1e-4 in ab (float64 skimage formulas on both
sides, rounded to fp32); forward_resident bit-identical to the host-plane forward of the same inputs; draws per
bin and suggestion centres/shares bit-exact (integer bin centres make every float64 sum exact)."""
import numpy as np
import pytest

from interactive_deep_colorization_amd import api, engine, workloads
from oracle import session, weights

pytestmark = pytest.mark.gpu

HINTS_AB = [(10, 12, 16, 18, 25.0, -40.0), (28, 48, 32, 52, -60.0, 10.0), (12, 14, 14, 16, 5.0, 5.0), (60, 60, 70, 70, 70.0, 70.0),
            (-4, 20, 2, 25, -20.0, 33.5), (90, 90, 95, 95, 1.0, 1.0), (40, 10, 36, 3, 9.0, -9.0)]
HINTS_RGB = [(5, 5, 9, 9, 255, 0, 0), (7, 7, 20, 12, 0, 128, 255), (-3, 30, 4, 80, 10, 200, 30), (40, 2, 38, 6, 90, 90, 90),
             (50, 50, 63, 63, 255, 255, 255), (52, 52, 54, 54, 0, 0, 0)]


@pytest.fixture(scope="module")
def sd(make_sd):
    return make_sd(0, "he")


@pytest.mark.parametrize("mode,hints,mask_value", [("ab", HINTS_AB, 1.0), ("rgb", HINTS_RGB, 1.0), ("rgb", HINTS_RGB, 110.0),
                                                   ("ab", [], 1.0)])
def test_hint_rasterisation(sd, mode, hints, mask_value):
    e = engine.HipColorizer(64, 64, max_batch=2, precision="fp32")
    e.set_hints([(0, 0, 63, 63, 1, 2, 3)], mode="ab", img=1)            # the other slot must be left alone
    e.set_hints(hints, mode=mode, img=0, mask_value=mask_value)
    ab, mask = e.hint_planes(0)
    ab_o, mask_o = session.raster_hints(hints, 64, 64, mode, mask_value)
    np.testing.assert_array_equal(mask, mask_o)
    if mode == "ab":
        np.testing.assert_array_equal(ab, ab_o)
    else:
        np.testing.assert_allclose(ab, ab_o, atol=1e-4)
        assert np.abs(ab[:, mask[0] > 0]).max() > 10.0
    ab1, mask1 = e.hint_planes(1)
    assert (mask1 == 1).all() and (ab1[0] == 1).all() and (ab1[1] == 2).all()
    with pytest.raises(Exception):
        e.set_hints([(0, 0, 1, 1, 300, 0, 0)], mode="rgb")
    with pytest.raises(Exception):
        e.set_hints(hints, img=2)
    e.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_resident_equals_host_plane_forward(sd, precision):
    L, _, _ = workloads.random_batch(2, 64, seed=4)
    e = engine.HipColorizer(64, 64, max_batch=2, precision=precision)
    e.load_state_dict(sd)
    planes = []
    for img, hints in enumerate((HINTS_AB, HINTS_AB[:3])):
        e.set_image_l(L[img], img)
        e.set_hints(hints, mode="ab", img=img)
        planes.append(session.raster_hints(hints, 64, 64, "ab"))
    out_r, rgb_r, lab_r = e.forward_resident(2, maskcent=0.5)
    ab = np.stack([p[0] for p in planes]); mask = np.stack([p[1] for p in planes])
    out_h, rgb_h, lab_h = e.forward_rgb(L, ab, mask, 0.5)
    np.testing.assert_array_equal(out_r, out_h)
    np.testing.assert_array_equal(rgb_r, rgb_h)
    np.testing.assert_array_equal(lab_r, lab_h)
    # lab2rgb in between does not disturb the resident planes
    e.lab2rgb(L + 50.0, ab)
    out_r2, _, _ = e.forward_resident(2, maskcent=0.5, want_rgb=False)
    np.testing.assert_array_equal(out_r2, out_h)
    e.close()


def _grid529():
    axis = np.arange(-110, 120, 10)
    return np.array(np.meshgrid(axis, axis)).reshape((2, 529)).T.astype(np.float32)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_resident_distribution_and_suggestions_529(sd, precision):
    L, ab, mask = workloads.random_batch(2, 64, seed=6)
    e = engine.HipColorizer(64, 64, max_batch=2, precision=precision, dist=True)
    e.load_state_dict(sd)
    with pytest.raises(Exception):
        e.dist_at(0, 0)                                                 # "Need to set prediction first"
    out, dq = e.forward_dist(L, ab, mask, 0.0)
    out2, none = e.forward_dist(L, ab, mask, 0.0, want_dist=False)
    assert none is None
    np.testing.assert_array_equal(out, out2)
    np.testing.assert_array_equal(e.get_dist(2), dq)
    c = _grid529()
    for img, y, x, K, N, seed in [(0, 0, 0, 5, 25000, 1), (1, 37, 22, 3, 4000, 99), (1, 63, 63, 8, 25000, 2 ** 31), (0, 10, 50, 1, 100, 5)]:
        pdf = e.dist_at(y, x, img)
        np.testing.assert_array_equal(pdf, dq[img, :, y // 4, x // 4])
        cen, conf, cnt = e.suggest_colors(y, x, c, K=K, N_draws=N, seed=seed, img=img, want_counts=True)
        cen_o, conf_o, cnt_o = session.suggest_colors(pdf, c, K=K, N=N, seed=seed, return_counts=True)
        np.testing.assert_array_equal(cnt, cnt_o)
        np.testing.assert_array_equal(cen, cen_o)
        np.testing.assert_array_equal(conf, conf_o)
        assert abs(conf.sum() - 1.0) < 1e-12 and np.all(np.diff(conf) <= 0)
    with pytest.raises(Exception):
        e.suggest_colors(0, 0, c, K=17)
    with pytest.raises(Exception):
        e.dist_at(64, 0)
    # a forward without the distribution invalidates the resident one
    e.forward(L, ab, mask, 0.0)
    with pytest.raises(Exception):
        e.dist_at(0, 0)
    e.close()


def test_suggestions_on_a_peaked_distribution(sd):
    """A distribution with three far-apart modes (written through a fabricated head bias is not possible from outside,
    so the resident tensor is exercised through a real forward and the mixture through the oracle-equality above);
    here: K larger than the number of drawn bins leaves empty clusters at the end with share 0."""
    L, ab, mask = workloads.random_batch(1, 64, seed=1)
    e = engine.HipColorizer(64, 64, max_batch=1, precision="fp32", dist=True)
    e.load_state_dict(sd)
    e.forward_dist(L, ab, mask, 0.0, want_dist=False)
    c = _grid529()
    cen, conf, cnt = e.suggest_colors(5, 5, c, K=6, N_draws=3, seed=4, want_counts=True)
    assert cnt.sum() == 3 and (conf > 0).sum() == (cnt > 0).sum() and abs(conf.sum() - 1.0) < 1e-12
    e.close()


def test_dist313_resident_suggestions():
    sd = weights.add_pred313_head(weights.make_state_dict(2, "he", include_class=False), 2)
    centres = weights.synthetic_ab_centres(2)
    L, ab, mask = workloads.random_batch(1, 64, seed=8)
    e = engine.HipColorizer(64, 64, max_batch=1, precision="fp32", dist313=True)
    e.load_state_dict(sd)
    _, pred, dist = e.forward_dist313(L, ab, mask, 0.0)
    e.forward_dist313(L, ab, mask, 0.0, want_dist=False)
    with pytest.raises(Exception):
        e.dist_at(0, 0)                                                 # not kept unless asked
    e.keep_dist(True)
    _, pred2, none = e.forward_dist313(L, ab, mask, 0.0, want_dist=False)
    assert none is None
    np.testing.assert_array_equal(pred, pred2)
    np.testing.assert_array_equal(e.get_dist(1), dist)
    for y, x in [(0, 0), (17, 45), (63, 63)]:
        pdf = e.dist_at(y, x)
        np.testing.assert_array_equal(pdf, dist[0, :, y, x])
        cen, conf, cnt = e.suggest_colors(y, x, centres, K=5, N_draws=25000, seed=y * 64 + x, want_counts=True)
        cen_o, conf_o, cnt_o = session.suggest_colors(pdf, centres, K=5, N=25000, seed=y * 64 + x, return_counts=True)
        np.testing.assert_array_equal(cnt, cnt_o)
        np.testing.assert_allclose(cen, cen_o, rtol=0, atol=1e-9)
        np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-12)
    e.close()


def test_api_click_session(sd, capsys):
    """The wrapper classes: net_forward_hints == net_forward on the rasterised planes; the distribution class keeps its
    tensor on the device, suggests colours there, and materialises dist_ab only when read."""
    rgb = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "mortar_pestle_256_rgb.npy"))
    small = rgb[::4, ::4].copy()
    sd1 = dict(sd)
    m = api.ColorizeImageTorch(Xd=64, maskcent=True)
    assert m.net_forward_hints(HINTS_RGB) == -1 and "I need to have an image!" in capsys.readouterr().out
    m.prep_net(path="", state_dict=sd1)
    m.set_image(small)
    out_h = m.net_forward_hints(HINTS_RGB, mode="rgb").copy()
    ab_dev, mask_dev = m.input_ab.copy(), m.input_mask.copy()           # read back lazily from the device planes
    ab_o, mask_o = session.raster_hints(HINTS_RGB, 64, 64, "rgb")
    np.testing.assert_allclose(ab_dev, ab_o, atol=1e-4)
    np.testing.assert_array_equal(mask_dev, mask_o)
    out_p = m.net_forward(ab_dev, mask_dev)
    np.testing.assert_array_equal(out_h, out_p)
    assert out_h.shape == (64, 64, 3) and out_h.dtype == np.uint8

    d = api.ColorizeImageTorchDist(Xd=64, maskcent=True)
    assert d.get_ab_reccs(3, 3) == 0                                    # "Need to set prediction first"
    d.prep_net(path="", state_dict=sd1)
    d.set_image(small)
    ab_ret = d.net_forward(ab_dev, mask_dev)
    assert ab_ret.shape == (2, 64, 64)
    np.random.seed(3)
    cen, conf = d.get_ab_reccs(20, 30, K=5, N=25000, return_conf=True)
    np.random.seed(3)
    cen2 = d.get_ab_reccs(20, 30, K=5, N=25000)
    np.testing.assert_array_equal(cen, cen2)
    assert cen.shape == (5, 2) and abs(conf.sum() - 1) < 1e-12
    pdf = d.dist_ab[:, 20, 30]                                          # materialised now: 529 x 64 x 64, x4 nearest
    assert d.dist_ab.shape == (529, 64, 64) and d.dist_ab_grid.shape == (23, 23, 64, 64)
    np.testing.assert_array_equal(d.dist_ab[:, 20, 30], d.dist_ab[:, 23, 31])
    np.random.seed(3)
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    cen_o, conf_o = session.suggest_colors(pdf, d.pts_in_hull, K=5, N=25000, seed=seed)
    np.testing.assert_array_equal(cen, cen_o)
    np.testing.assert_array_equal(conf, conf_o)
    ab_ret2 = d.net_forward_hints(HINTS_RGB)
    np.testing.assert_array_equal(ab_ret, ab_ret2)
    d.compute_entropy()
    assert d.dist_entropy.shape == (64, 64)

"""Host-side bookkeeping of the wrapper classes (api.py) with a stub in place of the HIP engine: the reference's
print-and-return--1 convention (data/colorize_image.py:85-90), the once-per-image upload of the L plane, and the lazy
read-back of device-resident planes / distributions.  No arithmetic of the path happens here (the stub returns zeros)."""
import numpy as np

from interactive_deep_colorization_amd import api


class StubNet(object):
    """Counts what the wrapper asks of the engine."""

    def __init__(self, X, bins=529):
        self.X, self.bins = X, bins
        self.calls = []

    def _rec(self, name):
        self.calls.append(name)

    def set_image_l(self, L_mc, img=0):
        assert np.asarray(L_mc).size == self.X * self.X
        self._rec("set_image_l")

    def set_hints(self, hints, mode="ab", img=0, mask_value=1.0):
        self._rec("set_hints:%s:%d:%g" % (mode, len(list(hints)), mask_value))

    def hint_planes(self, img=0):
        self._rec("hint_planes")
        return np.full((2, self.X, self.X), 3.0, np.float32), np.ones((1, self.X, self.X), np.float32)

    def forward_resident(self, n=1, maskcent=0.0, l_cent=50.0, want_ab=True, want_rgb=True, want_lab=True):
        self._rec("forward_resident:%g" % maskcent)
        X = self.X
        return (np.zeros((n, 2, X, X), np.float32), np.zeros((n, X, X, 3), np.uint8) if want_rgb else None,
                np.zeros((n, 3, X, X)) if want_rgb and want_lab else None)

    def forward_rgb(self, L_mc, ab, mask, maskcent=0.0, l_cent=50.0, want_lab=True):
        self._rec("forward_rgb")
        X = self.X
        return np.zeros((1, 2, X, X), np.float32), np.zeros((1, X, X, 3), np.uint8), np.zeros((1, 3, X, X))

    def forward_dist(self, L_mc, ab, mask, maskcent=0.0, want_dist=True):
        self._rec("forward_dist:%s" % want_dist)
        return np.zeros((1, 2, self.X, self.X), np.float32), None

    def get_dist(self, n=1):
        self._rec("get_dist")
        d = np.zeros((n, self.bins, self.X // 4, self.X // 4), np.float32)
        d[:, 7] = 1.0
        return d

    def suggest_colors(self, y, x, centres, K=5, N_draws=25000, seed=0, img=0, want_counts=False):
        self._rec("suggest:%d" % seed)
        return np.zeros((K, 2)), np.full(K, 1.0 / K)


def _model(cls, X=16, **kw):
    m = cls(Xd=X, **kw)
    m.net = StubNet(X)
    m.net_set = True
    return m


def test_guards_follow_the_reference_convention(capsys):
    m = api.ColorizeImageTorch(Xd=16)
    assert m.net_forward(np.zeros((2, 16, 16)), np.zeros((1, 16, 16))) == -1
    assert "I need to have an image!" in capsys.readouterr().out
    m.set_image(np.zeros((16, 16, 3), np.uint8))
    assert m.net_forward_hints([]) == -1
    assert "I need to have a net!" in capsys.readouterr().out


def test_edit_list_forward_uploads_l_once_and_reads_planes_lazily():
    m = _model(api.ColorizeImageTorch, maskcent=True)
    m.set_image(np.full((16, 16, 3), 128, np.uint8))
    out = m.net_forward_hints([(1, 1, 3, 3, 255, 0, 0)], mode="rgb")
    assert out.shape == (16, 16, 3)
    m.net_forward_hints([(1, 1, 3, 3, 255, 0, 0), (5, 5, 6, 6, 0, 255, 0)])
    assert m.net.calls == ["set_image_l", "set_hints:rgb:1:1", "forward_resident:0.5", "set_hints:rgb:2:1", "forward_resident:0.5"]
    ab = m.input_ab                                            # first read: one copy-back, then cached
    assert (ab == 3.0).all() and (m.input_mask == 1.0).all()
    assert m.net.calls.count("hint_planes") == 1
    m.set_image(np.full((16, 16, 3), 64, np.uint8))            # a new image: L goes up again
    m.net_forward_hints([])
    assert m.net.calls.count("set_image_l") == 2
    planes_ab, planes_m = np.ones((2, 16, 16)), np.zeros((1, 16, 16))
    m.net_forward(planes_ab, planes_m)                         # plane-valued forward: plain attributes again
    assert m.input_ab is planes_ab and m.input_mask is planes_m and m.net.calls[-1] == "forward_rgb"
    assert m.net.calls.count("hint_planes") == 1


def test_caffe_mask_mult_reaches_the_rasteriser():
    m = _model(api.ColorizeImageCaffe)
    m.set_image(np.zeros((16, 16, 3), np.uint8))
    m.net_forward_hints([(0, 0, 1, 1, 10.0, -10.0)], mode="ab")
    assert "set_hints:ab:1:110" in m.net.calls                  # mask_mult = 110 for the Caffe nets
    assert (m.input_mask == 1.0 / 110).all()                    # stub planes hold 1.0: divided back by mask_mult


def test_distribution_stays_on_the_device_until_read(capsys):
    d = _model(api.ColorizeImageTorchDist, maskcent=False)
    assert d.get_ab_reccs(1, 1) == 0 and "Need to set prediction first" in capsys.readouterr().out
    d.set_image(np.zeros((16, 16, 3), np.uint8))
    ret = d.net_forward(np.zeros((2, 16, 16)), np.zeros((1, 16, 16)))
    assert ret.shape == (2, 16, 16) and d.net.calls == ["forward_dist:False"]
    np.random.seed(1)
    c, conf = d.get_ab_reccs(3, 4, K=4, N=100, return_conf=True)
    assert c.shape == (4, 2) and abs(conf.sum() - 1) < 1e-12 and "get_dist" not in d.net.calls
    np.random.seed(1)
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    assert d.net.calls[-1] == "suggest:%d" % seed               # the draw is tied to numpy's global RNG state
    full = d.dist_ab                                            # now it is copied out and x4 nearest-upsampled
    assert full.shape == (529, 16, 16) and (full[7] == 1).all() and full.sum() == 256
    assert d.dist_ab_grid.shape == (23, 23, 16, 16) and d.net.calls.count("get_dist") == 1
    d.net_forward(np.zeros((2, 16, 16)), np.zeros((1, 16, 16)))
    _ = d.dist_ab_full
    assert d.net.calls.count("get_dist") == 2                   # a new forward invalidates the host copy

"""Seeded weight generator for the oracle (TEST INFRASTRUCTURE ONLY).

The reference ships no trained weights (``models/fetch_models.sh:2-6`` needs the
network), so every parity check runs on seeded random weights.  The generator
uses ``numpy.random.RandomState`` only, so the same seed gives the same bytes
on any box and any torch version.  Key names / layouts are exactly the
reference ``state_dict`` (``models/pytorch/model.py:13-109``):

* ``Conv2d.weight``          (Cout, Cin, kh, kw)
* ``ConvTranspose2d.weight`` (Cin, Cout, 4, 4)      <- note IOHW
* ``BatchNorm2d``            weight, bias, running_mean, running_var,
                             num_batches_tracked

BatchNorm buffers/affine are randomised too: the default init makes eval-BN an
identity and would hide epilogue bugs.
"""
import numpy as np

# (key prefix, kind, Cin, Cout, k)   kind: 'conv' | 'deconv' | 'bn'
LAYER_SPECS = [
    ("model1.0", "conv", 4, 64, 3), ("model1.2", "conv", 64, 64, 3), ("model1.4", "bn", 64, 64, 0),
    ("model2.0", "conv", 64, 128, 3), ("model2.2", "conv", 128, 128, 3), ("model2.4", "bn", 128, 128, 0),
    ("model3.0", "conv", 128, 256, 3), ("model3.2", "conv", 256, 256, 3), ("model3.4", "conv", 256, 256, 3),
    ("model3.6", "bn", 256, 256, 0),
    ("model4.0", "conv", 256, 512, 3), ("model4.2", "conv", 512, 512, 3), ("model4.4", "conv", 512, 512, 3),
    ("model4.6", "bn", 512, 512, 0),
    ("model5.0", "conv", 512, 512, 3), ("model5.2", "conv", 512, 512, 3), ("model5.4", "conv", 512, 512, 3),
    ("model5.6", "bn", 512, 512, 0),
    ("model6.0", "conv", 512, 512, 3), ("model6.2", "conv", 512, 512, 3), ("model6.4", "conv", 512, 512, 3),
    ("model6.6", "bn", 512, 512, 0),
    ("model7.0", "conv", 512, 512, 3), ("model7.2", "conv", 512, 512, 3), ("model7.4", "conv", 512, 512, 3),
    ("model7.6", "bn", 512, 512, 0),
    ("model8up.0", "deconv", 512, 256, 4), ("model3short8.0", "conv", 256, 256, 3),
    ("model8.1", "conv", 256, 256, 3), ("model8.3", "conv", 256, 256, 3), ("model8.5", "bn", 256, 256, 0),
    ("model9up.0", "deconv", 256, 128, 4), ("model2short9.0", "conv", 128, 128, 3),
    ("model9.1", "conv", 128, 128, 3), ("model9.3", "bn", 128, 128, 0),
    ("model10up.0", "deconv", 128, 128, 4), ("model1short10.0", "conv", 64, 128, 3),
    ("model10.1", "conv", 128, 128, 3),
    ("model_class.0", "conv", 256, 529, 1),
    ("model_out.0", "conv", 128, 2, 1),
]


GLOB_SPECS = [  # Global-Hints branch (models/global_model/deploy_nodist.prototxt:37-172); our key names
    ("glob.s_conv1", 2), ("glob.glob_conv1", 314), ("glob.glob_conv2", 512), ("glob.glob_conv3", 512),
    ("glob.glob_conv4", 512),
]


def add_global_branch(sd, seed=0):
    """Add seeded ``glob.*`` tensors (1x1 convs + BatchNorms of the Global-Hints branch) to ``sd`` in place.
    Gains are chosen so that the 512-vector added to conv4_3norm is O(1), i.e. it visibly changes the output."""
    rs = np.random.RandomState(seed + 4242)
    for key, cin in GLOB_SPECS:
        gain = 8.0 if cin == 314 else (1.0 if cin == 2 else np.sqrt(2.0))   # histogram entries are ~1/313
        sd[key + ".weight"] = (rs.standard_normal((512, cin, 1, 1)) * gain / np.sqrt(cin)).astype(np.float32)
        sd[key + ".bias"] = rs.uniform(-0.1, 0.3, 512).astype(np.float32)
    for i in range(1, 5):
        key = "glob.bn%d" % i
        sd[key + ".weight"] = rs.uniform(0.8, 1.2, 512).astype(np.float32)
        sd[key + ".bias"] = rs.uniform(-0.2, 0.2, 512).astype(np.float32)
        sd[key + ".running_mean"] = rs.uniform(0.0, 0.4, 512).astype(np.float32)
        sd[key + ".running_var"] = rs.uniform(0.25, 0.6, 512).astype(np.float32)
    return sd


def synthetic_ab_centres(seed=0):
    """A stand-in for data/color_bins/pts_in_hull.npy (313x2, the in-gamut ab bin centres on a 10-unit grid):
    313 distinct points of the 23x23 grid -110..110, seeded.  (The real table is reference DATA, loaded by the
    wrapper from the reference checkout exactly where the reference loads it, colorize_image.py:388-389.)"""
    rs = np.random.RandomState(seed + 313)
    grid = np.stack(np.meshgrid(np.arange(-110, 120, 10), np.arange(-110, 120, 10), indexing="ij"), -1).reshape(-1, 2)
    r2 = (grid ** 2).sum(1) + rs.uniform(0, 1, grid.shape[0])          # the 313 grid points closest to grey
    return grid[np.sort(np.argsort(r2)[:313])].astype(np.float32)


def add_pred313_head(sd, seed=0, centres=None):
    """Seeded ``pred.*`` tensors of the 313-bin head (models/reference_model/deploy_nopred.prototxt:650-850)."""
    rs = np.random.RandomState(seed + 31337)
    def conv(key, cin, k, deconv=False, gain=1.0):
        fan = cin * (4 if deconv else k * k)
        shape = (cin, 384, k, k) if deconv else (384, cin, k, k)
        sd[key + ".weight"] = (rs.standard_normal(shape) * gain / np.sqrt(fan)).astype(np.float32)
        sd[key + ".bias"] = rs.uniform(-0.1, 0.1, 384).astype(np.float32)
    g = 1.0 / np.sqrt(6.0) * np.sqrt(2.0)                               # six summed branches share the variance
    conv("pred.conv3_pred", 256, 3, gain=g); conv("pred.conv8_pred", 256, 3, gain=g)
    for i in (4, 5, 6, 7):
        conv("pred.conv%d_pred" % i, 512, 4, deconv=True, gain=g)
    sd["pred.pred_313.weight"] = (rs.standard_normal((313, 384, 1, 1)) * 6.0 / np.sqrt(384)).astype(np.float32)
    sd["pred.pred_313.bias"] = rs.uniform(-0.5, 0.5, 313).astype(np.float32)
    c = synthetic_ab_centres(seed) if centres is None else np.asarray(centres, np.float32)
    sd["pred.pred_ab.weight"] = np.ascontiguousarray(c.T)[:, :, None, None].astype(np.float32)   # (2,313,1,1)
    sd["pred.pred_ab.bias"] = rs.uniform(-1, 1, 2).astype(np.float32)
    return sd


def make_state_dict(seed=0, style="he", include_class=True):
    """Return ``{key: np.ndarray}`` with the reference key set.

    style='he'     : N(0, gain/sqrt(fan_in)) conv weights so activations stay
                     O(1) through all 30 layers and the tanh head is exercised
                     over its whole range (the hard case for parity).
    style='torch'  : U(+-1/sqrt(fan_in)) like ``torch.nn`` default init
                     (activations shrink; |out| stays small).
    """
    rs = np.random.RandomState(seed)
    sd = {}
    for key, kind, cin, cout, k in LAYER_SPECS:
        if key.startswith("model_class") and not include_class:
            continue
        if kind == "bn":
            c = cout
            sd[key + ".weight"] = rs.uniform(0.8, 1.2, c).astype(np.float32)
            sd[key + ".bias"] = rs.uniform(-0.2, 0.2, c).astype(np.float32)
            sd[key + ".running_mean"] = rs.uniform(0.2, 0.6, c).astype(np.float32)
            sd[key + ".running_var"] = rs.uniform(0.25, 0.6, c).astype(np.float32)
            sd[key + ".num_batches_tracked"] = np.array(1, dtype=np.int64)
            continue
        if kind == "deconv":
            # every output pixel sees 2x2 taps of Cin channels
            fan_in = cin * 4
            shape = (cin, cout, k, k)
        else:
            fan_in = cin * k * k
            shape = (cout, cin, k, k)
        if style == "he":
            gain = np.sqrt(2.0)
            if key in ("model8up.0", "model3short8.0", "model9up.0", "model2short9.0",
                       "model10up.0", "model1short10.0"):
                gain = 1.0          # the two summed branches share the variance
            if key == "model_out.0":
                gain = 0.6          # keep most pre-tanh logits inside +-2
            if key == "model_class.0":
                gain = 4.0          # make the 0.2-tempered softmax non-flat
            w = rs.standard_normal(shape) * (gain / np.sqrt(fan_in))
            b = rs.uniform(-0.1, 0.1, cout)
        elif style == "torch":
            bound = 1.0 / np.sqrt(fan_in)
            w = rs.uniform(-bound, bound, shape)
            b = rs.uniform(-bound, bound, cout)
        else:
            raise ValueError("unknown style %r" % style)
        sd[key + ".weight"] = w.astype(np.float32)
        sd[key + ".bias"] = b.astype(np.float32)
    return sd


def param_count(sd, include_class=True):
    n = 0
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if not include_class and k.startswith("model_class"):
            continue
        n += int(np.prod(v.shape))
    return n

"""Seeded weights for the parity tests (TEST INFRASTRUCTURE ONLY): the trunk generator is re-exported from the product
package's synthetic workloads, the Caffe-only branches' generators live here.

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

# The generator itself lives with the synthetic workloads of the product package (bench.py draws its random-init weights
# from it and must not import the oracle); the oracle and the tests use the very same function under its old name.
from interactive_deep_colorization_amd.workloads import LAYER_SPECS, param_count  # noqa: F401,E402
from interactive_deep_colorization_amd.workloads import random_state_dict as make_state_dict  # noqa: F401,E402


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

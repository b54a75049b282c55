"""Synthetic inputs for the BASELINE.json configs.

Pure numpy; shared by ``bench.py`` and the tests so both see the same bytes.
``put_point`` restates the notebook helper
(``DemoInteractiveColorization.ipynb:131-139``): it is the only "hint
rasteriser" the reference ships outside the Qt GUI.
"""
import numpy as np


def put_point(input_ab, mask, loc, p, val):
    """Write a (2p+1)x(2p+1) patch of colour ``val=(a,b)`` at ``loc=(h,w)``.

    In place, returns the same arrays, like the notebook helper.  One deliberate difference: a
    patch overlapping the top/left border is clipped to the image; the notebook's negative
    slice start silently drops such a point.
    """
    h0, w0 = max(loc[0] - p, 0), max(loc[1] - p, 0)
    input_ab[:, h0:loc[0] + p + 1, w0:loc[1] + p + 1] = np.array(val)[:, np.newaxis, np.newaxis]
    mask[:, h0:loc[0] + p + 1, w0:loc[1] + p + 1] = 1
    return (input_ab, mask)


def hints_config2(X=256, n_points=5, p=3, seed=0):
    """Config 2: ``n_points`` hints, 7x7 patches; the two notebook points first
    (``DemoInteractiveColorization.ipynb:181-184,225-228``), the rest from
    ``RandomState(seed)``: (h,w) ~ U{p..X-1-p}, (a,b) ~ U(-80,80)."""
    rs = np.random.RandomState(seed)
    input_ab = np.zeros((2, X, X))
    mask = np.zeros((1, X, X))
    pts = []
    if X == 256:
        pts += [((135, 160), (23., -69.)), ((100, 160), (0., 0.))]
    while len(pts) < n_points:
        loc = (int(rs.randint(p, X - p)), int(rs.randint(p, X - p)))
        val = (float(rs.uniform(-80, 80)), float(rs.uniform(-80, 80)))
        pts.append((loc, val))
    for loc, val in pts[:n_points]:
        put_point(input_ab, mask, loc, p, val)
    return input_ab, mask


def random_batch(N, H=256, W=None, seed=0, max_points=20, max_p=4, start=0):
    """Configs 3/4: N random L channels with random sparse hint patches.

    L_mc ~ U(-50,50) f32; per image k ~ U{0..max_points} patches with
    p ~ U{0..max_p} at uniform locations, ab ~ U(-80,80) constant per patch.
    Returns f32 arrays L_mc (N,1,H,W), ab (N,2,H,W), mask (N,1,H,W).
    Image ``i`` of the job draws from its own ``RandomState(seed*100003 + i)``, so a shard
    (``start`` = its first global index) is bit-identical to the same images of the full batch.
    """
    W = H if W is None else W
    L = np.empty((N, 1, H, W), np.float32)
    ab = np.zeros((N, 2, H, W), np.float32)
    mask = np.zeros((N, 1, H, W), np.float32)
    for i in range(N):
        rs = np.random.RandomState((seed * 100003 + start + i) % (2 ** 31 - 1))
        L[i, 0] = rs.uniform(-50, 50, (H, W)).astype(np.float32)
        k = int(rs.randint(0, max_points + 1))
        for _ in range(k):
            p = int(rs.randint(0, max_p + 1))
            h = int(rs.randint(0, H))
            w = int(rs.randint(0, W))
            val = rs.uniform(-80, 80, 2).astype(np.float32)
            h0, h1 = max(h - p, 0), min(h + p + 1, H)
            w0, w1 = max(w - p, 0), min(w + p + 1, W)
            ab[i, :, h0:h1, w0:w1] = val[:, None, None]
            mask[i, :, h0:h1, w0:w1] = 1.0
    return L, ab, mask


def global_hint_config5(N, seed=0):
    """Config 5 global input: normalised U(0,1) 313-bin histogram + flag 1;
    saturation pair left zero, as ``data/colorize_image.py:452-459`` leaves it."""
    rs = np.random.RandomState(seed + 7)
    hist = rs.uniform(0, 1, (N, 313)).astype(np.float64)
    hist /= hist.sum(axis=1, keepdims=True)
    glob = np.zeros((N, 314), np.float32)
    glob[:, :313] = hist
    glob[:, 313] = 1.0
    sat = np.zeros((N, 2), np.float32)
    return glob, sat


def shard_bounds(n_items, world_size, rank):
    """Contiguous split of ``n_items`` over ``world_size`` ranks:
    the first ``n_items % world_size`` ranks get one extra item."""
    base, rem = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
# Seeded random-init weights of the reference architecture (the reference ships no trained weights:
# ``models/fetch_models.sh:2-6`` needs the network).  ``numpy.random.RandomState`` only, so a seed gives the same bytes
# on any box and any torch version.  Key names / layouts are exactly the reference ``state_dict`` (SURVEY.md Appendix B;
# ``models/pytorch/model.py:13-109``): Conv2d.weight (Cout, Cin, kh, kw); ConvTranspose2d.weight (Cin, Cout, 4, 4);
# BatchNorm2d weight, bias, running_mean, running_var, num_batches_tracked -- randomised too (the default init makes
# eval-BN an identity and would hide epilogue bugs).  bench.py's synthetic weights and the parity tests' weights
# (``oracle.weights.make_state_dict`` is this function) come from here.
# ---------------------------------------------------------------------------------------------------------------
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


def random_state_dict(seed=0, style="he", include_class=True):
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

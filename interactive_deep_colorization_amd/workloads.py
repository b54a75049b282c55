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

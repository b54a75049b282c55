"""CPU restatement of the per-click host work around the network (TEST INFRASTRUCTURE ONLY).

* ``raster_hints``  -- ``UIControl.get_input`` (``ui/ui_control.py:177-187`` with ``PointEdit.updateInput``
  ``:52-63``: filled ``cv2.rectangle`` with inclusive corners on a black uint8 canvas and a uint8 mask, edits
  applied in list order) followed by ``rgb2lab`` of the canvas and ``mask > 0`` as in
  ``ui/gui_draw.py:273-277``; or, for ab-valued hints, the notebook's ``put_point``
  (``DemoInteractiveColorization.ipynb:131-139``).
* ``get_ab_reccs_reference`` -- ``ColorizeImageTorchDist.get_ab_reccs`` (``data/colorize_image.py:322-354``) as
  written: numpy global-RNG draws + sklearn ``KMeans`` (random seeding).  Not reproducible run to run.
* ``suggest_colors`` -- the deterministic form of the same computation that the device kernel implements
  (``include/ideepcolor.h: idc_suggest_colors``): identical cmf / digitize step, counter-based draws, k-means
  on the drawn colours with greedy k-means++ seeding.  Bit-exact restatement of the kernel for bin centres with
  integer coordinates (all sums are then exact in float64).

PARITY UNPINNED AGAINST cv2 for the rasteriser (cv2 is absent here; ``cv2.rectangle`` semantics -- inclusive corners, either
corner order, clipping -- are restated from its documentation; since round 6 the restatement is held against PIL's
``ImageDraw.rectangle``, an independent filled-rectangle rasteriser with the same inclusive / clipped / last-wins convention:
``tests/test_session_cpu.py``) and statistical only for the suggestions (the
reference is stochastic): ``tests/test_session_cpu.py`` checks that ``suggest_colors`` and
``get_ab_reccs_reference`` agree on well-separated mixtures.
"""
import numpy as np

from . import colorspace


def raster_hints(hints, H, W, mode="ab", mask_value=1.0):
    """hints: rows (y0, x0, y1, x1, c0, c1[, c2]) -> (ab (2,H,W) float32, mask (1,H,W) float32)."""
    mask = np.zeros((H, W), np.uint8)
    if mode == "rgb":
        canvas = np.zeros((H, W, 3), np.uint8)
    else:
        canvas = np.zeros((H, W, 2), np.float32)
    for r in hints:
        y0, y1 = sorted((int(r[0]), int(r[2])))
        x0, x1 = sorted((int(r[1]), int(r[3])))
        y0, x0, y1, x1 = max(y0, 0), max(x0, 0), min(y1, H - 1), min(x1, W - 1)
        if y0 > y1 or x0 > x1:
            continue
        mask[y0:y1 + 1, x0:x1 + 1] = 255
        if mode == "rgb":
            canvas[y0:y1 + 1, x0:x1 + 1] = np.array(r[4:7], np.uint8)
        else:
            canvas[y0:y1 + 1, x0:x1 + 1] = np.array(r[4:6], np.float32)
    if mode == "rgb":
        # rgb2lab only where painted: the black background is Lab (0,0,0) exactly
        ab = np.zeros((H, W, 2), np.float64)
        ys, xs = np.nonzero(mask)
        if len(ys):
            cols, inv = np.unique(canvas[ys, xs], axis=0, return_inverse=True)
            lab = colorspace.rgb2lab(cols[None])[0]
            ab[ys, xs] = lab[inv.reshape(-1), 1:]
        ab = ab.astype(np.float32)
    else:
        ab = canvas
    return ab.transpose(2, 0, 1).copy(), ((mask > 0).astype(np.float32) * np.float32(mask_value))[None]


def lowbias32(x):
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def draws(N, seed):
    """u_i in [0,1): 24-bit uniforms from the counter-based generator of idc_suggest_colors."""
    i = np.arange(N, dtype=np.uint64)
    x = (i * 0x9E3779B9 + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x85EBCA6B) + 0x165667B1) & 0xFFFFFFFF
    return (lowbias32(x) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def draw_counts(pdf, N, seed):
    """Step 1+2: cmf as the reference builds it (fp32 cumsum / last), np.digitize of the draws, counts per bin."""
    pdf = np.asarray(pdf, np.float32)
    cmf = np.cumsum(pdf, dtype=np.float32)
    cmf = cmf / cmf[-1]
    idx = np.digitize(draws(N, seed), bins=cmf)
    idx = np.minimum(idx, len(pdf) - 1)
    return np.bincount(idx, minlength=len(pdf)).astype(np.uint32)


def suggest_colors(pdf, centres, K=5, N=25000, seed=0, return_counts=False):
    """-> (centres (K,2) f64, conf (K,) f64) ordered by cluster occupancy, descending."""
    cnt = draw_counts(pdf, N, seed).astype(np.float64)
    pts = np.asarray(centres, np.float32).astype(np.float64)
    B = len(cnt)
    means = np.zeros((K, 2))
    for k in range(K):                                   # greedy k-means++ seeding
        if k == 0:
            d2 = np.ones(B)
        else:
            d = pts[:, None, :] - means[None, :k, :]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).min(axis=1)
        means[k] = pts[int(np.argmax(cnt * d2))]
    asg = -np.ones(B, np.int64)
    occ = np.zeros(K)
    for _ in range(100):                                 # Lloyd to a fixed point
        d = pts[:, None, :] - means[None, :, :]
        new = np.argmin(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1], axis=1)      # ties -> lowest index
        changed = bool(np.any((new != asg) & (cnt > 0)))
        asg = new
        for k in range(K):
            sel = (asg == k) & (cnt > 0)
            w = cnt[sel].sum()
            occ[k] = w
            if w > 0:
                means[k] = [np.sum(cnt[sel] * pts[sel, 0]) / w, np.sum(cnt[sel] * pts[sel, 1]) / w]
        if not changed:
            break
    order = sorted(range(K), key=lambda k: (-occ[k], k))
    out = (means[order], occ[order] / float(N))
    return out + (cnt.astype(np.uint32),) if return_counts else out


def get_ab_reccs_reference(pdf, pts_in_hull, K=5, N=25000, rng=None):
    """colorize_image.py:322-354 as written (sklearn KMeans on N random draws)."""
    from sklearn.cluster import KMeans
    rng = np.random if rng is None else rng
    cmf = np.cumsum(pdf)
    cmf = cmf / cmf[-1]
    rnd_pts = rng.uniform(low=0, high=1.0, size=N)
    inds = np.digitize(rnd_pts, bins=cmf)
    rnd_pts_ab = np.asarray(pts_in_hull)[inds, :]
    kmeans = KMeans(n_clusters=K, n_init=10).fit(rnd_pts_ab)
    k_label_cnt = np.histogram(kmeans.labels_, np.arange(0, K + 1))[0]
    k_inds = np.argsort(k_label_cnt, axis=0)[::-1]
    return kmeans.cluster_centers_[k_inds, :], 1. * k_label_cnt[k_inds] / N

"""CPU restatement of the display / full-resolution step that follows every ``net_forward`` -- TEST INFRASTRUCTURE ONLY
(imported by tests/; the product path never touches it).

* ``resize_cubic_cv2``: ``cv2.resize(src, (ow, oh), interpolation=cv2.INTER_CUBIC)`` on float64 data as OpenCV's
  ``resize.cpp`` computes it (``ui/gui_draw.py:281``): half-pixel centres, ``fx = (float)((dx + .5) * scale - .5)``,
  four taps ``sx-1 .. sx+2`` clamped to the image (replicated border), float32 Keys coefficients with A = -0.75
  (``interpolateCubic``), horizontal pass then vertical pass, sums in double.  cv2 is not installable here, so this follows
  the published algorithm, not a run of cv2 itself (**unpinned against cv2**); since round 6 it is held against an INDEPENDENT
  library implementation of the same algorithm -- ``torch.nn.functional.interpolate(mode="bicubic", align_corners=False)``:
  Keys kernel at A = -0.75, half-pixel centres, clamped indices -- to rounding at exact phases and to 2e-5 of the data range
  elsewhere (OpenCV's float32 coefficient arithmetic, which this file keeps): ``tests/test_round2_cpu.py``.
* ``zoom_linear`` / ``zoom_nearest``: ``scipy.ndimage.zoom(x, (1, fh, fw), order=1 | 0)`` of
  ``data/colorize_image.py:123-158`` -- **pinned**: scipy is present, the tests compare these with scipy itself.
* ``display_rgb``: the four lines of ``GUIDraw.compute_result`` (``ui/gui_draw.py:280-283``).
"""
import numpy as np

from . import colorspace


def _cubic_coeffs(x):
    A = np.float32(-0.75)
    x = x.astype(np.float32)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def _cubic_axis(n_in, n_out):
    scale = np.float64(n_in) / np.float64(n_out)
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, n_in - 1)
    return idx, _cubic_coeffs(f)


def resize_cubic_cv2(src, oh, ow):
    """src (H, W) or (H, W, C) float64 -> (oh, ow[, C])."""
    src = np.asarray(src, np.float64)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    H, W, C = src.shape
    xi, xc = _cubic_axis(W, ow)
    yi, yc = _cubic_axis(H, oh)
    xc = xc.astype(np.float64); yc = yc.astype(np.float64)
    # horizontal pass on every source row: v = S[x0]*a0 + S[x1]*a1 + S[x2]*a2 + S[x3]*a3 (left to right)
    rows = np.zeros((H, ow, C))
    for j in range(4):
        rows = rows + src[:, xi[:, j], :] * xc[None, :, j, None]
    out = np.zeros((oh, ow, C))
    for k in range(4):
        out = out + rows[yi[:, k]] * yc[:, k, None, None]
    return out[:, :, 0] if squeeze else out


def zoom_linear(x, oh, ow):
    """(C, H, W) -> (C, oh, ow) like scipy.ndimage.zoom(x, (1, oh/H, ow/W), order=1)."""
    x = np.asarray(x, np.float64)
    C, H, W = x.shape
    cy = np.arange(oh) * (np.float64(H - 1) / np.float64(oh - 1) if oh > 1 else 0.0)
    cx = np.arange(ow) * (np.float64(W - 1) / np.float64(ow - 1) if ow > 1 else 0.0)
    y0 = np.floor(cy).astype(np.int64); x0 = np.floor(cx).astype(np.int64)
    ty = cy - y0; tx = cx - x0
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    v00 = x[:, y0][:, :, x0]; v01 = x[:, y0][:, :, x1]; v10 = x[:, y1][:, :, x0]; v11 = x[:, y1][:, :, x1]
    wy0 = (1.0 - ty)[None, :, None]; wy1 = ty[None, :, None]; wx0 = (1.0 - tx)[None, None, :]; wx1 = tx[None, None, :]
    return v00 * (wy0 * wx0) + v01 * (wy0 * wx1) + v10 * (wy1 * wx0) + v11 * (wy1 * wx1)


def zoom_nearest(x, oh, ow):
    x = np.asarray(x)
    C, H, W = x.shape
    cy = np.arange(oh) * (np.float64(H - 1) / np.float64(oh - 1) if oh > 1 else 0.0)
    cx = np.arange(ow) * (np.float64(W - 1) / np.float64(ow - 1) if ow > 1 else 0.0)
    yi = np.minimum(np.floor(cy + 0.5).astype(np.int64), H - 1)
    xi = np.minimum(np.floor(cx + 0.5).astype(np.int64), W - 1)
    return x[:, yi][:, :, xi]


def display_rgb(output_ab, l_win):
    """ui/gui_draw.py:280-283: output_ab (2, X, X) float64, l_win (win_h, win_w) -> (win_h, win_w, 3) uint8."""
    win_h, win_w = l_win.shape
    ab = np.asarray(output_ab, np.float64).transpose((1, 2, 0))
    ab_win = resize_cubic_cv2(ab, win_h, win_w)
    pred_lab = np.concatenate((np.asarray(l_win, np.float64)[..., None], ab_win), axis=2)
    return (np.clip(colorspace.lab2rgb(pred_lab), 0, 1) * 255).astype('uint8')

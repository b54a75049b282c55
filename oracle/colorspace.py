"""float64 restatement of ``skimage.color.rgb2lab`` / ``lab2rgb`` (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: scikit-image is not installable here and is not vendored by
the reference (``README.md:117`` names ``scikit-image=0.13.0``).  The functions
follow the published formulas used by skimage (sRGB / D65 / 2-degree observer,
SURVEY.md Appendix E) and are checked against textbook known answers (sRGB
primaries, white, mid-grey) in ``tests/test_colorspace.py``.  Call sites in the
reference: ``data/colorize_image.py:27,36,172,178``.

Written per-pixel-formula style on purpose (clarity over speed): the product's
host code (``interactive_deep_colorization_amd/colorspace.py``) is a separate
vectorised implementation that the tests compare against this one.
"""
import numpy as np

# skimage.color.colorconv.xyz_from_rgb (sRGB -> XYZ, D65)
XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                         [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]])
RGB_FROM_XYZ = np.linalg.inv(XYZ_FROM_RGB)
WHITE_D65 = np.array([0.95047, 1.0, 1.08883])


def rgb2lab(rgb):
    """rgb HxWx3 uint8 (or float in [0,1]) -> Lab HxWx3 float64."""
    arr = np.asarray(rgb)
    if arr.dtype == np.uint8:
        arr = arr.astype(np.float64) / 255.0
    else:
        arr = arr.astype(np.float64)
    out = np.empty(arr.shape, np.float64)
    flat = arr.reshape(-1, 3)
    res = out.reshape(-1, 3)
    for i in range(flat.shape[0]):
        c = flat[i]
        lin = np.where(c > 0.04045, ((c + 0.055) / 1.055) ** 2.4, c / 12.92)
        xyz = XYZ_FROM_RGB.dot(lin) / WHITE_D65
        f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
        res[i, 0] = 116.0 * f[1] - 16.0
        res[i, 1] = 500.0 * (f[0] - f[1])
        res[i, 2] = 200.0 * (f[1] - f[2])
    return out


def lab2rgb(lab):
    """Lab HxWx3 float -> rgb HxWx3 float64 in [0,1] (clipped like skimage's xyz2rgb)."""
    arr = np.asarray(lab, np.float64)
    out = np.empty(arr.shape, np.float64)
    flat = arr.reshape(-1, 3)
    res = out.reshape(-1, 3)
    for i in range(flat.shape[0]):
        L, a, b = flat[i]
        fy = (L + 16.0) / 116.0
        fx = a / 500.0 + fy
        fz = fy - b / 200.0
        if fz < 0:
            fz = 0.0                      # skimage clamps negative z
        f = np.array([fx, fy, fz])
        xyz = np.where(f > 0.2068966, f ** 3, (f - 16.0 / 116.0) / 7.787) * WHITE_D65
        lin = RGB_FROM_XYZ.dot(xyz)
        srgb = np.where(lin > 0.0031308, 1.055 * np.power(np.maximum(lin, 0), 1 / 2.4) - 0.055, 12.92 * lin)
        res[i] = np.clip(srgb, 0.0, 1.0)
    return out


def lab2rgb_transpose(img_l, img_ab):
    """``data/colorize_image.py:20-28``: 1xXxX + 2xXxX -> XxXx3 uint8."""
    pred_lab = np.concatenate((img_l, img_ab), axis=0).transpose((1, 2, 0))
    return (np.clip(lab2rgb(pred_lab), 0, 1) * 255).astype("uint8")


def rgb2lab_transpose(img_rgb):
    """``data/colorize_image.py:31-36``: XxXx3 -> 3xXxX."""
    return rgb2lab(img_rgb).transpose((2, 0, 1))


def global_stats(rgb_u8, centres):
    """``models/global_model/global_stats.prototxt`` restated (PARITY UNPINNED: Caffe Python layers, not runnable
    here): rgb2lab -> 4x4 average pool of ab (``:101-111``) -> 1-nearest-neighbour hard assignment to the 313
    centres (``NNEncLayer`` with NN = 1, ``caffe_traininglayers.py:161-196`` / ``color_quantization.py:7-33``)
    -> global mean (``:224-233``); and the global mean of the HSV saturation (``:10-21,123-141``).
    (X, Y, 3) uint8 -> (hist (313,), s_avg)."""
    lab = rgb2lab(rgb_u8)
    X, Y = lab.shape[:2]
    ab = lab[..., 1:].astype(np.float32).reshape(X // 4, 4, Y // 4, 4, 2).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)
    c = np.asarray(centres, np.float32)
    d = ((ab[:, :, None, :] - c[None, None]) ** 2).sum(-1)
    idx = d.argmin(-1)
    hist = np.bincount(idx.ravel(), minlength=313).astype(np.float64) / idx.size
    v = np.asarray(rgb_u8, np.float64) / 255.0
    mx, mn = v.max(-1), v.min(-1)
    sat = np.where(mx > 0, (mx - mn) / np.where(mx > 0, mx, 1), 0.0)
    return hist.astype(np.float32), float(sat.mean())

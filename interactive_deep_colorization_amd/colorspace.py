"""Host-side colour maths and image IO for the drop-in API (vectorised numpy).

The reference does this with scikit-image / OpenCV / scipy
(``data/colorize_image.py:20-36,52-66,123-158``); neither skimage nor cv2 exists
in this image, so the formulas are restated.  This is
host plumbing around the HIP path, not part of it; it runs once per image load
(RGB->Lab) and once per ``net_forward`` (Lab->RGB of the 256x256 result).

sRGB, D65 white, 2-degree observer -- the constants skimage uses.
"""
import numpy as np

_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                          [0.212671, 0.715160, 0.072169],
                          [0.019334, 0.119193, 0.950227]], dtype=np.float64)
_RGB_FROM_XYZ = np.linalg.inv(_XYZ_FROM_RGB)
_WHITE = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)


def _as_float_rgb(rgb):
    arr = np.asarray(rgb)
    if arr.dtype == np.uint8:
        return arr.astype(np.float64) / 255.0
    return arr.astype(np.float64)


def rgb2lab(rgb):
    """``skimage.color.rgb2lab``: (...,3) uint8 or float[0,1] -> (...,3) float64 Lab."""
    arr = _as_float_rgb(rgb)
    lin = np.where(arr > 0.04045, np.power((arr + 0.055) / 1.055, 2.4), arr / 12.92)
    xyz = lin @ _XYZ_FROM_RGB.T
    xyz = xyz / _WHITE
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    return np.stack((116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)), axis=-1)


def lab2rgb(lab):
    """``skimage.color.lab2rgb``: (...,3) Lab -> (...,3) float64 sRGB clipped to [0,1]."""
    lab = np.asarray(lab, dtype=np.float64)
    fy = (lab[..., 0] + 16.0) / 116.0
    fx = lab[..., 1] / 500.0 + fy
    fz = np.maximum(fy - lab[..., 2] / 200.0, 0.0)          # skimage zeroes negative z
    f = np.stack((fx, fy, fz), axis=-1)
    xyz = np.where(f > 0.2068966, f ** 3, (f - 16.0 / 116.0) / 7.787) * _WHITE
    lin = xyz @ _RGB_FROM_XYZ.T
    srgb = np.where(lin > 0.0031308, 1.055 * np.power(np.maximum(lin, 0.0), 1.0 / 2.4) - 0.055, 12.92 * lin)
    return np.clip(srgb, 0.0, 1.0)


def lab2rgb_transpose(img_l, img_ab):
    """``data/colorize_image.py:20-28``: (1,X,X) L + (2,X,X) ab -> (X,X,3) uint8."""
    pred_lab = np.concatenate((img_l, img_ab), axis=0).transpose((1, 2, 0))
    return (np.clip(lab2rgb(pred_lab), 0, 1) * 255).astype("uint8")


def rgb2lab_transpose(img_rgb):
    """``data/colorize_image.py:31-36``: (X,X,3) -> (3,X,X)."""
    return rgb2lab(img_rgb).transpose((2, 0, 1))


def resize_bilinear_u8(img, out_h, out_w):
    """Bilinear resize with half-pixel centres and edge clamp, no antialiasing --
    the sampling rule of ``cv2.resize(im, (Xd, Xd))`` (INTER_LINEAR) used at
    ``data/colorize_image.py:58``.  Float arithmetic, round-half-up; OpenCV's
    11-bit fixed-point path can differ by one grey level."""
    img = np.asarray(img)
    in_h, in_w = img.shape[:2]
    ys = (np.arange(out_h) + 0.5) * (in_h / float(out_h)) - 0.5
    xs = (np.arange(out_w) + 0.5) * (in_w / float(out_w)) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    wy = (ys - y0)[:, None, None]; wx = (xs - x0)[None, :, None]
    y0c = np.clip(y0, 0, in_h - 1); y1c = np.clip(y0 + 1, 0, in_h - 1)
    x0c = np.clip(x0, 0, in_w - 1); x1c = np.clip(x0 + 1, 0, in_w - 1)
    f = img.astype(np.float64)
    if f.ndim == 2:
        f = f[:, :, None]
    top = f[y0c][:, x0c] * (1 - wx) + f[y0c][:, x1c] * wx
    bot = f[y1c][:, x0c] * (1 - wx) + f[y1c][:, x1c] * wx
    out = top * (1 - wy) + bot * wy
    out = np.floor(out + 0.5).clip(0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def imread_rgb(path):
    """``cv2.cvtColor(cv2.imread(path, 1), cv2.COLOR_BGR2RGB)`` -> (H,W,3) uint8 RGB."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB")).copy()

"""Golden sRGB <-> CIE Lab pairs from an INDEPENDENT evaluation (TEST INFRASTRUCTURE ONLY).

`oracle/colorspace.py` and the product's `colorspace.py` are both numpy float64 code, so agreeing with each other pins
neither.  This script evaluates the same published formulas -- IEC 61966-2-1 sRGB transfer function, CIE 1976 L*a*b* with
the constants scikit-image uses (`skimage/color/colorconv.py`: `xyz_from_rgb` to six decimals, its matrix INVERSE for the way
back, D65 2-degree white 0.95047 / 1 / 1.08883, thresholds 0.04045 / 0.0031308 / 0.008856 / 0.2068966, slope 7.787) -- in
50-digit `mpmath` arithmetic, with the 3x3 inverse taken exactly over the rationals, and stores the results rounded to
float64.  Call sites of the reference: data/colorize_image.py:20-36 (lab2rgb_transpose / rgb2lab_transpose).

    python oracle/make_golden_colorspace.py      ->  tests/golden/colorspace_pairs.npz   (1200 + 1200 pairs, ~60 KB)

skimage itself is not installable here (PARITY UNPINNED against the library); what this pins is the arithmetic of the two
restatements against the formulas, at 1e-9 instead of the five published colours at 0.02 of rounds 1-4.
"""
import os
from fractions import Fraction

import mpmath as mp
import numpy as np

mp.mp.dps = 50
M_RAT = [[Fraction(s) for s in row] for row in (("0.412453", "0.357580", "0.180423"),
                                                ("0.212671", "0.715160", "0.072169"),
                                                ("0.019334", "0.119193", "0.950227"))]
WHITE = [mp.mpf("0.95047"), mp.mpf(1), mp.mpf("1.08883")]


def _inv3(m):
    (a, b, c), (d, e, f), (g, h, i) = m
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    adj = [[e * i - f * h, c * h - b * i, b * f - c * e],
           [f * g - d * i, a * i - c * g, c * d - a * f],
           [d * h - e * g, b * g - a * h, a * e - b * d]]
    return [[x / det for x in row] for row in adj]


def _mpf(fr):
    return mp.mpf(fr.numerator) / mp.mpf(fr.denominator)


M = [[_mpf(x) for x in row] for row in M_RAT]
M_INV = [[_mpf(x) for x in row] for row in _inv3(M_RAT)]


def rgb2lab_exact(c01):
    """three mpf in [0, 1] -> (L, a, b) mpf"""
    lin = [((c + mp.mpf("0.055")) / mp.mpf("1.055")) ** mp.mpf("2.4") if c > mp.mpf("0.04045") else c / mp.mpf("12.92") for c in c01]
    xyz = [sum(M[r][k] * lin[k] for k in range(3)) / WHITE[r] for r in range(3)]
    f = [mp.cbrt(v) if v > mp.mpf("0.008856") else mp.mpf("7.787") * v + mp.mpf(16) / 116 for v in xyz]
    return 116 * f[1] - 16, 500 * (f[0] - f[1]), 200 * (f[1] - f[2])


def lab2rgb_exact(L, a, b):
    """(L, a, b) mpf -> sRGB in [0, 1] (clipped), skimage's negative-z clamp included"""
    fy = (L + 16) / 116
    fx = a / 500 + fy
    fz = fy - b / 200
    if fz < 0:
        fz = mp.mpf(0)
    f = [fx, fy, fz]
    xyz = [(v ** 3 if v > mp.mpf("0.2068966") else (v - mp.mpf(16) / 116) / mp.mpf("7.787")) * WHITE[k] for k, v in enumerate(f)]
    lin = [sum(M_INV[r][k] * xyz[k] for k in range(3)) for r in range(3)]
    out = []
    for v in lin:
        s = mp.mpf("1.055") * (v ** (1 / mp.mpf("2.4"))) - mp.mpf("0.055") if v > mp.mpf("0.0031308") else mp.mpf("12.92") * v
        out.append(min(max(s, mp.mpf(0)), mp.mpf(1)))
    return out


def main():
    rs = np.random.RandomState(20260922)
    # ---- sRGB -> Lab: 1000 random uint8 triples + the cube corners + values either side of the 0.04045 knee (10/255, 11/255) + 180 float triples
    u8 = np.concatenate((rs.randint(0, 256, (1000, 3)),
                         np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)]),
                         np.array([[10, 10, 10], [11, 11, 11], [10, 11, 200], [1, 0, 0], [0, 1, 0], [0, 0, 1], [128, 128, 128],
                                   [254, 255, 253], [11, 10, 0], [3, 200, 9], [90, 91, 10], [255, 11, 10]]))).astype(np.uint8)
    lab_of_u8 = np.array([[float(v) for v in rgb2lab_exact([mp.mpf(int(c)) / 255 for c in px])] for px in u8])
    rgbf = rs.uniform(0, 1, (180, 3))
    lab_of_f = np.array([[float(v) for v in rgb2lab_exact([mp.mpf(float(c)) for c in px])] for px in rgbf])
    # ---- Lab -> sRGB: in-gamut (from the pairs above, perturbed), out-of-gamut, negative-z, near-black
    lab_in = np.concatenate((
        lab_of_u8[:500] + rs.uniform(-0.3, 0.3, (500, 3)),
        np.stack((rs.uniform(0, 100, 500), rs.uniform(-110, 110, 500), rs.uniform(-110, 110, 500)), axis=1),      # the net's output range
        np.stack((rs.uniform(0, 12, 100), rs.uniform(-20, 20, 100), rs.uniform(-20, 120, 100)), axis=1),          # linear segment / z < 0
        np.array([[50.0, 120.0, -120.0], [5.0, 0.0, 100.0], [99.0, -110.0, 110.0], [0.0, 0.0, 0.0], [100.0, 0.0, 0.0]]),
        np.stack((rs.uniform(0, 100, 95), np.zeros(95), np.zeros(95)), axis=1)))                                   # the grey axis
    # values whose clipped result sits within 1e-7 of a branch knee would make 1e-9 a statement about the knee, not the formula
    rgb_of_lab = np.array([[float(v) for v in lab2rgb_exact(*[mp.mpf(float(c)) for c in px])] for px in lab_in])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "colorspace_pairs.npz")
    np.savez_compressed(out, rgb_u8=u8, lab_of_rgb_u8=lab_of_u8, rgb_f64=rgbf, lab_of_rgb_f64=lab_of_f, lab_in=lab_in, rgb_of_lab=rgb_of_lab,
                        digits=np.int64(mp.mp.dps))
    print("wrote", out, u8.shape, rgbf.shape, lab_in.shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

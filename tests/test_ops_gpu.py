"""GPU parity of the single operators (same kernels the network launches) through the C ABI.

Reference semantics: torch.nn.Conv2d / ConvTranspose2d / BatchNorm2d(eval) as used in
models/pytorch/model.py:13-109.  The oracle here is torch fp64 on the CPU (a floating-point
kernel keeps a plain torch reference); tolerances are stated per precision:
  fp32 path: |err| <= 2e-5 * (1 + max|ref|)      (exact-fp32 MFMA, only summation order differs)
  bf16 path: |err| <= 2.5e-2 * (1 + max|ref|)    (bf16 operands, fp32 accumulation)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from interactive_deep_colorization_amd import engine

pytestmark = pytest.mark.gpu
TOL = {"fp32": 2e-5, "bf16": 2.5e-2}
# every kernel variant is driven at the small test sizes: fp32 = conv_igemm<f32>, bf16 = conv_igemm<bf16>
# (small tiles), bf16-large = conv_igemm_v2 (32x32x16 MFMA, LDS-transposed stores) forced by the tile policy
PRECISIONS = ["fp32", "bf16", "bf16-large", "fp32-splitk", "bf16-splitk"]


@pytest.fixture(autouse=True)
def _reset_tile_policy():
    yield
    engine.set_tile_policy("auto")
    engine.set_splitk_policy("auto")


def _prec(precision):
    """'bf16-large' -> ('bf16', tile policy 'large'); 'x-splitk' -> small tiles with split-K forced (every cin
    chunk its own slice + splitk_epilogue); sets the policies for the coming op call."""
    engine.set_splitk_policy("never")
    if precision == "bf16-large":
        engine.set_tile_policy("large")
        return "bf16"
    engine.set_tile_policy("small")
    if precision.endswith("-splitk"):
        engine.set_splitk_policy("always")
        return precision.split("-")[0]
    return precision


def _ref_conv(x, w, b, dilation, in_stride, act, bn_s, bn_t, resid):
    xt = torch.from_numpy(x).double()[:, :, ::in_stride, ::in_stride]
    y = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=dilation * (w.shape[2] // 2),
                 dilation=dilation)
    if resid is not None:
        y = y + torch.from_numpy(resid).double()
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, 0.2)
    if bn_s is not None:
        y = y * torch.from_numpy(bn_s).double()[None, :, None, None] + torch.from_numpy(bn_t).double()[None, :, None, None]
    return y.numpy()


def _check(got, ref, precision, what):
    err = np.abs(got - ref).max()
    tol = TOL[precision] * (1 + np.abs(ref).max())
    assert np.isfinite(got).all(), what
    assert err <= tol, "%s: max-abs err %.3e > tol %.3e (max|ref| %.2f)" % (what, err, tol, np.abs(ref).max())


CONV_CASES = [
    # n, cin, cout, h,  w,  dil, stride, act, bn,   resid
    (1, 64, 64, 16, 16, 1, 1, 1, True, False),     # conv1_2 shape class (cout 64 -> one cout group)
    (2, 64, 128, 32, 48, 1, 2, 1, False, False),   # conv2_1: reads x[::2, ::2]
    (1, 128, 128, 24, 40, 1, 1, 0, False, False),  # shortcut conv: no activation; ragged tile edges
    (1, 128, 256, 8, 8, 2, 1, 1, True, False),     # dilated (model5/6) + BN after ReLU
    (3, 64, 128, 20, 36, 1, 1, 2, False, True),    # LeakyReLU + residual
    (1, 256, 128, 4, 4, 2, 1, 1, False, False),    # image smaller than the dilation halo
    (1, 64, 128, 8, 8, 1, 1, 1, True, False),
]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(case, precision):
    precision = _prec(precision)
    n, cin, cout, h, w, dil, stride, act, bn, use_res = case
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, cout).astype(np.float32)
    bn_s = rs.uniform(0.5, 2.0, cout).astype(np.float32) if bn else None
    bn_t = rs.uniform(-1, 1, cout).astype(np.float32) if bn else None
    resid = rs.standard_normal((n, cout, h // stride, w // stride)).astype(np.float32) if use_res else None
    got = engine.op_conv2d(x, wt, b, dilation=dil, in_stride=stride, act=act, bn_scale=bn_s, bn_shift=bn_t,
                           resid=resid, precision=precision)
    ref = _ref_conv(x, wt, b, dil, stride, act, bn_s, bn_t, resid)
    _check(got, ref, precision, "conv3x3 %s" % (case,))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[9]] + [(1, 128, 128, 16, 16, 1, 1, 1, False, True), (2, 256, 128, 8, 8, 1, 1, 0, True, True)])
def test_conv3x3_with_residual_under_the_default_policies(case, precision):
    """with the AUTO tile policy a small bf16 3x3 op with 1 / 2 / 4 / 8 cin chunks is a conv_kwave_bf16 candidate; that
    kernel takes no shortcut sum, and the single-op path used to decide before it knew about the residual (IDC_ERR_INTERNAL)."""
    engine.set_tile_policy("auto")
    engine.set_splitk_policy("auto")
    n, cin, cout, h, w, dil, stride, act, bn, _ = case
    rs = np.random.RandomState(7 + cin + cout + h)
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, cout).astype(np.float32)
    bn_s = rs.uniform(0.5, 2.0, cout).astype(np.float32) if bn else None
    bn_t = rs.uniform(-1, 1, cout).astype(np.float32) if bn else None
    resid = rs.standard_normal((n, cout, h // stride, w // stride)).astype(np.float32)
    got = engine.op_conv2d(x, wt, b, dilation=dil, in_stride=stride, act=act, bn_scale=bn_s, bn_shift=bn_t, resid=resid, precision=precision)
    _check(got, _ref_conv(x, wt, b, dil, stride, act, bn_s, bn_t, resid), precision, "conv3x3 + resid, auto policy %s" % (case,))
    got0 = engine.op_conv2d(x, wt, b, dilation=dil, in_stride=stride, act=act, bn_scale=bn_s, bn_shift=bn_t, resid=None, precision=precision)
    _check(got0, _ref_conv(x, wt, b, dil, stride, act, bn_s, bn_t, None), precision, "conv3x3, auto policy %s" % (case,))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv_is_transpose_detecting(precision):
    """Asymmetric single-tap weights: an (ky,kx) or (cin,cout) transpose cannot pass."""
    precision = _prec(precision)
    cin, cout, h, w = 64, 128, 8, 16
    x = np.zeros((1, cin, h, w), np.float32); x[0, 3, 2, 5] = 1.0
    wt = np.zeros((cout, cin, 3, 3), np.float32); wt[7, 3, 0, 2] = 2.0      # only tap (ky=0, kx=2)
    got = engine.op_conv2d(x, wt, np.zeros(cout, np.float32), precision=precision)
    exp = np.zeros((1, cout, h, w), np.float32); exp[0, 7, 3, 4] = 2.0      # oy = 2-(0-1), ox = 5-(2-1)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("shape", [(1, 128, 128, 8, 8), (2, 256, 128, 6, 20), (1, 512, 256, 4, 4)])
def test_deconv4x4s2(shape, precision):
    precision = _prec(precision)
    n, cin, cout, h, w = shape
    rs = np.random.RandomState(cin + cout + h)
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rs.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, cout).astype(np.float32)
    resid = rs.standard_normal((n, cout, 2 * h, 2 * w)).astype(np.float32)
    got = engine.op_deconv4x4s2(x, wt, b, act=1, resid=resid, precision=precision)
    ref = F.relu(F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                                    torch.from_numpy(b).double(), stride=2, padding=1) + torch.from_numpy(resid).double()).numpy()
    assert got.shape == (n, cout, 2 * h, 2 * w)
    _check(got, ref, precision, "deconv %s" % (shape,))
    got0 = engine.op_deconv4x4s2(x, wt, b, act=0, resid=None, precision=precision)
    ref0 = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(),
                              stride=2, padding=1).numpy()
    _check(got0, ref0, precision, "deconv (no act) %s" % (shape,))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv1x1_padded_cout(precision):
    """model_class shape class: Cout = 529 is padded to 640 inside; only 529 come back."""
    precision = _prec(precision)
    rs = np.random.RandomState(5)
    x = rs.standard_normal((1, 256, 8, 16)).astype(np.float32)
    wt = (rs.standard_normal((529, 256, 1, 1)) / 16).astype(np.float32)
    b = rs.uniform(-1, 1, 529).astype(np.float32)
    got = engine.op_conv2d(x, wt, b, precision=precision)
    ref = _ref_conv(x, wt, b, 1, 1, 0, None, None, None)
    assert got.shape == (1, 529, 8, 16)
    _check(got, ref, precision, "conv1x1 529")


def test_linearity_property_full_size():
    """Size-independent property at a BASELINE-sized layer (512->512 @32x32, N=4): conv(a*x) - a*conv(x)
    vanishes with zero bias (fp32 path)."""
    rs = np.random.RandomState(11)
    x = rs.standard_normal((4, 512, 32, 32)).astype(np.float32)
    wt = (rs.standard_normal((512, 512, 3, 3)) / np.sqrt(512 * 9)).astype(np.float32)
    z = np.zeros(512, np.float32)
    y1 = engine.op_conv2d(x, wt, z, dilation=2, precision="fp32")
    y2 = engine.op_conv2d(2.0 * x, wt, z, dilation=2, precision="fp32")
    np.testing.assert_array_equal(y2, 2.0 * y1)          # scaling by 2 is exact in binary fp
    assert np.abs(y1).max() > 1.0

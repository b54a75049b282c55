"""Round 6, GPU: the operand-split precisions (IDC_BF16X3 / IDC_BF16X6) -- the fp32 contract of
/root/reference/data/colorize_image.py:263 (ab map within 1e-3 of the reference's) carried on the bf16 matrix pipe.

An fp32 operand travels as hi + lo (bf16x3: three bf16 MFMA products per fp32 product) or hi + mid + lo (bf16x6: six products,
all 24 mantissa bits); products are accumulated in fp32; bias / BN / shortcut sums / tanh head stay fp32.  Oracle: torch float64
on the CPU (single ops) and oracle/siggraph_torch.py in float64 (whole network, models/pytorch/model.py:148-175).

Tolerances (stated here, used below):
  single ops    bf16x6: 2e-5 * (1 + max|ref|)   = the exact-fp32 kernels' bound (tests/test_ops_gpu.py)
                bf16x3: 1e-4 * (1 + max|ref|)   (operands rounded to 16 mantissa bits: 2^-17 relative per operand)
  whole network bf16x6: tests/bounds.py FP32_TOL (1e-3 torch-init, 3e-3 he-style) -- the fp32 path's bounds
                bf16x3: 1e-3 on torch-init weights (the north_star figure); he-style weights are outside its contract (5e-2 asserted)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from interactive_deep_colorization_amd import engine, workloads
from tests import bounds

pytestmark = pytest.mark.gpu
OP_TOL = {"bf16x6": 2e-5, "bf16x3": 1e-4, "fp16x3": 2e-5}      # fp16x3 (fp16 parts, 2^-23 relative per operand): the exact-fp32 kernels' bound


def _ref_conv(x, w, b, dilation, in_stride, act, bn_s, bn_t, resid):
    xt = torch.from_numpy(x).double()[:, :, ::in_stride, ::in_stride]
    y = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=dilation * (w.shape[2] // 2), dilation=dilation)
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
    tol = OP_TOL[precision] * (1 + np.abs(ref).max())
    assert np.isfinite(got).all(), what
    assert err <= tol, "%s: max-abs err %.3e > tol %.3e (max|ref| %.2f)" % (what, err, tol, np.abs(ref).max())
    return err


CONV_CASES = [
    # n, cin, cout, h,  w,  k, dil, stride, act, bn,   resid
    (2, 64, 128, 32, 48, 3, 1, 2, 1, False, False),    # conv2_1: reads x[::2, ::2]; <2,*> tile
    (1, 128, 128, 24, 40, 3, 1, 1, 0, False, False),   # shortcut conv: no activation; ragged tile edges
    (1, 128, 256, 8, 8, 3, 2, 1, 1, True, False),      # dilated (model5/6) + BN after ReLU; <4,2> tile
    (3, 64, 128, 20, 36, 3, 1, 1, 2, False, True),     # LeakyReLU + fp32 shortcut sum
    (1, 256, 128, 4, 4, 3, 2, 1, 1, False, False),     # image smaller than the dilation halo
    (1, 512, 512, 16, 16, 3, 1, 1, 1, True, False),    # trunk shape: 8 chunks x 3 / 6 segments
    (2, 256, 640, 8, 8, 1, 1, 1, 0, False, False),     # 1x1 (class logits shape class: 529 -> 640 padded couts)
]


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_split_conv(case, precision):
    n, cin, cout, h, w, k, dil, stride, act, bn, has_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    bn_s = (0.5 + rng.random(cout)).astype(np.float32) if bn else None
    bn_t = rng.standard_normal(cout).astype(np.float32) * 0.1 if bn else None
    resid = rng.standard_normal((n, cout, h // stride, w // stride)).astype(np.float32) if has_res else None
    got = engine.op_conv2d(x, wt, b, dilation=dil, in_stride=stride, act=act, bn_scale=bn_s, bn_shift=bn_t, resid=resid, precision=precision)
    ref = _ref_conv(x, wt, b, dil, stride, act, bn_s, bn_t, resid)
    _check(got, ref, precision, "conv %s %s" % (case, precision))


DECONV_CASES = [(1, 512, 256, 8, 8, True), (2, 256, 128, 12, 20, True), (1, 128, 128, 16, 16, False), (1, 512, 384, 8, 8, True)]


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
@pytest.mark.parametrize("case", DECONV_CASES)
def test_split_deconv(case, precision):
    n, cin, cout, h, w, has_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    resid = rng.standard_normal((n, cout, 2 * h, 2 * w)).astype(np.float32) if has_res else None
    got = engine.op_deconv4x4s2(x, wt, b, act=1, resid=resid, precision=precision)
    y = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=2, padding=1)
    if resid is not None:
        y = y + torch.from_numpy(resid).double()
    ref = F.relu(y).numpy()
    _check(got, ref, precision, "deconv %s %s" % (case, precision))


def _net_case(size, n, style, precision, maskcent=0.5, **kw):
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(0, style)
    L, ab, m = workloads.random_batch(n, size, seed=3, max_points=6, max_p=3)
    e = engine.HipColorizer(size, size, max_batch=n, precision=precision, **kw)
    try:
        e.load_state_dict(sd)
        out = e.forward(L, ab, m, maskcent)
        again = e.forward(L, ab, m, maskcent)
    finally:
        e.close()
    ref = siggraph_torch.forward(sd, L, ab, m, maskcent, dtype=torch.float64)
    return out, again, ref


@pytest.mark.parametrize("style", ["torch", "he"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
def test_split_network_small(precision, style):
    """64x64, batch 2 (ragged grids for every large tile), against the float64 oracle; run-to-run identical."""
    out, again, ref = _net_case(64, 2, style, precision)
    assert np.array_equal(out, again)
    err = float(np.abs(out - ref).max())
    tol = bounds.FP32_TOL[style] if precision in ("bf16x6", "fp16x3") else (1e-3 if style == "torch" else 5e-2)
    assert err <= tol, "%s %s: max-abs %.3e > %.1e" % (precision, style, err, tol)


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
def test_split_network_layer_by_layer(precision, fuse):
    """Every materialised activation of a 64x64 forward against the float64 oracle's (he-style weights: full dynamic range).
    fuse = 1 (default): the deconv + shortcut pairs are conv_ds_fused_ms launches and the shortcut sums are never stored; fuse = 0: two launches each."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(0, "he")
    L, ab, m = workloads.random_batch(2, 64, seed=5, max_points=6, max_p=3)
    ref_out, _, acts = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64, return_acts=True)
    names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3",
             "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv7_3", "conv3_3_short", "conv8_1", "conv8_2", "conv8_3", "conv2_2_short", "conv9_1",
             "conv9_2", "conv1_2_short", "conv10_1"]            # (conv10_2 is consumed by the tanh head inside its own launch: never stored)
    if fuse:
        names = [n_ for n_ in names if not n_.endswith("_short")]
    engine.set_option("split_ds_fuse", fuse)
    e = engine.HipColorizer(64, 64, max_batch=2, precision=precision)
    try:
        e.load_state_dict(sd)
        out = e.forward(L, ab, m, 0.0)
        rel = 2e-4 if precision in ("bf16x6", "fp16x3") else 2e-3
        for name in names:
            got, ref = e.activation(name, 2), acts[name]
            assert got.shape == ref.shape, name
            err = float(np.abs(got - ref).max())
            assert err <= rel * (1 + np.abs(ref).max()), "%s %s: %.3e (max|ref| %.2f)" % (precision, name, err, np.abs(ref).max())
        with pytest.raises(Exception):
            e.activation("conv10_2", 2)
        assert np.abs(out - ref_out).max() <= (bounds.FP32_TOL["he"] if precision in ("bf16x6", "fp16x3") else 5e-2)
        kernels = {r["name"]: r["kernel"] for r in e.layer_table()}
        assert ("conv_ds_fused_ms" in kernels["conv10_1"]) == bool(fuse), kernels["conv10_1"]
    finally:
        e.close()
        engine.set_option("split_ds_fuse", 1)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
def test_split_ds_fused_against_two_launches(precision):
    """conv_ds_fused_ms / _msh (deconv 4x4 s2 + the 3x3 shortcut conv in one launch, models/pytorch/model.py:156,170,172) on a ragged size (72 x 104: partial
    64 x 8 output tiles on both axes), batch 3: within the fp32 bounds of the float64 oracle, and as close to it as the two-launch form (same fp32 sums in
    another order)."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(1, "torch")
    H, W = 72, 104
    rs = np.random.RandomState(11)
    L = (rs.rand(3, 1, H, W) * 100).astype(np.float32)
    ab = np.zeros((3, 2, H, W), np.float32); m = np.zeros((3, 1, H, W), np.float32)
    ab[:, :, 10:16, 20:26] = rs.uniform(-60, 60, (3, 2, 1, 1)); m[:, :, 10:16, 20:26] = 1
    ref = siggraph_torch.forward(sd, L, ab, m, 0.5, dtype=torch.float64)
    outs = {}
    try:
        for fuse in (1, 0):
            engine.set_option("split_ds_fuse", fuse)
            e = engine.HipColorizer(H, W, max_batch=3, precision=precision)
            try:
                e.load_state_dict(sd)
                outs[fuse] = e.forward(L, ab, m, 0.5)
                assert np.array_equal(outs[fuse], e.forward(L, ab, m, 0.5))
            finally:
                e.close()
    finally:
        engine.set_option("split_ds_fuse", 1)
    tol = 1e-3
    e1, e0 = float(np.abs(outs[1] - ref).max()), float(np.abs(outs[0] - ref).max())
    assert e1 <= tol and e0 <= tol, (precision, e1, e0)
    assert e1 <= 2.0 * e0 + 1e-5, (precision, e1, e0)


def test_split_dist_head_and_global_hints_build():
    """The operand-split handle also runs the 529-bin head (fp32 logits from a split 1x1 launch) -- models/pytorch/model.py:159-160."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(0, "torch")
    L, ab, m = workloads.random_batch(1, 64, seed=9, max_points=4, max_p=2)
    e = engine.HipColorizer(64, 64, max_batch=1, precision="bf16x6", dist=True)
    try:
        e.load_state_dict(sd)
        out, dist = e.forward_dist(L, ab, m, 0.0)
    finally:
        e.close()
    ref_out, ref_dist = siggraph_torch.forward(sd, L, ab, m, 0.0, dist=True, dtype=torch.float64)
    assert np.abs(out - ref_out).max() <= 1e-3
    assert np.abs(dist - ref_dist[:, :, ::4, ::4]).max() <= 1e-5


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3"])
def test_conv1_1_split_kernel_against_the_generic_island(precision):
    """conv1_1 of an operand-split handle (exact-fp32 island, models/pytorch/model.py:13,139-148) on conv1_1_split_kernel (>= 128 tiles of 32 x 16: here
    batch 8 of 72 x 104, ragged on both axes) against the float64 oracle's conv1_1 and against conv_igemm<float> (option conv1_1_split = 0): both are fp32
    sums of the same 36 products -- 1e-5 relative -- and the network outputs agree to the fp32 contract."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(2, "he")
    H, W, n = 72, 104, 8
    rs = np.random.RandomState(5)
    L = (rs.rand(n, 1, H, W) * 100).astype(np.float32)
    ab = (rs.uniform(-80, 80, (n, 2, H, W)) * (rs.rand(n, 1, H, W) < 0.05)).astype(np.float32)
    m = (np.abs(ab).sum(1, keepdims=True) > 0).astype(np.float32)
    ref_out, _, acts = siggraph_torch.forward(sd, L, ab, m, 0.5, dtype=torch.float64, return_acts=True)
    got, outs = {}, {}
    try:
        for v in (1, 0):
            engine.set_option("conv1_1_split", v)
            e = engine.HipColorizer(H, W, max_batch=n, precision=precision)
            try:
                e.load_state_dict(sd)
                outs[v] = e.forward(L, ab, m, 0.5)
                got[v] = e.activation("conv1_1", n)
                kernels = {r["name"]: r["kernel"] for r in e.layer_table()}
                assert ("conv1_1_split_kernel" in kernels["conv1_1"]) == bool(v), kernels["conv1_1"]
            finally:
                e.close()
    finally:
        engine.set_option("conv1_1_split", 1)
    ref = acts["conv1_1"]
    scale = 1 + np.abs(ref).max()
    parts_tol = 2e-4 if precision == "bf16x3" else 1e-5          # the stored planes carry 16 (bf16x3) / 22-24 mantissa bits
    for v in (1, 0):
        assert np.abs(got[v] - ref).max() <= parts_tol * scale, (precision, v, float(np.abs(got[v] - ref).max()), scale)
    assert np.abs(got[1] - got[0]).max() <= parts_tol * scale
    # he-style weights (full tanh range): bf16x6 inside the fp32 bounds; fp16x3 measured 3.9e-3 at N = 32, bf16x3 outside its contract
    tol = {"bf16x6": bounds.FP32_TOL["he"], "fp16x3": 8e-3, "bf16x3": 5e-2}[precision]
    e1, e0 = float(np.abs(outs[1] - ref_out).max()), float(np.abs(outs[0] - ref_out).max())
    assert e1 <= tol and e0 <= tol and e1 <= 1.5 * e0 + 1e-5, (precision, e1, e0)


@pytest.mark.parametrize("style", ["torch", "he"])
def test_fp16_precision_is_the_split_machinery_with_one_part(style):
    """precision="fp16" (IDC_FP16): plain fp16 operands (11 significant bits against bf16's 8), fp32 accumulation, conv1_1 exact fp32 -- the operand-split
    graph with one part and one segment, its launches on the fp16 twins of the bf16 throughput kernels (conv_igemm_v2ph, conv_ds_fused_mh; option fp16_fast,
    0 = the one-segment split kernels).  Against the float64 oracle it must sit well inside the bf16 path's bounds (measured ~1 / 8 of bf16's error), run
    to run identical, and the two kernel sets must agree to fp16's own rounding."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(0, style)
    L, ab, m = workloads.random_batch(2, 64, seed=3, max_points=6, max_p=3)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.5, dtype=torch.float64)
    errs, outs = {}, {}
    try:
        for precision, fast in (("fp16", 1), ("fp16", 0), ("bf16", 1)):
            engine.set_option("fp16_fast", fast)
            e = engine.HipColorizer(64, 64, max_batch=2, precision=precision)
            try:
                e.load_state_dict(sd)
                out = e.forward(L, ab, m, 0.5)
                assert np.array_equal(out, e.forward(L, ab, m, 0.5))
                kernels = set(r["kernel"].split("<")[0].split("+")[0].split(" ")[0] for r in e.layer_table() if r["launches"] > 0 and r["kernel"].startswith("conv"))
                if precision == "fp16" and fast:
                    assert {"conv_igemm_v2ph", "conv_ds_fused_mh"} <= kernels <= {"conv_igemm_v2ph", "conv_ds_fused_mh", "conv_igemm_v2psh", "conv_igemm_v2sh",
                                                                                  "conv1_1_split_kernel", "conv1_2_split_kernel", "conv_igemm", "conv1_block_fused"}, kernels
                elif precision == "fp16":
                    assert kernels <= {"conv_igemm_v2psh", "conv_igemm_v2sh", "conv_ds_fused_msh", "conv1_1_split_kernel", "conv1_2_split_kernel", "conv_igemm"}, kernels
            finally:
                e.close()
            errs[(precision, fast)] = float(np.abs(out - ref).max())
            outs[(precision, fast)] = out
    finally:
        engine.set_option("fp16_fast", 1)
    for key in (("fp16", 1), ("fp16", 0)):
        assert errs[key] <= (0.05 if style == "torch" else 4.0), errs
        assert errs[key] <= 0.4 * errs[("bf16", 1)], errs
    assert np.abs(outs[("fp16", 1)] - outs[("fp16", 0)]).max() <= 2.0 * max(errs[("fp16", 1)], errs[("fp16", 0)])


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x6", "fp16x3", "fp16"])
def test_conv1_2_split_kernel_against_the_generic_tile(precision):
    """conv1_2 (64 -> 64 at full resolution, ReLU + eval-BN, models/pytorch/model.py:15-17) of the operand-split precisions on conv1_2_split_kernel (>= 256
    tiles of 32 x 12: batch 8 of 104 x 104, ragged on both axes) against conv_igemm_v2ps<1,4,1> (option conv1_2_split = 0) and the float64 oracle's conv1_2."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(3, "he")
    H, W, n = 104, 104, 8                     # (multiples of 8 for the net; 104 / 12 and 104 / 32 leave partial tiles)
    rs = np.random.RandomState(6)
    L = (rs.rand(n, 1, H, W) * 100).astype(np.float32)
    ab = (rs.uniform(-80, 80, (n, 2, H, W)) * (rs.rand(n, 1, H, W) < 0.05)).astype(np.float32)
    m = (np.abs(ab).sum(1, keepdims=True) > 0).astype(np.float32)
    ref_out, _, acts = siggraph_torch.forward(sd, L, ab, m, 0.5, dtype=torch.float64, return_acts=True)
    got, outs = {}, {}
    try:
        engine.set_option("fp16_fast", 0)            # (precision "fp16": model1 would otherwise run conv1_block_fused_th, conv1_2 inside it)
        for v in (1, 0):
            engine.set_option("conv1_2_split", v)
            e = engine.HipColorizer(H, W, max_batch=n, precision=precision)
            try:
                e.load_state_dict(sd)
                outs[v] = e.forward(L, ab, m, 0.5)
                assert np.array_equal(outs[v], e.forward(L, ab, m, 0.5))
                got[v] = e.activation("conv1_2", n)
                kernels = {r["name"]: r["kernel"] for r in e.layer_table()}
                assert ("conv1_2_split_kernel" in kernels["conv1_2"]) == bool(v), kernels["conv1_2"]
            finally:
                e.close()
    finally:
        engine.set_option("conv1_2_split", 1)
        engine.set_option("fp16_fast", 1)
    ref = acts["conv1_2"]
    scale = 1 + np.abs(ref).max()
    rel = {"bf16x3": 2e-3, "bf16x6": 2e-4, "fp16x3": 2e-4, "fp16": 4e-3}[precision]      # (test_split_network_layer_by_layer's per-layer bounds; fp16: one 11-bit part)
    for v in (1, 0):
        assert np.abs(got[v] - ref).max() <= rel * scale, (precision, v, float(np.abs(got[v] - ref).max()), scale)
    e1, e0 = float(np.abs(outs[1] - ref_out).max()), float(np.abs(outs[0] - ref_out).max())
    assert e1 <= 1.5 * e0 + 1e-5, (precision, e1, e0)


def test_fp16_conv1_block_twin():
    """IDC_FP16 at throughput size (>= 128 tiles: batch 16 of 64 x 64 gives 128 tiles of 32 x 8): model1 runs conv1_block_fused_th -- conv1_1 on ITS fp16 block of
    the blob, conv1_2's fp16 image, fp16 tile in LDS -- against fp16_fast = 0 (conv1_1's exact-fp32 island + conv1_2 on the split kernels) and the float64 oracle."""
    from oracle import siggraph_torch
    from tests.conftest import state_dict_for
    sd = state_dict_for(4, "he")
    n = 16
    L, ab, m = workloads.random_batch(n, 64, seed=8, max_points=5, max_p=3)
    ref_out, _, acts = siggraph_torch.forward(sd, L[:4], ab[:4], m[:4], 0.5, dtype=torch.float64, return_acts=True)
    got, outs = {}, {}
    try:
        for fast in (1, 0):
            engine.set_option("fp16_fast", fast)
            e = engine.HipColorizer(64, 64, max_batch=n, precision="fp16")
            try:
                e.load_state_dict(sd)
                outs[fast] = e.forward(L, ab, m, 0.5)
                got[fast] = e.activation("conv1_2", 4)
                kernels = {r["name"]: r["kernel"] for r in e.layer_table()}
                assert ("conv1_block_fused" in kernels["conv1_1"]) == bool(fast), kernels["conv1_1"]
            finally:
                e.close()
    finally:
        engine.set_option("fp16_fast", 1)
    ref = acts["conv1_2"]
    scale = 1 + np.abs(ref).max()
    for fast in (1, 0):
        assert np.abs(got[fast] - ref).max() <= 4e-3 * scale, (fast, float(np.abs(got[fast] - ref).max()), scale)
    e1, e0 = float(np.abs(outs[1][:4] - ref_out).max()), float(np.abs(outs[0][:4] - ref_out).max())
    assert e1 <= 4.0 and e0 <= 4.0 and e1 <= 2.0 * e0 + 1e-3, (e1, e0)


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x6"])
def test_split_precisions_through_the_caffe_only_heads(precision):
    """The operand-split handles also carry the Caffe-only branches (models/reference_model/deploy_nopred.prototxt:650-850 -- the 313-bin hyper-column head and
    its soft decode -- and models/global_model/deploy_nodist.prototxt:37-172,501-518 -- Global Hints): the ab map within 1e-3 of the float64 oracle, and the three
    outputs of forward_dist313 (decoded ab at two temperatures, the 313-bin distribution) within 1e-3 of the fp32 engine's."""
    from oracle import siggraph_torch, weights
    L, ab, m = workloads.random_batch(2, 64, seed=3, max_points=5, max_p=3)
    sd = weights.add_pred313_head(weights.add_global_branch(weights.make_state_dict(1, "torch"), 5), 7)
    g, s = workloads.global_hint_config5(2, seed=2)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64, glob=g, sat=s)
    res = {}
    for prec in ("fp32", precision):
        e = engine.HipColorizer(64, 64, max_batch=2, precision=prec, global_hints=True, dist313=True)
        try:
            e.load_state_dict(sd)
            e.set_global_hints(g, s)
            res[prec] = (e.forward(L, ab, m, 0.0), e.forward_dist313(L, ab, m, 0.0))
        finally:
            e.close()
    assert np.abs(res[precision][0] - ref).max() <= 1e-3
    for x32, xs in zip(res["fp32"][1], res[precision][1]):
        assert np.isfinite(xs).all() and np.abs(np.asarray(xs) - np.asarray(x32)).max() <= 1e-3 * (1 + np.abs(x32).max())

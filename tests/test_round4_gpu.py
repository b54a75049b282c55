"""Round-4 GPU tests: the deconv + shortcut launches and model1 in both MFMA shapes, against the float64 oracle per layer, the
reference's golden ab maps and each other.  Everything goes through the C ABI (ctypes)."""
import numpy as np
import pytest

import torch
import torch.nn.functional as F

from interactive_deep_colorization_amd import engine, workloads
from oracle import siggraph_torch

from bounds import FP32_TOL, bf16_bound, check_bf16_ab  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_options():
    yield
    for k, v in (("mfma16", 1), ("ds_mfma16", 1), ("v2p", 1), ("fuse_conv1", 1), ("kwave", 1), ("kwave_chain", 2)):
        try:
            engine.set_option(k, v)
        except Exception:
            pass
    engine.set_tile_policy("auto")


DS_LAYERS = ["conv8_1", "conv9_1", "conv10_1"]


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2", "net64_torch_s1_mc0"])
def test_deconv_shortcut_launch_in_both_mfma_shapes(golden, make_sd, name):
    """conv_ds_fused_m (16x16x32 MFMA, default) and conv_ds_fused (32x32x16, `ds_mfma16` = 0): model8up + model3short8, model9up +
    model2short9, model10up + model1short10 (model.py:156,170,172) of the large-tile bf16 path -- each launch's output against the
    float64 oracle at the bf16 per-layer tolerance (4 % of the layer's range), the ab map inside the bf16 bounds of the reference's
    golden output, and the two shapes within bf16 rounding of each other.  32x48 = ragged tiles (site grids of 4x6, 8x12, 16x24)."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    sd = make_sd(seed, style)
    _, _, acts = siggraph_torch.forward(sd, g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]), return_acts=True, dtype=torch.float64)
    outs, layers = {}, {}
    names = {0: "conv_ds_fused+shortcut", 1: "conv_ds_fused_m+shortcut"}
    from conftest import has_ab_partners
    shapes = (1, 0) if has_ab_partners() else (1,)            # conv_ds_fused (32x32x16 MFMA) exists in the -DIDC_AB_PARTNERS build only
    for shape in shapes:
        engine.set_tile_policy("large")
        engine.set_option("ds_mfma16", shape)
        e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
        e.load_state_dict(sd)
        outs[shape] = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
        table = {r["name"]: r["kernel"] for r in e.layer_table()}
        assert [table[k] for k in DS_LAYERS] == [names[shape]] * 3, table
        layers[shape] = {k: e.activation(k, n) for k in DS_LAYERS}
        np.testing.assert_array_equal(e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])), outs[shape])   # deterministic
        e.close()
    for shape in shapes:
        for k in DS_LAYERS:
            err = np.abs(layers[shape][k] - acts[k]).max()
            assert err <= 0.04 * (1 + np.abs(acts[k]).max()), "layer %s (ds_mfma16=%d): max-abs err %.3e" % (k, shape, err)
        check_bf16_ab(outs[shape] - g["out_ab"], style, tag="ds_mfma16=%d" % shape)
    if len(shapes) < 2:
        return
    for k in DS_LAYERS:                                       # same sums in a different order: a bf16 ulp here and there
        a, b = layers[1][k], layers[0][k]
        assert np.abs(a - b).max() <= 2.0 ** -6 * (1 + np.abs(b).max()), k
    assert np.abs(outs[1] - outs[0]).mean() <= bf16_bound(style)[1] / 2


def test_deconv_shortcut_m16_batch_is_per_image(make_sd):
    """A batch through conv_ds_fused_m equals its images run alone, bit for bit (no op mixes images; variant chosen per handle)."""
    engine.set_tile_policy("large")
    L, ab, m = workloads.random_batch(3, 64, seed=11)
    e = engine.HipColorizer(64, 64, max_batch=3, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    full = e.forward(L, ab, m, 0.0)
    assert any(r["kernel"].startswith("conv_ds_fused_m") for r in e.layer_table())
    for i in range(3):
        np.testing.assert_array_equal(e.forward(L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.0)[0], full[i])
    e.close()


V2_LAYERS = ["conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3",
             "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv7_3", "conv8_2", "conv8_3", "conv9_2"]


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2", "net64_torch_s1_mc0"])
def test_throughput_3x3_tile_without_address_arithmetic(golden, make_sd, name):
    """conv_igemm_v2p (padded halo rows, nine unrolled taps, buffer loads; default for the 3x3 convs of the large-tile bf16 path:
    dilation 1 and 2, the stride-2 first-of-block reads, conv10_2's fused head) issues conv_igemm_v2m's MFMAs in conv_igemm_v2m's
    order on the same operands: every layer and the ab map BIT-identical to `v2p` = 0, both inside the bf16 bounds of the golden."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    sd = make_sd(seed, style)
    outs, layers = {}, {}
    for v2p in (1, 0):
        engine.set_tile_policy("large")
        engine.set_option("v2p", v2p)
        e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
        e.load_state_dict(sd)
        outs[v2p] = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
        table = {r["name"]: r["kernel"] for r in e.layer_table()}
        # the 256-cout x (32 x 8) and the 4-wave 128-cout tiles have the form (the 8-wave 128-cout tile of tiny grids keeps conv_igemm_v2m)
        for k in V2_LAYERS + ["conv10_2"]:
            assert table[k].startswith("conv_igemm_v2<"), (k, table[k])
            if table[k].startswith("conv_igemm_v2<4,2>") or table[k].startswith("conv_igemm_v2<2,2>"):
                assert ("+m16p" in table[k]) == bool(v2p), (k, table[k])
        assert sum("+m16p" in table[k] for k in V2_LAYERS) >= (11 if v2p else 0), table
        layers[v2p] = {k: e.activation(k, n) for k in V2_LAYERS}
        e.close()
    engine.set_option("v2p", 1)
    for k in V2_LAYERS:
        np.testing.assert_array_equal(layers[1][k], layers[0][k], err_msg=k)
    np.testing.assert_array_equal(outs[1], outs[0])
    check_bf16_ab(outs[1] - g["out_ab"], style)


@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 208, 240), (1, 256, 256), (1, 200, 232)])
def test_model1_block_with_lds_weight_ring(make_sd, shape):
    """conv1_block_fused_t<4,3,true> (32x12 tiles, N >= 2 here) / <4,2,true> (32x8: the batch-1 click path) -- conv1_2's weight tiles through an
    LDS ring, two workgroups per CU: conv1_2's output (model.py:13-17) against the float64 oracle at the bf16 per-layer tolerance, ragged tile
    edges included (208 = 17 x 12 + 4 rows, 240 = 7.5 x 32 columns, 200 x 232), deterministic, and a batch equal to its images run alone (the
    single images take the 32x8 tile: the forms are bit-identical, which is what round 4 measured against the retired `conv1_lw` variants)."""
    n, H, W = shape
    sd = make_sd(0, "he")
    L, ab, m = workloads.random_batch(n, max(H, W), seed=3)
    L, ab, m = L[:, :, :H, :W].copy(), ab[:, :, :H, :W].copy(), m[:, :, :H, :W].copy()
    e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
    e.load_state_dict(sd)
    out = e.forward(L, ab, m, 0.0)
    table = {r["name"]: r["kernel"] for r in e.layer_table()}
    assert table["conv1_1"] == "conv1_block_fused", table
    c12 = e.activation("conv1_2", n)
    # model1 alone in float64 (model.py:13-17,139-148: pack, conv1_1 + ReLU, conv1_2 + ReLU, eval-BN) -- the whole-network oracle costs seconds here
    t = lambda k: torch.from_numpy(np.asarray(sd[k])).double()
    x = torch.cat((torch.from_numpy(L).double() / 100.0, torch.from_numpy(ab).double() / 110.0, torch.from_numpy(m).double()), dim=1)
    x = F.relu(F.conv2d(x, t("model1.0.weight"), t("model1.0.bias"), padding=1))
    x = F.relu(F.conv2d(x, t("model1.2.weight"), t("model1.2.bias"), padding=1))
    x = F.batch_norm(x, t("model1.4.running_mean"), t("model1.4.running_var"), t("model1.4.weight"), t("model1.4.bias"), False, 0.0, 1e-5)
    ref12 = x.numpy()
    err = np.abs(c12 - ref12).max()
    assert err <= 0.04 * (1 + np.abs(ref12).max()), err
    np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), out)
    e.close()
    e1 = engine.HipColorizer(H, W, max_batch=1, precision="bf16")          # the click path's tile choice (32x8)
    e1.load_state_dict(sd)
    for i in range(n):
        one = e1.forward(L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.0)
        np.testing.assert_array_equal(e1.activation("conv1_2", 1)[0], c12[i], err_msg="conv1_2 of image %d: 32x8 tile vs the batch's tile" % i)
        if n == 1:
            np.testing.assert_array_equal(one, out)
    e1.close()


# ------------------------------------------------------------------------------------------------ bf16 click path: conv_kwave_bf16
KW_LAYERS = ["conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "conv6_1", "conv6_2",
             "conv6_3", "conv7_1", "conv7_2", "conv7_3", "conv3_3_short", "conv8_2", "conv8_3", "conv2_2_short", "conv9_2"]


KW_DECONVS = ["conv8_1", "conv9_1", "conv10_1"]


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2", "net64_torch_s1_mc0"])
def test_kwave_click_path_layer_by_layer(golden, make_sd, name):
    """bf16, small launches (the batch-1 click path's kernel choice): every 3x3 stride-1 layer with 64 / 128 / 256 / 512 input channels
    (model.py:19-102; dilation 1 and 2, the x[::2, ::2] reads of conv2_1 / conv3_1 / conv4_1, ragged 32x48 geometry, batch 2) runs as
    conv_kwave_bf16 -- K split over the waves of a workgroup, no split-K launch.  Each layer against the float64 oracle at the bf16
    per-layer tolerance (4 % of the layer's range) and NOT worse than the Winograd form it replaces (`kwave` = 0), the ab map inside the
    stated bf16 bounds of the reference's golden output, deterministic, a batch equal to its images run alone."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    sd = make_sd(seed, style)
    _, _, acts = siggraph_torch.forward(sd, g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]), return_acts=True, dtype=torch.float64)
    err = {}
    engine.set_option("kwave_chain", 0)                       # one launch per layer here (the persistent trunk launch: tests/test_round5_gpu.py)
    for kw in (1, 0):
        engine.set_option("kwave", kw)
        e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
        e.load_state_dict(sd)
        out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
        table = {r["name"]: r["kernel"] for r in e.layer_table()}
        if kw:
            assert [k for k in KW_LAYERS if table[k] != "conv_kwave_bf16"] == [], table
            assert not any("splitK" in table[k] for k in KW_LAYERS)
        else:                                                  # round 2's kernels (the Winograd form in between was retired in round 5)
            assert not any(table[k].startswith("conv_kwave") for k in KW_LAYERS + KW_DECONVS), table
        if kw:                                                 # model8up / model9up / model10up + their shortcut sums (model.py:156,170,172)
            assert [table[k] for k in KW_DECONVS] == ["conv_kwave_deconv_bf16"] * 3, table
        err[kw] = {k: float(np.abs(e.activation(k, n) - acts[k]).mean()) for k in KW_LAYERS}
        for k in KW_LAYERS + (KW_DECONVS if kw else []):
            mx = np.abs(e.activation(k, n) - acts[k]).max()
            assert mx <= 0.04 * (1 + np.abs(acts[k]).max()), "layer %s (kwave=%d): max-abs err %.3e" % (k, kw, mx)
        d = np.abs(out - g["out_ab"])
        check_bf16_ab(d, style, tag="kwave=%d" % kw)
        np.testing.assert_array_equal(e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])), out)       # deterministic
        if kw:
            for i in range(n):
                one = e.forward(g["L_mc"][i:i + 1], g["ab"][i:i + 1], g["mask"][i:i + 1], float(g["maskcent"]))
                np.testing.assert_array_equal(one[0], out[i])
        e.close()
    engine.set_option("kwave_chain", 2)
    # the same plain bf16 products as conv_click's, summed in another order: the mean error over the layers is the same
    assert np.mean([err[1][k] for k in KW_LAYERS]) <= 1.05 * np.mean([err[0][k] for k in KW_LAYERS]), (err[1], err[0])


def test_kwave_click_config(golden, make_sd):
    """BASELINE configs[1] in bf16: 22 of the <= 28 launches of a click forward are conv_kwave_bf16, two conv_kwave_deconv_bf16 (model8up,
    model9up), none a reduction, the reference golden inside the torch-init bf16 bound; the N = 32 throughput path never selects the form."""
    g = golden("config2_mortar_5hints_torchinit")
    engine.set_option("kwave_chain", 0)                       # round 4's launch list: one launch per layer (round 5 chains eleven of them)
    e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
    e.load_state_dict(make_sd(int(g["weight_seed"]), str(g["weight_style"])))
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    engine.set_option("kwave_chain", 2)
    rows = [r for r in e.layer_table() if r["launches"] > 0]
    launches = sum(r["launches"] + (1 if "splitK" in r["kernel"] else 0) for r in rows)
    assert sum(r["kernel"] == "conv_kwave_deconv_bf16" for r in rows) == 2
    assert sum(r["kernel"] == "conv_kwave_bf16" for r in rows) == 22 and launches <= 28 and not any("splitK" in r["kernel"] for r in rows), (launches, [r["kernel"] for r in rows])
    d = np.abs(out - g["out_ab"])
    check_bf16_ab(d, "torch")
    e.close()
    e = engine.HipColorizer(256, 256, max_batch=32, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    L, ab, m = workloads.random_batch(1, 256, seed=3)
    e.forward(L, ab, m, 0.0)
    assert not any(r["kernel"].startswith("conv_kwave") for r in e.layer_table())
    e.close()


KW_OPS = [
    # n, cin, cout, h,  w,  dil, stride, act, bn
    (1, 64, 128, 32, 48, 1, 2, 1, False),      # one chunk (conv2_1's read of x[::2, ::2]): four waves = four tap ranges (2, 2, 2, 3 taps)
    (2, 128, 128, 24, 40, 1, 1, 0, False),     # two chunks x two tap ranges; ragged 16-pixel tile edges (40 = 2.5 tiles), no activation
    (1, 256, 256, 20, 36, 1, 1, 1, True),      # four chunks x (4 | 5 taps); eval-BN after the ReLU
    (1, 256, 512, 24, 24, 1, 2, 1, False),     # conv4_1: strided read, 512 couts
    (3, 512, 512, 8, 12, 2, 1, 1, True),       # dilated (model5 / model6): parity sub-grids of 4 x 6 pixels, batch 3
    (1, 512, 512, 5, 9, 2, 1, 2, False),       # odd sizes: parity sub-grids 3x5 / 2x4; LeakyReLU
    (1, 512, 256, 4, 4, 1, 1, 1, False),       # one partial tile
    (1, 256, 128, 3, 3, 2, 1, 1, False),       # image smaller than the dilation halo
]


@pytest.mark.parametrize("case", KW_OPS)
def test_kwave_single_conv(case):
    """conv_kwave_bf16 as a single operator (the launch the network would make at this size: tile policy auto, small grid) against
    torch float64 conv2d on the bf16-rounded operands' fp32 originals, at the bf16 operator tolerance of tests/test_ops_gpu.py
    (2.5e-2 * (1 + max|ref|)); `kwave` = 0 gives round 2's conv_click launch (the same plain bf16 products in another summation order)."""
    n, cin, cout, h, w, dil, stride, act, bn = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = rs.standard_normal((n, cin, h * stride, w * stride)).astype(np.float32)
    wt = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, cout).astype(np.float32)
    bn_s = rs.uniform(0.5, 2.0, cout).astype(np.float32) if bn else None
    bn_t = rs.uniform(-1, 1, cout).astype(np.float32) if bn else None
    y = F.conv2d(torch.from_numpy(x).double()[:, :, ::stride, ::stride], torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=dil, dilation=dil)
    y = F.relu(y) if act == 1 else (F.leaky_relu(y, 0.2) if act == 2 else y)
    if bn:
        y = y * torch.from_numpy(bn_s).double()[None, :, None, None] + torch.from_numpy(bn_t).double()[None, :, None, None]
    ref = y.numpy()
    got = {}
    for kw in (1, 0):
        engine.set_option("kwave", kw)
        got[kw] = engine.op_conv2d(x, wt, b, dilation=dil, in_stride=stride, act=act, bn_scale=bn_s, bn_shift=bn_t, precision="bf16")
        errv = np.abs(got[kw] - ref).max()
        assert np.isfinite(got[kw]).all() and errv <= 2.5e-2 * (1 + np.abs(ref).max()), (case, kw, errv)
    assert np.abs(got[1] - ref).mean() <= 1.1 * np.abs(got[0] - ref).mean()


KW_DECONV_OPS = [
    # n, cin, cout, h, w, act, resid
    (1, 512, 256, 8, 8, 1, True),        # model8up + shortcut sum + ReLU: a wave per chunk, all four phases
    (2, 256, 128, 12, 20, 1, True),      # model9up: two waves per chunk (phases r = 0 | 1); ragged 8x8 site tiles, batch 2
    (1, 128, 128, 9, 5, 2, True),        # model10up: four waves per chunk (one phase each); odd sizes, LeakyReLU(0.2) (model.py:99)
    (1, 256, 64, 3, 3, 0, False),        # no shortcut, no activation, 64 couts, one partial tile
]


@pytest.mark.parametrize("case", KW_DECONV_OPS)
def test_kwave_single_deconv(case):
    """conv_kwave_deconv_bf16 as a single operator against torch float64 conv_transpose2d (4x4, stride 2, pad 1) + shortcut sum + activation,
    at the bf16 operator tolerance; `kwave` = 0 gives the round-2 launch (conv_click): different bits."""
    n, cin, cout, h, w, act, use_res = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rs.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, cout).astype(np.float32)
    resid = rs.standard_normal((n, cout, 2 * h, 2 * w)).astype(np.float32) if use_res else None
    y = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=2, padding=1)
    if use_res:
        y = y + torch.from_numpy(resid).double()
    ref = (F.relu(y) if act == 1 else (F.leaky_relu(y, 0.2) if act == 2 else y)).numpy()
    got = {}
    for kw in (1, 0):
        engine.set_option("kwave", kw)
        got[kw] = engine.op_deconv4x4s2(x, wt, b, act=act, resid=resid, precision="bf16")
        errv = np.abs(got[kw] - ref).max()
        assert np.isfinite(got[kw]).all() and errv <= 2.5e-2 * (1 + np.abs(ref).max()), (case, kw, errv)
    if cin >= 256:                                             # (two chunks: conv_click's sums differ from this kernel's only below the bf16 rounding of the output)
        assert np.abs(got[1] - ref).mean() <= 1.1 * np.abs(got[0] - ref).mean()

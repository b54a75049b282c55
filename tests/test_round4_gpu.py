"""Round-4 GPU tests: the deconv + shortcut launches and model1 in both MFMA shapes, against the float64 oracle per layer, the
reference's golden ab maps and each other.  Everything goes through the C ABI (ctypes)."""
import numpy as np
import pytest

import torch

from interactive_deep_colorization_amd import engine, workloads
from oracle import siggraph_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_options():
    yield
    for k, v in (("mfma16", 1), ("ds_mfma16", 1), ("v2p", 1), ("fuse_conv1", 1), ("conv1_lw", 3)):
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
    for shape in (1, 0):
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
    bound = (20.0, 2.0) if style == "he" else (0.6, 0.06)
    for shape in (1, 0):
        for k in DS_LAYERS:
            err = np.abs(layers[shape][k] - acts[k]).max()
            assert err <= 0.04 * (1 + np.abs(acts[k]).max()), "layer %s (ds_mfma16=%d): max-abs err %.3e" % (k, shape, err)
        d = np.abs(outs[shape] - g["out_ab"])
        assert d.max() <= bound[0] and d.mean() <= bound[1], (shape, d.max(), d.mean())
    for k in DS_LAYERS:                                       # same sums in a different order: a bf16 ulp here and there
        a, b = layers[1][k], layers[0][k]
        assert np.abs(a - b).max() <= 2.0 ** -6 * (1 + np.abs(b).max()), k
    assert np.abs(outs[1] - outs[0]).mean() <= bound[1] / 2


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
    bound = (20.0, 2.0) if style == "he" else (0.6, 0.06)
    d = np.abs(outs[1] - g["out_ab"])
    assert d.max() <= bound[0] and d.mean() <= bound[1], (d.max(), d.mean())


@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 208, 240)])
def test_model1_block_with_lds_weight_ring(make_sd, shape):
    """conv1_block_fused_t<4,2,true> / <4,3,true> (`conv1_lw` = 2 / 3: 32x8 / 32x12 tiles, conv1_2's weight tiles through an LDS ring,
    two workgroups per CU) against the 32x32-tile form (itself checked per layer against the oracle in test_net_gpu / test_parity_record):
    conv1_2's output (model.py:13-17) and the ab map bit-identical -- the same MFMAs in the same order per accumulator -- ragged
    tile edges included (208 = 17 x 12 + 4 rows, 240 = 7.5 x 32 columns)."""
    n, H, W = shape
    sd = make_sd(0, "he")
    L, ab, m = workloads.random_batch(n, max(H, W), seed=3)
    L, ab, m = L[:, :, :H, :W].copy(), ab[:, :, :H, :W].copy(), m[:, :, :H, :W].copy()
    got = {}
    try:
        for lw in (0, 2, 3):
            engine.set_option("conv1_lw", lw)
            e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
            e.load_state_dict(sd)
            out = e.forward(L, ab, m, 0.0)
            table = {r["name"]: r["kernel"] for r in e.layer_table()}
            assert table["conv1_1"] == "conv1_block_fused", table
            got[lw] = (e.activation("conv1_2", n), out)
            e.close()
    finally:
        engine.set_option("conv1_lw", 3)
    assert np.isfinite(got[0][1]).all() and np.abs(got[0][0]).max() > 0.1
    for lw in (2, 3):
        np.testing.assert_array_equal(got[lw][0], got[0][0], err_msg="conv1_2, conv1_lw=%d" % lw)
        np.testing.assert_array_equal(got[lw][1], got[0][1], err_msg="ab map, conv1_lw=%d" % lw)

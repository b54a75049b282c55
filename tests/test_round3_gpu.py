"""Round-3 GPU tests: zero-copy (mapped pinned host memory) I/O of idc_forward_device, per-stage pipeline timing,
residency flags after a device-pointer forward, pinned-array lifetime across close(), wrapper fallbacks.
Everything goes through the C ABI (ctypes); the oracle is not needed here (bit-for-bit self-consistency)."""
import gc
import os

import numpy as np
import pytest

import torch

from interactive_deep_colorization_amd import _native as N
from interactive_deep_colorization_amd import api, engine, workloads
from oracle import siggraph_torch

from bounds import FP32_TOL, bf16_bound, check_bf16_ab  # noqa: F401

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WINO_LAYERS = ["conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "conv6_1", "conv6_2",
               "conv6_3", "conv7_1", "conv7_2", "conv7_3", "conv3_3_short", "conv8_2", "conv8_3", "conv2_2_short", "conv9_2",
               "conv1_2_short", "conv10_2"]


@pytest.fixture(autouse=True)
def _reset_options():
    yield
    engine.set_option("winograd", 1)           # (round 5: one option -- 0 off, 1 automatic, 2 every deconv too, 12 / 21 / 22 a forced 3x3 form)
    engine.set_option("kwave", 1)
    engine.set_option("mfma16", 1)
    engine.set_tile_policy("auto")
    engine.set_splitk_policy("auto")


# ------------------------------------------------------------------------------------------------ fp32 Winograd F(2x2,3x3)
@pytest.mark.parametrize("form", [0, 12, 21, 22])
@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2", "net64_torch_s1_mc0"])
def test_winograd_fp32_layer_by_layer(golden, make_sd, name, form):
    """Every 3x3 stride-1 layer of the fp32 path runs as conv_wino_f32 (dilation 1 and 2, ragged 32x48 geometry with 3x... pixel
    trunks, batch 2, every <TB,CB> form forced): each against the float64 oracle at the tolerance of the direct fp32 kernels,
    the ab map against the reference golden."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    _, _, acts = siggraph_torch.forward(make_sd(seed, style), g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]),
                                        return_acts=True, dtype=torch.float64)
    engine.set_option("winograd", form if form else 1)
    e = engine.HipColorizer(H, W, max_batch=n, precision="fp32")
    e.load_state_dict(make_sd(seed, style))
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    table = {r["name"]: r["kernel"] for r in e.layer_table()}
    assert [k for k in WINO_LAYERS if table[k] != "conv_wino_f32"] == [], table
    assert not any("splitK" in v for k, v in table.items() if k in WINO_LAYERS)
    for k in WINO_LAYERS:
        ref = acts[k]
        err = np.abs(e.activation(k, n) - ref).max()
        assert err <= 2e-4 * (1 + np.abs(ref).max()), "layer %s (form %d): max-abs err %.3e" % (k, form, err)
    d = np.abs(out - g["out_ab"])
    assert d.max() <= (3e-3 if style == "he" else 1e-3), d.max()
    np.testing.assert_array_equal(e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])), out)       # deterministic
    # the direct kernels compute the same network (different summation order only)
    engine.set_option("winograd", 0)
    base = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    assert not any(v == "conv_wino_f32" for v in (r["kernel"] for r in e.layer_table()))
    assert np.abs(out - base).max() <= (3e-3 if style == "he" else 5e-4)
    e.close()


# (round 5: conv_wino_bf16 / conv_wino_deconv_bf16 -- the bf16 Winograd kernels of the round-3 click path -- were retired to
#  docs/experiments/conv_wino_bf16_round3.hip.txt together with their tests; conv_kwave_* are tested in test_round4_gpu.py / test_round5_gpu.py)


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2"])
def test_winograd_deconv_fp32_layer_by_layer(golden, make_sd, name):
    """fp32 ConvTranspose 4x4 s2 layers as Winograd F(2x2,2x2) over their four phases (conv_wino_deconv_f32, forced on every deconv):
    conv8_1 / conv9_1 / conv10_1 (deconv + fp32 shortcut sum + ReLU) against the float64 oracle at the direct kernels' tolerance."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    _, _, acts = siggraph_torch.forward(make_sd(seed, style), g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]),
                                        return_acts=True, dtype=torch.float64)
    engine.set_option("winograd", 2)
    e = engine.HipColorizer(H, W, max_batch=n, precision="fp32")
    e.load_state_dict(make_sd(seed, style))
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    table = {r["name"]: r["kernel"] for r in e.layer_table()}
    for k in ("conv8_1", "conv9_1", "conv10_1"):
        assert table[k] == "conv_wino_deconv_f32", table
        ref = acts[k]
        err = np.abs(e.activation(k, n) - ref).max()
        assert err <= 2e-4 * (1 + np.abs(ref).max()), "layer %s: max-abs err %.3e" % (k, err)
    assert np.abs(out - g["out_ab"]).max() <= 3e-3
    engine.set_option("winograd", 0)
    base = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    assert not any(r["kernel"].startswith("conv_wino") for r in e.layer_table())
    assert np.abs(out - base).max() <= 3e-3
    e.close()


def test_winograd_fp32_click_config(golden, make_sd):
    """BASELINE configs[1] (one 256x256 image, 5 hints), fp32 default = Winograd: the reference golden at 1e-3, batch == images alone."""
    g = golden("config2_mortar_5hints_torchinit")
    e = engine.HipColorizer(256, 256, max_batch=1, precision="fp32")
    e.load_state_dict(make_sd(int(g["weight_seed"]), str(g["weight_style"])))
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    kernels = [r["kernel"] for r in e.layer_table() if r["launches"] > 0]
    assert sum(k == "conv_wino_f32" for k in kernels) >= 20 and sum("splitK" in k for k in kernels) <= 1, kernels      # no reduction launches left (model10up runs un-split)
    assert np.abs(out - g["out_ab"]).max() <= 1e-3
    e.close()
    L, ab, m = workloads.random_batch(3, 64, seed=9)
    e = engine.HipColorizer(64, 64, max_batch=3, precision="fp32")
    e.load_state_dict(make_sd(0, "he"))
    whole = e.forward(L, ab, m, 0.0)
    for i in range(3):
        np.testing.assert_array_equal(e.forward(L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.0)[0], whole[i])
    for form in (12, 21, 22):                 # the forms differ in tiling only: same sums in the same order, bit for bit
        engine.set_option("winograd", form)
        np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), whole)
    e.close()


@pytest.mark.parametrize("precision,size,nb", [("bf16", 64, 4), ("fp32", 64, 2), ("bf16", 256, 32)])
def test_mapped_io_equals_idc_forward_bit_for_bit(make_sd, precision, size, nb):
    """idc_forward_device on PINNED HOST pointers (mapped into the device's address space): conv1's operand staging
    reads L / ab / mask over PCIe, the head writes out_ab into host memory.  Same bits as the copying idc_forward --
    at the bench geometry (N=32, 256x256: conv1_block_fused + the fused tanh head) and on the small-tile kernels."""
    e = engine.HipColorizer(size, size, max_batch=nb, precision=precision)
    e.load_state_dict(make_sd(0, "he"))
    L, ab, m = workloads.random_batch(nb, size, seed=11)
    ref = e.forward(L, ab, m, 0.5)
    pin = [e.pinned_empty(x.shape) for x in (L, ab, m)] + [e.pinned_empty((nb, 2, size, size))]
    for dst, src in zip(pin[:3], (L, ab, m)):
        dst[...] = src
    pin[3][...] = -7.0
    for _ in range(2):                                        # back to back on the stream, then one sync
        e.forward_device(nb, pin[0], pin[1], pin[2], pin[3], 0.5, sync=False)
    e.sync()
    np.testing.assert_array_equal(pin[3], ref)
    e.close()
    gc.collect()
    assert float(np.abs(pin[3] - ref).max()) == 0.0           # the pinned arrays outlive the engine


def test_pipeline_times_are_ordered_and_plausible(make_sd):
    e = engine.HipColorizer(256, 256, max_batch=8, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    with pytest.raises(N.IdcError):
        e.pipeline_times(0)                                   # nothing ran yet
    bufs = []
    for k in range(2):
        L, ab, m = workloads.random_batch(8, 256, seed=3 + k)
        arrs = [e.pinned_empty(x.shape) for x in (L, ab, m)] + [e.pinned_empty((8, 2, 256, 256))]
        for dst, src in zip(arrs[:3], (L, ab, m)):
            dst[...] = src
        bufs.append(arrs)
    for i in range(6):
        k = i & 1
        if i >= 2:
            e.wait(k)
        e.forward_async(k, *bufs[k], 0.0)
    with pytest.raises(N.IdcError):
        e.pipeline_times(1)                                   # still in flight
    e.wait(0); e.wait(1)
    t0, t1 = e.pipeline_times(0), e.pipeline_times(1)
    for t in (t0, t1):
        assert np.all(np.diff(t) >= -1e-3), t                # h2d start <= h2d end <= compute start <= ... <= d2h end
        assert 0.0 < t[3] - t[2] < 50.0 and t[1] - t[0] < 50.0 and t[5] - t[4] < 50.0, t
    assert t1[3] > t0[3]                                      # batch 5 (slot 1) computed after batch 4 (slot 0)
    e.close()


def test_forward_device_invalidates_the_resident_result(make_sd):
    """after idc_forward_device the result is in the CALLER's buffer; the display step must not serve the map an
    older forward left in d_out / d_labq."""
    import torch
    e = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    L, ab, m = workloads.random_batch(2, 64, seed=5)
    e.forward_rgb(L, ab, m, 0.0)
    Lw = np.full((64, 64), 50.0)
    e.upsample_lab2rgb(Lw, "output_ab", "linear")            # resident: fine
    e.upsample_lab2rgb(Lw, "output_ab_raw", "nearest")
    dev = torch.device("cuda", 0)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
    dout = torch.empty((2, 2, 64, 64), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    e.forward_device(2, dL, dab, dm, dout, 0.0, sync=True)
    for src in ("output_ab", "output_ab_raw"):
        with pytest.raises(N.IdcError):
            e.upsample_lab2rgb(Lw, src, "linear")
    e.close()


def test_wrapper_getters_fall_back_when_the_engine_lost_the_map(make_sd):
    """the Python-side token can outlive the engine's resident map (a direct engine call in between); the
    full-resolution getters then take the host route instead of raising; GlobDist keeps its device route."""
    rgb = np.load(os.path.join(HERE, "golden", "mortar_pestle_256_rgb.npy"))
    model = api.ColorizeImageTorch(Xd=256, precision="bf16")
    model.prep_net(path="", state_dict=make_sd(0, "he"))
    model.set_image(rgb)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    model.net_forward(hab, hm)
    dev_img = model.get_img_fullres()
    L, ab, m = workloads.random_batch(1, 256, seed=2)
    model.net.forward(L, ab, m, 0.0)                           # behind the wrapper's back: labq no longer resident
    host_img = model.get_img_fullres()                         # must not raise
    d = np.abs(dev_img.astype(np.int32) - host_img.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 2e-4
    with pytest.raises(RuntimeError):
        model.get_result_window(np.full((300, 300), 50.0))
    model.net.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_winograd_odd_trunk_geometry(make_sd, precision):
    """40x72 input: the trunk runs at 5x9 pixels (odd in both directions), the dilated layers' parity sub-grids are 3x5 / 2x4 --
    ragged tiles in every Winograd kernel (conv, strided-view conv, deconv); batch 3 so that image != block boundaries."""
    sd = make_sd(2, "torch")
    L, ab, m = workloads.random_batch(3, 72, seed=4)
    L, ab, m = L[:, :, :40, :], ab[:, :, :40, :], m[:, :, :40, :]
    L, ab, m = (np.ascontiguousarray(x) for x in (L, ab, m))
    ref = siggraph_torch.forward(sd, L, ab, m, 0.5)
    engine.set_option("winograd", 2)
    e = engine.HipColorizer(40, 72, max_batch=3, precision=precision)
    e.load_state_dict(sd)
    out = e.forward(L, ab, m, 0.5)
    kw_ = lambda k: k.startswith("conv_kwave") or k.startswith("chained into")      # (round 5: trunk layers may ride in a conv_kwave_chain_bf16 launch)
    assert sum(r["kernel"].startswith("conv_wino") or kw_(r["kernel"]) for r in e.layer_table()) >= 22
    if precision == "bf16":                                     # round 4: the 3x3 layers of the bf16 click path are conv_kwave_bf16, the deconvs stay Winograd
        assert sum(kw_(r["kernel"]) for r in e.layer_table()) >= 19
    d = np.abs(out - ref)
    assert d.max() <= (1e-3 if precision == "fp32" else bf16_bound("torch")[0]), d.max()
    for i in range(3):
        np.testing.assert_array_equal(e.forward(L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.5)[0], out[i])
    e.close()


def test_pooled_pinned_results_keep_value_semantics():
    """Results of the blocking calls live in recycled pinned buffers (no staging memcpy); an array still referenced is
    never overwritten by a later call, and pinned / pageable inputs give the same bits."""
    import gc
    from interactive_deep_colorization_amd import engine, workloads
    sd = workloads.random_state_dict(0, "he")
    e = engine.HipColorizer(64, 64, max_batch=2, precision="fp32")
    e.load_state_dict(sd)
    L, ab, m = workloads.random_batch(2, 64, seed=3)
    out1, rgb1, lab1 = e.forward_rgb(L[:1], ab[:1], m[:1], 0.0)
    keep = (out1.copy(), rgb1.copy(), lab1.copy())
    out2, rgb2, lab2 = e.forward_rgb(L[1:], ab[1:], m[1:], 0.0)
    assert not np.shares_memory(out1, out2) and not np.shares_memory(lab1, lab2) and not np.shares_memory(rgb1, rgb2)
    assert np.array_equal(out1, keep[0]) and np.array_equal(rgb1, keep[1]) and np.array_equal(lab1, keep[2])
    assert not np.array_equal(out1, out2)
    # recycled after the last view dies
    addr = out2.ctypes.data
    del out2, rgb2, lab2
    gc.collect()
    out3, rgb3, lab3 = e.forward_rgb(L[:1], ab[:1], m[:1], 0.0)
    assert addr in (out1.ctypes.data, out3.ctypes.data) or e._pool.retained >= 0
    assert np.array_equal(out3, keep[0]) and np.array_equal(rgb3, keep[1]) and np.array_equal(lab3, keep[2])
    # pinned inputs (uploaded in place) and float64 inputs (converted into pooled pinned buffers) == pageable float32 inputs
    pL, pab, pm = (e.pinned_empty(x[:1].shape) for x in (L, ab, m))
    pL[...], pab[...], pm[...] = L[:1], ab[:1], m[:1]
    out4 = e.forward(pL, pab, pm, 0.0)
    out5 = e.forward(L[:1].astype(np.float64), ab[:1].astype(np.float64), m[:1].astype(np.float64), 0.0)
    assert np.array_equal(out4, keep[0]) and np.array_equal(out5, keep[0])
    # mixed: one pinned, two pageable
    out6 = e.forward(pL, ab[:1], m[:1], 0.0)
    assert np.array_equal(out6, keep[0])
    # a result outlives the engine
    e.close()
    gc.collect()
    assert np.array_equal(out1, keep[0]) and np.array_equal(lab3, keep[2])


# ------------------------------------------------------------------------------------------------ the two MFMA shapes of the throughput tile
@pytest.mark.parametrize("mfma16", [1, 0])
@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2"])
def test_throughput_tile_in_both_mfma_shapes(golden, make_sd, name, mfma16):
    """conv_igemm_v2m (16x16x32 MFMA, the default where it applies) and conv_igemm_v2 (32x32x16, `mfma16` = 0) compute the same
    layers: both within the bf16 bound of the reference's golden output and within bf16 rounding of each other."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    sd = make_sd(seed, style)
    outs = {}
    from conftest import has_ab_partners
    partners = has_ab_partners()
    if not partners and mfma16 == 0:
        pytest.skip("conv_igemm_v2 (32x32x16 MFMA) is an A/B partner: not in the default library (make EXTRA=-DIDC_AB_PARTNERS)")
    for shape in ((mfma16, 1 - mfma16) if partners else (1,)):
        engine.set_tile_policy("large")
        engine.set_option("mfma16", shape)
        e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
        e.load_state_dict(sd)
        outs[shape] = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
        kernels = [r["kernel"] for r in e.layer_table() if r["kernel"].startswith("conv_igemm_v2")]
        assert kernels, e.layer_table()
        assert any("+m16" in k for k in kernels) == bool(shape), kernels
        if shape:                                    # what stays on the 32x32 kernel: launches with a shortcut sum
            assert all("+m16" in k or "+shortcut" in k for k in kernels), kernels
        e.close()
    ref = g["out_ab"]
    bound = bf16_bound(style)
    for shape, out in outs.items():
        check_bf16_ab(out - ref, style, tag="mfma16=%d" % shape)
    if partners:
        d = np.abs(outs[1] - outs[0])
        assert d.mean() <= bound[1] / 2, (d.max(), d.mean())

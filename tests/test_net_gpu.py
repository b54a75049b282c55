"""GPU parity of the whole forward pass through the C ABI (idc_forward) against
(a) the reference-generated golden vectors in tests/golden/ and (b) the oracle re-run here.

Stated tolerances (max-abs on the ab output, range +-110):
  fp32 path, torch-default-style weights (|ab| <~ 25):   1e-3   <- the BASELINE.json target
  fp32 path, he-style weights (full tanh range):          3e-3   = the reference's own fp32 noise floor
        (its output moves 4.8e-4 between oneDNN blockings at 64x64 -- golden field
         batched_vs_single_f32 -- and the fp32-vs-fp64 gap is 3.7e-4)
  bf16 path (bf16 activations+weights, fp32 accumulate; ~0.3 % rounding noise per layer, 30 layers):
        stated once in tests/bounds.py (max / mean / q99.9): he-style 16 / 1.5 / 11 (measured 13.1-13.9 / 1.22 / 8.0 at 256x256,
        4.2 / 0.61 at 64x64), torch-style 0.3 / 0.04 / 0.15 (measured 0.14 / 0.023 / 0.094)
Per-layer activations are compared too, so a failure names the first bad layer.
"""
import numpy as np
import pytest
import torch

from interactive_deep_colorization_amd import api, engine, workloads
from oracle import siggraph_torch

pytestmark = pytest.mark.gpu

from bounds import FP32_TOL, bf16_bound, check_bf16_ab
ACT_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2",
             "conv4_3", "conv5_1", "conv5_2", "conv5_3", "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2",
             "conv7_3", "conv3_3_short", "conv8_1", "conv8_2", "conv8_3", "conv2_2_short", "conv9_1", "conv9_2",
             "conv1_2_short", "conv10_1", "conv10_2"]

_ENGINES = {}


def get_engine(H, W, max_batch, precision, seed, style, dist=False, make_sd=None):
    key = (H, W, max_batch, precision, seed, style, dist)
    if key not in _ENGINES:
        e = engine.HipColorizer(H, W, max_batch=max_batch, precision=precision, dist=dist)
        e.load_state_dict(make_sd(seed, style))
        _ENGINES[key] = e
    return _ENGINES[key]


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net64_torch_s1_mc0", "net32x48_he_s2"])
def test_fp32_matches_reference_golden_layer_by_layer(golden, make_sd, name):
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    e = get_engine(H, W, n, "fp32", seed, style, make_sd=make_sd)
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    _, _, acts = siggraph_torch.forward(make_sd(seed, style), g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]),
                                        return_acts=True, dtype=torch.float64)
    for k in ACT_NAMES:                                   # first failing layer is reported
        got = e.activation(k, n)
        ref = acts[k]
        assert got.shape == ref.shape, k
        err = np.abs(got - ref).max()
        assert err <= 2e-4 * (1 + np.abs(ref).max()), "layer %s: max-abs err %.3e" % (k, err)
    err = np.abs(out - g["out_ab"]).max()
    assert err <= FP32_TOL[style], "fp32 vs reference golden: %.3e > %.1e" % (err, FP32_TOL[style])
    # both fp32 implementations sit at the same distance from the float64 truth
    e64_ref = float(g["noise_floor_f32_vs_f64"])
    e64_hip = np.abs(out - g["out_ab_f64"]).max()
    assert e64_hip <= 4 * e64_ref + 1e-4, "HIP fp32 is %.2e from fp64, the reference %.2e" % (e64_hip, e64_ref)


@pytest.fixture(autouse=True)
def _reset_options():
    yield
    engine.set_option("fuse_conv1", 1)


@pytest.fixture(autouse=True)
def _reset_tile_policy():
    yield
    engine.set_tile_policy("auto")
    engine.set_splitk_policy("auto")


@pytest.mark.parametrize("tiles", ["small", "large"])
@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net64_torch_s1_mc0", "net32x48_he_s2"])
def test_bf16_within_stated_tolerance(golden, make_sd, name, tiles):
    """Both bf16 kernel families (conv_igemm small tiles / conv_igemm_v2 large tiles, forced through the
    tile policy because these images are tiny) against the reference golden output, and layer by layer
    against the float64 oracle (|err| <= 4 % of the layer's range: bf16 rounding accumulated over <= 29 layers)."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    engine.set_tile_policy(tiles)
    e = get_engine(H, W, n, "bf16", seed, style, make_sd=make_sd)
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    kernels = set(r["kernel"].split("<")[0] for r in e.layer_table() if r["kernel"].startswith("conv"))
    assert ("conv_igemm_v2" in kernels) == (tiles == "large"), kernels
    _, _, acts = siggraph_torch.forward(make_sd(seed, style), g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]),
                                        return_acts=True, dtype=torch.float64)
    # large tiles: the tanh head rides in conv10_2's epilogue and each shortcut conv in its deconv's K loop;
    # those tensors are never materialised (their consumers are checked instead)
    table = e.layer_table()
    fused = set(r["name"] for r in table if r["kernel"].endswith("+head") or r["kernel"].startswith("fused into"))
    assert (fused == {"conv10_2", "conv3_3_short", "conv2_2_short", "conv1_2_short"}) == (tiles == "large"), fused
    for k in ACT_NAMES:
        if k in fused:                                         # never materialised: only the head output leaves the kernel
            continue
        got = e.activation(k, n)
        err = np.abs(got - acts[k]).max()
        assert err <= 0.04 * (1 + np.abs(acts[k]).max()), "layer %s (%s tiles): max-abs err %.3e" % (k, tiles, err)
    d = np.abs(out - g["out_ab"])
    assert np.isfinite(out).all() and np.abs(out).max() <= 110.0
    check_bf16_ab(d, style, tag=name)


@pytest.mark.parametrize("name,precision", [("config1_mortar_zero_hints", "fp32"), ("config2_mortar_5hints", "fp32"),
                                            ("config2_mortar_5hints_torchinit", "fp32"),
                                            ("config2_mortar_5hints", "bf16")])
def test_baseline_configs_1_and_2(golden, make_sd, name, precision):
    """BASELINE.json configs[0]/[1]: mortar_pestle.jpg at 256x256, zero hints / 5 hint points."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    e = get_engine(256, 256, 1, precision, seed, style, make_sd=make_sd)
    out = e.forward(g["L_mc"][0], g["ab"][0], g["mask"][0], float(g["maskcent"]))     # 3-D call shape of the reference
    d = np.abs(out - g["out_ab"])
    if precision == "fp32":
        assert d.max() <= FP32_TOL[style], "max-abs %.3e" % d.max()
    else:
        check_bf16_ab(d, style, tag=name)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_batching_is_exactly_per_image(make_sd, precision):
    """No op mixes images (eval-BN): a batch equals its images run alone, bit for bit, and a
    2-way shard of the batch equals the whole batch (the multi-GPU equivalence)."""
    L, ab, m = workloads.random_batch(5, 64, seed=21)
    e = get_engine(64, 64, 5, precision, 0, "he", make_sd=make_sd)
    full = e.forward(L, ab, m, 0.0)
    for i in (0, 3, 4):
        np.testing.assert_array_equal(e.forward(L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.0)[0], full[i])
    parts = [e.forward(L[lo:hi], ab[lo:hi], m[lo:hi], 0.0) for lo, hi in
             (workloads.shard_bounds(5, 2, r) for r in range(2))]
    np.testing.assert_array_equal(np.concatenate(parts), full)
    np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), full)         # run-to-run deterministic
    # hints matter, masks accepted as bool (GUI passes bool: ui/gui_draw.py:274-275)
    out_b = e.forward(L[:1], ab[:1], m[:1].astype(bool), 0.0)
    np.testing.assert_array_equal(out_b[0], full[0])


def test_config3_full_size_properties(make_sd):
    """BASELINE.json configs[2]: N=32 random 256x256, bf16 -- size-independent properties only
    (the oracle needs ~10 s per image here): finite, bounded by 110*tanh, deterministic, and
    image 7 of the batch equals image 7 run alone."""
    L, ab, m = workloads.random_batch(32, 256, seed=0)
    e = get_engine(256, 256, 32, "bf16", 0, "he", make_sd=make_sd)
    out = e.forward(L, ab, m, 0.0)
    assert out.shape == (32, 2, 256, 256) and np.isfinite(out).all() and np.abs(out).max() <= 110.0
    assert out.std() > 1.0
    np.testing.assert_array_equal(e.forward(L[7:8], ab[7:8], m[7:8], 0.0)[0], out[7])
    np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), out)
    # (the comparison with the oracle at this size lives in tests/test_parity_record_gpu.py, several images, measured errors recorded)


def test_512_fp32(make_sd):
    """512x512 input (BASELINE configs[4] geometry, local-hints net): trunk runs at 64x64."""
    L, ab, m = workloads.random_batch(1, 512, seed=5)
    e = get_engine(512, 512, 1, "fp32", 1, "torch", make_sd=make_sd)
    out = e.forward(L, ab, m, 0.5)
    ref = siggraph_torch.forward(make_sd(1, "torch"), L, ab, m, 0.5)
    assert np.abs(out - ref).max() <= FP32_TOL["torch"]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_splitk_matches_unsplit(golden, make_sd, precision):
    """Split-K (the batch-1 click path: slices of the cin chunks + a fixed-order reduction) computes the same
    network: against the golden output at the usual tolerance, and within summation-order noise of the unsplit run."""
    g = golden("net64_he_s0_mc05")
    n, _, H, W = g["L_mc"].shape
    engine.set_tile_policy("small")
    e = get_engine(H, W, n, precision, 0, "he", make_sd=make_sd)
    engine.set_splitk_policy("never")
    base = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    assert not any("splitK" in r["kernel"] for r in e.layer_table())
    engine.set_splitk_policy("always")
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    assert sum("splitK" in r["kernel"] for r in e.layer_table()) >= 20
    np.testing.assert_array_equal(e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])), out)   # deterministic
    np.testing.assert_array_equal(e.forward(g["L_mc"][:1], g["ab"][:1], g["mask"][:1], float(g["maskcent"]))[0], out[0])
    d = np.abs(out - g["out_ab"])
    if precision == "fp32":
        assert d.max() <= FP32_TOL["he"] and np.abs(out - base).max() <= 2e-3
    else:
        check_bf16_ab(d, "he")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dist_head(golden, make_sd, precision):
    g = golden("dist64_he_s0")
    e = get_engine(64, 64, 1, precision, 0, "he", dist=True, make_sd=make_sd)
    out, dq = e.forward_dist(g["L_mc"], g["ab"], g["mask"], 0.0)
    assert dq.shape == (1, 529, 16, 16)
    np.testing.assert_allclose(dq.sum(axis=1), 1.0, atol=1e-4)
    tol = 2e-4 if precision == "fp32" else 2e-2
    assert np.abs(dq - g["class_probs_lowres"]).max() <= tol
    assert np.abs(out - g["out_ab"]).max() <= (FP32_TOL["he"] if precision == "fp32" else bf16_bound("he")[0])


def test_errors_through_the_abi(make_sd):
    from interactive_deep_colorization_amd import _native as N
    e = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
    z = np.zeros((1, 1, 64, 64), np.float32)
    with pytest.raises(N.IdcError) as ei:                       # "I need to have a net!"
        e.forward(z, np.zeros((1, 2, 64, 64)), z)
    assert ei.value.status == -4
    e.load_state_dict(make_sd(1, "torch"))
    with pytest.raises(N.IdcError) as ei:
        e.forward(np.zeros((3, 1, 64, 64)), np.zeros((3, 2, 64, 64)), np.zeros((3, 1, 64, 64)))
    assert ei.value.status == -6                                # batch > max_batch
    with pytest.raises(ValueError):
        e.forward(np.zeros((1, 1, 32, 32)), np.zeros((1, 2, 32, 32)), np.zeros((1, 1, 32, 32)))
    with pytest.raises(N.IdcError):
        e.forward_dist(z, np.zeros((1, 2, 64, 64)), z)          # no dist head on this handle
    e.close()


def test_drop_in_api_end_to_end(golden, make_sd):
    """The reference call sequence (ideepcolor.py:68-72, notebook cells) against the HIP backend."""
    rgb = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "mortar_pestle_256_rgb.npy"))
    g = golden("config2_mortar_5hints")
    model = api.ColorizeImageTorch(Xd=256, maskcent=False, precision="fp32")
    model.prep_net(path="", state_dict=make_sd(0, "he"))
    model.set_image(rgb)
    input_ab, mask = workloads.hints_config2(256, 5, 3, 0)
    img = model.net_forward(input_ab, mask)
    assert img.shape == (256, 256, 3) and img.dtype == np.uint8
    assert np.abs(model.output_ab_raw[None] - g["out_ab"]).max() <= FP32_TOL["he"]
    assert model.output_ab.shape == (2, 256, 256) and model.input_ab is input_ab
    assert model.get_img_fullres().shape == (256, 256, 3)
    assert model.get_result_PSNR() > 5
    # Caffe i/o convention on the same kernels: raw inputs, x100 head
    caffe = api.ColorizeImageCaffe(Xd=256, precision="fp32")
    sd = dict(make_sd(0, "he"))
    w0 = sd["model1.0.weight"].copy()           # fold the torch input normalisation into conv1_1
    w0[:, 0] /= 100.0; w0[:, 1:3] /= 110.0; w0[:, 3] /= 110.0
    sd["model1.0.weight"] = w0
    caffe.prep_net(0, state_dict=sd)
    caffe.set_image(rgb)
    caffe.net_forward(input_ab, mask)
    np.testing.assert_allclose(caffe.output_ab_raw * 1.1, model.output_ab_raw, atol=5e-3)
    # distribution model + colour suggestions
    dist = api.ColorizeImageTorchDist(Xd=256, precision="bf16")
    dist.prep_net(path="", dist=True, state_dict=make_sd(0, "he"))
    dist.set_image(rgb)
    ret = dist.net_forward(input_ab, mask)
    assert ret.shape == (2, 256, 256) and dist.dist_ab.shape == (529, 256, 256)
    assert dist.dist_ab_grid.shape == (23, 23, 256, 256)
    reccs, conf = dist.get_ab_reccs(135, 160, K=3, N=2000, return_conf=True)
    assert reccs.shape == (3, 2) and abs(conf.sum() - 1) < 1e-6


def test_device_colour_post_matches_host_formulas(golden, make_sd):
    """idc_lab2rgb / idc_forward_rgb against the numpy restatement of skimage's lab2rgb /
    rgb2lab (oracle/colorspace.py): uint8 RGB equal up to the truncation knife edge (<= 1 LSB on <= 0.01 % of the
    values), refreshed Lab = rgb2lab of the device's own uint8 image to 1e-9."""
    from oracle import colorspace as ocs
    g = golden("config2_mortar_5hints")
    e = get_engine(256, 256, 1, "fp32", 0, "he", make_sd=make_sd)
    L = g["L_mc"] + 50.0
    rs = np.random.RandomState(3)
    ab = rs.uniform(-100, 100, (1, 2, 256, 256)).astype(np.float32)         # includes far out-of-gamut colours
    rgb, labq = e.lab2rgb(L, ab)
    ref = (np.clip(ocs.lab2rgb(np.concatenate((L[0], ab[0]), 0).transpose(1, 2, 0).astype(np.float64)), 0, 1) * 255).astype("uint8")
    diff = np.abs(rgb[0].astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4, (diff.max(), (diff > 0).mean())
    np.testing.assert_allclose(labq[0], ocs.rgb2lab(rgb[0]).transpose(2, 0, 1), atol=1e-9)
    out, rgb2, labq2 = e.forward_rgb(g["L_mc"], g["ab"], g["mask"], 0.0, l_cent=50.0)
    assert np.abs(out - g["out_ab"]).max() <= FP32_TOL["he"]
    # the fused form adds l_cent to the float32 L_mc plane in float64; the two-call form is handed float32(L_mc + 50):
    # the last-bit difference in L may flip a value sitting exactly on a truncation edge (1 LSB, a handful of values)
    rgb3, labq3 = e.lab2rgb(L, out)
    dd = np.abs(rgb2.astype(int) - rgb3.astype(int))
    assert dd.max() <= 1 and (dd > 0).mean() <= 1e-4, (dd.max(), (dd > 0).mean())
    same = (dd.max(axis=-1) == 0)[:, None]                                   # pixels whose uint8 triple agrees: identical refresh
    np.testing.assert_array_equal(np.where(same, labq2, 0), np.where(same, labq3, 0))


@pytest.mark.parametrize("precision,tiles", [("fp32", "auto"), ("bf16", "small"), ("bf16", "large")])
def test_ragged_geometry_40x72_batch3(make_sd, precision, tiles):
    """H, W multiples of 8 only (40 x 72: every level has partial tiles, the 5x9 trunk is smaller than one tile and
    than the dilation-2 halo), odd batch; against the oracle run here."""
    L, ab, m = workloads.random_batch(3, 40, 72, seed=31, max_points=4, max_p=2)
    sd = make_sd(2, "he")
    engine.set_tile_policy(tiles)
    e = engine.HipColorizer(40, 72, max_batch=3, precision=precision)
    e.load_state_dict(sd)
    out = e.forward(L, ab, m, 0.5)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.5)
    d = np.abs(out - ref)
    if precision == "fp32":
        assert d.max() <= FP32_TOL["he"], d.max()
    else:
        check_bf16_ab(d, "he")
    np.testing.assert_array_equal(e.forward(L[2:3], ab[2:3], m[2:3], 0.5)[0], out[2])
    e.close()


def test_conv1_1_throughput_kernel(make_sd):
    """conv1_1's 32x32-tile form (conv1_1_bf16_kernel, used from 128 tiles up) against a float64 conv of the packed
    input (model.py:139-148, :13) and against the small-tile path that single-image forwards take.  72 x 104: partial
    tiles on both axes, 3 x 4 tiles x 12 images = 144 workgroups.  Tolerance = one bf16 ulp of the largest value (both
    the operands and the stored result are bf16)."""
    import torch
    H, W, N = 72, 104, 12
    L, ab, m = workloads.random_batch(N, H, W, seed=17, max_points=6, max_p=3)
    sd = make_sd(4, "he")
    engine.set_option("fuse_conv1", 0)                         # conv1_1 as its own launch, so that its output exists
    e = engine.HipColorizer(H, W, max_batch=N, precision="bf16")
    e.load_state_dict(sd)
    e.forward(L, ab, m, 0.5)
    big = e.activation("conv1_1", N)
    c12_two = e.activation("conv1_2", N)
    x = np.concatenate([L / 100.0, ab / 110.0, m - 0.5], axis=1).astype(np.float64)
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(sd["model1.0.weight"].astype(np.float64)),
                                                torch.from_numpy(sd["model1.0.bias"].astype(np.float64)), padding=1)).numpy()
    tol = 2.0 ** -7 * max(1.0, np.abs(ref).max())
    assert np.abs(big - ref).max() <= tol, (np.abs(big - ref).max(), tol)
    e.forward(L[5:6], ab[5:6], m[5:6], 0.5)                    # same handle: same kernel whatever the batch
    np.testing.assert_array_equal(e.activation("conv1_1", 1)[0], big[5])
    e.close()
    e1 = engine.HipColorizer(H, W, max_batch=1, precision="bf16")   # 12 tiles: the small-tile kernel
    e1.load_state_dict(sd)
    e1.forward(L[5:6], ab[5:6], m[5:6], 0.5)
    small = e1.activation("conv1_1", 1)
    assert np.abs(small[0] - big[5]).max() <= tol
    assert (small[0] == big[5]).mean() > 0.98                   # same operands, same fp32 sums up to their order
    e1.close()
    # model1 as one launch (conv1_block_fused): same operands and the same fp32 sums up to their order, so conv1_2
    # agrees with the two-launch result to one bf16 ulp and is identical almost everywhere
    engine.set_option("fuse_conv1", 1)
    e2 = engine.HipColorizer(H, W, max_batch=N, precision="bf16")
    e2.load_state_dict(sd)
    out2 = e2.forward(L, ab, m, 0.5)
    c12 = e2.activation("conv1_2", N)
    assert "conv1_block_fused" in [r["kernel"] for r in e2.layer_table()]
    with pytest.raises(Exception):                             # conv1_1 itself is never written now: refused, not stale
        e2.activation("conv1_1", N)
    tol2 = 2.0 ** -7 * max(1.0, np.abs(c12_two).max())
    assert np.abs(c12 - c12_two).max() <= tol2, (np.abs(c12 - c12_two).max(), tol2)
    assert (c12 == c12_two).mean() > 0.97
    assert np.isfinite(out2).all()
    e2.close()

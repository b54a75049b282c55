"""Round 5, GPU side."""
import contextlib
import io
import os

import numpy as np
import pytest

from interactive_deep_colorization_amd import api, caffe_io, engine, workloads
from oracle import siggraph_torch

from bounds import FP32_TOL, bf16_bound, check_bf16_ab  # noqa: F401

pytestmark = pytest.mark.gpu
CHAIN_DEFAULT = 2               # the library's default for "kwave_chain" (csrc/idc_engine.hip)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _caffe_style(sd):
    sd = {k: np.array(v, copy=True) for k, v in sd.items() if not k.startswith("model_class") and not k.endswith("num_batches_tracked")}
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k[:-len("running_mean")] + "weight"] = np.ones_like(sd[k])
            sd[k[:-len("running_mean")] + "bias"] = np.zeros_like(sd[k])
    return sd


def test_caffe_class_loads_a_caffemodel_with_the_reference_argument_order(tmp_path, make_sd):
    """VERDICT r4 item 6 / missing #1: `ColorizeImageCaffe(Xd).prep_net(gpu_id, prototxt_path, caffemodel_path)` with a real
    `.caffemodel` (data/colorize_image.py:392-403; ideepcolor.py:60-66 passes ./models/reference_model/model.caffemodel under its
    DEFAULT --backend caffe).  The file is written by caffe_io from Caffe-representable weights (BatchNorm without affine, conv1_1
    split into bw_conv1_1 + ab_conv1_1, the final Scale 100); the class must (a) load it, (b) produce what the same class produces
    from the equivalent state_dict, bit for bit up to the BatchNorm statistics' one float32 rounding through Caffe's scale_factor,
    and (c) agree with the oracle run the Caffe way (inputs L - 50, ab, mask * 110; output tanh * 100)."""
    sd = _caffe_style(make_sd(3, "he"))
    w0 = sd["model1.0.weight"]                  # a Caffe-trained conv1_1 sees raw L - 50, ab and mask * 110: keep the activations O(1)
    w0[:, 0] /= 100.0; w0[:, 1:3] /= 110.0; w0[:, 3] /= 110.0
    path = str(tmp_path / "model.caffemodel")
    caffe_io.write_caffemodel(path, caffe_io.state_dict_to_caffe_layers(sd, net="nodist", out_mul=100.0))
    rgb = np.load(os.path.join(REPO, "tests", "golden", "mortar_pestle_256_rgb.npy"))
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    outs = {}
    for how in ("caffemodel", "state_dict"):
        with contextlib.redirect_stdout(io.StringIO()):
            m = api.ColorizeImageCaffe(256)
            if how == "caffemodel":
                m.prep_net(0, "./models/reference_model/deploy_nodist.prototxt", path)          # the reference's positional order
            else:
                m.prep_net(0, state_dict=sd)
            m.set_image(rgb)
            out = m.net_forward(hab, hm)
        assert isinstance(out, np.ndarray) and out.shape == (256, 256, 3) and out.dtype == np.uint8
        outs[how] = (np.array(m.output_ab_raw, copy=True), out.copy(), np.array(m.img_l_mc, copy=True))
        m.net.close()
    assert np.abs(outs["caffemodel"][0] - outs["state_dict"][0]).max() <= 2e-3          # BN statistics differ by one fp32 rounding
    assert (outs["caffemodel"][1] != outs["state_dict"][1]).mean() <= 1e-3
    # (c) the Caffe conventions against the oracle run the Caffe way (l_div = ab_div = 1, mask * 110, head x 100)
    L_mc = outs["caffemodel"][2][None].astype(np.float32)
    ref = siggraph_torch.forward(sd, L_mc, hab[None].astype(np.float32), hm[None].astype(np.float32) * 110.0, 0.0,
                                 l_div=1., ab_div=1., out_mul=100.)
    assert np.abs(outs["caffemodel"][0][None] - ref).max() <= 3e-3, np.abs(outs["caffemodel"][0][None] - ref).max()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("style", ["torch", "he"])
def test_kwave_chain_equals_the_eleven_launches(make_sd, mode, style):
    """the 512 -> 512 trunk of the bf16 click forward (conv4_2 .. conv7_3 at batch 1) as ONE persistent launch
    (conv_kwave_chain_bf16: a grid barrier between layers instead of a launch floor; mode 1 = hipLaunchCooperativeKernel, 2 = plain launch
    after an occupancy check, the default).  Same arithmetic in the same order as the eleven conv_kwave_bf16 launches: the ab map and every trunk
    activation are IDENTICAL, call after call; 17 launches instead of 27."""
    sd = make_sd(0, style)
    L = workloads.random_batch(1, 256, seed=7)[0].astype(np.float32)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    ab, m = hab[None].astype(np.float32), hm[None].astype(np.float32)
    trunk = ["conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv7_3"]
    res = {}
    try:
        for chain in (0, mode):
            engine.set_option("kwave_chain", chain)
            e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
            e.load_state_dict(sd)
            outs = [e.forward(L, ab, m, 0.0) for _ in range(4)]
            rows = {r["name"]: r for r in e.layer_table()}
            acts = {k: e.activation(k, 1) for k in trunk + ["conv8_1"]}
            launches = sum(r["launches"] for r in rows.values())
            res[chain] = (outs, rows, acts, launches)
            e.close()
    finally:
        engine.set_option("kwave_chain", CHAIN_DEFAULT)
    (o0, rows0, a0, n0), (o1, rows1, a1, n1) = res[0], res[mode]
    assert [rows0[k]["kernel"] for k in trunk] == ["conv_kwave_bf16"] * 11
    assert rows1["conv4_2"]["kernel"] == "conv_kwave_chain_bf16 x11" and rows1["conv4_2"]["launches"] == 1, rows1["conv4_2"]
    assert [rows1[k]["kernel"] for k in trunk[1:]] == ["chained into conv4_2"] * 10 and all(rows1[k]["launches"] == 0 for k in trunk[1:])
    assert n0 - n1 == 10 and n1 <= 18, (n0, n1)
    for k in a0:
        np.testing.assert_array_equal(a1[k], a0[k], err_msg=k)
    for o in o1 + o0[1:]:
        np.testing.assert_array_equal(o, o0[0])
    ref = siggraph_torch.forward(sd, L, ab, m, 0.0)
    check_bf16_ab(o1[0] - ref, style, tag="chain mode %d" % mode)


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2", "net64_torch_s1_mc0"])
def test_kwave_chain_small_and_ragged_images(golden, make_sd, name):
    """The chain on the reference-generated goldens (64 x 64 and 32 x 48, batch 2): the trunk is 8 x 8 / 4 x 6 pixels there, the dilated
    layers need four times the workgroups of the others (one tile per parity), so a forward holds THREE chains (conv4_2-3, conv5_1-conv6_3,
    conv7_1-3) with different grids -- the barrier counters are re-based between them.  Identical to one launch per layer, batch == singles."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    sd = make_sd(seed, style)
    res = {}
    try:
        for chain in (0, 2):
            engine.set_option("kwave_chain", chain)
            e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
            e.load_state_dict(sd)
            outs = [e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])) for _ in range(3)]
            kernels = {r["name"]: r["kernel"] for r in e.layer_table()}
            acts = {k: e.activation(k, n) for k in ("conv4_3", "conv6_3", "conv7_3")}
            ones = [e.forward(g["L_mc"][i:i + 1], g["ab"][i:i + 1], g["mask"][i:i + 1], float(g["maskcent"]))[0] for i in range(n)]
            res[chain] = (outs, kernels, acts, ones)
            e.close()
    finally:
        engine.set_option("kwave_chain", CHAIN_DEFAULT)
    heads = [k for k, v in res[2][1].items() if v.startswith("conv_kwave_chain_bf16")]
    assert heads == ["conv4_2", "conv5_1", "conv7_1"], res[2][1]
    assert [res[2][1][k] for k in heads] == ["conv_kwave_chain_bf16 x2", "conv_kwave_chain_bf16 x6", "conv_kwave_chain_bf16 x3"]
    for o in res[2][0]:
        np.testing.assert_array_equal(o, res[0][0][0])
    for k in res[0][2]:
        np.testing.assert_array_equal(res[2][2][k], res[0][2][k], err_msg=k)
    for i in range(n):
        np.testing.assert_array_equal(res[2][3][i], res[2][0][0][i])
    check_bf16_ab(res[2][0][0] - g["out_ab"], style, tag=name)


def test_kwave_chain_is_not_taken_where_it_does_not_apply(make_sd):
    """Per-launch profiling wants eleven event pairs (the chain is one launch): it falls back; so does a handle whose trunk carries the
    Global-Hints shift in conv4_3's epilogue (the chain starts behind it), and a batch that needs more workgroups than the chip holds."""
    sd = make_sd(0, "torch")
    L, ab, m = workloads.random_batch(2, 256, seed=3)
    e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
    e.load_state_dict(sd)
    base = e.forward(L[:1], ab[:1], m[:1], 0.0)
    assert any(r["kernel"].startswith("conv_kwave_chain_bf16") for r in e.layer_table())
    e.set_profiling(True)
    prof = e.forward(L[:1], ab[:1], m[:1], 0.0)
    assert not any(r["kernel"].startswith("conv_kwave_chain_bf16") for r in e.layer_table())
    idx = [r["index"] for r in e.layer_table() if r["name"] == "conv5_2"][0]
    assert float(e.layer_times_ms()[idx]) > 0                   # its own event pair again
    e.set_profiling(False)
    np.testing.assert_array_equal(prof, base)
    np.testing.assert_array_equal(e.forward(L[:1], ab[:1], m[:1], 0.0), base)
    assert any(r["kernel"].startswith("conv_kwave_chain_bf16") for r in e.layer_table())
    e.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_reference_api_outputs_are_fetched_on_read(make_sd, precision):
    """net_forward returns the uint8 image; output_ab_raw / output_lab / output_ab -- which the reference fills on the
    host inside every call (colorize_image.py:263-267,196-198) -- are computed by the same device call and copied over when first READ.
    Same values as the eager three-output call (engine.forward_rgb), plain numpy arrays afterwards, each net_forward's attributes its own;
    assigning one makes it a plain attribute; a direct engine call in between first lets the object fetch what it has not read yet."""
    sd = make_sd(0, "torch")
    rgb = np.load(os.path.join(REPO, "tests", "golden", "mortar_pestle_256_rgb.npy"))
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    hab2 = hab.copy(); hab2[:, 40:47, 40:47] = 33.0
    hm2 = hm.copy(); hm2[:, 40:47, 40:47] = 1.0
    with contextlib.redirect_stdout(io.StringIO()):
        m = api.ColorizeImageTorch(Xd=256, precision=precision)
        m.prep_net(gpu_id=0, state_dict=sd)
        m.set_image(rgb)
    img1 = m.net_forward(hab, hm)
    assert m._out_pending == set(api._OUT_ATTRS)                     # nothing but the image has crossed PCIe
    L = m.img_l_mc[None].astype(np.float32)
    other = workloads.random_batch(1, 256, seed=2)
    m.net.forward(*other, 0.0)                                       # behind the object's back: the pending maps are fetched BEFORE they are replaced
    assert not m._out_pending
    raw_e, rgb_e, lab_e = m.net.forward_rgb(L, hab[None].astype(np.float32), hm[None].astype(np.float32), m.mask_cent, l_cent=50.0)
    np.testing.assert_array_equal(m.output_ab_raw, raw_e[0]); np.testing.assert_array_equal(m.output_ab, lab_e[0][1:])
    img1b = m.net_forward(hab, hm)
    np.testing.assert_array_equal(img1b, img1)
    np.testing.assert_array_equal(img1, rgb_e[0])
    ab1, lab1, raw1 = m.output_ab, m.output_lab, m.output_ab_raw
    assert not m._out_pending and isinstance(ab1, np.ndarray) and ab1.dtype == np.float64 and ab1.shape == (2, 256, 256)
    np.testing.assert_array_equal(raw1, raw_e[0]); np.testing.assert_array_equal(lab1, lab_e[0]); np.testing.assert_array_equal(ab1, lab_e[0][1:])
    assert m.output_ab is ab1                                        # cached: one fetch per forward
    img2 = m.net_forward(hab2, hm2)
    assert (img2 != img1).any()
    ab2 = m.output_ab
    assert (ab2 != ab1).any() and np.array_equal(ab1, lab_e[0][1:])  # the first call's array is the caller's to keep
    win = m.get_result_window(np.full((300, 280), 60.0))             # the display step still runs on the resident map
    assert win.shape == (300, 280, 3)
    m.net_forward(hab, hm)
    m.output_ab = np.zeros((2, 256, 256))                            # caller-supplied map (what _set_out_ab_ does): a plain attribute now
    assert not m._out_on_device() and (m.output_ab == 0).all()
    np.testing.assert_array_equal(m.output_ab_raw, raw_e[0])         # ... the other two are still fetched from the device
    m.net.close()


@pytest.mark.parametrize("shape", [(1, 256, 256), (1, 200, 232), (2, 128, 160)])
def test_deconv_shortcut_half_workgroups_on_small_grids(make_sd, shape):
    """conv_ds_fused_m<1> (round 5): when model10up + shortcut would launch fewer 128-cout workgroups than the chip has CUs (ONE 256x256 image: 128) it
    runs as 64-cout, 4-wave workgroups (256 of them) whose shortcut halo chunk goes to LDS by LDS-DMA.  Same MFMAs in the same order per accumulator:
    conv10_1 and the ab map bit-identical to the 8-wave form (`ds_mfma16` = 2), ragged tiles included."""
    n, H, W = shape
    sd = make_sd(0, "he")
    L, ab, m = workloads.random_batch(n, max(H, W), seed=6)
    L, ab, m = L[:, :, :H, :W].copy(), ab[:, :, :H, :W].copy(), m[:, :, :H, :W].copy()
    res = {}
    try:
        if (H, W) != (256, 256):
            engine.set_tile_policy("large")                   # small images hand model10up to conv_kwave_deconv_bf16: force the throughput family
        for mode in (2, 1):
            engine.set_option("ds_mfma16", mode)
            e = engine.HipColorizer(H, W, max_batch=n, precision="bf16")
            e.load_state_dict(sd)
            out = e.forward(L, ab, m, 0.0)
            kern = {r["name"]: r["kernel"] for r in e.layer_table()}
            res[mode] = (out, e.activation("conv10_1", n), kern["conv10_1"], e.forward(L, ab, m, 0.0))
            e.close()
    finally:
        engine.set_option("ds_mfma16", 1)
        engine.set_tile_policy("auto")
    assert res[1][2].startswith("conv_ds_fused_m"), res[1][2]
    np.testing.assert_array_equal(res[1][1], res[2][1])
    np.testing.assert_array_equal(res[1][0], res[2][0])
    np.testing.assert_array_equal(res[1][3], res[1][0])


def test_kwave_chain_gives_up_instead_of_hanging():
    """The safety net of the plain-launch default: a workgroup that never sees the others at the grid barrier (a partitioned / shared device; played here by
    the test hook `kw_force_abort` = 1 -- unreachable barrier, tiny poll budget) sets the host-visible flag and leaves; the blocking call that waited for that forward
    returns IDC_ERR_INTERNAL, the handle falls back to one launch per layer, and the next call gives the right answer.  Own process: the hook is process-wide."""
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from interactive_deep_colorization_amd import engine, workloads, _native
sd = workloads.random_state_dict(0, "torch")
L, ab, m = workloads.random_batch(1, 256, seed=3)
engine.set_option("kw_force_abort", 1)
e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
e.load_state_dict(sd)
try:
    e.forward(L, ab, m, 0.0)
    print("NO_ERROR")
except _native.IdcError as ex:
    print("ERROR:", str(ex)[:160].replace("\n", " "))
out = e.forward(L, ab, m, 0.0)
kern = [r["kernel"] for r in e.layer_table()]
print("CHAIN_AFTER:", any(k.startswith("conv_kwave_chain") for k in kern), "KW:", sum(k == "conv_kwave_bf16" for k in kern))
np.save(sys.argv[1], out)
e.close()
''' % REPO
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "out.npy")
        p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        assert "ERROR:" in p.stdout and "timed" in p.stdout and "NO_ERROR" not in p.stdout, p.stdout
        assert "CHAIN_AFTER: False KW: 22" in p.stdout, p.stdout
        got = np.load(path)
    engine.set_option("kwave_chain", 0)
    try:
        e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
        e.load_state_dict(workloads.random_state_dict(0, "torch"))
        L, ab, m = workloads.random_batch(1, 256, seed=3)
        np.testing.assert_array_equal(got, e.forward(L, ab, m, 0.0))
        e.close()
    finally:
        engine.set_option("kwave_chain", CHAIN_DEFAULT)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_click_with_resident_l_plane_equals_click_with_l_passed(make_sd, precision):
    """idc_forward_rgb_lazy(L_mc=NULL) = "the L plane idc_set_image_l left in the handle" (api.py uploads it once per image): same bits as passing L; NULL
    without a resident plane is refused; the wrapper uploads again when the engine was used with another L in between (engine.l_serial)."""
    sd = make_sd(0, "torch")
    rgb = np.load(os.path.join(REPO, "tests", "golden", "mortar_pestle_256_rgb.npy"))
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = api.ColorizeImageTorch(Xd=256, precision=precision)
        m.prep_net(gpu_id=0, state_dict=sd)
        with pytest.raises(RuntimeError, match="resident L"):
            m.net.forward_rgb_lazy(None, hab, hm, 0.5)                 # nothing uploaded yet
        m.set_image(rgb)
    e = m.net
    L = m.img_l_mc[None].astype(np.float32)
    want_rgb = e.forward_rgb_lazy(L, hab[None].astype(np.float32), hm[None].astype(np.float32), m.mask_cent, l_cent=50.0).copy()
    want_ab, want_lab = [a.copy() for a in e.fetch_outputs(1)]
    serial = e.l_serial
    for _ in range(3):
        np.testing.assert_array_equal(m.net_forward(hab, hm), want_rgb[0])
    assert e.l_serial == serial + 1 + 3                                # ONE set_image_l for the three clicks
    np.testing.assert_array_equal(m.output_ab_raw, want_ab[0]); np.testing.assert_array_equal(m.output_lab, want_lab[0])
    e.forward_rgb_lazy(np.full((1, 1, 256, 256), 11.0, np.float32), hab[None].astype(np.float32), hm[None].astype(np.float32), 0.5)   # another L in the slot
    np.testing.assert_array_equal(m.net_forward(hab, hm), want_rgb[0])
    np.testing.assert_array_equal(m.output_ab_raw, want_ab[0])
    e.close()


def test_blocking_wait_switch_gives_the_same_click():
    """Option `spin_sync` = 0 (a fresh process each) = the blocking hipStreamSynchronize of rounds 1-4 instead of the bounded poll the one-image
    calls use since round 5; `pcie_kernel` = 0 = the copy engines instead of pcie_copy_kernel for their transfers; same bytes in every combination.
    (Options since round 6: the default library reads no tuning knob from the environment.)"""
    import subprocess
    import sys
    code = """
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from interactive_deep_colorization_amd import engine, workloads
engine.set_option("spin_sync", int(sys.argv[1])); engine.set_option("pcie_kernel", int(sys.argv[2]))
e = engine.HipColorizer(64, 64, max_batch=1, precision="fp32")
e.load_state_dict(workloads.random_state_dict(0, "torch"))
L, ab, m = workloads.random_batch(1, 64, seed=3)
out = e.forward(L, ab, m, 0.0)
rgb = e.forward_rgb_lazy(L, ab, m, 0.0)
oab, lab = e.fetch_outputs(1)
print("SUM", hashlib.sha1(out.tobytes() + rgb.tobytes() + oab.tobytes() + lab.tobytes()).hexdigest())
""" % REPO
    sums = []
    for spin, by_kernel in (("0", "0"), ("1", "1"), ("1", "0")):          # ... and pcie_kernel = 0: every transfer through hipMemcpyAsync, as before round 5
        r = subprocess.run([sys.executable, "-c", code, spin, by_kernel], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        sums.append([l for l in r.stdout.splitlines() if l.startswith("SUM")][0])
    assert sums[0] == sums[1] == sums[2]

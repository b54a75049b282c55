"""Round-2 GPU tests: the batch-1 click kernel, the transfer pipeline, stream ordering, device-blob adoption and the
RCCL broadcast entry point, the display / full-resolution step, the weight-FILE loader, wrapper state after prep_net.
Everything goes through the C ABI (ctypes); the oracle is only the checker."""
import collections
import os

import numpy as np
import pytest
import torch

from interactive_deep_colorization_amd import _native as N
from interactive_deep_colorization_amd import api, engine, workloads
from oracle import colorspace as ocs
from oracle import display, siggraph_torch

from bounds import FP32_TOL, bf16_bound, check_bf16_ab  # noqa: F401

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(autouse=True)
def _reset():
    yield
    engine.set_option("click", -1)
    engine.set_option("winograd", 1)
    engine.set_option("kwave", 1)
    engine.set_option("fuse_conv1", 1)
    engine.set_tile_policy("auto")
    engine.set_splitk_policy("auto")


# ------------------------------------------------------------------------------------------------ conv_click
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_click_kernel_whole_network(golden, make_sd, precision):
    """BASELINE configs[1] (one 256x256 image, 5 hints) runs on conv_click by default; it computes the same network
    as conv_igemm (click off) and sits at the usual distance from the reference golden."""
    g = golden("config2_mortar_5hints_torchinit")
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    engine.set_option("winograd", 0)          # (the fp32 default runs the 3x3 stride-1 layers as Winograd: tests/test_round3_gpu.py)
    engine.set_option("kwave", 0)             # (... and the bf16 default, round 4, as conv_kwave_bf16: tests/test_round4_gpu.py)
    e = engine.HipColorizer(256, 256, max_batch=1, precision=precision)
    e.load_state_dict(make_sd(seed, style))
    out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    kernels = [r["kernel"] for r in e.layer_table() if r["launches"] > 0 and r["kernel"].startswith("conv")]
    assert sum(k.startswith("conv_click") for k in kernels) >= 20, kernels
    np.testing.assert_array_equal(e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])), out)    # deterministic
    engine.set_option("click", 0)
    base = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
    assert not any(r["kernel"].startswith("conv_click") for r in e.layer_table())
    d = np.abs(out - g["out_ab"])
    if precision == "fp32":
        assert d.max() <= 1e-3, d.max()
        assert np.abs(out - base).max() <= 5e-4          # different split of K: summation-order noise only
    else:
        check_bf16_ab(d, "torch")
        assert np.abs(out - base).max() <= bf16_bound("torch")[0]
    e.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net32x48_he_s2"])
def test_click_kernel_layer_by_layer(golden, make_sd, name, precision):
    """Small / ragged geometries (64x64 batch 2, 32x48): every layer that runs on conv_click against the float64 oracle,
    with K split as far as it goes and not at all."""
    g = golden(name)
    style, seed = str(g["weight_style"]), int(g["weight_seed"])
    n, _, H, W = g["L_mc"].shape
    _, _, acts = siggraph_torch.forward(make_sd(seed, style), g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]),
                                        return_acts=True, dtype=torch.float64)
    engine.set_option("winograd", 0)          # conv_click is what this test is about (fp32 default: Winograd, test_round3_gpu.py)
    engine.set_option("kwave", 0)             # (bf16 default since round 4: conv_kwave_bf16 / conv_kwave_deconv_bf16, test_round4_gpu.py)
    e = engine.HipColorizer(H, W, max_batch=n, precision=precision)
    e.load_state_dict(make_sd(seed, style))
    for sk in ("auto", "always", "never"):
        engine.set_splitk_policy(sk)
        out = e.forward(g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]))
        table = e.layer_table()
        clicked = [r["name"] for r in table if r["kernel"].startswith("conv_click")]
        assert len(clicked) >= 20, (sk, clicked)
        if sk == "never":
            assert not any("splitK" in r["kernel"] for r in table)
        if sk == "always":
            assert sum("splitK" in r["kernel"] for r in table) >= 15
        for k in clicked:
            got = e.activation(k, n)
            ref = acts[k]
            tol = 2e-4 * (1 + np.abs(ref).max()) if precision == "fp32" else 0.04 * (1 + np.abs(ref).max())
            err = np.abs(got - ref).max()
            assert err <= tol, "layer %s (split-K %s): max-abs err %.3e" % (k, sk, err)
        d = np.abs(out - g["out_ab"])
        assert d.max() <= (3e-3 if precision == "fp32" else bf16_bound("he")[0])
    e.close()


# ------------------------------------------------------------------------------------------------ transfer pipeline
def test_forward_async_pipeline_equals_blocking(make_sd):
    """Two-slot overlapped transfers: six batches through idc_forward_async / idc_wait, pinned and pageable host
    buffers, equal the blocking idc_forward bit for bit."""
    e = engine.HipColorizer(64, 64, max_batch=4, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    batches = [workloads.random_batch(4 if i % 3 else 3, 64, seed=40 + i) for i in range(6)]
    ref = [e.forward(*b, 0.0) for b in batches]
    for pinned in (True, False):
        outs = [None] * len(batches)
        bufs = []
        for i, (L, ab, m) in enumerate(batches):
            if pinned:
                arrs = [e.pinned_empty(x.shape) for x in (L, ab, m)] + [e.pinned_empty((L.shape[0], 2, 64, 64))]
                for dst, src in zip(arrs[:3], (L, ab, m)):
                    dst[...] = src
            else:
                arrs = [L.copy(), ab.copy(), m.copy(), np.empty((L.shape[0], 2, 64, 64), np.float32)]
            bufs.append(arrs)
        for i in range(len(batches)):
            slot = i & 1
            if i >= 2:
                e.wait(slot)
                outs[i - 2] = bufs[i - 2][3].copy()
            e.forward_async(slot, bufs[i][0], bufs[i][1], bufs[i][2], bufs[i][3], 0.0)
        e.wait(0); e.wait(1)
        outs[-2] = bufs[-2][3].copy(); outs[-1] = bufs[-1][3].copy()
        for o, r in zip(outs, ref):
            np.testing.assert_array_equal(o, r)
    # a slot in flight must be waited for; the blocking call drains the pipeline by itself
    e.forward_async(0, *bufs[0], 0.0)
    with pytest.raises(N.IdcError):
        e.forward_async(0, *bufs[1], 0.0)
    np.testing.assert_array_equal(e.forward(*batches[2], 0.0), ref[2])
    np.testing.assert_array_equal(bufs[0][3], ref[0])
    e.close()


def test_stream_wait_and_signal_order_a_torch_stream(make_sd):
    """Inputs produced and outputs consumed on a torch side stream, ordered against the handle's stream by events only."""
    e = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    L, ab, m = workloads.random_batch(2, 64, seed=9)
    ref = e.forward(L, ab, m, 0.0)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    hL, hab, hm = (torch.from_numpy(x).pin_memory() for x in (L, ab, m))
    for _ in range(3):
        with torch.cuda.stream(side):
            dL = hL.to(dev, non_blocking=True) * 1.0
            dab = hab.to(dev, non_blocking=True) * 1.0
            dm = hm.to(dev, non_blocking=True) * 1.0
            dout = torch.empty((2, 2, 64, 64), dtype=torch.float32, device=dev)
        e.stream_wait(side.cuda_stream)
        e.forward_device(2, dL, dab, dm, dout, 0.0, sync=False)
        e.stream_signal(side.cuda_stream)
        with torch.cuda.stream(side):
            got = (dout + 0.0).cpu()
        side.synchronize()
        np.testing.assert_array_equal(got.numpy(), ref)
    e.close()


# ------------------------------------------------------------------------------------------------ weights on the device
def test_set_weights_device_adopt_and_copy(make_sd):
    """The multi-GPU adoption path on one GPU: a torch-owned uint8 device tensor holding the packed blob (what the RCCL
    broadcast leaves) adopted in place (copy=0) or copied (copy=1) gives the outputs of load_state_dict; a corrupted
    payload is refused by the checksum."""
    sd = make_sd(1, "torch")
    L, ab, m = workloads.random_batch(2, 64, seed=3)
    e0 = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
    e0.load_state_dict(sd)
    ref = e0.forward(L, ab, m, 0.5)
    blob = engine.pack_weights(sd, "bf16")
    dev = torch.device("cuda", 0)
    for copy in (False, True):
        t = torch.from_numpy(blob).to(dev)
        e = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
        e.set_weights_device(t.data_ptr(), t.numel(), copy=copy, keepalive=t)
        if copy:
            t.zero_()                                         # the handle owns its own copy
            torch.cuda.synchronize(dev)
        np.testing.assert_array_equal(e.forward(L, ab, m, 0.5), ref)
        e.close()
    bad = torch.from_numpy(blob).to(dev)
    bad[len(blob) // 2] ^= 0x40
    e = engine.HipColorizer(64, 64, max_batch=2, precision="bf16")
    with pytest.raises(N.IdcError) as ei:
        e.set_weights_device(bad.data_ptr(), bad.numel(), copy=False, keepalive=bad)
    assert "checksum" in str(ei.value)
    e.close(); e0.close()


def test_broadcast_weights_through_the_c_abi_world1(make_sd):
    """idc_comm_unique_id + idc_broadcast_weights (librccl opened by the library itself): a one-rank communicator on
    this GPU -- ncclCommInitRank, ncclBroadcast of the 68 MB blob on the handle's stream, ncclCommDestroy.  (Two ranks
    cannot share one GPU under RCCL; the N > 1 control flow is covered over gloo in test_sharded_gloo.py and by the
    single-GPU dry run of bench.py.)"""
    sd = make_sd(1, "torch")
    L, ab, m = workloads.random_batch(1, 64, seed=4)
    e = engine.HipColorizer(64, 64, max_batch=1, precision="bf16")
    with pytest.raises(N.IdcError):                           # the root needs weights
        e.broadcast_weights(e.comm_unique_id(), 0, 1, 0)
    e.load_state_dict(sd)
    ref = e.forward(L, ab, m, 0.0)
    uid = e.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e.broadcast_weights(uid, 0, 1, 0)
    np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), ref)
    with pytest.raises(N.IdcError):
        e.broadcast_weights(uid, 2, 1, 0)                      # rank outside the world
    e.close()


# ------------------------------------------------------------------------------------------------ display step
def test_upsample_lab2rgb_display_and_fullres(make_sd):
    """idc_upsample_lab2rgb against the CPU restatements: cv2 INTER_CUBIC display resize (unpinned restatement) and
    scipy.ndimage.zoom order 1 / 0 (pinned: scipy itself), each followed by skimage's lab2rgb formulas; at most one
    uint8 level on at most 0.02 % of the values (float64 on both sides, pow/cbrt differ in the last bit)."""
    from scipy.ndimage import zoom
    rgb = np.load(os.path.join(HERE, "golden", "mortar_pestle_256_rgb.npy"))
    model = api.ColorizeImageTorch(Xd=256, precision="bf16")
    model.prep_net(path="", state_dict=make_sd(0, "he"))
    model.set_image(rgb)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    model.net_forward(hab, hm)
    out_ab = np.array(model.output_ab)                        # refreshed, float64
    rs = np.random.RandomState(5)

    def close(a, b, frac=2e-4):
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() <= frac, (d.max(), (d > 0).mean())

    for (wh, ww) in ((345, 410), (256, 256), (180, 200)):          # (512 x 512 dropped in round 5: 5 s of numpy cubic checker for a third up-scale case)
        l_win = rs.uniform(0, 100, (wh, ww))
        close(model.get_result_window(l_win), display.display_rgb(out_ab, l_win))
    for (fh, fw) in ((507, 600), (256, 256), (300, 280)):
        Lf = rs.uniform(0, 100, (fh, fw))
        ab_lin = zoom(out_ab, (1, 1. * fh / 256, 1. * fw / 256), order=1)
        close(model.net.upsample_lab2rgb(Lf, "output_ab", "linear"), ocs.lab2rgb_transpose(Lf[None], ab_lin))
        raw_near = zoom(model.output_ab_raw.astype(np.float64), (1, 1. * fh / 256, 1. * fw / 256), order=0)
        close(model.net.upsample_lab2rgb(Lf, "output_ab_raw", "nearest"), ocs.lab2rgb_transpose(Lf[None], raw_near))
    # the wrapper's full-resolution getter takes the device route and matches its own host route
    dev_img = model.get_img_fullres()
    model.output_ab = np.array(model.output_ab)               # a replaced attribute: host route
    close(dev_img, model.get_img_fullres())
    # edit-list path: the hint planes exist on the device only
    hints = [(100 + 7 * i, 60 + 9 * i, 106 + 7 * i, 66 + 9 * i, 30 * i % 256, 200 - 20 * i, 40 + 15 * i) for i in range(5)]
    model.net_forward_hints(hints)
    dev_in = model.get_input_img_fullres(); dev_sup = model.get_sup_fullres()     # (reading input_mask pulls the planes back)
    ab_in, mask_in = model.input_ab, model.input_mask         # read back: host route from here on
    close(dev_in, ocs.lab2rgb_transpose(model.img_l_fullres, zoom(ab_in, (1, 1., 1.), order=1)))
    close(dev_sup, ocs.lab2rgb_transpose(50 * mask_in, ab_in))
    model.net.close()


def test_resident_forward_needs_an_image_and_prep_net_resets_the_session(make_sd):
    """a fresh handle has no L plane (idc_forward_resident refuses instead of reading uninitialised memory);
    prep_net swaps the engine handle, so the wrapper uploads the L plane again."""
    e = engine.HipColorizer(64, 64, max_batch=1, precision="bf16")
    e.load_state_dict(make_sd(0, "he"))
    with pytest.raises(N.IdcError) as ei:
        e.forward_resident(1)
    assert "image" in str(ei.value)
    ab0, m0 = e.hint_planes(0)
    assert not ab0.any() and not m0.any()                     # planes start zeroed = "no hints"
    e.close()
    rgb = np.load(os.path.join(HERE, "golden", "mortar_pestle_256_rgb.npy"))[:64, :64].copy()
    hints = [(10, 12, 16, 18, 200, 30, 40), (40, 40, 44, 44, 10, 220, 90)]
    model = api.ColorizeImageTorch(Xd=64, precision="bf16")
    model.prep_net(path="", state_dict=make_sd(0, "he"))
    model.set_image(rgb)
    first = model.net_forward_hints(hints).copy()
    model.prep_net(path="", state_dict=make_sd(0, "he"))      # new handle, same image: L must be re-uploaded
    second = model.net_forward_hints(hints)
    np.testing.assert_array_equal(first, second)
    model.net.close()


# ------------------------------------------------------------------------------------------------ weight file
def _write_reference_style_pth(path, keys, metadata_keys, dtypes, sd):
    """A .pth with the structure torch.save(net.state_dict()) has for the reference module: OrderedDict in module
    order, int64 num_batches_tracked entries, a _metadata attribute (version records) -- values from the seeded
    generator (the real 137 MB file is not a fixture; oracle/make_golden_pth.py recorded its structure)."""
    od = collections.OrderedDict()
    for k, dt in zip(keys, dtypes):
        v = np.asarray(sd[str(k)])
        od[str(k)] = torch.from_numpy(v.astype(np.int64 if "int64" in str(dt) else np.float32))
    od._metadata = collections.OrderedDict((str(k), {"version": 1}) for k in metadata_keys)
    torch.save(od, path)


@pytest.mark.parametrize("dist", [False, True])
def test_prep_net_from_a_pth_file(golden, make_sd, tmp_path, dist):
    """data/colorize_image.py:216-233 end to end: ColorizeImageTorch(.Dist).prep_net(path=...) reads a torch-saved
    state_dict (with _metadata, num_batches_tracked, model_class.*), and net_forward reproduces what the REFERENCE
    module returned for the same file contents (fixture written by oracle/make_golden_pth.py)."""
    g = golden("pth64_torch_s3")
    tag = "dist" if dist else "reg"
    sd = make_sd(int(g["weight_seed"]), str(g["weight_style"]))
    path = str(tmp_path / "caffemodel.pth")
    _write_reference_style_pth(path, g["keys_" + tag], g["metadata_keys_" + tag], g["dtypes_" + tag], sd)
    loaded = api.read_state_dict(path)
    assert not hasattr(loaded, "_metadata") and list(loaded.keys()) == [str(k) for k in g["keys_" + tag]]
    if dist:
        model = api.ColorizeImageTorchDist(Xd=64, maskcent=True, precision="fp32")
        model.prep_net(path=path, dist=True)
    else:
        model = api.ColorizeImageTorch(Xd=64, precision="fp32")
        model.prep_net(path=path)
    model.set_image(g["rgb"])
    ret = model.net_forward(g["input_ab"], g["input_mask"])
    if dist:
        # the reference's dist forward hands back out_reg*110*110 (model.py:164-166); so does the wrapper
        assert np.abs(ret - g["out_dist"]).max() <= 1e-3 * 110
        assert np.abs(model.output_ab_raw - g["out_dist"] / 110.0).max() <= 1e-3
        assert np.abs(model.dist_ab[:, ::4, ::4] - g["class_probs_lowres"]).max() <= 2e-4
    else:
        assert ret.shape == (64, 64, 3) and ret.dtype == np.uint8
        assert np.abs(model.output_ab_raw - g["out_reg"]).max() <= 1e-3
    model.net.close()

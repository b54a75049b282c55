"""CPU tests of the C-ABI boundary: the library loads, exports every symbol of include/ideepcolor.h,
refuses to run without a gfx950 device, and packs weights into exactly the documented layout."""
import ctypes
import os
import re

import numpy as np
import pytest

from interactive_deep_colorization_amd import _native as N
from interactive_deep_colorization_amd import api, engine
from oracle import weights as oweights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = N.load()
    header = open(os.path.join(REPO, "include", "ideepcolor.h")).read()
    declared = set(re.findall(r"\b(idc_[a-z0-9_]+)\s*\(", header))
    declared -= {"idc_status", "idc_precision"}
    assert declared == set(N.EXPORTED_SYMBOLS), declared ^ set(N.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), "libideepcolor_hip.so does not export %s" % sym
    assert lib.idc_version() == 2


@pytest.mark.skipif(N.load().idc_device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback():
    """Without a device the product path fails loudly -- it never computes on the CPU."""
    lib = N.load()
    h = ctypes.c_void_p()
    assert lib.idc_create(0, 256, 256, 1, N.IDC_BF16, 0, ctypes.byref(h)) == -2      # IDC_ERR_NO_DEVICE
    assert b"no HIP device" in lib.idc_last_error(None)
    with pytest.raises(N.IdcError):
        engine.HipColorizer(64, 64)
    with pytest.raises(N.IdcError):
        engine.op_conv2d(np.zeros((1, 32, 8, 8)), np.zeros((64, 32, 3, 3)), np.zeros(64))
    m = api.ColorizeImageTorch(Xd=64)
    with pytest.raises(N.IdcError):
        m.prep_net(state_dict=oweights.make_state_dict(0, "torch"))
    assert lib.idc_create(0, 250, 256, 1, 0, 0, ctypes.byref(h)) == -1               # H % 8 != 0


def test_api_guards_match_reference(capsys):
    """net_forward returns -1 and prints, like data/colorize_image.py:85-90."""
    m = api.ColorizeImageTorch(Xd=32)
    assert m.net_forward(np.zeros((2, 32, 32)), np.zeros((1, 32, 32))) == -1
    assert "I need to have an image!" in capsys.readouterr().out
    m.set_image(np.random.RandomState(0).randint(0, 256, (32, 32, 3)).astype(np.uint8))
    assert m.img_l_mc.shape == (1, 32, 32) and abs((m.img_l - 50.0) - m.img_l_mc).max() < 1e-12
    assert m.net_forward(np.zeros((2, 32, 32)), np.zeros((1, 32, 32), bool)) == -1
    assert "I need to have a net!" in capsys.readouterr().out
    assert m.get_img_gray().shape == (32, 32, 3) and m.get_img_gray().dtype == np.uint8
    d = api.ColorizeImageTorchDist(Xd=32)
    assert d.get_ab_reccs(3, 3) == 0 and d.pts_grid.shape == (529, 2)
    assert tuple(d.pts_grid[1]) == (-100, -110)          # a varies fastest
    c = api.ColorizeImageCaffe(Xd=32)
    assert c.mask_mult == 110.


# ---- packed-blob layout, re-derived independently of the C++ packer --------------------------------
LAYERS = [  # (wkey, bnkey, kind, cin, cout)   order of interactive_deep_colorization_amd/csrc/idc_net.h
    ("model1.0", None, "im2col", 4, 64), ("model1.2", "model1.4", "c3", 64, 64),
    ("model2.0", None, "c3", 64, 128), ("model2.2", "model2.4", "c3", 128, 128),
    ("model3.0", None, "c3", 128, 256), ("model3.2", None, "c3", 256, 256), ("model3.4", "model3.6", "c3", 256, 256),
    ("model4.0", None, "c3", 256, 512), ("model4.2", None, "c3", 512, 512), ("model4.4", "model4.6", "c3", 512, 512),
    ("model5.0", None, "c3", 512, 512), ("model5.2", None, "c3", 512, 512), ("model5.4", "model5.6", "c3", 512, 512),
    ("model6.0", None, "c3", 512, 512), ("model6.2", None, "c3", 512, 512), ("model6.4", "model6.6", "c3", 512, 512),
    ("model7.0", None, "c3", 512, 512), ("model7.2", None, "c3", 512, 512), ("model7.4", "model7.6", "c3", 512, 512),
    ("model3short8.0", None, "c3", 256, 256), ("model8up.0", None, "dc", 512, 256),
    ("model8.1", None, "c3", 256, 256), ("model8.3", "model8.5", "c3", 256, 256),
    ("model_class.0", None, "c1", 256, 529),
    ("model2short9.0", None, "c3", 128, 128), ("model9up.0", None, "dc", 256, 128),
    ("model9.1", "model9.3", "c3", 128, 128),
    ("model1short10.0", None, "c3", 64, 128), ("model10up.0", None, "dc", 128, 128),
    ("model10.1", None, "c3", 128, 128),
]


def _al(v, a=256):
    return (v + a - 1) // a * a


def _plan(precision, dist):
    from conftest import has_ab_partners
    partners = has_ab_partners()          # round 6: the layout-2 images (conv_igemm_v2's) exist only in the -DIDC_AB_PARTNERS build; default bf16 blob 68 MB
    kc = 64 if precision == "bf16" else 32
    off, plan = 64, []
    for wkey, bnkey, kind, cin, cout in LAYERS:
        if wkey == "model_class.0" and not dist:
            continue
        cpad = 64 if cout <= 64 else _al(cout, 128)
        nkc = (64 // kc) if kind == "im2col" else -(-cin // kc)
        ntap = {"c3": 9, "dc": 16, "c1": 1, "im2col": 1}[kind]
        e = dict(wkey=wkey, bnkey=bnkey, kind=kind, cin=cin, cout=cout, cpad=cpad, nkc=nkc, ncg=cpad // 64, ntap=ntap)
        off = _al(off); e["w_off"] = off; off += ntap * nkc * e["ncg"] * 8192
        if partners and precision == "bf16" and kind != "im2col" and cpad >= 128:   # second image: layout 2 (conv_igemm_v2)
            off = _al(off); e["w2_off"] = off; off += ntap * nkc * e["ncg"] * 8192
        if kind == "c3" and precision == "fp32":
            # third image, fp32 blob only (round 5: the bf16 click path's Winograd kernels were retired, a bf16 blob is 68 MB, 136 MB in the partner build):
            # Winograd F(2x2,3x3) weights U = G g G^T of the 3x3 stride-1 layers (idc_wino.hip)
            off = _al(off); e["w3_off"] = off; off += cin * cpad * 16 * 4
        if kind == "dc" and precision == "fp32":     # deconvs: Winograd F(2x2,2x2) over the four phases, 36 values per (cin, cout)
            off = _al(off); e["w3d_off"] = off; off += cin * cpad * 36 * 4
        off = _al(off); e["b_off"] = off; off += cpad * 4
        if bnkey:
            off = _al(off); e["s_off"] = off; off += cpad * 4
            off = _al(off); e["t_off"] = off; off += cpad * 4
        if kind == "dc":           # deconv layers sum a shortcut conv: bias of the fused launch = own + shortcut bias
            off = _al(off); e["fb_off"] = off; off += cpad * 4
        plan.append(e)
    off = _al(off); head_w = off; off += 1024
    off = _al(off); head_b = off; off += 8
    return plan, head_w, head_b, _al(off)


def _bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)


def _read_w(blob, e, precision, tw, co, k, layout=1):
    """Element (tap tw, cout co, K index k) of a layer's packed image, per idc_layout.h."""
    eb = 2 if precision == "bf16" else 4
    kc_e, eps = 128 // eb, 16 // eb
    kc, kin = divmod(k, kc_e)
    s, el = divmod(kin, eps)
    cg, col = divmod(co, 64)
    if layout == 1:      # 16x16 MFMA D layout: row ci*16 + g*4 + reg holds cout g*16 + ci*4 + reg; swizzle row&7
        g, ci, reg = col >> 4, (col >> 2) & 3, col & 3
        lam = ci * 16 + g * 4 + reg
        sw = lam & 7
    else:                # 32x32 MFMA D layout: row mi*32 + (r>>2)*8 + hh*4 + (r&3) holds cout hh*32 + mi*16 + r
        hh, mi, r = col >> 5, (col >> 4) & 1, col & 15
        lam = mi * 32 + (r >> 2) * 8 + hh * 4 + (r & 3)
        sw = (lam >> 1) & 7
    base = e["w_off"] if layout == 1 else e["w2_off"]
    off = base + ((tw * e["nkc"] + kc) * e["ncg"] + cg) * 8192 + lam * 128 + ((s ^ sw) * 16) + el * eb
    if precision == "bf16":
        return int(blob[off:off + 2].view(np.uint16)[0])
    return float(blob[off:off + 4].view(np.float32)[0])


@pytest.mark.parametrize("precision,dist", [("bf16", False), ("fp32", True)])
def test_pack_weights_layout(make_sd, precision, dist):
    sd = make_sd(0, "he")
    blob = engine.pack_weights(sd, precision, dist=dist)
    plan, head_w, head_b, total = _plan(precision, dist)
    assert blob.size == total == N.load().idc_weights_blob_bytes(1 if precision == "bf16" else 0, 1 if dist else 0)
    hdr = blob[:64]
    assert hdr[:4].view(np.uint32)[0] == 0x43444931 and hdr[8:12].view(np.uint32)[0] == (1 if precision == "bf16" else 0)
    assert int(hdr[16:24].view(np.uint64)[0]) == total
    rs = np.random.RandomState(0)
    for e in plan:
        w = sd[e["wkey"] + ".weight"]
        for _ in range(40):
            co, ci = rs.randint(e["cout"]), rs.randint(e["cin"])
            if e["kind"] == "c3":
                ky, kx = rs.randint(3), rs.randint(3); tw, k, val = ky * 3 + kx, ci, w[co, ci, ky, kx]
            elif e["kind"] == "im2col":
                ky, kx = rs.randint(3), rs.randint(3); tw, k, val = 0, (ky * 3 + kx) * 4 + ci, w[co, ci, ky, kx]
            elif e["kind"] == "c1":
                tw, k, val = 0, ci, w[co, ci, 0, 0]
            else:                                           # ConvTranspose weight is (Cin, Cout, 4, 4)
                ky, kx = rs.randint(4), rs.randint(4); tw, k, val = ky * 4 + kx, ci, w[ci, co, ky, kx]
            got = _read_w(blob, e, precision, tw, co, k)
            if precision == "bf16":
                assert got == int(_bf16_bits(np.float32(val)).ravel()[0]), (e["wkey"], co, ci)
                if "w2_off" in e:
                    assert _read_w(blob, e, precision, tw, co, k, layout=2) == got, (e["wkey"], co, ci, "layout 2")
            else:
                assert got == float(val), (e["wkey"], co, ci)
            if "w3_off" in e:
                # U[i][j] of (co, ci) in MFMA A-operand order [chunk = ci/32][pos = i*4+j][co/16][ks][lane = g*16 + co%16][e]
                G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
                U = G @ w[co, ci].astype(np.float64) @ G.T
                i, j = rs.randint(4), rs.randint(4)
                kc3, eps3 = (64, 8) if precision == "bf16" else (32, 4)        # channels per 128-byte chunk, elements per 16-byte slot
                c, within = divmod(ci, kc3)
                slot, el = divmod(within, eps3)
                ks, gq = divmod(slot, 4)
                idx = ((((c * 16 + i * 4 + j) * (e["cpad"] // 16) + co // 16) * 2 + ks) * 64 + gq * 16 + co % 16) * eps3 + el
                if precision == "bf16":
                    o3 = e["w3_off"] + idx * 2
                    assert int(blob[o3:o3 + 2].view(np.uint16)[0]) == int(_bf16_bits(np.float32(U[i, j])).ravel()[0]), (e["wkey"], co, ci, i, j)
                else:
                    o3 = e["w3_off"] + idx * 4
                    assert float(blob[o3:o3 + 4].view(np.float32)[0]) == np.float32(U[i, j]), (e["wkey"], co, ci, i, j)
            if "w3d_off" in e:
                # phase (r,s): g[a][b] = W[ci][co][KY[r][a]][KY[s][b]], KY = ((3,1),(2,0)); U = G g G^T, G = [[1,0],[1,1],[0,1]]
                KY = ((3, 1), (2, 0))
                G2 = np.array([[1, 0], [1, 1], [0, 1]], np.float64)
                r_, s_, i, j = rs.randint(2), rs.randint(2), rs.randint(3), rs.randint(3)
                g = np.array([[w[ci, co, KY[r_][a_], KY[s_][b_]] for b_ in (0, 1)] for a_ in (0, 1)], np.float64)
                U = G2 @ g @ G2.T
                kc3, eps3 = (64, 8) if precision == "bf16" else (32, 4)
                c, within = divmod(ci, kc3)
                slot, el = divmod(within, eps3)
                ks, gq = divmod(slot, 4)
                pidx = ((r_ * 2 + s_) * 3 + i) * 3 + j
                idx = ((((c * 36 + pidx) * (e["cpad"] // 16) + co // 16) * 2 + ks) * 64 + gq * 16 + co % 16) * eps3 + el
                if precision == "bf16":
                    o3 = e["w3d_off"] + idx * 2
                    assert int(blob[o3:o3 + 2].view(np.uint16)[0]) == int(_bf16_bits(np.float32(U[i, j])).ravel()[0]), (e["wkey"], co, ci)
                else:
                    o3 = e["w3d_off"] + idx * 4
                    assert float(blob[o3:o3 + 4].view(np.float32)[0]) == np.float32(U[i, j]), (e["wkey"], co, ci, r_, s_, i, j)
        np.testing.assert_array_equal(blob[e["b_off"]:e["b_off"] + e["cout"] * 4].view(np.float32), sd[e["wkey"] + ".bias"])
        if e["bnkey"]:
            g_, b_ = sd[e["bnkey"] + ".weight"].astype(np.float64), sd[e["bnkey"] + ".bias"].astype(np.float64)
            mu, var = sd[e["bnkey"] + ".running_mean"].astype(np.float64), sd[e["bnkey"] + ".running_var"].astype(np.float64)
            s = g_ / np.sqrt(var + 1e-5)
            np.testing.assert_allclose(blob[e["s_off"]:e["s_off"] + e["cout"] * 4].view(np.float32), s, rtol=1e-6)
            np.testing.assert_allclose(blob[e["t_off"]:e["t_off"] + e["cout"] * 4].view(np.float32), b_ - mu * s, rtol=1e-5, atol=1e-6)
    for dc, short in (("model8up.0", "model3short8.0"), ("model9up.0", "model2short9.0"), ("model10up.0", "model1short10.0")):
        e = [x for x in plan if x["wkey"] == dc][0]
        np.testing.assert_array_equal(blob[e["fb_off"]:e["fb_off"] + e["cout"] * 4].view(np.float32),
                                      sd[dc + ".bias"] + sd[short + ".bias"])
    np.testing.assert_array_equal(blob[head_w:head_w + 1024].view(np.float32), sd["model_out.0.weight"].ravel())
    np.testing.assert_array_equal(blob[head_b:head_b + 8].view(np.float32), sd["model_out.0.bias"])
    # the im2col operand of conv1_1 is 64 wide: K slots 36..63 are zero padding
    e0 = plan[0]
    assert all(_read_w(blob, e0, precision, 0, 5, k) == 0 for k in range(36, 64))


def test_pack_weights_errors(make_sd):
    sd = dict(make_sd(0, "he"))
    missing = {k: v for k, v in sd.items() if k != "model7.2.weight"}
    with pytest.raises(N.IdcError) as ei:
        engine.pack_weights(missing, "bf16")
    assert ei.value.status == -5 and "model7.2.weight" in str(ei.value)
    bad = dict(sd); bad["model8up.0.weight"] = np.zeros((256, 512, 4, 4), np.float32)    # Conv2d-style shape: wrong
    with pytest.raises(N.IdcError):
        engine.pack_weights(bad, "fp32")
    no_class = {k: v for k, v in sd.items() if not k.startswith("model_class")}
    engine.pack_weights(no_class, "bf16", dist=False)                  # model_class is optional without dist
    with pytest.raises(N.IdcError):
        engine.pack_weights(no_class, "bf16", dist=True)
    a = engine.pack_weights(sd, "bf16"); b = engine.pack_weights(sd, "bf16")
    assert np.array_equal(a, b)                                        # deterministic bytes (checksummed header)


def test_pack_global_hints_section(make_sd):
    """IDC_FLAG_GLOBAL_HINTS appends the branch parameters (fp32, transposed [k][512] + bias / BN scale / shift
    per stage); missing glob.* keys are IDC_ERR_MISSING_KEY."""
    from oracle import weights
    sd = weights.add_global_branch(dict(make_sd(0, "he")), 0)
    plain = engine.pack_weights(sd, "fp32")
    blob = engine.pack_weights(sd, "fp32", global_hints=True)
    n_f = 316 * 512 + 3 * 512 + 3 * (512 * 512 + 3 * 512)
    off = (plain.size + 255) // 256 * 256
    assert blob.size == (off + n_f * 4 + 255) // 256 * 256
    sec = blob[off:off + n_f * 4].view(np.float32)
    wg, ws = sd["glob.glob_conv1.weight"][:, :, 0, 0], sd["glob.s_conv1.weight"][:, :, 0, 0]
    np.testing.assert_array_equal(sec[:314 * 512].reshape(314, 512), wg.T)
    np.testing.assert_array_equal(sec[314 * 512:316 * 512].reshape(2, 512), ws.T)
    q = sec[316 * 512:]
    np.testing.assert_array_equal(q[:512], sd["glob.glob_conv1.bias"] + sd["glob.s_conv1.bias"])
    s1 = sd["glob.bn1.weight"].astype(np.float64) / np.sqrt(sd["glob.bn1.running_var"].astype(np.float64) + 1e-5)
    np.testing.assert_allclose(q[512:1024], s1, rtol=1e-6)
    w2 = q[1536:1536 + 512 * 512].reshape(512, 512)
    np.testing.assert_array_equal(w2, sd["glob.glob_conv2.weight"][:, :, 0, 0].T)
    with pytest.raises(N.IdcError) as ei:
        engine.pack_weights(make_sd(0, "he"), "fp32", global_hints=True)
    assert ei.value.status == -5

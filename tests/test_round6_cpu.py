"""Round 6, CPU side (no GPU): the operand-split precisions' weight blob, and this round's host-logic changes.

Blob of IDC_BF16X3 / IDC_BF16X6 (csrc/idc_engine.hip make_blob_plan, re-derived here independently): conv1_1 is an fp32 island (fp32
layout-1 image, 32-channel chunks); every other layer carries 2 / 3 layout-1 bf16 images back to back -- part 0 = rne(w), part 1 =
rne(w - part 0), part 2 = rne(w - part 0 - part 1) -- so that hi + lo reproduces w to 2^-16 relative and hi + mid + lo reproduces it exactly.
Reference semantics of the weights: torch layouts of models/pytorch/model.py:13-109.
"""
import os

import numpy as np
import pytest

from interactive_deep_colorization_amd import _native as N
from interactive_deep_colorization_amd import engine

from test_abi_cpu import LAYERS, _al, _bf16_bits, _read_w


def _split_plan(parts, dist=False, fp16_block=False):
    off, plan = 64, []
    for wkey, bnkey, kind, cin, cout in LAYERS:
        if wkey == "model_class.0" and not dist:
            continue
        island = kind == "im2col"
        kc = 32 if island else 64
        cpad = 64 if cout <= 64 else _al(cout, 128)
        nkc = (64 // kc) if kind == "im2col" else -(-cin // kc)
        ntap = {"c3": 9, "dc": 16, "c1": 1, "im2col": 1}[kind]
        e = dict(wkey=wkey, bnkey=bnkey, kind=kind, cin=cin, cout=cout, cpad=cpad, nkc=nkc, ncg=cpad // 64, ntap=ntap, island=island)
        e["w_bytes"] = ntap * nkc * e["ncg"] * 8192
        off = _al(off); e["w_off"] = off; off += e["w_bytes"] * (1 if island else parts)
        if island and fp16_block:           # IDC_FP16: conv1_1 also as one fp16 layout-1 block (conv1_block_fused_th's operand)
            off = _al(off); e["w2_off"] = off; off += 8192
        off = _al(off); e["b_off"] = off; off += cpad * 4
        if bnkey:
            off = _al(off); off += cpad * 4
            off = _al(off); off += cpad * 4
        if kind == "dc":
            off = _al(off); off += cpad * 4
        if not island:                      # one fp32 = 2^-s: the layer's weight parts hold w * 2^s (IDC_FP16X3; 1.0 for bf16 parts)
            off = _al(off); e["ws_off"] = off; off += 4
        plan.append(e)
    off = _al(off); off += 1024
    off = _al(off); off += 8
    return plan, _al(off)


def _bf16_to_f32(bits):
    return (np.uint32(bits) << np.uint32(16)).view(np.float32) if isinstance(bits, np.ndarray) else \
        np.array([int(bits) << 16], np.uint32).view(np.float32)[0]


@pytest.mark.parametrize("precision,parts", [("bf16x3", 2), ("bf16x6", 3)])
def test_split_blob_layout_and_parts(make_sd, precision, parts):
    sd = make_sd(0, "he")
    blob = engine.pack_weights(sd, precision)
    plan, total = _split_plan(parts)
    assert blob.size == total == N.load().idc_weights_blob_bytes(engine._PREC[precision], 0)
    assert blob[:4].view(np.uint32)[0] == 0x43444931 and blob[8:12].view(np.uint32)[0] == engine._PREC[precision]
    rs = np.random.RandomState(1)
    for e in plan:
        w = sd[e["wkey"] + ".weight"]
        if not e["island"]:
            assert blob[e["ws_off"]:e["ws_off"] + 4].view(np.float32)[0] == 1.0       # bf16 parts keep fp32's exponent range: no weight scale
        for _ in range(25):
            co, ci = rs.randint(e["cout"]), rs.randint(e["cin"])
            if e["kind"] == "c3":
                ky, kx = rs.randint(3), rs.randint(3); tw, k, val = ky * 3 + kx, ci, w[co, ci, ky, kx]
            elif e["kind"] == "im2col":
                ky, kx = rs.randint(3), rs.randint(3); tw, k, val = 0, (ky * 3 + kx) * 4 + ci, w[co, ci, ky, kx]
            elif e["kind"] == "c1":
                tw, k, val = 0, ci, w[co, ci, 0, 0]
            else:
                ky, kx = rs.randint(4), rs.randint(4); tw, k, val = ky * 4 + kx, ci, w[ci, co, ky, kx]
            val = np.float32(val)
            if e["island"]:                                   # conv1_1: the fp32 image, bit for bit
                assert _read_w(blob, e, "fp32", tw, co, k) == float(val), e["wkey"]
                continue
            rem, total_v = val, np.float32(0)
            for p in range(parts):
                ep = dict(e, w_off=e["w_off"] + p * e["w_bytes"])
                got = _read_w(blob, ep, "bf16", tw, co, k)
                assert got == int(_bf16_bits(np.float32(rem)).ravel()[0]), (e["wkey"], co, ci, "part %d" % p)
                piece = _bf16_to_f32(got)
                rem = np.float32(rem - piece)                 # exact: the remainder of a round-to-nearest is representable
                total_v = np.float32(total_v + piece)
            if parts == 3:
                assert total_v == val, (e["wkey"], co, ci)    # hi + mid + lo = all 24 mantissa bits
            else:
                assert abs(float(total_v) - float(val)) <= 2.0 ** -16 * abs(float(val)) + 1e-30


FUSED_PAIRS = {"model8up.0": "model3short8.0", "model9up.0": "model2short9.0", "model10up.0": "model1short10.0"}     # deconv -> the shortcut conv it is summed with


def _f16_exponents(sd, plan):
    """the packer's per-layer power of two: max|w| * 2^s in [8192, 16384); a deconv and its shortcut conv share the smaller one (one accumulator set)"""
    ex = {}
    for e in plan:
        if e["island"]:
            continue
        mx = float(np.abs(sd[e["wkey"] + ".weight"]).max())
        ex[e["wkey"]] = 14 - int(np.frexp(np.float32(mx))[1]) if mx > 0 else 0
    for d, c in FUSED_PAIRS.items():
        ex[d] = ex[c] = min(ex[d], ex[c])
    return ex


@pytest.mark.parametrize("style", ["he", "torch"])
def test_fp16x3_blob_carries_fp16_parts(make_sd, style):
    """IDC_FP16X3: bf16x3's blob layout (two layout-1 images per layer, conv1_1 an fp32 image) with FP16 parts of w * 2^s -- part 0 = rne16(w 2^s), part 1 =
    rne16(w 2^s - part 0), s the layer's power of two (stored as 2^-s beside the bias): hi + lo reproduces w 2^s to 2^-22 relative (fp16 has 11 significant
    bits) for every weight down to 2^-13 of the layer's largest -- unscaled, he-style weights (~0.02) had SUBNORMAL lo parts (6e-8 absolute)."""
    sd = make_sd(0, style)
    blob = engine.pack_weights(sd, "fp16x3")
    plan, total = _split_plan(2)
    assert blob.size == total == N.load().idc_weights_blob_bytes(N.IDC_FP16X3, 0) == N.load().idc_weights_blob_bytes(N.IDC_BF16X3, 0)
    assert blob[8:12].view(np.uint32)[0] == N.IDC_FP16X3
    rs = np.random.RandomState(2)
    ex = _f16_exponents(sd, plan)
    for e in plan:
        if e["island"]:
            continue
        w = sd[e["wkey"] + ".weight"]
        sexp = ex[e["wkey"]]
        assert blob[e["ws_off"]:e["ws_off"] + 4].view(np.float32)[0] == np.float32(2.0 ** -sexp), e["wkey"]
        assert 8192 <= float(np.abs(w).max()) * 2.0 ** sexp or e["wkey"] in FUSED_PAIRS or e["wkey"] in FUSED_PAIRS.values()
        assert float(np.abs(w).max()) * 2.0 ** sexp < 16384
        for _ in range(25):
            co, ci = rs.randint(e["cout"]), rs.randint(e["cin"])
            if e["kind"] == "c3":
                ky, kx = rs.randint(3), rs.randint(3); tw, k, val = ky * 3 + kx, ci, w[co, ci, ky, kx]
            elif e["kind"] == "c1":
                tw, k, val = 0, ci, w[co, ci, 0, 0]
            else:
                ky, kx = rs.randint(4), rs.randint(4); tw, k, val = ky * 4 + kx, ci, w[ci, co, ky, kx]
            val = np.float32(np.float32(val) * np.float32(2.0 ** sexp))          # exact: a power of two
            hi = np.float16(val)
            lo = np.float16(np.float32(val - np.float32(hi)))
            got0 = _read_w(blob, e, "bf16", tw, co, k)
            got1 = _read_w(blob, dict(e, w_off=e["w_off"] + e["w_bytes"]), "bf16", tw, co, k)
            assert got0 == int(hi.view(np.uint16)) and got1 == int(lo.view(np.uint16)), (e["wkey"], co, ci)
            assert abs(float(np.float32(hi) + np.float32(lo)) - float(val)) <= 2.0 ** -21 * abs(float(val)) + 1e-7
            if abs(float(val)) >= 2.0:                       # i.e. |w| >= 2^-13 of the layer's largest: the lo part is a normal fp16 number
                assert abs(float(np.float32(hi) + np.float32(lo)) - float(val)) <= 2.0 ** -22 * abs(float(val))


def test_split_precisions_are_refused_where_they_do_not_apply():
    lib = N.load()
    assert lib.idc_weights_blob_bytes(6, 0) == 0 and lib.idc_weights_blob_bytes(-1, 0) == 0        # (5 = IDC_FP16 since the one-part form exists)
    assert lib.idc_weights_blob_bytes(N.IDC_BF16X3, 0) < lib.idc_weights_blob_bytes(N.IDC_BF16X6, 0)
    with pytest.raises(KeyError):
        engine.pack_weights({}, "bf16x9")


def test_global_net_skips_a_checkpoints_ab_conv1_1():
    """the Global-Hints prototxt has no ab_conv1_1 (models/global_model/deploy_nodist.prototxt:28-32,189-202) and
    Caffe skips source layers the net lacks; a checkpoint that still carries the layer must not feed local ab / mask planes into the
    global net.  The class passes its net kind; judged by the layers alone (net=None) the merge still happens."""
    from interactive_deep_colorization_amd import caffe_io
    from test_round5_cpu import _caffe_style_sd
    sd = _caffe_style_sd(include_glob=True)
    layers = caffe_io.state_dict_to_caffe_layers(sd, net="global")
    rs = np.random.RandomState(3)
    wa, ba = rs.randn(64, 3, 3, 3).astype(np.float32), rs.randn(64).astype(np.float32)
    layers.insert(0, {"name": "ab_conv1_1", "type": "Convolution", "blobs": [wa, ba]})
    back = caffe_io.read_caffemodel(caffe_io.write_caffemodel(None, layers))
    got, info = caffe_io.caffe_layers_to_state_dict(back, net="global")
    assert info["net"] == "global" and "ab_conv1_1" in info["ignored"]
    assert np.abs(got["model1.0.weight"][:, 1:]).max() == 0 and np.array_equal(got["model1.0.bias"], sd["model1.0.bias"])
    merged, info2 = caffe_io.caffe_layers_to_state_dict(back)
    assert np.array_equal(merged["model1.0.weight"][:, 1:], wa) and info2["net_in_file"] == "global"
    # the shared model.caffemodel carries several heads: 'net' reports what the class asked for, not what the file happens to hold
    assert caffe_io.caffe_layers_to_state_dict(back, net="nodist")[1]["net"] == "nodist"


# ---- the first 8-GPU run, made boring ------------------------------------------------------------------------------
def test_eight_ranks_through_the_c_abi_transport():
    """Eight gloo ranks with --transport c_abi (round 5 ran eight ranks on the torch transport only): every rank's librccl pre-flight, the
    agreed fallback, the real blob on every rank, one line, rc 0."""
    from test_round5_cpu import _bench
    p, line = _bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--control-flow-only", "--transport", "c_abi")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["ranks_reporting"] == 8 and line["every_rank_holds_rank0_blob"] is True
    assert line["transport_requested"] == "c_abi" and line["transport_used"] == "torch" and line["transport_fallback_reason"]
    assert len(line["weights_broadcast_ms_per_rank"]) == 8 and all(x is not None for x in line["weights_broadcast_ms_per_rank"])


def test_strong_scaling_splits_a_fixed_global_batch():
    """SURVEY.md 8(d) config 4, second form: --scaling strong keeps 256 images in total (32 per rank at 8, 64 at 4); weak keeps 32 per rank."""
    from test_round5_cpu import _bench
    p, line = _bench("--gpus", "4", "--steps", "2", "--warmup", "1", "--control-flow-only", "--scaling", "strong")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["scaling"] == "strong" and line["per_gpu_batch"] == 64 and line["global_batch"] == 256 and line["n_gpus"] == 4
    p, line = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--control-flow-only")
    assert line["scaling"] == "weak" and line["per_gpu_batch"] == 32 and line["global_batch"] == 64
    p, _ = _bench("--gpus", "3", "--steps", "1", "--control-flow-only", "--scaling", "strong")
    assert p.returncode != 0                                   # 256 does not divide by 3: refused, not rounded


def test_cpu_baseline_times_the_reference_module_where_it_exists():
    """kind 'reference' (models/pytorch/model.py imported untouched) in the authoring container, 'port' (the oracle's
    restatement: bit-identical at N = 1) where the reference tree does not exist -- the GPU box."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from interactive_deep_colorization_amd import workloads
    from oracle import siggraph_torch
    sd = workloads.random_state_dict(0, "torch")
    fwd, kind = bench._cpu_forward(sd)
    assert kind == ("reference" if os.path.isdir("/root/reference/models/pytorch") else "port")
    L, ab, m = workloads.random_batch(1, 256, seed=0)
    out = np.asarray(fwd(L, ab, m)).reshape(2, 256, 256)
    assert np.array_equal(out, siggraph_torch.forward(sd, L, ab, m, 0.0)[0])
    wh = bench.cpu_whole_host(2, 4, 1.0, "torch")              # fewer than two workers fit: says so instead of inventing a number
    assert wh["value"] is None and "whole host" in wh["note"]


def test_fp16_blob_is_one_fp16_image_per_layer(make_sd):
    """IDC_FP16: the split plan with ONE part -- an fp16 image of w per layer, UNSCALED (one part keeps its 11 bits down to |w| = 6e-5, and the fp16 twins of
    the bf16 throughput kernels start their accumulators at the bias: no factor to take back), the scale slot 1.0, conv1_1 an fp32 image."""
    sd = make_sd(0, "he")
    blob = engine.pack_weights(sd, "fp16")
    plan, total = _split_plan(1, fp16_block=True)
    assert blob.size == total == N.load().idc_weights_blob_bytes(N.IDC_FP16, 0)
    assert blob[8:12].view(np.uint32)[0] == N.IDC_FP16
    rs = np.random.RandomState(3)
    for e in plan:
        if e["island"]:                      # conv1_1: the fp32 island image AND one fp16 block with K = tap * 4 + channel (nkc = 1)
            w = sd[e["wkey"] + ".weight"]
            for _ in range(20):
                co, ci, ky, kx = rs.randint(e["cout"]), rs.randint(e["cin"]), rs.randint(3), rs.randint(3)
                k = (ky * 3 + kx) * 4 + ci
                assert _read_w(blob, e, "fp32", 0, co, k) == float(np.float32(w[co, ci, ky, kx]))
                assert _read_w(blob, dict(e, w_off=e["w2_off"], nkc=1), "bf16", 0, co, k) == int(np.float16(np.float32(w[co, ci, ky, kx])).view(np.uint16))
            continue
        if e["kind"] != "c3":
            continue
        w = sd[e["wkey"] + ".weight"]
        assert blob[e["ws_off"]:e["ws_off"] + 4].view(np.float32)[0] == 1.0
        for _ in range(10):
            co, ci, ky, kx = rs.randint(e["cout"]), rs.randint(e["cin"]), rs.randint(3), rs.randint(3)
            assert _read_w(blob, e, "bf16", ky * 3 + kx, co, ci) == int(np.float16(np.float32(w[co, ci, ky, kx])).view(np.uint16)), (e["wkey"], co, ci)


def test_fp16_weight_scale_is_what_closes_the_gap_to_fp32(make_sd):
    """Why IDC_FP16X3's weight parts hold w * 2^s: on he-style weights (~0.02) the lo part of an UNSCALED weight is a subnormal fp16 number (6e-8 absolute =
    2^-18 of the weight).  oracle/emulate.py restates both arithmetics on the CPU (fp16 parts, three products, fp32 accumulation; conv1_1 exact fp32 as on
    the GPU): the scaled form must sit at the fp32 arithmetic's own distance from the float64 oracle, the unscaled one well above it (GPU, N = 32 256^2:
    3.9e-3 unscaled, 1.9e-3 scaled, 1.7e-3 fp32 -- profiles/parity_r06_gpu.json, profiles/r06_fp16_wscale_study.txt)."""
    import torch
    from interactive_deep_colorization_amd import workloads
    from oracle import emulate, siggraph_torch
    sd = make_sd(0, "he")
    L, ab, m = workloads.random_batch(1, 64, seed=4, max_points=4, max_p=2)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64)
    err = {}
    for mode in ("fp32", "splitf2_fp32", "splitf2s_fp32"):
        out = emulate.forward(sd, L, ab, m, 0.0, default=mode, modes={"conv1_1": "fp32"})
        err[mode] = emulate.error_stats(out, ref)["mean_abs"]
    assert err["splitf2s_fp32"] <= 1.5 * err["fp32"], err
    assert err["splitf2_fp32"] >= 1.8 * err["splitf2s_fp32"], err


def _caffe_subset_messages():
    """caffe.proto's NetParameter / LayerParameter / V1LayerParameter / BlobProto / BlobShape subset built with Google's protobuf runtime (descriptor_pb2 ->
    message classes): the reference implementation of the wire format, independent of caffe_io's hand-written reader and writer."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="caffe_subset_test.proto", package="caffe_subset_test", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, extra in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if extra.get("type_name"):
                f.type_name = ".caffe_subset_test." + extra["type_name"]
            if extra.get("packed"):
                f.options.packed = True
    R, O = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, R, {"packed": True})])
    msg("BlobProto", [("shape", 7, F.TYPE_MESSAGE, O, {"type_name": "BlobShape"}), ("data", 5, F.TYPE_FLOAT, R, {"packed": True}),
                      ("double_data", 8, F.TYPE_DOUBLE, R, {"packed": True}), ("num", 1, F.TYPE_INT32, O, {}), ("channels", 2, F.TYPE_INT32, O, {}),
                      ("height", 3, F.TYPE_INT32, O, {}), ("width", 4, F.TYPE_INT32, O, {})])
    msg("BlobProtoUnpacked", [("shape", 7, F.TYPE_MESSAGE, O, {"type_name": "BlobShape"}), ("data", 5, F.TYPE_FLOAT, R, {})])
    msg("LayerParameter", [("name", 1, F.TYPE_STRING, O, {}), ("type", 2, F.TYPE_STRING, O, {}), ("bottom", 3, F.TYPE_STRING, R, {}),
                           ("top", 4, F.TYPE_STRING, R, {}), ("phase", 10, F.TYPE_INT32, O, {}), ("blobs", 7, F.TYPE_MESSAGE, R, {"type_name": "BlobProto"})])
    msg("LayerParameterU", [("name", 1, F.TYPE_STRING, O, {}), ("type", 2, F.TYPE_STRING, O, {}), ("blobs", 7, F.TYPE_MESSAGE, R, {"type_name": "BlobProtoUnpacked"})])
    msg("V1LayerParameter", [("bottom", 2, F.TYPE_STRING, R, {}), ("top", 3, F.TYPE_STRING, R, {}), ("name", 4, F.TYPE_STRING, O, {}),
                             ("type", 5, F.TYPE_INT32, O, {}), ("blobs", 6, F.TYPE_MESSAGE, R, {"type_name": "BlobProto"})])
    msg("NetParameter", [("name", 1, F.TYPE_STRING, O, {}), ("layers", 2, F.TYPE_MESSAGE, R, {"type_name": "V1LayerParameter"}),
                         ("force_backward", 5, F.TYPE_BOOL, O, {}), ("layer", 100, F.TYPE_MESSAGE, R, {"type_name": "LayerParameter"})])
    msg("NetParameterU", [("name", 1, F.TYPE_STRING, O, {}), ("layer", 100, F.TYPE_MESSAGE, R, {"type_name": "LayerParameterU"})])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe_subset_test." + n))
    return {n: get(n) for n in ("NetParameter", "NetParameterU")}


def test_caffemodel_wire_format_against_googles_protobuf_runtime():
    """Round 6: caffe_io.py reads and writes the protobuf wire format by hand (Caffe is absent).  Google's protobuf runtime IS in the image: files it
    serialises from the published caffe.proto subset -- packed and unpacked float data, BlobShape and the legacy num/channels/height/width, V2 `layer` and
    V1 `layers` records, unknown fields in between, double_data -- must come back from read_caffemodel value for value, and what write_caffemodel emits
    must parse in the runtime to the same layers.  (Pins the wire-format code to the format's reference implementation; a real checkpoint stays absent.)"""
    pytest.importorskip("google.protobuf")
    from interactive_deep_colorization_amd import caffe_io
    M = _caffe_subset_messages()
    rs = np.random.RandomState(0)
    w = rs.randn(5, 3, 3, 3).astype(np.float32); b = rs.randn(5).astype(np.float32); legacy = rs.randn(2, 3, 1, 1).astype(np.float32)
    dd = rs.randn(4).astype(np.float64)
    net = M["NetParameter"](name="n", force_backward=True)
    L = net.layer.add(name="conv1_2", type="Convolution", phase=1); L.bottom.append("x"); L.top.append("y")
    for arr in (w, b):
        bl = L.blobs.add(); bl.shape.dim.extend(arr.shape); bl.data.extend(arr.ravel().tolist())
    L2 = net.layer.add(name="relu", type="ReLU"); L2.bottom.append("y"); L2.top.append("y")
    L3 = net.layer.add(name="legacy_dims", type="Convolution")
    bl = L3.blobs.add(num=2, channels=3, height=1, width=1); bl.data.extend(legacy.ravel().tolist())
    L4 = net.layer.add(name="doubles", type="Scale")
    bl = L4.blobs.add(); bl.shape.dim.append(4); bl.double_data.extend(dd.tolist())
    V1 = net.layers.add(name="old_conv", type=4); V1.bottom.append("a"); V1.top.append("b")
    bl = V1.blobs.add(); bl.shape.dim.extend(b.shape); bl.data.extend(b.tolist())
    got = caffe_io.read_caffemodel(net.SerializeToString())
    by = {g["name"]: g for g in got}
    assert set(by) == {"conv1_2", "relu", "legacy_dims", "doubles", "old_conv"}
    assert by["conv1_2"]["type"] == "Convolution" and by["conv1_2"]["bottom"] == ["x"] and by["conv1_2"]["top"] == ["y"]
    assert np.array_equal(by["conv1_2"]["blobs"][0], w) and np.array_equal(by["conv1_2"]["blobs"][1], b)
    assert by["relu"]["blobs"] == [] and np.array_equal(by["legacy_dims"]["blobs"][0], legacy)
    assert np.array_equal(by["doubles"]["blobs"][0], dd.astype(np.float32))
    assert by["old_conv"]["type"] == "Convolution" and np.array_equal(by["old_conv"]["blobs"][0], b)
    # unpacked repeated floats (a writer may emit one 32-bit record per value)
    netu = M["NetParameterU"](name="u")
    Lu = netu.layer.add(name="conv1_2", type="Convolution")
    bl = Lu.blobs.add(); bl.shape.dim.extend(w.shape); bl.data.extend(w.ravel().tolist())
    gotu = caffe_io.read_caffemodel(netu.SerializeToString())
    assert np.array_equal(gotu[0]["blobs"][0], w)
    # our writer -> Google's parser
    raw = caffe_io.write_caffemodel(None, [{"name": "conv1_2", "type": "Convolution", "bottom": ["x"], "top": ["y"], "blobs": [w, b]},
                                           {"name": "relu", "type": "ReLU", "bottom": ["y"], "top": ["y"], "blobs": []}])
    back = M["NetParameter"]()
    back.ParseFromString(raw)
    assert [l.name for l in back.layer] == ["conv1_2", "relu"] and back.layer[0].type == "Convolution"
    assert list(back.layer[0].blobs[0].shape.dim) == list(w.shape)
    assert np.array_equal(np.array(back.layer[0].blobs[0].data, np.float32).reshape(w.shape), w)
    assert np.array_equal(np.array(back.layer[0].blobs[1].data, np.float32), b)

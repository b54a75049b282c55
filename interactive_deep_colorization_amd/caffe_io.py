"""`.caffemodel` ingestion for the Caffe classes -- no Caffe needed.

The reference's default backend builds ``caffe.Net(prototxt_path, caffemodel_path, caffe.TEST)``
(``data/colorize_image.py:392-403``; ``ideepcolor.py:60-66`` passes ``./models/reference_model/model.caffemodel``).  A
``.caffemodel`` is a serialised ``caffe.NetParameter`` protobuf; the subset needed to get the learned blobs out is small
(BVLC Caffe ``src/caffe/proto/caffe.proto``, restated here -- Caffe is not in this image, so this follows the published
schema; PARITY UNPINNED against a real checkpoint, ``models/fetch_models.sh`` needs the network; the wire-format code itself is held
against Google's protobuf runtime serialising / parsing that subset, ``tests/test_round6_cpu.py``):

    NetParameter      1 name (string)   100 layer (LayerParameter, repeated)   2 layers (V1LayerParameter, repeated, legacy)
    LayerParameter    1 name   2 type (string)   3 bottom   4 top   7 blobs (BlobProto, repeated)
    V1LayerParameter  4 name   5 type (enum)     2 bottom   3 top   6 blobs
    BlobProto         7 shape (BlobShape)   5 data (float, packed or not)   8 double_data   1-4 num/channels/height/width (legacy)
    BlobShape         1 dim (int64, packed or not)

``read_caffemodel`` walks the wire format (varint / 64-bit / length-delimited / 32-bit records) and returns the layers in file
order with their blobs as float32 arrays; ``write_caffemodel`` emits the same subset (used by the tests' round trip and to
turn a torch-key ``state_dict`` into a Caffe file).  ``caffe_layers_to_state_dict`` maps the reference's Caffe layer names
onto the key set the engine packs (``include/ideepcolor.h``):

* ``bw_conv1_1`` (64,1,3,3) + ``ab_conv1_1`` (64,3,3,3), summed by an Eltwise (``deploy_nodist.prototxt:19-51``) -> ONE 4-channel
  conv ``model1.0`` (weights concatenated along Cin in the order L, a, b, mask; biases added); the Global-Hints net has
  ``bw_conv1_1`` only (``global_model/deploy_nodist.prototxt:176-209``) -> zero ab / mask columns;
* Caffe ``BatchNorm`` blobs (mean, variance, scale_factor) (``:78-87``; test mode = ``(x - mean/sf) / sqrt(var/sf + 1e-5)``,
  no affine -- there is no Scale layer behind it) -> ``running_mean = mean/sf``, ``running_var = var/sf``, ``weight = 1``,
  ``bias = 0`` (torch's eval BatchNorm with eps 1e-5 is then the same function; sf = 0 means "no statistics": Caffe divides by 1);
* the depthwise all-ones 1x1 stride-2 convs ``conv*_norm_ss`` (``:88-103``: the ``[::2, ::2]`` subsample) carry a (C,1,1,1) blob:
  folded into the input channels of the conv that reads them (a no-op when it is the filler's 1);
* ``Deconvolution`` blobs are (Cin, Cout, 4, 4) like ``ConvTranspose2d``; ``conv10_ab`` -> ``model_out.0``; the final ``Scale``
  layer ``pred_ab`` of the regression net (``:812-821``, filler 100) -> the engine's output multiplier (``out_mul``);
* 313-bin net (``deploy_nopred.prototxt:650-850``): ``conv3_pred`` .. ``conv8_pred``, ``pred_313`` -> ``pred.*``; ``pred_313_us`` /
  ``pred_313_rs`` (shared bilinear kernel ``kern_us``), ``scale_S`` and the cluster-centre conv ``pred_ab`` are what the reference
  OVERWRITES at load time (``colorize_image.py:405-413,482-485``): the file's values are ignored exactly as the reference ignores
  them and the injected ones are used (bilinear kernel in closed form inside ``dist313_kernel``, S through
  ``idc_set_dist_temperature``, centres from ``pts_in_hull``); ``scale_T`` must be the prototxt's 2.6;
* Global-Hints branch (``global_model/deploy_nodist.prototxt:37-172``): ``s_conv1``, ``glob_conv1`` .. ``glob_conv4`` -> ``glob.*``,
  their BatchNorms ``s_glob_conv1norm``, ``glob_conv2norm`` .. ``glob_conv4norm`` -> ``glob.bn1`` .. ``glob.bn4``.

Host plumbing of the drop-in class; nothing here touches the device.
"""
import struct

import numpy as np

__all__ = ["read_caffemodel", "write_caffemodel", "caffe_layers_to_state_dict", "state_dict_to_caffe_layers",
           "read_caffemodel_state_dict", "is_caffemodel", "CaffeModelError"]


class CaffeModelError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------- wire format
def _varint(buf, pos):
    x = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise CaffeModelError("truncated varint at byte %d" % pos)
        b = buf[pos]
        pos += 1
        x |= (b & 0x7F) << shift
        if not b & 0x80:
            return x, pos
        shift += 7
        if shift > 70:
            raise CaffeModelError("varint longer than 10 bytes at byte %d" % pos)


def _fields(buf):
    """yield (field number, wire type, value) over one message; value = int (varint / fixed) or a memoryview (length-delimited)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > n:
                raise CaffeModelError("truncated 64-bit field %d" % num)
            val = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise CaffeModelError("length-delimited field %d runs past its message (%d > %d)" % (num, pos + ln, n))
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            if pos + 4 > n:
                raise CaffeModelError("truncated 32-bit field %d" % num)
            val = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise CaffeModelError("unsupported wire type %d (field %d) -- not a caffemodel?" % (wt, num))
        yield num, wt, val


def _packed_varints(val, wt):
    if wt == 0:
        return [val]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


def _blob(buf):
    dims, legacy, chunks, dchunks = [], {}, [], []
    for num, wt, val in _fields(buf):
        if num == 7 and wt == 2:                                   # BlobShape
            for n2, w2, v2 in _fields(val):
                if n2 == 1:
                    dims += _packed_varints(v2, w2)
        elif num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = val
        elif num == 5:                                             # float data
            chunks.append(np.frombuffer(bytes(val), dtype="<f4") if wt == 2 else np.frombuffer(val, dtype="<f4"))
        elif num == 8:                                             # double data
            dchunks.append(np.frombuffer(bytes(val), dtype="<f8") if wt == 2 else np.frombuffer(val, dtype="<f8"))
    if chunks:
        data = np.concatenate(chunks).astype(np.float32)
    elif dchunks:
        data = np.concatenate(dchunks).astype(np.float32)
    else:
        data = np.zeros(0, np.float32)
    if not dims and legacy:
        dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    if not dims:
        dims = [data.size]
    if int(np.prod(dims)) != data.size:
        raise CaffeModelError("blob of shape %s carries %d values" % (dims, data.size))
    return data.reshape([int(d) for d in dims])


_V1_TYPES = {4: "Convolution", 39: "Deconvolution", 18: "ReLU", 23: "TanH", 25: "Eltwise", 20: "Softmax", 33: "Slice", 36: "Silence"}


def read_caffemodel(path_or_bytes):
    """-> list of {'name', 'type', 'bottom', 'top', 'blobs': [float32 arrays]} in file order (layers without blobs included)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        raw = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            raw = f.read()
    layers = []
    for num, wt, val in _fields(memoryview(raw)):
        if wt != 2 or num not in (100, 2):
            continue
        v1 = num == 2
        L = {"name": "", "type": "", "bottom": [], "top": [], "blobs": []}
        for n2, w2, v2 in _fields(val):
            if v1:
                if n2 == 4 and w2 == 2:
                    L["name"] = bytes(v2).decode("utf-8")
                elif n2 == 5 and w2 == 0:
                    L["type"] = _V1_TYPES.get(v2, "V1:%d" % v2)
                elif n2 == 2 and w2 == 2:
                    L["bottom"].append(bytes(v2).decode("utf-8"))
                elif n2 == 3 and w2 == 2:
                    L["top"].append(bytes(v2).decode("utf-8"))
                elif n2 == 6 and w2 == 2:
                    L["blobs"].append(_blob(v2))
            else:
                if n2 == 1 and w2 == 2:
                    L["name"] = bytes(v2).decode("utf-8")
                elif n2 == 2 and w2 == 2:
                    L["type"] = bytes(v2).decode("utf-8")
                elif n2 == 3 and w2 == 2:
                    L["bottom"].append(bytes(v2).decode("utf-8"))
                elif n2 == 4 and w2 == 2:
                    L["top"].append(bytes(v2).decode("utf-8"))
                elif n2 == 7 and w2 == 2:
                    L["blobs"].append(_blob(v2))
        layers.append(L)
    if not layers:
        raise CaffeModelError("no layer records (fields 100 / 2 of caffe.NetParameter) found -- not a caffemodel")
    return layers


def is_caffemodel(path):
    """True when the file parses as a NetParameter with at least one layer (cheap: stops at the first layer record)."""
    try:
        with open(path, "rb") as f:
            head = f.read(1 << 16)
        pos = 0
        key, pos = _varint(head, pos)
        return (key >> 3) in (1, 2, 100) and (key & 7) == 2 and not head.startswith(b"PK") and not head.startswith(b"\x80\x02")
    except Exception:
        return False


def _enc_varint(x):
    out = bytearray()
    x = int(x)
    if x < 0:
        x += 1 << 64
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path, layers, net_name="ideepcolor"):
    """Serialise [{'name', 'type', 'bottom', 'top', 'blobs'}] as a V2 caffe.NetParameter (packed float data, BlobShape)."""
    out = bytearray(_enc_ld(1, net_name.encode("utf-8")))
    for L in layers:
        msg = bytearray(_enc_ld(1, L["name"].encode("utf-8")) + _enc_ld(2, L.get("type", "").encode("utf-8")))
        for b in L.get("bottom", []):
            msg += _enc_ld(3, b.encode("utf-8"))
        for t in L.get("top", []):
            msg += _enc_ld(4, t.encode("utf-8"))
        for blob in L.get("blobs", []):
            a = np.ascontiguousarray(np.asarray(blob), dtype="<f4")
            shape = _enc_ld(1, b"".join(_enc_varint(d) for d in (a.shape if a.ndim else (1,))))
            msg += _enc_ld(7, _enc_ld(7, shape) + _enc_ld(5, a.tobytes()))
        out += _enc_ld(100, bytes(msg))
    if path is None:
        return bytes(out)
    with open(path, "wb") as f:
        f.write(out)
    return path


# ---------------------------------------------------------------------------------------------- layer names -> state_dict keys
_CONV = {"conv1_2": "model1.2", "conv2_1": "model2.0", "conv2_2": "model2.2", "conv3_1": "model3.0", "conv3_2": "model3.2",
         "conv3_3": "model3.4", "conv4_1": "model4.0", "conv4_2": "model4.2", "conv4_3": "model4.4", "conv5_1": "model5.0",
         "conv5_2": "model5.2", "conv5_3": "model5.4", "conv6_1": "model6.0", "conv6_2": "model6.2", "conv6_3": "model6.4",
         "conv7_1": "model7.0", "conv7_2": "model7.2", "conv7_3": "model7.4", "conv8_1": "model8up.0",
         "conv3_3_short": "model3short8.0", "conv8_2": "model8.1", "conv8_3": "model8.3", "conv9_1": "model9up.0",
         "conv2_2_short": "model2short9.0", "conv9_2": "model9.1", "conv10_1": "model10up.0",
         "conv1_2_short": "model1short10.0", "conv10_2": "model10.1", "conv10_ab": "model_out.0",
         "conv3_pred": "pred.conv3_pred", "conv4_pred": "pred.conv4_pred", "conv5_pred": "pred.conv5_pred",
         "conv6_pred": "pred.conv6_pred", "conv7_pred": "pred.conv7_pred", "conv8_pred": "pred.conv8_pred",
         "pred_313": "pred.pred_313",
         "s_conv1": "glob.s_conv1", "glob_conv1": "glob.glob_conv1", "glob_conv2": "glob.glob_conv2",
         "glob_conv3": "glob.glob_conv3", "glob_conv4": "glob.glob_conv4"}
_BN = {"conv1_2norm": "model1.4", "conv2_2norm": "model2.4", "conv3_3norm": "model3.6", "conv4_3norm": "model4.6",
       "conv5_3norm": "model5.6", "conv6_3norm": "model6.6", "conv7_3norm": "model7.6", "conv8_3norm": "model8.5",
       "conv9_2norm": "model9.3",
       "s_glob_conv1norm": "glob.bn1", "glob_conv2norm": "glob.bn2", "glob_conv3norm": "glob.bn3", "glob_conv4norm": "glob.bn4"}
_SS = {"conv1_2norm_ss": "model2.0", "conv2_2norm_ss": "model3.0", "conv3_3norm_ss": "model4.0"}      # folded into this conv's Cin
_INJECTED = ("pred_313_us", "pred_313_rs", "scale_S")     # overwritten by the reference at load time (colorize_image.py:405-413,482-485)
_DECONV = ("model8up.0", "model9up.0", "model10up.0", "pred.conv4_pred", "pred.conv5_pred", "pred.conv6_pred", "pred.conv7_pred")
BN_EPS = 1e-5


def caffe_layers_to_state_dict(layers, net=None):
    """Layers of ``read_caffemodel`` -> ``(state_dict, info)``.  ``state_dict``: the engine's keys (float32 arrays);
    ``info``: {'out_mul': the regression net's final Scale (100) or None, 'net': 'nodist' | 'nopred' | 'global',
    'ignored': names of blob-carrying layers not used (injected at load time by the reference, or unknown)}.

    ``net`` = the prototxt the caller's class stands for ('nodist' | 'nopred' | 'global'; None = judge by the layers the file
    holds).  Caffe copies blobs BY LAYER NAME into the net the prototxt defines and ignores source layers the net does not have
    (``caffe.Net(prototxt, caffemodel)``, ``data/colorize_image.py:392-403``): the Global-Hints prototxt comments ``ab_conv1_1``
    out (``models/global_model/deploy_nodist.prototxt:28-32,189-202``), so a checkpoint that still carries that layer must NOT feed
    local ab / mask planes into the global net."""
    by_name = {}
    for L in layers:
        if L["blobs"]:
            if L["name"] in by_name:
                raise CaffeModelError("layer '%s' appears twice with blobs" % L["name"])
            by_name[L["name"]] = [np.asarray(b, np.float32) for b in L["blobs"]]
    sd, ignored = {}, []

    def conv_blobs(name):
        bl = by_name[name]
        if len(bl) not in (1, 2) or bl[0].ndim != 4:
            raise CaffeModelError("layer '%s': expected (weight[, bias]) with a 4-d weight, got shapes %s" % (name, [b.shape for b in bl]))
        w = bl[0]
        b = bl[1].reshape(-1) if len(bl) == 2 else None
        return w, b

    # conv1_1 = bw_conv1_1 (+ ab_conv1_1): one 4-channel conv, input order (L, a, b, mask)
    if "bw_conv1_1" not in by_name:
        raise CaffeModelError("no 'bw_conv1_1' layer: not one of the reference's colorization nets")
    wl, bl_ = conv_blobs("bw_conv1_1")
    if wl.shape[1:] != (1, 3, 3):
        raise CaffeModelError("bw_conv1_1 weight %s, expected (64, 1, 3, 3)" % (wl.shape,))
    w4 = np.zeros((wl.shape[0], 4, 3, 3), np.float32)
    w4[:, :1] = wl
    b4 = np.zeros(wl.shape[0], np.float32) if bl_ is None else bl_.copy()
    if "ab_conv1_1" in by_name and net == "global":
        ignored.append("ab_conv1_1")                              # not a layer of the Global-Hints net: Caffe would skip it
    elif "ab_conv1_1" in by_name:
        wa, ba = conv_blobs("ab_conv1_1")
        if wa.shape != (wl.shape[0], 3, 3, 3):
            raise CaffeModelError("ab_conv1_1 weight %s, expected (%d, 3, 3, 3)" % (wa.shape, wl.shape[0]))
        w4[:, 1:] = wa
        if ba is not None:
            b4 = b4 + ba                                          # Eltwise SUM of the two conv outputs
    sd["model1.0.weight"], sd["model1.0.bias"] = w4, b4

    out_mul = None
    for name, bl in by_name.items():
        if name in ("bw_conv1_1", "ab_conv1_1"):
            continue
        if name in _CONV:
            w, b = conv_blobs(name)
            key = _CONV[name]
            sd[key + ".weight"] = w.copy()
            cout = w.shape[1] if key in _DECONV else w.shape[0]
            sd[key + ".bias"] = b.copy() if b is not None else np.zeros(cout, np.float32)
        elif name in _BN:
            if len(bl) != 3:
                raise CaffeModelError("BatchNorm '%s': expected (mean, variance, scale_factor), got %d blobs" % (name, len(bl)))
            mean, var, sf = bl[0].reshape(-1), bl[1].reshape(-1), float(bl[2].reshape(-1)[0])
            s = 0.0 if sf == 0.0 else 1.0 / sf                    # batch_norm_layer.cpp: scale_factor 0 -> multiply by 0
            key = _BN[name]
            sd[key + ".running_mean"] = (mean * s).astype(np.float32)
            sd[key + ".running_var"] = (var * s).astype(np.float32)
            sd[key + ".weight"] = np.ones(mean.size, np.float32)   # Caffe BatchNorm has no affine (no Scale layer follows)
            sd[key + ".bias"] = np.zeros(mean.size, np.float32)
        elif name == "pred_ab":
            if bl[0].ndim == 4 and bl[0].shape[1] == 313:          # nopred: the cluster-centre conv, injected from pts_in_hull by the caller
                ignored.append(name)
            else:                                                  # nodist: Scale (filler 100) behind the TanH
                v = bl[0].reshape(-1)
                if v.size == 0 or np.abs(v - v[0]).max() > 1e-6 * max(1.0, abs(float(v[0]))):
                    raise CaffeModelError("Scale layer 'pred_ab' with per-channel values %s is not representable (one output multiplier)" % v)
                out_mul = float(v[0])
        elif name == "scale_T":
            v = bl[0].reshape(-1)
            if v.size and np.abs(v - 2.6).max() > 1e-4:
                raise CaffeModelError("scale_T = %s: the annealed-mean temperature is the prototxt's 2.6 in this engine" % v[:4])
        elif name in _SS or name in _INJECTED:
            pass
        else:
            ignored.append(name)
    for name, key in _SS.items():                                  # depthwise 1x1 stride-2 'subsample' convs: fold a non-unit blob
        if name in by_name:
            s = by_name[name][0].reshape(-1)
            if key + ".weight" not in sd or sd[key + ".weight"].shape[1] != s.size:
                raise CaffeModelError("'%s' has %d channels but its consumer %s is missing or mismatched" % (name, s.size, key))
            if len(by_name[name]) > 1:
                raise CaffeModelError("'%s' carries a bias (the reference's layer has bias_term: false)" % name)
            sd[key + ".weight"] = sd[key + ".weight"] * s[None, :, None, None]
    ignored += [n for n in _INJECTED if n in by_name]
    found = "nopred" if "pred.pred_313.weight" in sd else ("global" if "glob.glob_conv1.weight" in sd else "nodist")
    # 'net': what the caller's class asked for (the shared model.caffemodel carries several heads); 'net_in_file': what the layers say
    return sd, {"out_mul": out_mul, "net": net or found, "net_in_file": found, "ignored": sorted(ignored)}


def state_dict_to_caffe_layers(sd, net="nodist", out_mul=100.0, scale_factor=999.98236):
    """The inverse map, for tests and for turning a torch-key state_dict into the file the reference's default invocation reads:
    BatchNorm affine is folded away first (Caffe's BatchNorm cannot carry it) -- weight/bias must be 1/0 unless ``fold_affine``
    is possible, which it is NOT behind a ReLU; so a state_dict with a non-trivial BN affine raises.  ``scale_factor`` is the
    moving-average normaliser Caffe stores (mean and variance blobs are multiplied by it)."""
    g = lambda k: np.asarray(sd[k], np.float32)
    layers = []
    w1, b1 = g("model1.0.weight"), g("model1.0.bias")
    if net == "global":
        if np.abs(w1[:, 1:]).max() > 0:
            raise CaffeModelError("global net: conv1_1 sees L only (bw_conv1_1); the state_dict has ab / mask weights")
        layers.append({"name": "bw_conv1_1", "type": "Convolution", "blobs": [w1[:, :1], b1]})
    else:
        layers.append({"name": "ab_conv1_1", "type": "Convolution", "blobs": [w1[:, 1:], np.zeros_like(b1)]})
        layers.append({"name": "bw_conv1_1", "type": "Convolution", "blobs": [w1[:, :1], b1]})
    inv_conv = {v: k for k, v in _CONV.items()}
    inv_bn = {v: k for k, v in _BN.items()}
    inv_ss = {v: k for k, v in _SS.items()}
    order = [k[:-len(".weight")] for k in sd if k.endswith(".weight") and k != "model1.0.weight"]
    for key in order:
        if key in inv_bn:
            if np.abs(g(key + ".weight") - 1).max() > 0 or np.abs(g(key + ".bias")).max() > 0:
                raise CaffeModelError("%s has an affine part; Caffe's BatchNorm (no Scale layer in the prototxt) cannot carry it" % key)
            layers.append({"name": inv_bn[key], "type": "BatchNorm",
                           "blobs": [g(key + ".running_mean") * scale_factor, g(key + ".running_var") * scale_factor,
                                     np.array([scale_factor], np.float32)]})
        elif key in inv_conv:
            if key in inv_ss:
                c = g(key + ".weight").shape[1]
                layers.append({"name": inv_ss[key], "type": "Convolution", "blobs": [np.ones((c, 1, 1, 1), np.float32)]})
            name = inv_conv[key]
            layers.append({"name": name, "type": "Deconvolution" if key in _DECONV else "Convolution",
                           "blobs": [g(key + ".weight"), g(key + ".bias")]})
        elif key in ("model_class.0", "pred.pred_ab"):
            continue                                               # torch-only 529 head / injected centres: not in the Caffe files
        else:
            raise CaffeModelError("no Caffe layer for state_dict key '%s'" % key)
    if net == "nodist":
        layers.append({"name": "pred_ab", "type": "Scale", "blobs": [np.full(2, out_mul, np.float32)]})
    if net == "nopred":
        k = np.array(((.25, .5, .25, 0), (.5, 1., .5, 0), (.25, .5, .25, 0), (0, 0, 0, 0)), np.float32)
        layers.append({"name": "pred_313_us", "type": "Deconvolution", "blobs": [np.zeros((313, 1, 4, 4), np.float32)]})   # injected at load
        layers.append({"name": "scale_S", "type": "Scale", "blobs": [np.full(313, 0.2, np.float32)]})
        layers.append({"name": "scale_T", "type": "Scale", "blobs": [np.full(313, 2.6, np.float32)]})
        layers.append({"name": "pred_ab", "type": "Convolution", "blobs": [np.zeros((2, 313, 1, 1), np.float32), np.zeros(2, np.float32)]})
        del k
    return layers


def read_caffemodel_state_dict(path, net=None):
    """``.caffemodel`` -> ``(state_dict, info)`` (see ``caffe_layers_to_state_dict``)."""
    return caffe_layers_to_state_dict(read_caffemodel(path), net=net)

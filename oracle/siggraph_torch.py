"""The oracle proper: batched torch-CPU restatement of ``SIGGRAPHGenerator.forward``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The arithmetic of the
reference path lives in third-party PyTorch (``torch.nn.Conv2d`` & co. --
SURVEY.md section 8c), so the restatement calls the same ATen CPU kernels
through ``torch.nn.functional`` on explicit weight tensors and follows
``models/pytorch/model.py`` line by line:

    :139-142  f64 numpy -> f32, add batch dim, ``mask - maskcent``
    :148      cat(L/100, ab/110, mask) -> model1
    :149-151  ``[:, :, ::2, ::2]`` subsampling between blocks
    :152-154  model5/6 (dilation 2) and model7
    :156-157  model8up(conv7_3) + model3short8(conv3_3) -> model8
    :160      (dist) softmax(0.2 * model_class(conv8_3)), nearest x4
    :170-175  model9/model10, 1x1 -> tanh -> x110

Differences from the shipped module, all deliberate: any batch size N (the
shipped forward hard-codes ``[None]``); optional float64; optional dict of
block activations for per-layer parity.  ``oracle/make_golden.py`` pins it
against the untouched reference module (max-abs diff 0.0 in fp32).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, models/pytorch/model.py:10


def _t(sd, key, dtype):
    return torch.from_numpy(np.ascontiguousarray(sd[key])).to(dtype)


def _conv(x, sd, key, dtype, dilation=1):
    w = _t(sd, key + ".weight", dtype)
    b = _t(sd, key + ".bias", dtype)
    pad = dilation * (w.shape[-1] // 2)
    return F.conv2d(x, w, b, stride=1, padding=pad, dilation=dilation)


def _deconv(x, sd, key, dtype):
    w = _t(sd, key + ".weight", dtype)
    b = _t(sd, key + ".bias", dtype)
    return F.conv_transpose2d(x, w, b, stride=2, padding=1)


def _bn(x, sd, key, dtype):
    return F.batch_norm(x, _t(sd, key + ".running_mean", dtype), _t(sd, key + ".running_var", dtype),
                        _t(sd, key + ".weight", dtype), _t(sd, key + ".bias", dtype),
                        training=False, eps=BN_EPS)


BILINEAR_US = ((.25, .5, .25, 0.), (.5, 1., .5, 0.), (.25, .5, .25, 0.), (0., 0., 0., 0.))   # colorize_image.py:410-413


def pred313_head(sd, feats, dtype=torch.float32, S=0.2, T=2.6):
    """313-bin head, restated from ``models/reference_model/deploy_nopred.prototxt:650-850`` (Caffe only in the
    reference; PARITY UNPINNED).  ``feats`` = dict of the post-BN block outputs conv3_3 .. conv8_3.
    Returns (pred_313 logits at H/4, dist_ab_S (N,313,H,W), pred_ab (N,2,H,W))."""
    hyper = _conv(feats["conv3_3"], sd, "pred.conv3_pred", dtype)                       # :650-665
    for i in (4, 5, 6, 7):                                                             # :666-729 Deconvolution 4x4 s2 p1
        hyper = hyper + _deconv(feats["conv%d_3" % i], sd, "pred.conv%d_pred" % i, dtype)
    hyper = F.relu(hyper + _conv(feats["conv8_3"], sd, "pred.conv8_pred", dtype))      # :730-763 Eltwise SUM, ReLU
    logits = _conv(hyper, sd, "pred.pred_313", dtype)                                  # :764-775
    k = torch.tensor(BILINEAR_US, dtype=dtype)[None, None].repeat(313, 1, 1, 1)        # grouped, shared kernel, no bias
    up = F.conv_transpose2d(logits, k, None, stride=2, padding=1, groups=313)          # pred_313_us :776-790
    up = F.conv_transpose2d(up, k, None, stride=2, padding=1, groups=313)              # pred_313_rs :791-805
    dist_S = F.softmax(up * S, dim=1)                                                  # scale_S + Softmax :807-821
    dist_T = F.softmax(up * T, dim=1)                                                  # scale_T + Softmax :826-840
    pred_ab = _conv(dist_T, sd, "pred.pred_ab", dtype)                                 # :842-850, weight = pts_in_hull.T
    return logits, dist_S, pred_ab


def global_branch(sd, glob, sat, dtype=torch.float32):
    """Global-Hints branch, restated from ``models/global_model/deploy_nodist.prototxt:37-172`` (Caffe only in the
    reference; PARITY UNPINNED: no Caffe here to run it against).  ``glob`` (N,314), ``sat`` (N,2) ->
    (N,512): ``relu(glob_conv1(g) + s_conv1(s))`` -> BN, then three ``1x1 conv -> ReLU -> BN`` stages
    (conv, ReLU, BatchNorm order of the prototxt; Eltwise default op = SUM ``:66-72``)."""
    g = torch.from_numpy(np.ascontiguousarray(np.asarray(glob, np.float32))).to(dtype)[:, :, None, None]
    s = torch.from_numpy(np.ascontiguousarray(np.asarray(sat, np.float32))).to(dtype)[:, :, None, None]
    y = F.relu(_conv(g, sd, "glob.glob_conv1", dtype) + _conv(s, sd, "glob.s_conv1", dtype))
    y = _bn(y, sd, "glob.bn1", dtype)
    for i in (2, 3, 4):
        y = _bn(F.relu(_conv(y, sd, "glob.glob_conv%d" % i, dtype)), sd, "glob.bn%d" % i, dtype)
    return y                                                   # (N,512,1,1)


def forward(sd, L_mc, ab, mask, maskcent=0.0, dist=False, dtype=torch.float32,
            return_acts=False, num_threads=None, glob=None, sat=None, l_div=100., ab_div=110., out_mul=110.,
            dist313=False, S=0.2):
    """Batched restatement.

    L_mc (N,1,H,W) in [-50,50]; ab (N,2,H,W) raw Lab ab; mask (N,1,H,W) in {0,1}
    (any array-like; bool accepted like the GUI passes, ``ui/gui_draw.py:274-275``).
    Returns ab (N,2,H,W) numpy of ``dtype`` -- the value the reference produces at
    ``data/colorize_image.py:263`` *before* lab2rgb -- and, if ``dist``, also the
    529-bin distribution (N,529,H,W).  H, W must be multiples of 8.
    """
    if num_threads is not None:
        torch.set_num_threads(int(num_threads))
    f32 = torch.float32
    with torch.no_grad():
        # model.py:139-142 -- torch.Tensor(np) casts to f32 first, whatever dtype follows
        A = torch.from_numpy(np.ascontiguousarray(np.asarray(L_mc, dtype=np.float64))).to(f32)
        B = torch.from_numpy(np.ascontiguousarray(np.asarray(ab, dtype=np.float64))).to(f32)
        M = torch.from_numpy(np.ascontiguousarray(np.asarray(mask, dtype=np.float64))).to(f32)
        M = M - maskcent
        x = torch.cat((A / l_div, B / ab_div, M), dim=1).to(dtype)         # :148 (Caffe twin: l_div = ab_div = 1)
        acts = {}
        relu = F.relu

        x = relu(_conv(x, sd, "model1.0", dtype)); acts["conv1_1"] = x
        x = relu(_conv(x, sd, "model1.2", dtype))
        conv1_2 = _bn(x, sd, "model1.4", dtype); acts["conv1_2"] = conv1_2
        x = conv1_2[:, :, ::2, ::2]                                      # :149
        x = relu(_conv(x, sd, "model2.0", dtype)); acts["conv2_1"] = x
        x = relu(_conv(x, sd, "model2.2", dtype))
        conv2_2 = _bn(x, sd, "model2.4", dtype); acts["conv2_2"] = conv2_2
        x = conv2_2[:, :, ::2, ::2]                                      # :150
        x = relu(_conv(x, sd, "model3.0", dtype)); acts["conv3_1"] = x
        x = relu(_conv(x, sd, "model3.2", dtype)); acts["conv3_2"] = x
        x = relu(_conv(x, sd, "model3.4", dtype))
        conv3_3 = _bn(x, sd, "model3.6", dtype); acts["conv3_3"] = conv3_3
        x = conv3_3[:, :, ::2, ::2]                                      # :151
        x = relu(_conv(x, sd, "model4.0", dtype)); acts["conv4_1"] = x
        x = relu(_conv(x, sd, "model4.2", dtype)); acts["conv4_2"] = x
        x = relu(_conv(x, sd, "model4.4", dtype))
        x = _bn(x, sd, "model4.6", dtype)
        if glob is not None:       # Global Hints: SpatialRepLayer broadcast + Eltwise SUM onto conv4_3norm
            if sat is None:        # (deploy_nodist.prototxt:501-518; caffe_traininglayers.py:45-46)
                sat = np.zeros((np.asarray(glob).shape[0], 2), np.float32)
            gvec = global_branch(sd, glob, sat, dtype)
            acts["glob_conv4norm"] = gvec
            x = x + gvec
        acts["conv4_3"] = x
        for blk, d in (("5", 2), ("6", 2), ("7", 1)):                   # :152-154
            for j, idx in enumerate(("0", "2", "4")):
                x = relu(_conv(x, sd, "model%s.%s" % (blk, idx), dtype, dilation=d))
                if j < 2:
                    acts["conv%s_%d" % (blk, j + 1)] = x
            x = _bn(x, sd, "model%s.6" % blk, dtype); acts["conv%s_3" % blk] = x
        conv7_3 = x
        short8 = _conv(conv3_3, sd, "model3short8.0", dtype); acts["conv3_3_short"] = short8
        up8 = _deconv(conv7_3, sd, "model8up.0", dtype) + short8         # :156
        x = relu(up8); acts["conv8_1"] = x                               # model8[0] ReLU (in-place)
        x = relu(_conv(x, sd, "model8.1", dtype)); acts["conv8_2"] = x
        x = relu(_conv(x, sd, "model8.3", dtype))
        conv8_3 = _bn(x, sd, "model8.5", dtype); acts["conv8_3"] = conv8_3
        if dist313:
            feats = {"conv3_3": conv3_3, "conv4_3": acts["conv4_3"], "conv5_3": acts["conv5_3"],
                     "conv6_3": acts["conv6_3"], "conv7_3": conv7_3, "conv8_3": conv8_3}
            lg, dS, pab = pred313_head(sd, feats, dtype, S=S)
            acts["pred_313"] = lg; acts["dist_ab_S"] = dS; acts["pred_ab"] = pab
        out_cl = None
        if dist:                                                         # :160
            logits = _conv(conv8_3, sd, "model_class.0", dtype)
            acts["class_logits"] = logits
            out_cl = F.interpolate(F.softmax(logits * .2, dim=1), scale_factor=4, mode="nearest")
        short9 = _conv(conv2_2, sd, "model2short9.0", dtype); acts["conv2_2_short"] = short9
        up9 = _deconv(conv8_3, sd, "model9up.0", dtype) + short9         # :170
        x = relu(up9); acts["conv9_1"] = x
        x = relu(_conv(x, sd, "model9.1", dtype))
        conv9_3 = _bn(x, sd, "model9.3", dtype); acts["conv9_2"] = conv9_3
        short10 = _conv(conv1_2, sd, "model1short10.0", dtype); acts["conv1_2_short"] = short10
        up10 = _deconv(conv9_3, sd, "model10up.0", dtype) + short10      # :172
        x = relu(up10); acts["conv10_1"] = x
        x = F.leaky_relu(_conv(x, sd, "model10.1", dtype), 0.2); acts["conv10_2"] = x   # :101-102
        out = torch.tanh(_conv(x, sd, "model_out.0", dtype)) * out_mul  # :108-109,174-175 (Caffe twin: x100)
        acts["out_ab"] = out
    res = out.numpy()
    if return_acts:
        acts = {k: v.numpy() for k, v in acts.items()}
        return (res, out_cl.numpy() if dist else None, acts)
    if dist:
        # NB the shipped dist branch returns out_reg*110*110 (model.py:166,168, a
        # reference bug the GUI never reads); the oracle returns the sane x110.
        return res, out_cl.numpy()
    return res


def load_into_reference_module(net, sd):
    """Load a numpy state dict into an instance of the *reference* nn.Module."""
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    net.load_state_dict(tsd, strict=True)
    net.eval()
    return net

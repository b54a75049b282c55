"""Independent float64 numpy restatement of the reference forward (TEST INFRASTRUCTURE ONLY).

Shares no arithmetic with torch: convolution is written out as the textbook
sum over taps, the transposed convolution as the textbook scatter, eval-BN as
``(x-mean)/sqrt(var+eps)*w+b``.  It exists to (1) cross-check the torch
restatement (so the oracle does not silently inherit a torch quirk) and
(2) provide the float64 "truth" against which both fp32 implementations (the
reference's ATen kernels and the HIP path) are measured.  Follows
``models/pytorch/model.py:134-175``; layer semantics per SURVEY.md Appendix C.
Use at small sizes (<= 64x64): it is O(seconds) there.
"""
import numpy as np

BN_EPS = 1e-5


def conv2d(x, w, b, dilation=1):
    """x (N,Cin,H,W), w (Cout,Cin,k,k) cross-correlation, zero pad = dilation*(k//2), stride 1."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    p = dilation * (k // 2)
    xp = np.zeros((n, cin, h + 2 * p, wd + 2 * p))
    xp[:, :, p:p + h, p:p + wd] = x
    out = np.zeros((n, cout, h, wd))
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, :, ky * dilation:ky * dilation + h, kx * dilation:kx * dilation + wd]
            out += np.einsum("oc,nchw->nohw", w[:, :, ky, kx], xs, optimize=True)
    return out + np.asarray(b, np.float64)[None, :, None, None]


def conv_transpose_4x4_s2_p1(x, w, b):
    """Scatter definition: out[co, 2i-1+ky, 2j-1+kx] += x[ci,i,j] * w[ci,co,ky,kx]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    n, cin, h, wd = x.shape
    cout = w.shape[1]
    full = np.zeros((n, cout, 2 * h + 2, 2 * wd + 2))
    for ky in range(4):
        for kx in range(4):
            contrib = np.einsum("co,nchw->nohw", w[:, :, ky, kx], x, optimize=True)
            full[:, :, ky:ky + 2 * h:2, kx:kx + 2 * wd:2] += contrib
    return full[:, :, 1:2 * h + 1, 1:2 * wd + 1] + np.asarray(b, np.float64)[None, :, None, None]


def batchnorm_eval(x, sd, key):
    mean = np.asarray(sd[key + ".running_mean"], np.float64)[None, :, None, None]
    var = np.asarray(sd[key + ".running_var"], np.float64)[None, :, None, None]
    g = np.asarray(sd[key + ".weight"], np.float64)[None, :, None, None]
    b = np.asarray(sd[key + ".bias"], np.float64)[None, :, None, None]
    return (x - mean) / np.sqrt(var + BN_EPS) * g + b


def relu(x):
    return np.maximum(x, 0.0)


def forward(sd, L_mc, ab, mask, maskcent=0.0, dist=False, return_acts=False):
    f32 = np.float32
    # model.py:139-148; the reference casts its inputs to f32 before normalising
    A = np.asarray(L_mc, np.float64).astype(f32)
    B = np.asarray(ab, np.float64).astype(f32)
    M = np.asarray(mask, np.float64).astype(f32) - f32(maskcent)
    x = np.concatenate((A / f32(100.), B / f32(110.), M), axis=1).astype(np.float64)
    acts = {}

    def c(x, key, d=1):
        return conv2d(x, sd[key + ".weight"], sd[key + ".bias"], d)

    x = relu(c(x, "model1.0")); acts["conv1_1"] = x
    conv1_2 = batchnorm_eval(relu(c(x, "model1.2")), sd, "model1.4"); acts["conv1_2"] = conv1_2
    x = relu(c(conv1_2[:, :, ::2, ::2], "model2.0")); acts["conv2_1"] = x
    conv2_2 = batchnorm_eval(relu(c(x, "model2.2")), sd, "model2.4"); acts["conv2_2"] = conv2_2
    x = relu(c(conv2_2[:, :, ::2, ::2], "model3.0")); acts["conv3_1"] = x
    x = relu(c(x, "model3.2")); acts["conv3_2"] = x
    conv3_3 = batchnorm_eval(relu(c(x, "model3.4")), sd, "model3.6"); acts["conv3_3"] = conv3_3
    x = relu(c(conv3_3[:, :, ::2, ::2], "model4.0")); acts["conv4_1"] = x
    x = relu(c(x, "model4.2")); acts["conv4_2"] = x
    x = batchnorm_eval(relu(c(x, "model4.4")), sd, "model4.6"); acts["conv4_3"] = x
    for blk, d in (("5", 2), ("6", 2), ("7", 1)):
        x = relu(c(x, "model%s.0" % blk, d)); acts["conv%s_1" % blk] = x
        x = relu(c(x, "model%s.2" % blk, d)); acts["conv%s_2" % blk] = x
        x = batchnorm_eval(relu(c(x, "model%s.4" % blk, d)), sd, "model%s.6" % blk); acts["conv%s_3" % blk] = x
    short8 = c(conv3_3, "model3short8.0"); acts["conv3_3_short"] = short8
    x = relu(conv_transpose_4x4_s2_p1(x, sd["model8up.0.weight"], sd["model8up.0.bias"]) + short8)
    acts["conv8_1"] = x
    x = relu(c(x, "model8.1")); acts["conv8_2"] = x
    conv8_3 = batchnorm_eval(relu(c(x, "model8.3")), sd, "model8.5"); acts["conv8_3"] = conv8_3
    out_cl = None
    if dist:
        logits = c(conv8_3, "model_class.0") * 0.2
        acts["class_logits"] = logits / 0.2
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        p = e / e.sum(axis=1, keepdims=True)
        out_cl = p.repeat(4, axis=2).repeat(4, axis=3)     # nearest x4, model.py:131,160
    short9 = c(conv2_2, "model2short9.0"); acts["conv2_2_short"] = short9
    x = relu(conv_transpose_4x4_s2_p1(conv8_3, sd["model9up.0.weight"], sd["model9up.0.bias"]) + short9)
    acts["conv9_1"] = x
    conv9_3 = batchnorm_eval(relu(c(x, "model9.1")), sd, "model9.3"); acts["conv9_2"] = conv9_3
    short10 = c(conv1_2, "model1short10.0"); acts["conv1_2_short"] = short10
    x = relu(conv_transpose_4x4_s2_p1(conv9_3, sd["model10up.0.weight"], sd["model10up.0.bias"]) + short10)
    acts["conv10_1"] = x
    x = c(x, "model10.1")
    x = np.where(x > 0, x, 0.2 * x); acts["conv10_2"] = x
    out = np.tanh(c(x, "model_out.0")) * 110.0
    acts["out_ab"] = out
    if return_acts:
        return out, out_cl, acts
    if dist:
        return out, out_cl
    return out

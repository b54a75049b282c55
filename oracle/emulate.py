"""CPU emulation of the HIP engine's REDUCED-PRECISION arithmetic (TEST / STUDY INFRASTRUCTURE ONLY, see oracle/__init__.py).

The bf16 path of the engine computes every conv with bf16 operands (stored activations and packed weights rounded to
bf16, RNE), fp32 accumulation, an fp32 epilogue (bias, shortcut sum, activation, eval-BN affine) and rounds the result
to bf16 for storage; conv10_2 is never rounded (the tanh head runs on the fp32 accumulators) and the shortcut convs'
sums never leave the accumulators (conv_ds_fused).  This module restates ``models/pytorch/model.py:148-175`` with those
roundings switchable PER LAYER, so that questions the GPU cannot answer cheaply get a number:

* which stored tensors carry the bf16 error of the ab map (``mode`` 'fp32' for a group of layers, 'bf16' for the rest --
  VERDICT r2 item 5; ``tools/bf16_attribution.py`` writes profiles/parity_r03.json);
* what a Winograd form of the 3x3 stride-1 convs would cost in accuracy BEFORE anyone writes the kernel
  (``mode`` 'wino2d' = F(2x2,3x3): input transform in fp32 then rounded to bf16, kernel transform in float64 from the fp32
  master weights then rounded to bf16, 16 position-GEMMs accumulated in fp32, output transform in fp32;
  'wino1d' = F(2,3) along x with the three dy taps direct) -- VERDICT r2 item 2.

Not bit-exact with the GPU (the fp32 summation order differs: ~1e-6 relative, far below the 2^-9 operand rounding that
is being studied); the all-'bf16' configuration is checked against the GPU's measured error in profiles/parity_r02.json.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .siggraph_torch import BN_EPS

# (name, state_dict key, kind, dilation, group).  kind: c = 3x3 conv, d = deconv 4x4 s2 (+ the shortcut conv `short`)
LAYERS = [
    ("conv1_1", "model1.0", "c", 1, "encoder"), ("conv1_2", "model1.2", "c", 1, "encoder"),
    ("conv2_1", "model2.0", "c", 1, "encoder"), ("conv2_2", "model2.2", "c", 1, "encoder"),
    ("conv3_1", "model3.0", "c", 1, "encoder"), ("conv3_2", "model3.2", "c", 1, "encoder"), ("conv3_3", "model3.4", "c", 1, "encoder"),
    ("conv4_1", "model4.0", "c", 1, "trunk"), ("conv4_2", "model4.2", "c", 1, "trunk"), ("conv4_3", "model4.4", "c", 1, "trunk"),
    ("conv5_1", "model5.0", "c", 2, "trunk"), ("conv5_2", "model5.2", "c", 2, "trunk"), ("conv5_3", "model5.4", "c", 2, "trunk"),
    ("conv6_1", "model6.0", "c", 2, "trunk"), ("conv6_2", "model6.2", "c", 2, "trunk"), ("conv6_3", "model6.4", "c", 2, "trunk"),
    ("conv7_1", "model7.0", "c", 1, "trunk"), ("conv7_2", "model7.2", "c", 1, "trunk"), ("conv7_3", "model7.4", "c", 1, "trunk"),
    ("conv8_1", "model8up.0", "d", 1, "decoder"), ("conv8_2", "model8.1", "c", 1, "decoder"), ("conv8_3", "model8.3", "c", 1, "decoder"),
    ("conv9_1", "model9up.0", "d", 1, "decoder"), ("conv9_2", "model9.1", "c", 1, "decoder"),
    ("conv10_1", "model10up.0", "d", 1, "decoder"), ("conv10_2", "model10.1", "c", 1, "decoder"),
]
LAYER_NAMES = [l[0] for l in LAYERS]
GROUPS = {g: [l[0] for l in LAYERS if l[4] == g] for g in ("encoder", "trunk", "decoder")}
SHORT_OF = {"conv8_1": "model3short8.0", "conv9_1": "model2short9.0", "conv10_1": "model1short10.0"}
WINO_ELIGIBLE = [l[0] for l in LAYERS if l[2] == "c" and l[0] not in ("conv1_1", "conv2_1", "conv3_1", "conv4_1")]   # 3x3, input stride 1


def q(x):
    """fp32 -> bf16 (RNE) -> fp32: the value a bf16 store / operand holds."""
    return x.to(torch.bfloat16).to(torch.float32)


def _w(sd, key):
    return torch.from_numpy(np.ascontiguousarray(sd[key + ".weight"])).float(), torch.from_numpy(np.ascontiguousarray(sd[key + ".bias"])).float()


def _bn(x, sd, key):
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sd[key + "." + k])).double()
    s = t("weight") / torch.sqrt(t("running_var") + BN_EPS)          # folded on the host in float64 like the packer
    sh = t("bias") - t("running_mean") * s
    return x * s.float()[None, :, None, None] + sh.float()[None, :, None, None]


# ---- Winograd F(2x2,3x3) / F(2,3) with bf16 operands ---------------------------------------------------------------
_BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
_G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)
_AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def _wino2d_d1(x, w, round_ops):
    """3x3, pad 1, stride 1, no bias.  x (N,C,H,W) fp32 (already operand-rounded), w (Co,C,3,3) fp32 master weights."""
    N, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    U = torch.einsum("ik,ockl,jl->ocij", _G, w.double(), _G).float()              # G g G^T in float64, rounded once
    xp = F.pad(x, (1, 1, 1, 1))
    P = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                          # (N,C,H/2,W/2,4,4)
    V = torch.einsum("ik,ncyxkl,jl->ncyxij", _BT, P, _BT)                           # B^T d B in fp32
    if round_ops:
        U, V = q(U), q(V)
    th, tw = H // 2, W // 2
    M = torch.empty((N, w.shape[0], th, tw, 4, 4))
    Vf = V.permute(4, 5, 1, 0, 2, 3).reshape(4, 4, C, N * th * tw)
    for i in range(4):
        for j in range(4):
            M[:, :, :, :, i, j] = (U[:, :, i, j] @ Vf[i, j]).reshape(w.shape[0], N, th, tw).permute(1, 0, 2, 3)
    Y = torch.einsum("ik,noyxkl,jl->noyxij", _AT, M, _AT)                           # (N,Co,th,tw,2,2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)


def _wino1d_d1(x, w, round_ops):
    """F(2,3) along x, the three dy taps direct."""
    N, C, H, W = x.shape
    assert W % 2 == 0
    U = torch.einsum("pk,ocyk->ocyp", _G, w.double()).float()                       # (Co,C,3,4)
    xp = F.pad(x, (1, 1, 1, 1))
    P = xp.unfold(3, 4, 2)                                                          # (N,C,H+2,W/2,4)
    V = torch.einsum("pk,ncyxk->ncyxp", _BT, P)                                     # (N,C,H+2,W/2,4)
    if round_ops:
        U, V = q(U), q(V)
    M = torch.stack([F.conv2d(V[..., p], U[..., p].unsqueeze(-1)) for p in range(4)], dim=-1)    # (N,Co,H,W/2,4)
    Y = torch.einsum("ip,noyxp->noyxi", _AT, M)                                     # (N,Co,H,W/2,2)
    return Y.reshape(N, w.shape[0], H, W)


_BT3 = torch.tensor([[1., -1., 0.], [0., 1., 0.], [0., -1., 1.]])
_G2 = torch.tensor([[1., 0.], [1., 1.], [0., 1.]], dtype=torch.float64)
_AT2 = torch.tensor([[1., 1., 0.], [0., 1., 1.]])
_KY = ((3, 1), (2, 0))          # deconv taps of output phase r in ascending input offset


def _wino_deconv(x, w, round_ops):
    """ConvTranspose2d 4x4 s2 p1 without bias as Winograd F(2x2,2x2) over its four output phases (conv_wino_deconv_*):
    x (N,C,H,W), w (C,Co,4,4) fp32 master weights; operands rounded to bf16 when ``round_ops``."""
    N, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    Co = w.shape[1]
    xp = F.pad(x, (1, 2, 1, 2))                                     # sites -1 .. H+1
    out = torch.empty((N, Co, 2 * H, 2 * W))
    for r in range(2):
        for s in range(2):
            g = torch.stack([torch.stack([w[:, :, _KY[r][a], _KY[s][b]] for b in range(2)], -1) for a in range(2)], -2)   # (C,Co,2,2)
            U = torch.einsum("ia,coab,jb->coij", _G2, g.double(), _G2).float()
            P = xp[:, :, r:r + H + 1, s:s + W + 1].unfold(2, 3, 2).unfold(3, 3, 2)           # (N,C,H/2,W/2,3,3)
            V = torch.einsum("ik,ncyxkl,jl->ncyxij", _BT3, P, _BT3)
            if round_ops:
                U, V = q(U), q(V)
            M = torch.einsum("coij,ncyxij->noyxij", U, V)
            Y = torch.einsum("ai,noyxij,bj->noyxab", _AT2, M, _AT2)                             # (N,Co,H/2,W/2,2,2)
            out[:, :, r::2, s::2] = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, Co, H, W)
    return out


def split_bf16(x, terms):
    """x (fp32) as a sum of `terms` bf16 values, largest first: x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1).  Three terms carry
    all 24 mantissa bits of an fp32 value (8 + 8 + 8), two carry 16."""
    parts, r = [], x
    for _ in range(terms):
        p = q(r)
        parts.append(p)
        r = r - p
    return parts


def split_f16(x, terms, scale_exp=0):
    """x (fp32) * 2^scale_exp as a sum of `terms` fp16 values (RNE, clamped to +-65504; subnormals as the hardware keeps them), each divided back by
    2^scale_exp (powers of two: exact).  IDC_FP16X3 (round 6): activations unscaled, a layer's weights with the power of two that brings max|w| into
    [8192, 16384) -- ``f16_weight_exponent`` below, csrc/idc_engine.hip's packer -- so that the lo part of a small weight is a NORMAL fp16 number."""
    s = 2.0 ** scale_exp
    parts, r = [], x * s
    for _ in range(terms):
        p = r.clamp(-65504., 65504.).to(torch.float16).to(torch.float32)
        parts.append(p / s)
        r = r - p
    return parts


def f16_weight_exponent(*ws):
    """the power of two shared by the given weight tensors (a deconv and the shortcut conv summed with it share one: one accumulator set)"""
    mx = max(float(w.abs().max()) for w in ws)
    if mx == 0.0:
        return 0
    return min(40, max(-10, 14 - int(np.frexp(np.float32(mx))[1])))


def _splits(mode, x, w, *more_w):
    """operand parts of one layer in a split mode: 'split2_fp32' / 'split3_fp32' (bf16 parts), 'splitf2_fp32' (fp16 parts, unscaled weights: what round 6
    first shipped), 'splitf2s_fp32' (fp16 parts, per-layer weight scale: IDC_FP16X3 as shipped)"""
    if mode.startswith("splitf"):
        terms = int(mode[6])
        e = f16_weight_exponent(w, *more_w) if mode[7:8] == "s" else 0
        return split_f16(x, terms), split_f16(w, terms, e), e
    terms = int(mode[5])
    return split_bf16(x, terms), split_bf16(w, terms), 0


def _split_products(xs, ws, op):
    """sum over the operand-term pairs (i, j) with i + j < terms of op(x_i, w_j), every product exact in fp32 (bf16 x bf16), accumulated in fp32
    smallest terms first: terms = 3 -> six bf16 products per fp32 product (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi), terms = 2 -> three."""
    terms = len(xs)
    pairs = sorted(((i, j) for i in range(terms) for j in range(terms) if i + j < terms), key=lambda ij: -(ij[0] + ij[1]))
    y = None
    for i, j in pairs:
        t = op(xs[i], ws[j])
        y = t if y is None else y + t
    return y


def _conv3(x, w, b, dilation, mode):
    if mode.startswith("split"):                         # 'split3_fp32' / 'split2_fp32' / 'splitf2[s]_fp32': an fp32 conv from 16-bit MFMAs on split operands
        xs, ws, _ = _splits(mode, x, w)
        y = _split_products(xs, ws, lambda a, c: F.conv2d(a, c, None, padding=dilation, dilation=dilation))
        return y + b[None, :, None, None]
    if mode in ("fp32", "bf16"):
        ww = q(w) if mode == "bf16" else w
        return F.conv2d(x, ww, b, padding=dilation, dilation=dilation)
    f = _wino2d_d1 if mode.startswith("wino2d") else _wino1d_d1
    round_ops = not mode.endswith("_fp32")
    if dilation == 1:
        y = f(x, w, round_ops)
    else:                                                # dilation 2 = four independent dilation-1 convs on the parity sub-grids
        y = torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]))
        for i in range(2):
            for j in range(2):
                y[:, :, i::2, j::2] = f(x[:, :, i::2, j::2].contiguous(), w, round_ops)
    return y + b[None, :, None, None]


def forward(sd, L_mc, ab, mask, maskcent=0.0, modes=None, default="bf16", l_div=100., ab_div=110., out_mul=110., return_acts=False):
    """modes: {layer name: 'fp32' | 'bf16' | 'wino2d' | 'wino1d' | 'wino2d_fp32' | 'wino1d_fp32' | 'split3_fp32' | 'split2_fp32'} (missing = ``default``).
    A layer in any mode but 'fp32' / '*_fp32' rounds its operands (input activation, weights) and its stored output to
    bf16; an 'fp32' layer rounds nothing.  Returns the ab map (N,2,H,W) float32 numpy."""
    modes = dict(modes or {})
    md = lambda n: modes.get(n, default)
    low = lambda n: not md(n).endswith("fp32")          # the layer's operands / output are bf16
    with torch.no_grad():
        A = torch.from_numpy(np.ascontiguousarray(np.asarray(L_mc, dtype=np.float64))).float()
        B = torch.from_numpy(np.ascontiguousarray(np.asarray(ab, dtype=np.float64))).float()
        M = torch.from_numpy(np.ascontiguousarray(np.asarray(mask, dtype=np.float64))).float() - maskcent
        x = torch.cat((A / l_div, B / ab_div, M), dim=1)
        acts = {}

        def conv(name, key, x, dil=1, act="relu", bn=None, store=True):
            w, b = _w(sd, key)
            xin = q(x) if low(name) else x
            y = _conv3(xin, w, b, dil, md(name))
            y = F.relu(y) if act == "relu" else (F.leaky_relu(y, 0.2) if act == "leaky" else y)
            if bn:
                y = _bn(y, sd, bn)
            if store and low(name):
                y = q(y)
            acts[name] = y
            return y

        def up(name, key, x, skip):
            w, b = _w(sd, key)
            ws, bs = _w(sd, SHORT_OF[name])
            if md(name).startswith("split"):
                xs_, wd_, _ = _splits(md(name), x, w, ws)               # (the deconv and its shortcut conv share the weight exponent)
                sk_, wsh_, _ = _splits(md(name), skip, ws, w)
                y = _split_products(xs_, wd_, lambda a, c: F.conv_transpose2d(a, c, None, stride=2, padding=1)) + b[None, :, None, None]
                y = y + _split_products(sk_, wsh_, lambda a, c: F.conv2d(a, c, None, padding=1)) + bs[None, :, None, None]
            elif md(name).startswith("wino"):                   # deconv as F(2x2,2x2), shortcut conv as F(2x2,3x3)
                ro = not md(name).endswith("_fp32")
                xin, sk = (q(x), q(skip)) if ro else (x, skip)
                sc = _wino2d_d1(sk, ws, ro) + bs[None, :, None, None]
                if ro:
                    sc = q(sc)                                  # the shortcut conv is its own launch on the click path: stored bf16
                y = _wino_deconv(xin, w, ro) + b[None, :, None, None] + sc
            elif low(name):
                y = F.conv_transpose2d(q(x), q(w), b, stride=2, padding=1) + F.conv2d(q(skip), q(ws), bs, padding=1)
            else:
                y = F.conv_transpose2d(x, w, b, stride=2, padding=1) + F.conv2d(skip, ws, bs, padding=1)
            y = F.relu(y)
            if low(name):
                y = q(y)
            acts[name] = y
            return y

        x = conv("conv1_1", "model1.0", x)
        c12 = conv("conv1_2", "model1.2", x, bn="model1.4")
        x = conv("conv2_1", "model2.0", c12[:, :, ::2, ::2])
        c22 = conv("conv2_2", "model2.2", x, bn="model2.4")
        x = conv("conv3_1", "model3.0", c22[:, :, ::2, ::2])
        x = conv("conv3_2", "model3.2", x)
        c33 = conv("conv3_3", "model3.4", x, bn="model3.6")
        x = conv("conv4_1", "model4.0", c33[:, :, ::2, ::2])
        x = conv("conv4_2", "model4.2", x)
        x = conv("conv4_3", "model4.4", x, bn="model4.6")
        for blk, d in (("5", 2), ("6", 2), ("7", 1)):
            x = conv("conv%s_1" % blk, "model%s.0" % blk, x, d)
            x = conv("conv%s_2" % blk, "model%s.2" % blk, x, d)
            x = conv("conv%s_3" % blk, "model%s.4" % blk, x, d, bn="model%s.6" % blk)
        x = up("conv8_1", "model8up.0", x, c33)
        x = conv("conv8_2", "model8.1", x)
        c83 = conv("conv8_3", "model8.3", x, bn="model8.5")
        x = up("conv9_1", "model9up.0", c83, c22)
        c92 = conv("conv9_2", "model9.1", x, bn="model9.3")
        x = up("conv10_1", "model10up.0", c92, c12)
        x = conv("conv10_2", "model10.1", x, act="leaky", store=False)        # never rounded: the head reads the accumulators
        wo, bo = _w(sd, "model_out.0")
        out = torch.tanh(F.conv2d(x, wo, bo)) * out_mul
    if return_acts:
        return out.numpy(), {k: v.numpy() for k, v in acts.items()}
    return out.numpy()


def error_stats(out, ref):
    d = np.abs(out.astype(np.float64) - ref.astype(np.float64))
    return {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "q999": float(np.quantile(d, 0.999)),
            "rel_rms": float(np.sqrt((d ** 2).mean()) / max(np.sqrt((ref.astype(np.float64) ** 2).mean()), 1e-30))}

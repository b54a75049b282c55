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

#!/usr/bin/env python3
"""Fixtures for the two Caffe-only branches of the reference: the Global-Hints fusion
(models/global_model/deploy_nodist.prototxt) and the 313-bin distribution / soft-decode head
(models/reference_model/deploy_nopred.prototxt).

PARITY UNPINNED: Caffe cannot be installed here and the reference ships no weights or outputs for these nets,
so these vectors come from the oracle's own torch restatement of the prototxt (oracle/siggraph_torch.py) with
seeded weights -- they pin the HIP path to the restatement, not to Caffe.

    python oracle/make_golden_caffe_branches.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import siggraph_torch, weights  # noqa: E402
from interactive_deep_colorization_amd import workloads  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def global_hints_case(name="glob64_he_s3", seed=3, n=2, X=64):
    sd = weights.add_global_branch(weights.make_state_dict(seed, "he", include_class=False), seed)
    L, ab, mask = workloads.random_batch(n, X, seed=11, max_points=4, max_p=2)
    ab = ab * 0; mask = mask * 0                                  # the global net silences the local hint planes
    glob, sat = workloads.global_hint_config5(n, seed=1)
    glob[1] = 0.0                                                 # image 1: "no histogram" (all-zero input, flag 0)
    out, _, acts = siggraph_torch.forward(sd, L, ab, mask, 0.0, glob=glob, sat=sat, return_acts=True)
    out64 = siggraph_torch.forward(sd, L, ab, mask, 0.0, glob=glob, sat=sat, dtype=torch.float64)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), L_mc=L.astype(np.float32), ab=ab.astype(np.float32),
                        mask=mask.astype(np.float32), glob=glob, sat=sat, out_ab=out.astype(np.float32),
                        out_ab_f64=out64, glob_vec=acts["glob_conv4norm"][:, :, 0, 0].astype(np.float32),
                        conv4_3=acts["conv4_3"].astype(np.float32), weight_seed=np.int64(seed),
                        weight_style=np.array("he"))
    print("%s: out [%.1f, %.1f], f32 vs f64 %.2e" % (name, out.min(), out.max(), np.abs(out - out64).max()))


def dist313_case(name="dist313_64_he_s2", seed=2, X=64):
    sd = weights.add_pred313_head(weights.make_state_dict(seed, "he", include_class=False), seed)
    L, ab, mask = workloads.random_batch(1, X, seed=12, max_points=5, max_p=2)
    out, _, acts = siggraph_torch.forward(sd, L, ab, mask, 0.0, dist313=True, return_acts=True)
    o64, _, a64 = siggraph_torch.forward(sd, L, ab, mask, 0.0, dist313=True, return_acts=True, dtype=torch.float64)
    rs = np.random.RandomState(0)
    pos = rs.randint(0, X * X, 48)                                # sampled pixels of the 313 x X x X distribution
    dS = acts["dist_ab_S"][0].reshape(313, -1)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), L_mc=L.astype(np.float32), ab=ab.astype(np.float32),
                        mask=mask.astype(np.float32), out_ab=out.astype(np.float32),
                        pred_313=acts["pred_313"].astype(np.float32), pred_ab=acts["pred_ab"].astype(np.float32),
                        pred_ab_f64=a64["pred_ab"], dist_pos=pos, dist_samples=dS[:, pos].astype(np.float32),
                        dist_entropy=-(dS * np.log(dS)).sum(0).astype(np.float32), weight_seed=np.int64(seed))
    print("%s: pred_ab [%.1f, %.1f], f32 vs f64 %.2e, logits std %.2f" % (
        name, acts["pred_ab"].min(), acts["pred_ab"].max(), np.abs(acts["pred_ab"] - a64["pred_ab"]).max(),
        acts["pred_313"].std()))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    global_hints_case()
    dist313_case()

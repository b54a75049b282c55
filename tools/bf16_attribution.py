#!/usr/bin/env python3
"""Where does the bf16 error of the ab map come from, and what would a Winograd form add?  

CPU study on the bench workload (BASELINE configs[2]: 256x256, the first `--n` images of the N=32 bench batch, he-style
and torch-init weights): oracle/emulate.py restates the engine's bf16 arithmetic (bf16 operands and stored activations,
fp32 accumulation and epilogue) with the rounding switchable per layer.  For each stored-tensor group the group is kept
fp32 (operands, weights and outputs unrounded) while everything else stays bf16; then selected layers are switched to the
Winograd forms with bf16 operands.  Error = against the float64 oracle.  Writes profiles/parity_r03.json.

    python tools/bf16_attribution.py [--n 4] [--out profiles/parity_r03.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from interactive_deep_colorization_amd import workloads   # noqa: E402
from oracle import emulate, siggraph_torch, weights       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "parity_r03.json"))
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    G = emulate.GROUPS
    E = emulate.WINO_ELIGIBLE
    small = [n for n in E if n in ("conv1_2", "conv2_2", "conv9_2", "conv10_2")]                  # <= 128 couts: where a fused form is feasible
    trunk = [n for n in E if n in G["trunk"]]
    variants = [
        ("all_bf16", {}),
        ("encoder_fp32", {n: "fp32" for n in G["encoder"]}),
        ("trunk_fp32", {n: "fp32" for n in G["trunk"]}),
        ("decoder_fp32", {n: "fp32" for n in G["decoder"]}),
        ("conv9_2_conv10_1_fp32", {"conv9_2": "fp32", "conv10_1": "fp32"}),
        ("last_block_fp32 (conv10_1, conv10_2)", {"conv10_1": "fp32", "conv10_2": "fp32"}),
        ("model9_model10_fp32", {n: "fp32" for n in ("conv9_1", "conv9_2", "conv10_1", "conv10_2")}),
        ("encoder_and_trunk_fp32", {n: "fp32" for n in G["encoder"] + G["trunk"]}),
        ("wino2d_trunk", {n: "wino2d" for n in trunk}),
        ("wino2d_all_3x3_s1", {n: "wino2d" for n in E}),
        ("click_path_as_shipped (wino2d on conv2_1 .. conv9_2 incl. the strided-view convs, F(2x2,2x2) on conv8_1 / conv9_1)",
         dict({n: "wino2d" for n in E + ["conv2_1", "conv3_1", "conv4_1"] if n not in ("conv1_2", "conv10_2")}, conv8_1="wino2d", conv9_1="wino2d")),
        ("wino1d_small (conv1_2, conv2_2, conv9_2, conv10_2)", {n: "wino1d" for n in small}),
        ("wino1d_all_3x3_s1", {n: "wino1d" for n in E}),
    ]
    res = {"config": "BASELINE configs[2] geometry: %dx%d, first %d images of the bench batch (workloads.random_batch(n, 256, seed=0)), "
                     "maskcent 0; error of the ab map (+-110) against the float64 oracle" % (args.size, args.size, args.n),
           "how": "oracle/emulate.py: CPU restatement of the engine's bf16 arithmetic with the rounding switchable per layer; "
                  "validated by the all_bf16 row against the GPU's measured error (profiles/parity_r02.json: he 13.1 / 1.2, torch-init 0.136 / 0.022 at N=32)",
           "groups": G, "wino_eligible": E, "weights": {}}
    L, ab, m = workloads.random_batch(args.n, args.size, seed=0)
    for style in ("he", "torch"):
        sd = weights.make_state_dict(0, style)
        t0 = time.time()
        ref64 = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64)
        ref32 = siggraph_torch.forward(sd, L, ab, m, 0.0)
        rows = {"fp32_oracle_vs_fp64": emulate.error_stats(ref32, ref64)}
        print(style, "oracles %.1f s" % (time.time() - t0), rows["fp32_oracle_vs_fp64"], flush=True)
        for name, modes in variants:
            t0 = time.time()
            out = emulate.forward(sd, L, ab, m, 0.0, modes=modes)
            rows[name] = emulate.error_stats(out, ref64)
            rows[name]["layers"] = sorted(modes)
            print(style, name, {k: round(v, 4) for k, v in rows[name].items() if k != "layers"}, "%.1f s" % (time.time() - t0), flush=True)
        res["weights"][style] = rows
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()

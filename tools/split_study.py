#!/usr/bin/env python3
"""CPU study (oracle/emulate.py): what would an fp32 path built from bf16 MFMAs on SPLIT operands cost in accuracy?

fp32 value = hi + mid + lo in bf16 (8 + 8 + 8 mantissa bits).  'split3': six bf16 products per fp32 product (terms below 2^-24 dropped), fp32
accumulation -- 6/16 of the exact-fp32 MFMA's time on gfx950 (v_mfma_f32_16x16x32_bf16 does 16x the MACs per cycle of v_mfma_f32_16x16x4_f32);
'split2': hi + lo (16 bits), three products -- 3/16.  Activations stay fp32 between layers, as on the fp32 path.  Prints the ab-map error against
the float64 oracle beside the plain fp32 arithmetic's, BASELINE configs[1] (one 256x256 image, 5 hints), both weight styles.

    python tools/split_study.py > profiles/r04_split_study.txt
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from interactive_deep_colorization_amd import workloads  # noqa: E402
from oracle import emulate, siggraph_torch, weights  # noqa: E402

torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print("# %dx%d, one image, 5 hints; error of the ab map (range +-110) against the float64 oracle" % (size, size))
print("%-8s %-14s %12s %12s %12s %12s" % ("weights", "arithmetic", "max_abs", "mean_abs", "q99.9", "rel_rms"))
for style in ("torch", "he"):
    sd = weights.make_state_dict(0, style)
    L = workloads.random_batch(1, size, seed=7)[0].astype(np.float32)
    hab, hm = workloads.hints_config2(size, 5, 3, 0)
    ab, m = hab[None].astype(np.float32), hm[None].astype(np.float32)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64)
    # round 6: 'splitf2' = fp16 parts (IDC_FP16X3 as first shipped), 'splitf2s' = with the per-layer power-of-two weight scale (as shipped now);
    # conv1_1 is an exact-fp32 island in every split mode, as on the GPU
    for mode in ("fp32", "split3_fp32", "split2_fp32", "splitf2_fp32", "splitf2s_fp32", "bf16"):
        t0 = time.time()
        out = emulate.forward(sd, L, ab, m, 0.0, default=mode, modes={"conv1_1": "fp32"} if mode.startswith("split") else None)
        st = emulate.error_stats(out, ref)
        print("%-8s %-14s %12.3e %12.3e %12.3e %12.3e   (%.0f s)" % (style, mode, st["max_abs"], st["mean_abs"], st["q999"], st["rel_rms"], time.time() - t0))

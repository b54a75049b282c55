#!/usr/bin/env python3
"""Is the bench forward limited by cycles or by the power cap?  Same engine, same launches, three operand sets:
  he      the bench's weights and inputs (full-range activations)
  small   the same weights scaled so that activations die out (ReLU outputs mostly 0 after a few layers)
  zero    all-zero weights and inputs: every MFMA operand is 0, the data path does not toggle
The instruction streams are identical (nothing in the kernels branches on data), so any time difference is the
DVFS response to switching activity.  Prints ms per N=32 forward and the conv-stack TFLOP/s of each.
Usage: python tools/power_probe.py [steps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (first: see tests/conftest.py on the two HIP runtimes)

torch.cuda.init()
from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

FLOP = 4811437113344.0
N, H = 32, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(sd, L, ab, m, tag):
    e = engine.HipColorizer(H, H, max_batch=N, precision="bf16", device=0)
    e.load_state_dict(sd)
    dev = torch.device("cuda", 0)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
    dout = torch.empty((N, 2, H, H), dtype=torch.float32, device=dev)
    for _ in range(5):
        e.forward_device(N, dL, dab, dm, dout, 0.0, sync=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.forward_device(N, dL, dab, dm, dout, 0.0, sync=False)
    e.forward_device(N, dL, dab, dm, dout, 0.0, sync=True)
    ms = (time.perf_counter() - t0) / (steps + 1) * 1e3
    e.close()
    return {"operands": tag, "ms_per_forward": round(ms, 4), "tflops": round(FLOP / ms / 1e9, 1),
            "frac_of_2500": round(FLOP / ms / 1e9 / 2500.0, 4)}


sd = workloads.random_state_dict(0, "he")
L, ab, m = workloads.random_batch(N, H, seed=0)
res = [run(sd, L, ab, m, "he (bench)")]
sd_small = {k: (v * 0.05 if (k.endswith(".weight") and v.ndim == 4) else v) for k, v in sd.items()}
res.append(run(sd_small, L, ab, m, "conv weights x0.05"))
sd_zero = {k: np.zeros_like(v) for k, v in sd.items()}
res.append(run(sd_zero, np.zeros_like(L), np.zeros_like(ab), np.zeros_like(m), "all zero"))
res.append(run(sd, L, ab, m, "he (bench), again"))
for r in res:
    print(json.dumps(r))

#!/usr/bin/env python3
"""Per-layer times of the bench workload (bf16, N=32, 256x256) for the build under <root> (default: this tree), as
one JSON line -- for same-box A/B of two worktrees:  python tools/quick_layers.py _ab ; python tools/quick_layers.py ."""
import json
import os
import sys

import numpy as np

# usage: quick_layers.py [root] [precision=bf16] [batch=32]
root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), ".."))
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sys.path.insert(0, root)
import torch                                                           # noqa: E402
from interactive_deep_colorization_amd import engine, workloads        # noqa: E402

nb = NB
e = engine.HipColorizer(256, 256, max_batch=nb, precision=PREC)
e.load_state_dict(workloads.random_state_dict(0, "he"))
L, ab, m = workloads.random_batch(nb, 256, seed=0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
dout = torch.empty((nb, 2, 256, 256), dtype=torch.float32, device=dev)
for _ in range(5):
    e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=True)
import time                                                            # noqa: E402
t0 = time.perf_counter()
for _ in range(30):
    e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
e.sync()
whole = (time.perf_counter() - t0) / 30 * 1e3
e.set_profiling(True)
for _ in range(10):
    e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
e.sync()
ms = e.layer_times_ms()
rows = {r["name"]: [round(float(ms[r["index"]]), 4), r["kernel"]] for r in e.layer_table() if ms[r["index"]] > 0}
print(json.dumps({"root": os.path.basename(root), "precision": PREC, "batch": nb, "ms_per_forward": round(whole, 4), "img_s": round(nb / whole * 1e3, 1), "layers": rows}))

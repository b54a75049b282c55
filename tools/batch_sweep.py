#!/usr/bin/env python3
"""Images/sec of the 256x256 forward against the per-GPU batch (device-resident, torch-init weights), one precision per call.
usage: batch_sweep.py [precision=bf16] [batches=1,2,4,8,16,32,64,128]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch                                                           # noqa: E402
from interactive_deep_colorization_amd import engine, workloads        # noqa: E402

PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"
NS = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16,32,64,128").split(",")]
sd = workloads.random_state_dict(0, "torch")
dev = torch.device("cuda", 0)
rows = []
for nb in NS:
    L, ab, m = workloads.random_batch(nb, 256, seed=0)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
    dout = torch.empty((nb, 2, 256, 256), dtype=torch.float32, device=dev)
    e = engine.HipColorizer(256, 256, max_batch=nb, precision=PREC)
    e.load_state_dict(sd)
    for _ in range(5):
        e.forward_device(nb, dL, dab, dm, dout, 0.5, sync=True)
    steps = max(10, min(200, 2000 // nb))
    t0 = time.perf_counter()
    for _ in range(steps):
        e.forward_device(nb, dL, dab, dm, dout, 0.5, sync=False)
    e.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    rows.append({"batch": nb, "ms_per_forward": round(ms, 4), "img_s": round(nb / ms * 1e3, 1), "steps": steps})
    e.close()
    del dL, dab, dm, dout
print(json.dumps({"precision": PREC, "rows": rows}))

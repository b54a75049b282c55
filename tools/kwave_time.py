#!/usr/bin/env python3
"""Per-launch times of the bf16 click path (one 256x256 image) with `kwave` = argv[1] (0 | 1): median of 30 profiled forwards."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

kw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
engine.set_option("kwave", kw)
sd = workloads.random_state_dict(0, "he")
L = workloads.random_batch(1, size, seed=7)[0].astype(np.float32)
hab, hm = workloads.hints_config2(size, 5, 3, 0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (L, hab[None], hm[None]))
dout = torch.empty((1, 2, size, size), dtype=torch.float32, device=dev)
e = engine.HipColorizer(size, size, max_batch=1, precision=prec)
e.load_state_dict(sd)
for _ in range(30):
    e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
import time
ts = []
for _ in range(300):
    t0 = time.perf_counter()
    e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
    ts.append(time.perf_counter() - t0)
p50 = float(np.median(ts)) * 1e3
e.set_profiling(True)
for _ in range(30):
    e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
lo, med, hi = e.layer_times_stats()
rows = [r for r in e.layer_table() if r["launches"] > 0]
tag = "%s %d kwave=%d side_stream=%s" % (prec, size, kw, os.environ.get("IDC_SIDE_STREAM", "1"))
pick = ["conv1_1", "conv2_2", "conv3_2", "conv5_2", "conv7_3", "conv8_1", "conv8_2", "conv9_1", "conv9_2", "conv10_1", "conv10_2"]
print(tag, " ".join("%s:%.1f" % (r["name"], med[r["index"]] * 1e3) for r in rows if r["name"] in pick),
      "| sum of launches %.1f us | forward p50 %.4f ms" % (sum(med[r["index"]] for r in rows) * 1e3, p50))
e.close()

"""Same-box A/B of the bf16 click forward with model10up + shortcut as 128-cout 8-wave workgroups (ds_mfma16 = 2) and as 64-cout 4-wave ones (1)."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

sd = workloads.random_state_dict(0, "torch")
L = workloads.random_batch(1, 256, seed=7)[0].astype(np.float32)
hab, hm = workloads.hints_config2(256, 5, 3, 0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, hab[None].astype(np.float32), hm[None].astype(np.float32)))
dout = torch.empty((1, 2, 256, 256), dtype=torch.float32, device=dev)
for rep in range(3):
    for mode in (2, 1):
        engine.set_option("ds_mfma16", mode)
        e = engine.HipColorizer(256, 256, max_batch=1, precision="bf16")
        e.load_state_dict(sd)
        for _ in range(30):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
            ts.append(time.perf_counter() - t0)
        e.set_profiling(True)
        for _ in range(20):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        _, med, _ = e.layer_times_stats()
        tab = {r["name"]: float(med[r["index"]]) for r in e.layer_table()}
        print("ds_mfma16=%d pass %d: click p50 %.4f ms   conv10_1 %.1f us (per-launch events)  conv10_2 %.1f us" % (
            mode, rep, statistics.median(ts) * 1e3, tab["conv10_1"] * 1e3, tab["conv10_2"] * 1e3), flush=True)
        e.close()

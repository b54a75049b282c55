#!/usr/bin/env python3
"""conv_kwave_bf16 against conv_wino_bf16 on the bf16 click path (one 256x256 image, 5 hints): per-layer error against the float64
oracle, the ab map, and the device-resident forward time (p50 of 200, alternating handles).  GPU only; the oracle is the checker."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from interactive_deep_colorization_amd import engine, workloads  # noqa: E402
from oracle import siggraph_torch  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
style = sys.argv[2] if len(sys.argv) > 2 else "he"
sd = workloads.random_state_dict(0, style)
L = workloads.random_batch(1, size, seed=7)[0].astype(np.float32)
hab, hm = workloads.hints_config2(size, 5, 3, 0)
ab, m = hab[None].astype(np.float32), hm[None].astype(np.float32)
ref, _, acts = siggraph_torch.forward(sd, L, ab, m, 0.0, return_acts=True, dtype=torch.float64)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (L, ab, m))
dout = torch.empty((1, 2, size, size), dtype=torch.float32, device=dev)
res = {}
handles = {}
for kw in (0, 1):
    engine.set_option("kwave", kw)
    e = engine.HipColorizer(size, size, max_batch=1, precision="bf16")
    e.load_state_dict(sd)
    out = e.forward(L, ab, m, 0.0)
    table = [(r["name"], r["kernel"]) for r in e.layer_table() if r["launches"] > 0]
    errs = {}
    for name, kern in table:
        if name in acts and (kern.startswith("conv_kwave") or kern.startswith("conv_wino_")):
            got = e.activation(name, 1)
            errs[name] = (kern, float(np.abs(got - acts[name]).max()), float(np.abs(got - acts[name]).mean()), float(np.abs(acts[name]).max()))
    d = np.abs(out - ref)
    res[kw] = {"ab_max": float(d.max()), "ab_mean": float(d.mean()), "layers": errs, "n_kw": sum(k.startswith("conv_kwave") for _, k in table)}
    np.testing.assert_array_equal(e.forward(L, ab, m, 0.0), out)
    handles[kw] = e
for kw in (0, 1):
    print("kwave=%d  launches on the form: %d   ab map vs float64 oracle: max %.4f mean %.5f" % (kw, res[kw]["n_kw"], res[kw]["ab_max"], res[kw]["ab_mean"]))
print("%-14s %-22s %10s %10s | %-22s %10s %10s | %8s" % ("layer", "kernel", "max", "mean", "kernel", "max", "mean", "|ref|max"))
for name in res[1]["layers"]:
    a0 = res[0]["layers"].get(name, ("-", 0, 0, 0)); a1 = res[1]["layers"][name]
    print("%-14s %-22s %10.4f %10.5f | %-22s %10.4f %10.5f | %8.2f" % (name, a0[0][:22], a0[1], a0[2], a1[0][:22], a1[1], a1[2], a1[3]))
ts = {0: [], 1: []}
for rep in range(4):
    for kw in (0, 1):
        e = handles[kw]
        for _ in range(20):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        for _ in range(200):
            t0 = time.perf_counter()
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
            ts[kw].append(time.perf_counter() - t0)
for kw in (0, 1):
    print("kwave=%d  forward p50 %.4f ms  (min %.4f)" % (kw, float(np.median(ts[kw])) * 1e3, float(np.min(ts[kw])) * 1e3))
for kw in (0, 1):
    e = handles[kw]
    e.set_profiling(True)
    for _ in range(20):
        e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
    lo, med, hi = e.layer_times_stats()
    e.set_profiling(False)
    rows = e.layer_table()
    print("kwave=%d per launch (us, median of 20):" % kw, " ".join("%s:%.1f" % (r["name"], med[r["index"]] * 1e3) for r in rows if r["launches"] > 0))
    e.close()

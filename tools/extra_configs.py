#!/usr/bin/env python3
"""Throughput of the non-headline BASELINE.json configs on one MI355X (informational; bench.py stays on configs[2]):
configs[4] 512x512 N=8 bf16 with Global Hints, and the fp32 path at N=32 / 256x256."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactive_deep_colorization_amd import engine, workloads
from oracle import weights

dev = torch.device("cuda", 0)
def run(H, nb, prec, glob):
    sd = weights.make_state_dict(0, "he", include_class=False)
    if glob:
        sd = weights.add_global_branch(sd, 0)
    e = engine.HipColorizer(H, H, max_batch=nb, precision=prec, global_hints=glob)
    e.load_state_dict(sd)
    if glob:
        g, s = workloads.global_hint_config5(nb, seed=0)
        e.set_global_hints(g, s)
    L, ab, m = workloads.random_batch(nb, H, seed=0)
    t = [torch.from_numpy(x).to(dev) for x in (L, ab, m)]
    o = torch.empty((nb, 2, H, H), dtype=torch.float32, device=dev)
    for _ in range(3): e.forward_device(nb, t[0], t[1], t[2], o, 0.0, sync=True)
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps): e.forward_device(nb, t[0], t[1], t[2], o, 0.0, sync=False)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    gflop = 150.391 * (H / 256.0) ** 2 * nb
    print("%dx%d N=%d %s%s: %.3f ms/forward, %.1f img/s, %.1f TFLOP/s" % (H, H, nb, prec, " +global hints" if glob else "",
                                                                          dt * 1e3, nb / dt, gflop / dt / 1e3))
    e.close()
run(512, 8, "bf16", True)
run(256, 32, "fp32", False)
run(256, 32, "bf16", False)

#!/usr/bin/env python3
"""Probe: does running the batch as two half-batches on two engine streams (kernels of both interleave on the
chip) beat one batch-32 stream?  (tails / launch gaps / burst spreading).  Tuning experiment, not the product path."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactive_deep_colorization_amd import engine, workloads

sd = workloads.random_state_dict(0, "he")
dev = torch.device("cuda", 0)
def mk(nb, seed):
    e = engine.HipColorizer(256, 256, max_batch=nb, precision="bf16")
    e.load_state_dict(sd)
    L, ab, m = workloads.random_batch(nb, 256, seed=seed)
    t = [torch.from_numpy(x).to(dev) for x in (L, ab, m)]
    out = torch.empty((nb, 2, 256, 256), dtype=torch.float32, device=dev)
    return e, t, out
def run(engs, steps=20):
    for e, t, o in engs:
        for _ in range(3): e.forward_device(t[0].shape[0], t[0], t[1], t[2], o, 0.0, sync=False)
    for e, _, _ in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, t, o in engs: e.forward_device(t[0].shape[0], t[0], t[1], t[2], o, 0.0, sync=False)
    for e, _, _ in engs: e.sync()
    dt = time.perf_counter() - t0
    n = sum(t[0].shape[0] for _, t, _ in engs)
    return n * steps / dt
one = [mk(32, 0)]
print("1 x 32      : %.0f img/s" % run(one))
two = [mk(16, 0), mk(16, 1)]
print("2 x 16 (2 streams): %.0f img/s" % run(two))
four = [mk(8, i) for i in range(4)]
print("4 x 8  (4 streams): %.0f img/s" % run(four))
print("1 x 32 again: %.0f img/s" % run(one))

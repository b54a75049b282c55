#!/usr/bin/env python3
"""One batch of 32 images per step as ONE N=32 forward against TWO concurrent N=16 forwards on two handles (two streams):
does space-sharing the chip hide launch floors and tile tails?  bf16, 256x256, device-resident, K steps per timed region."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

K = 20
sd = workloads.random_state_dict(0, "he")
dev = torch.device("cuda", 0)
L, ab, m = workloads.random_batch(32, 256, seed=0)
dL, dab, dm = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (L, ab, m))
dout = torch.empty((32, 2, 256, 256), dtype=torch.float32, device=dev)


def make(nb):
    e = engine.HipColorizer(256, 256, max_batch=nb, precision="bf16", throughput_blob=True)
    e.load_state_dict(sd)
    return e


def region(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / K)
    return best


e32 = make(32)
parts = {}
for split in (2, 4):
    hs = [make(32 // split) for _ in range(split)]
    parts[split] = hs


def one():
    e32.forward_device(32, dL, dab, dm, dout, 0.0, sync=False)
    e32.sync()


def multi(split):
    hs = parts[split]
    nb = 32 // split

    def f():
        for i, h in enumerate(hs):
            s = slice(i * nb, (i + 1) * nb)
            h.forward_device(nb, dL[s], dab[s], dm[s], dout[s], 0.0, sync=False)
        for h in hs:
            h.sync()
    return f


for rep in range(2):
    r = region(one)
    print("one N=32 forward per step      : %s ms  -> %.0f img/s" % (" ".join("%.4f" % (x * 1e3) for x in r), 32 / min(r)))
    for split in (2, 4):
        r = region(multi(split))
        print("%d concurrent N=%d forwards      : %s ms  -> %.0f img/s" % (split, 32 // split, " ".join("%.4f" % (x * 1e3) for x in r), 32 / min(r)))
ref = dout.clone()
one()
torch.cuda.synchronize()
a = dout.clone()
multi(2)()
torch.cuda.synchronize()
print("max |one - two-stream| over the batch: %.3e" % float((a - dout).abs().max()))

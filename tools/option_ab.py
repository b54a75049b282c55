#!/usr/bin/env python3
"""Same-box A/B of one engine option inside the N = 32 bf16 forward: alternating passes, whole-forward time and per-layer medians.
usage: option_ab.py <option> <value A> <value B> [layer name substrings to print ...]     e.g.  option_ab.py ds_order 0 1 conv8_1 conv9_1 conv10_1"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
show = sys.argv[4:] or ["conv8_1", "conv9_1", "conv10_1"]
nb = int(os.environ.get("AB_BATCH", "32"))
style = os.environ.get("AB_WEIGHTS", "torch")
sd = workloads.random_state_dict(0, style)
L, ab, m = workloads.random_batch(nb, 256, seed=0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
dout = torch.empty((nb, 2, 256, 256), dtype=torch.float32, device=dev)
outs = {}
best = {va: {}, vb: {}}
wholes = {va: [], vb: []}
for rep in range(int(os.environ.get("AB_PASSES", "3"))):
    for v in (va, vb):
        engine.set_option(opt, v)
        e = engine.HipColorizer(256, 256, max_batch=nb, precision="bf16")
        e.load_state_dict(sd)
        for _ in range(5):
            e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=True)
        t0 = time.perf_counter()
        for _ in range(30):
            e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
        e.sync()
        whole = (time.perf_counter() - t0) / 30 * 1e3
        e.set_profiling(True)
        for _ in range(20):
            e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=True)
        _, med, _ = e.layer_times_stats()
        tab = {r["name"]: (float(med[r["index"]]), r["kernel"]) for r in e.layer_table()}
        outs[v] = dout.cpu().numpy().copy()
        wholes[v].append(whole)
        for k in tab:
            best[v][k] = min(best[v].get(k, 1e9), tab[k][0])
        print("%s=%d pass %d: forward %.4f ms  %s" % (opt, v, rep, whole, "  ".join("%s %.4f" % (k, tab[k][0]) for k in tab if any(s in k for s in show))), flush=True)
        if rep == 0:
            print("   kernels: " + "; ".join("%s = %s" % (k, tab[k][1]) for k in tab if any(s in k for s in show)), flush=True)
        e.close()
import numpy as np  # noqa: E402
print("max |out(%s=%d) - out(%s=%d)| = %.3e   (sum of squares %.6e vs %.6e)" % (opt, va, opt, vb, float(np.abs(outs[va] - outs[vb]).max()),
                                                                                  float((outs[va].astype(np.float64) ** 2).sum()), float((outs[vb].astype(np.float64) ** 2).sum())))
print("forward ms: %s=%d %s | %s=%d %s" % (opt, va, ["%.4f" % x for x in wholes[va]], opt, vb, ["%.4f" % x for x in wholes[vb]]))
for k in best[va]:
    a_, b_ = best[va][k], best[vb].get(k, 0.0)
    if a_ > 0.003:
        print("%-16s %.4f -> %.4f  %+5.1f%%" % (k, a_, b_, 100 * (b_ - a_) / a_))

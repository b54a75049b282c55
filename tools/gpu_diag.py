#!/usr/bin/env python3
"""Bring-up diagnostics for a gpurun call: per-layer error table (fp32 + bf16) against the float64
oracle on the 64x64 golden case, and a per-layer time table at the bench workload.

    python tools/gpu_diag.py [--skip-bench]
"""
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def layer_errors():
    import torch
    from interactive_deep_colorization_amd import engine
    from oracle import siggraph_torch, weights
    g = dict(np.load(os.path.join(REPO, "tests", "golden", "net64_he_s0_mc05.npz")))
    sd = weights.make_state_dict(int(g["weight_seed"]), str(g["weight_style"]))
    L, ab, m, mc = g["L_mc"], g["ab"], g["mask"], float(g["maskcent"])
    _, _, acts = siggraph_torch.forward(sd, L, ab, m, mc, return_acts=True, dtype=torch.float64)
    for prec in ("fp32", "bf16"):
        e = engine.HipColorizer(64, 64, max_batch=2, precision=prec)
        e.load_state_dict(sd)
        out = e.forward(L, ab, m, mc)
        print("== %s: out_ab max-abs err vs reference golden %.4e   vs f64 %.4e  (range %.1f..%.1f)" % (
            prec, np.abs(out - g["out_ab"]).max(), np.abs(out - g["out_ab_f64"]).max(), out.min(), out.max()))
        for row in e.layer_table():
            k = row["name"]
            if k not in acts:
                continue
            got = e.activation(k, 2)
            ref = acts[k]
            err = np.abs(got - ref)
            print("   %-14s max|ref| %8.3f  max err %.3e  mean err %.3e  %s" % (
                k, np.abs(ref).max(), err.max(), err.mean(),
                "" if err.max() < 0.05 * (1 + np.abs(ref).max()) else "  <-- BAD"))
        e.close()


def layer_times(precision="bf16", nb=32):
    import torch
    from interactive_deep_colorization_amd import engine, workloads
    from oracle import weights
    sd = weights.make_state_dict(0, "he")
    e = engine.HipColorizer(256, 256, max_batch=nb, precision=precision)
    t0 = time.time(); e.load_state_dict(sd); print("pack+upload weights: %.2f s" % (time.time() - t0))
    L, ab, m = workloads.random_batch(nb, 256, seed=0)
    dev = torch.device("cuda", 0)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
    dout = torch.empty((nb, 2, 256, 256), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=True)
    e.set_profiling(True)
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    ms = e.layer_times_ms()
    print("== %s N=%d: %.3f ms/forward  -> %.1f img/s ; sum of layers %.3f ms" % (precision, nb, dt * 1e3, nb / dt, ms.sum()))
    peak = 2500.0 if precision == "bf16" else 157.3
    tot_f = 0
    for row in e.layer_table():
        t = float(ms[row["index"]])
        fl = row["flops"] * nb
        tot_f += fl
        tf = fl / (t * 1e-3) / 1e12 if t > 0 and fl > 0 else 0.0
        gbs = row["min_bytes"] * nb / (t * 1e-3) / 1e9 if t > 0 else 0.0
        print("   %-14s %8.4f ms  %8.1f TFLOP/s (%5.1f%% of peak)  %8.1f GB/s algorithmic" % (row["name"], t, tf, 100 * tf / peak, gbs))
    print("   conv stack: %.1f TFLOP/s overall = %.1f%% of %.0f" % (tot_f / (ms.sum() * 1e-3) / 1e12, 100 * tot_f / (ms.sum() * 1e-3) / 1e12 / peak, peak))
    e.close()


if __name__ == "__main__":
    for fn, args in ((layer_errors, ()), (layer_times, ("bf16", 32)), (layer_times, ("fp32", 1)), (layer_times, ("bf16", 1))):
        if "--skip-bench" in sys.argv and fn is layer_times:
            continue
        try:
            fn(*args)
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()

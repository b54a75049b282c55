#!/usr/bin/env python3
"""Sweep the small-tile / split-K tuning knobs of the N=1 click path on the GPU box (one gpurun call):

    python tools/click_sweep.py            # runs every combination below in a child process, prints a table
    python tools/click_sweep.py --child    # one measurement under the current environment (JSON line)

Per combination: device-resident p50 of the whole forward and the per-layer times (event pair per launch, untimed
pass), bf16 and fp32.  The winner per layer shape is what set_geometry()'s defaults encode."""
import itertools
import json
import os
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child():
    import numpy as np
    import torch
    from interactive_deep_colorization_amd import engine, workloads
    sd = workloads.random_state_dict(0, "he")
    L = workloads.random_batch(1, 256, seed=7)[0].astype(np.float32)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    dev = torch.device("cuda", 0)
    out = {}
    for prec in os.environ.get("SWEEP_PREC", "bf16,fp32").split(","):
        e = engine.HipColorizer(256, 256, max_batch=1, precision=prec)
        e.load_state_dict(sd)
        dL, dab, dm = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (L, hab[None], hm[None]))
        dout = torch.empty((1, 2, 256, 256), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        for _ in range(15):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
            ts.append(time.perf_counter() - t0)
        e.set_profiling(True)
        for _ in range(20):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ms = e.layer_times_ms()
        rows = {r["name"]: round(float(ms[r["index"]]) * 1e3, 1) for r in e.layer_table() if ms[r["index"]] > 0.003}
        out[prec] = {"p50_us": round(statistics.median(ts) * 1e6, 1), "sum_layers_us": round(sum(rows.values()), 1), "layers_us": rows}
        e.close()
    print(json.dumps(out))


def main():
    combos = []
    for rep in range(2):                                            # interleaved repeats: boxes and clock states drift
        combos.append(("default #%d" % rep, {}))
        combos.append(("v2 half tiles off #%d" % rep, {"IDC_V2_HALF_TILES": "0"}))
        combos.append(("click off #%d" % rep, {"IDC_CLICK": "0"}))
        combos.append(("click off, half tiles off #%d" % rep, {"IDC_CLICK": "0", "IDC_V2_HALF_TILES": "0"}))
        combos.append(("click goal 512 #%d" % rep, {"IDC_CLICK_GOAL": "512"}))
    results = []
    for name, env in combos:
        e = dict(os.environ); e.update(env)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            results.append((name, env, json.loads(line)))
        except Exception as ex:
            results.append((name, env, {"error": repr(ex)[:200]}))
    print("# whole-forward p50 (us) and sum of per-layer times")
    for name, env, r in results:
        if "error" in r:
            print("%-28s ERROR %s" % (name, r["error"])); continue
        print("%-28s " % name + "  ".join("%s p50 %7.1f sum %7.1f" % (p, r[p]["p50_us"], r[p]["sum_layers_us"]) for p in r))
    for prec in ("bf16", "fp32"):
        ok = [(n, r[prec]["layers_us"]) for n, _, r in results if prec in r]
        if not ok:
            continue
        layers = list(ok[0][1].keys())
        print("# %s per layer (us): best combination" % prec)
        best_total = 0.0
        for l in layers:
            vals = [(v.get(l, 1e9), n) for n, v in ok]
            b = min(vals)
            best_total += b[0]
            print("%-14s default %7.1f  best %7.1f  (%s)" % (l, ok[0][1].get(l, 0), b[0], b[1]))
        print("# %s sum of per-layer bests: %.1f us" % (prec, best_total))
    print(json.dumps([{"name": n, "env": e, "result": r} for n, e, r in results]))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()

#!/usr/bin/env python3
"""The N=1 click path (BASELINE.json configs[1]: one 256x256 image, 5 hint points) for a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --stats -d <dir> -o x -- python tools/click_trace.py bf16 [--graph 0|1]
    python tools/click_trace.py --gaps <results.db> [--forwards 60]

First form: 20 warm-up + 60 device-resident forwards of one precision (nothing else runs in the process).
Second form: reads the rocpd database of such a run and prints, per forward, the launch count, the sum of kernel
durations, the wall span first-start .. last-end and the inter-kernel gaps -- i.e. what a hipGraph / fewer launches
could and could not remove.
"""
import argparse
import collections
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(prec, graph, n_fw=60):
    import numpy as np
    import torch
    from interactive_deep_colorization_amd import engine, workloads
    if graph is not None:
        engine.set_option("hip_graph", int(graph))
    sd = workloads.random_state_dict(0, "he")
    L = workloads.random_batch(1, 256, seed=7)[0].astype(np.float32)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    e = engine.HipColorizer(256, 256, max_batch=1, precision=prec)
    e.load_state_dict(sd)
    dev = torch.device("cuda", 0)
    dL, dab, dm = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (L, hab[None], hm[None]))
    dout = torch.empty((1, 2, 256, 256), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    import time
    for _ in range(20):
        e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
    ts = []
    for _ in range(n_fw):
        t0 = time.perf_counter()
        e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"precision": prec, "graph": graph, "host_p50_ms": round(float(np.median(ts)) * 1e3, 4)}))
    e.close()


def gaps(db, n_fw):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    names = [r[0] for r in rows]
    # a forward = the span between two consecutive launches of the first conv kernel of the graph
    first = None
    for nm in names:
        if "conv" in nm:
            first = nm
            break
    starts = [i for i, nm in enumerate(names) if nm == first]
    # the first kernel name may occur several times per forward (same template, different layers): find the period
    per = None
    for p in range(1, 400):
        if len(names) > 3 * p and names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p]:
            per = p
            break
    if per is None:
        print("no repeating forward found (%d dispatches)" % len(rows))
        return
    fw = [rows[len(rows) - (k + 1) * per: len(rows) - k * per] for k in range(min(n_fw, len(rows) // per - 1))]
    ksum = [sum(r[2] - r[1] for r in f) / 1e3 for f in fw]
    span = [(f[-1][2] - f[0][1]) / 1e3 for f in fw]
    gap = [[(f[i + 1][1] - f[i][2]) / 1e3 for i in range(len(f) - 1)] for f in fw]
    med = lambda v: sorted(v)[len(v) // 2]
    print("# %d forwards of %d launches each" % (len(fw), per))
    print("launches per forward        %d" % per)
    print("sum of kernel durations     p50 %.1f us" % med(ksum))
    print("span first start..last end  p50 %.1f us" % med(span))
    print("sum of inter-kernel gaps    p50 %.1f us  (mean gap %.2f us, max gap p50 %.2f us)" %
          (med([sum(g) for g in gap]), sum(sum(g) for g in gap) / max(1, sum(len(g) for g in gap)), med([max(g) for g in gap])))
    by = collections.OrderedDict()
    f = fw[len(fw) // 2]
    print("# one forward, launch by launch (us): start offset, duration, gap before")
    t0 = f[0][1]
    for i, r in enumerate(f):
        nm = r[0].replace("void ", "").replace("idc::", "")
        print("%3d %9.2f %8.2f %6.2f  %s" % (i, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - f[i - 1][2]) / 1e3 if i else 0.0, nm[:70]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("precision", nargs="?", default="bf16")
    ap.add_argument("--graph", type=int, default=None)
    ap.add_argument("--gaps", default=None)
    ap.add_argument("--forwards", type=int, default=60)
    a = ap.parse_args()
    if a.gaps:
        gaps(a.gaps, a.forwards)
    else:
        run(a.precision, a.graph, a.forwards)

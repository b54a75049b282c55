#!/usr/bin/env python3
"""Same-process A/B of an idc_set_option switch on the N = 32 256x256 forward of one precision: whole-forward ms and the per-layer times with the switch at each
value.  usage: split_ds_ab.py [precision=fp16x3] [option=split_ds_fuse] [values=1,0] [batch=32]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch                                                           # noqa: E402
from interactive_deep_colorization_amd import engine, workloads        # noqa: E402

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
OPT = sys.argv[2] if len(sys.argv) > 2 else "split_ds_fuse"
VALS = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,0").split(",")]
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 32
sd = workloads.random_state_dict(0, "torch")
L, ab, m = workloads.random_batch(NB, 256, seed=0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
dout = torch.empty((NB, 2, 256, 256), dtype=torch.float32, device=dev)
res, outs = {}, {}
for rep in range(2):
    for v in VALS:
        engine.set_option(OPT, v)
        e = engine.HipColorizer(256, 256, max_batch=NB, precision=PREC)
        e.load_state_dict(sd)
        for _ in range(5):
            e.forward_device(NB, dL, dab, dm, dout, 0.5, sync=True)
        t0 = time.perf_counter()
        for _ in range(20):
            e.forward_device(NB, dL, dab, dm, dout, 0.5, sync=False)
        e.sync()
        whole = (time.perf_counter() - t0) / 20 * 1e3
        outs[v] = dout.cpu().numpy().copy()
        e.set_profiling(True)
        for _ in range(5):
            e.forward_device(NB, dL, dab, dm, dout, 0.5, sync=False)
        e.sync()
        ms = e.layer_times_ms()
        rows = {r["name"]: [round(float(ms[r["index"]]), 4), r["kernel"]] for r in e.layer_table() if ms[r["index"]] > 0.01}
        res.setdefault(v, []).append(dict(ms_per_forward=round(whole, 4), img_s=round(NB / whole * 1e3, 1), layers=rows))
        e.close()
import numpy as np                                                     # noqa: E402
diff = float(np.abs(outs[VALS[0]] - outs[VALS[-1]]).max())
print(json.dumps({"precision": PREC, "option": OPT, "batch": NB, "max_abs_diff_between_values": diff, "runs": {str(k): v for k, v in res.items()}}))

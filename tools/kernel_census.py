#!/usr/bin/env python3
"""Which kernel each layer launches, per configuration, with the DEFAULT options (round 6: decides what the default library has to contain).
One line per (configuration, kernel): the layers it served."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from interactive_deep_colorization_amd import engine, workloads  # noqa: E402
from oracle import weights  # noqa: E402

CONFIGS = [  # (label, H, N, precision, kwargs, max_batch)
    ("bf16 N=32 256", 256, 32, "bf16", {}, 32), ("bf16 N=1 256 (click)", 256, 1, "bf16", {}, 1), ("bf16 N=2 256", 256, 2, "bf16", {}, 2),
    ("bf16 N=4 256", 256, 4, "bf16", {}, 4), ("bf16 N=8 256", 256, 8, "bf16", {}, 8), ("bf16 N=1 64", 64, 1, "bf16", {}, 1), ("bf16 N=3 40x72", (40, 72), 3, "bf16", {}, 3),
    ("fp32 N=1 256 (click)", 256, 1, "fp32", {}, 1), ("fp32 N=8 256", 256, 8, "fp32", {}, 8), ("fp32 N=1 64", 64, 1, "fp32", {}, 1),
    ("bf16x3 N=8 256", 256, 8, "bf16x3", {}, 8), ("bf16x6 N=1 256", 256, 1, "bf16x6", {}, 1), ("fp16x3 N=8 256", 256, 8, "fp16x3", {}, 8),
    ("fp16x3 N=32 256", 256, 32, "fp16x3", {}, 32), ("fp16 N=32 256", 256, 32, "fp16", {}, 32), ("fp16 N=1 256", 256, 1, "fp16", {}, 1),
    ("bf16 N=8 512 global", 512, 8, "bf16", {"global_hints": True}, 8), ("fp32 N=1 512 global", 512, 1, "fp32", {"global_hints": True}, 1),
    ("bf16 N=1 256 global", 256, 1, "bf16", {"global_hints": True}, 1),
    ("bf16 N=1 256 dist", 256, 1, "bf16", {"dist": True}, 1), ("fp32 N=1 256 dist", 256, 1, "fp32", {"dist": True}, 1), ("bf16 N=8 256 dist", 256, 8, "bf16", {"dist": True}, 8),
    ("bf16 N=1 256 dist313", 256, 1, "bf16", {"dist313": True}, 1), ("fp32 N=1 256 dist313", 256, 1, "fp32", {"dist313": True}, 1),
    ("bf16 N=1 256 global+dist313", 256, 1, "bf16", {"global_hints": True, "dist313": True}, 1),
]
total = collections.Counter()
for label, hw, n, prec, kw, mb in CONFIGS:
    H, W = (hw, hw) if isinstance(hw, int) else hw
    try:
        sd = weights.make_state_dict(1, "torch", include_class=bool(kw.get("dist")))
        if kw.get("global_hints"):
            sd = weights.add_global_branch(sd, 5)
        if kw.get("dist313"):
            sd = weights.add_pred313_head(sd, 7)
        e = engine.HipColorizer(H, W, max_batch=mb, precision=prec, **kw)
        e.load_state_dict(sd)
        L, ab, m = workloads.random_batch(n, max(H, W), seed=1)
        L, ab, m = L[:, :, :H, :W], ab[:, :, :H, :W], m[:, :, :H, :W]
        if kw.get("global_hints"):
            glob, sat = workloads.global_hint_config5(n, seed=2)
            e.set_global_hints(glob, sat)
        L, ab, m = (np.ascontiguousarray(x) for x in (L, ab, m))
        if kw.get("dist313"):
            e.forward_dist313(L, ab, m, 0.0)
        elif kw.get("dist"):
            e.forward_dist(L, ab, m, 0.0)
        else:
            e.forward(L, ab, m, 0.0)
        by = collections.defaultdict(list)
        for r in e.layer_table():
            if r["launches"] > 0 or "chain" in r["kernel"]:
                by[r["kernel"]].append(r["name"])
        for k in sorted(by):
            total[k.split("+")[0]] += 1
            print("%-28s %-44s %s" % (label, k, " ".join(by[k])))
        e.close()
    except Exception as ex:
        print("%-28s FAILED: %s" % (label, str(ex)[:300]))
print("\nkernels over all configurations:")
for k, v in sorted(total.items()):
    print("  %-40s in %d configurations" % (k, v))

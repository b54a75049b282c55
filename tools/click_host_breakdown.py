#!/usr/bin/env python3
"""Where the host side of a reference-API click goes (round 5): p50 of the same one-image forward through entry points that differ by one step each.
usage: python tools/click_host_breakdown.py [fp32|bf16]"""
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from interactive_deep_colorization_amd import engine, workloads  # noqa: E402


def p50(f, n=150, warm=15):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def main():
    import torch
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    e = engine.HipColorizer(256, 256, max_batch=1, precision=prec)
    e.load_state_dict(workloads.random_state_dict(0, "torch"))
    L, ab, m = workloads.random_batch(1, 256, seed=7)
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    ab, m = hab[None].astype(np.float32), hm[None].astype(np.float32)
    dev = torch.device("cuda", 0)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
    dout = torch.empty((1, 2, 256, 256), dtype=torch.float32, device=dev)
    pab, pm = e.pinned_empty(ab.shape), e.pinned_empty(m.shape)
    pab[...] = ab; pm[...] = m
    e.set_image_l(L[0], 0)
    rows = [("device-resident forward (idc_forward_device, sync)", lambda: e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)),
            ("+ colour step + image back, hints rasterised on the device (forward_resident)", None),
            ("forward_rgb_lazy, resident L, PINNED ab/mask (H2D 768 KB + D2H 196 KB)", lambda: e.forward_rgb_lazy(None, pab, pm, 0.0)),
            ("forward_rgb_lazy, resident L, pageable ab/mask (+ staging memcpy)", lambda: e.forward_rgb_lazy(None, ab, m, 0.0)),
            ("forward_rgb_lazy, L passed, pageable (+ 256 KB)", lambda: e.forward_rgb_lazy(L, ab, m, 0.0)),
            ("forward (ab map back, 512 KB), pageable", lambda: e.forward(L, ab, m, 0.0))]
    hints = np.array([[100, 100, 106, 106, 30.0, -20.0]], np.float32)
    e.set_hints(hints, mode="ab", img=0)
    rows[1] = (rows[1][0], lambda: e.forward_resident(1, 0.0, want_ab=False, want_lab=False))
    t = time.perf_counter(); buf = np.empty_like(ab)
    for _ in range(200):
        np.copyto(buf, ab)
    print("# host memcpy of 512 KB: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
    for name, f in rows:
        print("%-90s %.4f ms" % (name, p50(f)), flush=True)
    e.close()


if __name__ == "__main__":
    main()

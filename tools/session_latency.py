#!/usr/bin/env python3
"""Click-path latency of the wrapper classes on one MI355X (p50 over 200 clicks, 256x256, batch 1):
planes-in `net_forward` vs edit-list `net_forward_hints`; the distribution model's `net_forward` and `get_ab_reccs`
with the distribution resident on the device vs materialised on the host the way the reference does it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactive_deep_colorization_amd import api, workloads          # noqa: E402


def p50(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return float(np.median(t) * 1e3)


def main():
    sd = workloads.random_state_dict(0, "he")
    rgb = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                               "mortar_pestle_256_rgb.npy"))
    hints = [(100 + 7 * i, 60 + 9 * i, 106 + 7 * i, 66 + 9 * i, 30 * i % 256, 200 - 20 * i, 40 + 15 * i) for i in range(8)]
    res = {}
    for prec in ("bf16", "fp32"):
        m = api.ColorizeImageTorch(Xd=256, maskcent=True, precision=prec)
        m.prep_net(path="", state_dict=sd); m.set_image(rgb)
        m.net_forward_hints(hints)
        ab, mask = m.input_ab.copy(), m.input_mask.copy()
        res["net_forward_planes_p50_ms_" + prec] = p50(lambda: m.net_forward(ab, mask))
        res["net_forward_hints_p50_ms_" + prec] = p50(lambda: m.net_forward_hints(hints))
        d = api.ColorizeImageTorchDist(Xd=256, maskcent=True, precision=prec)
        d.prep_net(path="", state_dict=sd); d.set_image(rgb)
        res["dist_net_forward_resident_p50_ms_" + prec] = p50(lambda: d.net_forward(ab, mask))
        res["dist_net_forward_hints_p50_ms_" + prec] = p50(lambda: d.net_forward_hints(hints))

        def eager():                                   # what the reference does on every call: copy out + x4 upsample
            d.net_forward(ab, mask)
            return d.dist_ab_full
        res["dist_net_forward_materialised_p50_ms_" + prec] = p50(eager, n=20, warm=3)
        d.net_forward(ab, mask)
        res["get_ab_reccs_device_p50_ms_" + prec] = p50(lambda: d.get_ab_reccs(120, 130, K=5, N=25000), n=100, warm=5)
        if prec == "bf16":
            from oracle import session
            pdf = d.dist_ab[:, 120, 130]
            res["get_ab_reccs_reference_sklearn_p50_ms"] = p50(
                lambda: session.get_ab_reccs_reference(pdf, d.pts_in_hull, K=5, N=25000), n=5, warm=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

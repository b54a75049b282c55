"""Same-box A/B of the bf16 click forward with the trunk as eleven launches (kwave_chain = 0), as one cooperative launch (1) and as one
plain launch (2): device-resident p50 over 300 forwards each, alternating, two passes.  python tools/chain_ab.py [precision]"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interactive_deep_colorization_amd import engine, workloads  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = workloads.random_state_dict(0, "torch")
L = workloads.random_batch(1, 256, seed=7)[0].astype(np.float32)
hab, hm = workloads.hints_config2(256, 5, 3, 0)
dev = torch.device("cuda", 0)
dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, hab[None].astype(np.float32), hm[None].astype(np.float32)))
dout = torch.empty((1, 2, 256, 256), dtype=torch.float32, device=dev)
for rep in range(2):
    for mode in (0, 2, 1):
        engine.set_option("kwave_chain", mode)
        e = engine.HipColorizer(256, 256, max_batch=1, precision=prec)
        e.load_state_dict(sd)
        for _ in range(30):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
            ts.append(time.perf_counter() - t0)
        # back-to-back (no host sync between forwards): the device-side period of a click forward
        e.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=False)
        e.sync()
        bb = (time.perf_counter() - t0) / 200
        nl = sum(r["launches"] for r in e.layer_table())
        chain = [r["kernel"] for r in e.layer_table() if "chain" in r["kernel"]][:1]
        print("kwave_chain=%d pass %d: p50 %.4f ms  p10 %.4f  back-to-back %.4f ms  launches %d %s" % (
            mode, rep, statistics.median(ts) * 1e3, sorted(ts)[30] * 1e3, bb * 1e3, nl, chain), flush=True)
        e.close()

import sys, time, statistics, numpy as np
sys.path.insert(0, '.')
import torch
from interactive_deep_colorization_amd import engine, workloads
from oracle import siggraph_torch, weights
H=256
L = workloads.random_batch(1, H, seed=7)[0].astype(np.float32)
hab, hm = workloads.hints_config2(256, 5, 3, 0)
ab = hab[None].astype(np.float32); m = hm[None].astype(np.float32)
dev = torch.device("cuda", 0)
res = {}
for style in ("he", "torch"):
    sd = weights.make_state_dict(0, style)
    ref = siggraph_torch.forward(sd, L, ab, m, 0.0, dtype=torch.float64)
    for wb in (1, 0, 1, 0):
        engine.set_option("winograd_bf16", wb)
        e = engine.HipColorizer(H, H, max_batch=1, precision="bf16")
        e.load_state_dict(sd)
        dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
        dout = torch.empty((1, 2, H, H), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        for _ in range(20): e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True); ts.append(time.perf_counter() - t0)
        out = dout.cpu().numpy()
        d = np.abs(out - ref)
        n_l = sum(r["launches"] + (1 if "splitK" in r["kernel"] else 0) for r in e.layer_table())
        print(style, "winograd_bf16", wb, "p50 %.4f ms" % (statistics.median(ts) * 1e3), "launches", n_l, "err max %.4f mean %.5f" % (d.max(), d.mean()), flush=True)
        e.close()

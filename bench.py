#!/usr/bin/env python3
"""bench.py -- 256x256 images/sec of the Local-Hints forward pass on MI355X (+ p50 click latency).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (the 29 conv/deconv layers as 25 implicit-GEMM launches: input pack, model1, the
shortcut convs and the tanh head fused into them, i.e. SIGGRAPHGenerator.forward, models/pytorch/model.py:134-175) over one batch of 32
synthetic 256x256 inputs per GPU, bf16 MFMA path -- BASELINE.json configs[2] ("Batch 32 random
256x256 L-channels with random sparse hint masks, 1x MI355X bf16"), the configuration the
images/sec target is quoted on.  Inputs and outputs are resident in HBM during the timed region.
With N ranks every rank runs its own 32 images (weak scaling, no data-path collective); the
weights are packed on rank 0 and broadcast once over RCCL before the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      : the conv stack (kernels conv_igemm_v2p / conv_ds_fused_m / conv1_block_fused_t; --precision: their split / fp16 forms) against the dense bf16 MFMA peak,
                  from per-layer HIP events recorded on the engine's stream inside the timed region;
  cpu_baseline  : the torch-CPU oracle (same ATen kernels as the reference's PyTorch backend)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only);
  fp32_contract_path : (N = 1 GPU, default run) the same batch, weights and step count on precision fp16x3 -- the fastest path inside the reference's
                  1e-3 contract (data/colorize_image.py:263) -- and `parity.max_abs_err`: the distance of the bf16 and fp16x3 outputs of images 0, 1 from the
                  CPU baseline's own fp32 outputs of those images; never `value`;
  latency       : p50 per-click recolor latency, BASELINE.json configs[1] (one 256x256 image,
                  5 hint points), fp32 and bf16: kernels only (device-resident), the blocking C-ABI call with
                  host buffers, and the whole reference-API net_forward (forward + Lab->RGB + refresh).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FLOP_PER_IMAGE_256 = 150.391e9          # 75.1955 GMAC, conv/deconv MACs x 2
PEAK_BF16_DENSE_TFLOPS = 2500.0         # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
LAYER_PASS_FORWARDS = 20                # per-launch event pass (untimed): forwards recorded, median / min per layer reported
PEAK_FP32_MFMA_TFLOPS = 157.3
PER_GPU_BATCH = 32
H = W = 256


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)      # 40 ms: the clock ramp from idle ends inside the warm-up, not inside the timed region
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-contract-leg", action="store_true",
                    help="skip `fp32_contract_path`: the same N = 32 workload on precision fp16x3 (the fastest path inside the 1e-3 contract of "
                         "data/colorize_image.py:263) and the distance of both precisions' outputs from the CPU baseline's own fp32 outputs")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "bf16x6", "fp16x3", "fp16"],
                    help="bf16x3 / bf16x6 (round 6): the fp32 contract on the bf16 matrix pipe -- fp32 operands as 2 / 3 bf16 parts, 3 / 6 bf16 MFMA products "
                         "per fp32 product, fp32 accumulation; `roofline.peak` is then the dense bf16 peak / products")
    ap.add_argument("--setup-forwards", type=int, default=0,
                    help="untimed forwards run BEFORE the W warm-up steps of the contract.  Default 0 since round 5: the state the timed region "
                         "starts from is the one the contract's own warm-up leaves (same-box A/B 0 vs 40: 8612 / 8619 vs 8798 / 8477 img/s -- inside "
                         "the run-to-run spread, profiles/r05_setup_forwards.txt); reported in the line")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU per step (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the contract's form): --batch images per GPU whatever N.  strong (SURVEY.md 8(d) config 4's second "
                         "form): --global-batch images in total, global-batch / N per GPU")
    ap.add_argument("--global-batch", type=int, default=256, help="total images per step under --scaling strong (BASELINE configs[3]: 256)")
    ap.add_argument("--cpu-worker", nargs=4, metavar=("THREADS", "T_START", "SECONDS", "STYLE"), default=None,
                    help=argparse.SUPPRESS)       # internal: one process of cpu_baseline's whole-host leg
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) pipeline leg")
    ap.add_argument("--no-peak-probe", action="store_true", help="skip the 3 s MFMA-peak probe (tools/ubench/mfma_peak)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps forwards each: `value` is the FIRST (the contract's K steps); the others "
                         "only feed `repeats` (min / median / max), the run-to-run spread on this box")
    ap.add_argument("--transport", default="torch", choices=["torch", "c_abi"],
                    help="weight broadcast at N>1: torch.distributed.broadcast (RCCL) or the library's own "
                         "idc_broadcast_weights (ncclBroadcast through the C ABI; experimental, see include/ideepcolor.h)")
    ap.add_argument("--control-flow-only", action="store_true",
                    help="no GPU, no engine, no number: every rank runs the launcher / process-group / weight-broadcast / barrier / "
                         "MAX-reduce / per-rank gather control flow of the N>1 job over gloo with a stand-in that launches nothing "
                         "(the real packed blob still travels).  Prints a line with value null and control_flow_only true.  For "
                         "tests/test_round5_cpu.py on boxes without a GPU; never a measurement.")
    ap.add_argument("--weights", default="torch", choices=["torch", "he"],
                    help="weights `value` is quoted on: 'torch' = torch-default init + randomised BN buffers, the weights SURVEY.md "
                         "8(d) config 3 names; 'he' = full-range he-style weights (harder data for a power-capped chip).  The "
                         "other style is timed too and reported beside it (N=1).")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="idc_set_option(NAME, VALUE) before the engine is created (A/B runs: --option winograd=0, --option kwave_chain=0 ...); "
                         "recorded in the line under config.options")
    ap.add_argument("--dryrun-single-gpu", action="store_true",
                    help="N>1 control flow on ONE GPU: every rank uses device 0, the process group is gloo (host "
                         "broadcast of the packed blob -> idc_set_weights_host); for exercising barriers, the MAX-reduce "
                         "and the sharded workload with the real HIP engine where only one GPU exists.  The number it "
                         "prints is NOT a scaling result (ranks share the GPU) and says so.")
    return ap.parse_args()


def seeded_weights(style="torch"):
    """Random-init weights of the reference architecture (no checkpoint ships / no network): numpy-seeded, reference
    state_dict key set.  'torch' = torch-default init + randomised BN buffers, the weights SURVEY.md
    8(d) config 3 specifies for the headline; 'he' = he-style full-range weights (the secondary figure)."""
    from interactive_deep_colorization_amd import workloads
    return workloads.random_state_dict(0, style)


def _cpu_forward(sd):
    """(callable(L, ab, mask) -> ab map, kind): the reference's own nn.Module (models/pytorch/model.py:134-175, imported untouched) when the
    reference tree is present -- the authoring container; it never travels to the GPU box -- else the oracle's restatement of it (bit-identical
    at N = 1: oracle/make_golden.py asserts it).  Checker / baseline only: nothing here is on the product path."""
    import torch
    from oracle import siggraph_torch
    ref_root = "/root/reference"
    if os.path.isdir(os.path.join(ref_root, "models", "pytorch")):
        try:
            sys.path.insert(0, ref_root)
            from models.pytorch import model as ref_model                 # noqa: E402
            net = siggraph_torch.load_into_reference_module(ref_model.SIGGRAPHGenerator(dist=False), sd)

            def run(L, ab, m):
                with torch.no_grad():
                    return net(L[0].astype(np.float64), ab[0].astype(np.float64), m[0].astype(np.float64), 0.0)[0].numpy()
            return run, "reference"
        except Exception as ex:
            print("bench.py: reference module not usable for the CPU baseline (%s): timing the port" % str(ex)[:120], file=sys.stderr)
        finally:
            if sys.path and sys.path[0] == ref_root:
                sys.path.pop(0)
    return (lambda L, ab, m: siggraph_torch.forward(sd, L, ab, m, 0.0)), "port"


def cpu_worker(threads, t_start, seconds, style):
    """One process of the whole-host leg: `threads` threads, N = 1 forwards of the bench workload from t_start (absolute) for `seconds`."""
    import torch
    from interactive_deep_colorization_amd import workloads
    torch.set_num_threads(threads)
    sd = workloads.random_state_dict(0, style)
    fwd, kind = _cpu_forward(sd)
    L, ab, m = workloads.random_batch(1, H, seed=0)
    fwd(L, ab, m)                                   # warm-up (oneDNN primitive creation)
    late = max(0.0, time.time() - t_start)
    while time.time() < t_start:
        time.sleep(0.01)
    t0, n = time.time(), 0
    while time.time() < t_start + seconds:
        fwd(L, ab, m)
        n += 1
    print(json.dumps({"images": n, "elapsed": time.time() - t0, "late_s": late, "kind": kind}))


def cpu_baseline(sd, budget_s=12.0, style="torch", keep_outputs=None):
    """The reference path on the host cores: the reference's nn.Module when its tree is present (kind "reference"), else the torch-CPU oracle
    (the same ATen/oneDNN kernels, models/pytorch/model.py:148-175 restated; kind "port"), N=1 per call as the reference runs it, fp32.
    Two figures: (1) ONE stream -- a short probe picks the thread count (oneDNN thrashes when oversubscribed: 256 threads on one 256x256
    image take 22 s), then a bounded sample of about `budget_s` seconds of forwards at that count: `value`, the latency-style baseline;
    (2) the WHOLE HOST -- P concurrent processes x T threads with P x T ~ os.cpu_count(), each pinned to its own T
    cpus, all counting inside the same `budget_s`-second window: `whole_host`, the host's throughput."""
    import torch
    from interactive_deep_colorization_amd import workloads
    try:
        ncpu = len(os.sched_getaffinity(0))          # the cpus this process may run on (a container's mask can be much smaller than os.cpu_count())
    except Exception:
        ncpu = os.cpu_count() or 1
    fwd, kind = _cpu_forward(sd)
    L, ab, m = workloads.random_batch(1, H, seed=0)
    probe = {}
    for cores in sorted(set(c for c in (8, 16, 32, 64) if c <= ncpu) or {ncpu}):
        torch.set_num_threads(cores)
        fwd(L, ab, m)                                               # warm-up
        t0 = time.perf_counter()
        fwd(L, ab, m)
        probe[cores] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    n_img = max(5, min(200, int(budget_s / probe[cores])))
    ts = []
    for i in range(n_img):
        Li, abi, mi = workloads.random_batch(1, H, seed=0, start=i)
        t0 = time.perf_counter()
        o_ = fwd(Li, abi, mi)
        ts.append(time.perf_counter() - t0)
        if keep_outputs is not None and i < 2:                        # (outside the timed interval) the reference's answer for images 0, 1 of the batch
            keep_outputs.append(np.asarray(o_, np.float32).reshape(-1, 2, H, W)[0])
    total = sum(ts)
    res = {"value": round(n_img / total, 3), "unit": "images/sec", "cores": cores, "kind": kind,
           "sample": "%d distinct 256x256 images of the bench workload, one per call (N=1, fp32, %s), %d threads (best of 8/16/32/64 probed) "
                     "on a %d-cpu host (%d usable by this process): %.1f s, p50 %.3f s per image" % (
                         n_img, "the reference's SIGGRAPHGenerator.forward on torch CPU" if kind == "reference"
                         else "torch CPU oracle = the reference's ATen kernels", cores, os.cpu_count() or ncpu, ncpu, total, statistics.median(ts))}
    try:
        res["whole_host"] = cpu_whole_host(ncpu, min(cores, 8), budget_s, style)
    except Exception as ex:                                           # a diagnostic leg: never sinks the line
        res["whole_host"] = {"value": None, "error": str(ex)[:200]}
    return res


def cpu_limits():
    """What the container may actually use of the host's cpus: the affinity mask and the cgroup CPU-time quota (v2 cpu.max / v1 cfs_quota_us).
    A box that shows 256 cpus under a 16-core quota runs P x T = 256 threads at the speed of 16 -- the whole-host figure then says so."""
    out = {"affinity_cpus": None, "cgroup_quota_cores": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["cgroup_quota_cores"] = None if q == "max" else round(float(q) / float(per), 2)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_quota_cores"] = None if q <= 0 else round(q / per, 2)
        except Exception:
            pass
    return out


def cpu_whole_host(ncpu, threads, seconds, style):
    """P = ncpu // threads processes x `threads` threads, started together, counting N = 1 forwards inside one common window."""
    import subprocess
    procs_n = ncpu // max(threads, 1)
    if procs_n < 2:
        return {"value": None, "note": "%d cpus: one %d-thread stream already is the whole host" % (ncpu, threads)}
    t_start = time.time() + 20.0 + 0.25 * procs_n                    # every worker imports torch, draws the weights, warms up before this
    procs = []
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(ncpu))
    for i in range(procs_n):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), repr(t_start), repr(float(seconds)), style]
        cpus = set(allowed[i * threads:(i + 1) * threads]) or set(allowed)

        def pin(c=cpus):
            try:
                os.sched_setaffinity(0, c)
            except Exception:
                pass
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, preexec_fn=pin))
    rows = []
    for p_ in procs:
        try:
            out, _ = p_.communicate(timeout=seconds + 180)
            rows.append(json.loads(out.decode().strip().splitlines()[-1]))
        except Exception:
            p_.kill()
    if not rows:
        return {"value": None, "note": "no worker reported"}
    imgs = sum(r["images"] for r in rows)
    return {"value": round(imgs / seconds, 2), "unit": "images/sec", "processes": procs_n, "threads_per_process": threads,
            "cores": procs_n * threads, "host_cpus": os.cpu_count(), "usable_cpus": ncpu, "container_limits": cpu_limits(), "workers_reporting": len(rows), "workers_late": sum(1 for r in rows if r["late_s"] > 0),
            "kind": rows[0]["kind"],
            "sample": "%d processes x %d threads, each pinned to its own cpus, N=1 forwards of one 256x256 image counted inside a common "
                      "%.0f s window: %d images" % (procs_n, threads, seconds, imgs)}


def measure_latency(sd, device):
    """BASELINE configs[1]: single 256x256 image, 5 hint points; p50 over 200 calls after 20 warm-ups."""
    import torch
    from interactive_deep_colorization_amd import engine, workloads
    out = {}
    rgb_free_L = workloads.random_batch(1, H, seed=7)[0]
    hab, hm = workloads.hints_config2(256, 5, 3, 0)
    L = rgb_free_L.astype(np.float32); ab = hab[None].astype(np.float32); m = hm[None].astype(np.float32)
    for prec in ("fp32", "bf16"):
        e = engine.HipColorizer(H, W, max_batch=1, precision=prec, device=device)
        e.load_state_dict(sd)
        dev = torch.device("cuda", device)
        dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (L, ab, m))
        dout = torch.empty((1, 2, H, W), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        for _ in range(20):
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
        tk = []
        for _ in range(200):
            t0 = time.perf_counter()
            e.forward_device(1, dL, dab, dm, dout, 0.0, sync=True)
            tk.append(time.perf_counter() - t0)
        th = []
        for _ in range(100):
            t0 = time.perf_counter()
            e.forward(L, ab, m, 0.0)
            th.append(time.perf_counter() - t0)
        out[prec] = {"device_resident_p50_ms": round(statistics.median(tk) * 1e3, 4),
                     "c_abi_host_call_p50_ms": round(statistics.median(th) * 1e3, 4)}
        e.close()
        # (iii) the whole reference-API call: ColorizeImageTorch.net_forward = guards + forward + Lab->RGB +
        # the rgb->Lab refresh of output_ab (colorize_image.py:249-268), colour steps on the device
        try:
            import contextlib
            import io
            from interactive_deep_colorization_amd import api
            rgb = np.load(os.path.join(REPO, "tests", "golden", "mortar_pestle_256_rgb.npy"))
            with contextlib.redirect_stdout(io.StringIO()):
                model = api.ColorizeImageTorch(Xd=256, precision=prec)
                model.prep_net(gpu_id=device, state_dict=sd)
                model.set_image(rgb)
            for _ in range(10):
                model.net_forward(hab, hm)
            ta = []
            for _ in range(100):
                t0 = time.perf_counter()
                model.net_forward(hab, hm)
                ta.append(time.perf_counter() - t0)
            out[prec]["api_net_forward_p50_ms"] = round(statistics.median(ta) * 1e3, 4)
            # ... and the call followed by a read of output_ab, as ui/gui_draw.py:280 does after every net_forward: since round 5 the
            # ab map and the refreshed Lab stay on the device until an attribute read fetches them (2.0 of the 2.2 MB a click sent back)
            tb = []
            for _ in range(100):
                t0 = time.perf_counter()
                model.net_forward(hab, hm)
                _ = model.output_ab
                tb.append(time.perf_counter() - t0)
            out[prec]["api_net_forward_then_output_ab_p50_ms"] = round(statistics.median(tb) * 1e3, 4)
            model.net.close()
        except Exception as ex:                      # the latency leg must never sink the bench line
            out[prec]["api_net_forward_p50_ms"] = None
            out[prec]["api_error"] = repr(ex)[:200]
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves, exactly the way
    the contract's launcher would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free port> bench.py <the same arguments>` -- and hand its exit status back.  Rank 0 of the child job prints
    the JSON line on the stdout this process inherited."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n_ranks, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher -- starting %d ranks: %s" % (n_ranks, n_ranks, " ".join(cmd[1:9])), file=sys.stderr)
    return subprocess.call(cmd, env=env)


class _NoLaunchEngine(object):
    """--control-flow-only: the surface ShardedColorizer and the timed loop touch, launching nothing and computing nothing
    (no GPU, no oracle).  The blob size and the RCCL probe are the real library's (host code of libideepcolor_hip.so)."""

    def __init__(self, precision, throughput_blob):
        from interactive_deep_colorization_amd import engine as _e
        self._e = _e
        self.precision = precision
        self.flags = _e._flags(throughput_blob=throughput_blob)
        self.blob = None
        self.device = 0

    def blob_bytes(self):
        return int(self._e.N.load().idc_weights_blob_bytes(int(self._e._PREC[self.precision]), self.flags))

    def comm_unique_id(self):                       # the library's own dlopen of librccl + ncclGetUniqueId: no handle needed
        import ctypes
        N = self._e.N
        buf = (ctypes.c_char * N.IDC_UNIQUE_ID_BYTES)()
        N.check(N.load().idc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def broadcast_weights(self, unique_id, rank, world, root=0):
        raise RuntimeError("control-flow-only run: no device handle exists for ncclBroadcast (reached after the librccl dlopen)")

    def set_weights_blob(self, blob):
        self.blob = (int(blob.size), int(blob[:4096].astype(np.uint64).sum()))

    def set_weights_device(self, ptr, nbytes, copy=False, keepalive=None):
        raise AssertionError("gloo groups move host bytes")

    def forward_device(self, *a, **kw):
        pass

    def sync(self):
        pass


def control_flow_only(args, rank, world, dist, sharded, engine):
    """The N>1 job's control flow without a GPU (see --control-flow-only)."""
    import torch
    tblob = args.precision == "bf16" and args.batch >= 8
    e = _NoLaunchEngine(args.precision, tblob)
    sc = sharded.ShardedColorizer(e, rank=rank, world_size=world)
    blob = None
    if rank == 0:
        from interactive_deep_colorization_amd import workloads
        blob = engine.pack_weights(workloads.random_state_dict(0, args.weights), args.precision, throughput_blob=tblob)
    sc.broadcast_weights(blob, transport=args.transport)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e.forward_device()
    e.sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank, blobs, bcast_ms = [elapsed], [e.blob], [sc.weights_broadcast_ms]
    if args.scaling == "strong" and args.global_batch % world:
        print("bench.py: --global-batch %d does not divide by %d ranks" % (args.global_batch, world), file=sys.stderr)
        sys.exit(2)
    nb_cf = args.global_batch // world if args.scaling == "strong" else args.batch
    if world > 1:
        bcast_ms = [None] * world
        dist.all_gather_object(bcast_ms, sc.weights_broadcast_ms)
        t = torch.tensor([elapsed], dtype=torch.float64)
        mine = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [float(q[0].item()) for q in parts]
        blobs = [None] * world
        dist.all_gather_object(blobs, e.blob)
        assert float(t[0].item()) >= max(per_rank) - 1e-12
    if rank == 0:
        print(json.dumps({"metric": "256x256 images/sec", "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "control_flow_only": True, "ranks_reporting": len(per_rank),
                          "scaling": args.scaling, "per_gpu_batch": nb_cf, "global_batch": nb_cf * world,
                          "weights_broadcast_ms_per_rank": [None if x is None else round(float(x), 3) for x in bcast_ms],
                          "launched_by": os.environ.get("IDC_BENCH_LAUNCHED_BY", "external launcher"),
                          "process_group_backend": dist.get_backend() if world > 1 else None,
                          "transport_requested": args.transport, "transport_used": sc.transport_used or args.transport,
                          "transport_fallback_reason": sc.transport_fallback_reason,
                          "weights_blob_bytes": e.blob_bytes(), "every_rank_holds_rank0_blob": len(set(blobs)) == 1 and blobs[0] is not None,
                          "note": "no GPU work ran: launcher, rendezvous, weight broadcast, barriers and the rank reductions only"}))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()


def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(int(args.cpu_worker[0]), float(args.cpu_worker[1]), float(args.cpu_worker[2]), args.cpu_worker[3])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["IDC_BENCH_LAUNCHED_BY"] = "bench.py self_launch"
        sys.exit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    from interactive_deep_colorization_amd import engine, sharded, workloads

    for kv in args.option:
        name, _, val = kv.partition("=")
        engine.set_option(name.strip(), int(val))
    rank, local_rank, world = sharded.init_process_group(
        backend="gloo" if (args.dryrun_single_gpu or args.control_flow_only) else None)
    if args.dryrun_single_gpu:
        local_rank = 0
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; reporting n_gpus=%d"
              % (args.gpus, world, world), file=sys.stderr)
    if args.control_flow_only:
        return control_flow_only(args, rank, world, dist, sharded, engine)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible -- the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    nb = args.batch
    if args.scaling == "strong":                      # fixed total work: --global-batch images split evenly
        if args.global_batch % world:
            print("bench.py: --global-batch %d does not divide by %d ranks" % (args.global_batch, world), file=sys.stderr)
            sys.exit(2)
        nb = args.global_batch // world

    # ---- engine + weights (rank 0 packs, RCCL broadcast of the packed blob) --------------------------
    # bf16 throughput job: the blob without the Winograd images of the batch-1 / fp32 kernels (IDC_FLAG_THROUGHPUT_BLOB) -- the
    # N = 32 bf16 handle never selects those kernels, (a bf16 blob is 68 MB since round 6: one image per layer)
    tblob = args.precision == "bf16" and nb >= 8
    e = engine.HipColorizer(H, W, max_batch=nb, precision=args.precision, device=local_rank, throughput_blob=tblob)
    sc = sharded.ShardedColorizer(e, rank=rank, world_size=world)
    sd = seeded_weights(args.weights) if rank == 0 else None
    blob = engine.pack_weights(sd, args.precision, throughput_blob=tblob) if rank == 0 else None
    sc.broadcast_weights(blob, transport=args.transport)

    # ---- synthetic inputs, resident in HBM (each rank owns its own contiguous shard of the job) -----
    Lh, abh, mh = workloads.random_batch(nb, H, seed=0, start=rank * nb)
    dL, dab, dm = (torch.from_numpy(x).to(dev) for x in (Lh, abh, mh))
    dout = torch.empty((nb, 2, H, W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(max(args.setup_forwards, 0)):     # engine setup (disclosed as `setup_forwards`), not the contract's warm-up
        e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
    e.sync()
    for _ in range(args.warmup):
        e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
    e.sync()
    # Timed region: one HIP-event pair per forward on the engine's own stream (first conv launch .. end of the last
    # kernel) gives the conv family's duration live; per-launch pairs would slow the region by ~4 % and are taken in
    # a separate, untimed pass below for the per-layer table.
    e.set_profiling("forward")

    def timed_region():
        barrier(); torch.cuda.synchronize(dev); e.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
        e.sync(); torch.cuda.synchronize(dev); barrier()
        return time.perf_counter() - t0

    elapsed = timed_region()                    # THE timed region of the contract: exactly --steps forwards
    forward_ms = float(e.layer_times_ms()[0])
    my_elapsed = elapsed
    extra = [timed_region() for _ in range(max(args.repeats, 1) - 1)]     # spread only, never `value`
    e.set_profiling(True)                       # untimed: per-launch events, LAYER_PASS_FORWARDS forwards (median per layer)
    for _ in range(LAYER_PASS_FORWARDS):
        e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
    e.sync()
    layer_min, layer_ms, layer_max = e.layer_times_stats()
    e.set_profiling(False)
    out_head = dout[:2].cpu().numpy().copy() if world == 1 else None      # images 0 and 1 of the timed workload (for `fp32_contract_path.parity`)

    per_rank = [my_elapsed]
    bcast_ms = [sc.weights_broadcast_ms]
    if world > 1:
        bcast_ms = [None] * world
        dist.all_gather_object(bcast_ms, sc.weights_broadcast_ms)
        on_cpu = dist.get_backend() == "gloo"
        t = torch.tensor([elapsed] + extra, dtype=torch.float64, device="cpu" if on_cpu else dev)
        mine = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, extra = float(t[0].item()), [float(x) for x in t[1:].tolist()]
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [float(q[0].item()) for q in parts]
    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    # ---- accounting -------------------------------------------------------------------------------------
    table = e.layer_table()
    # every conv / deconv launch of the graph: conv_igemm<..>, conv_igemm_v2<..>[+head|+shortcut], conv1_block_fused
    # (rows of layers that ran fused inside another launch say "fused into ..." and carry no FLOP of their own)
    conv_rows = [(r, float(layer_ms[r["index"]])) for r in table if r["kernel"].startswith("conv") and r["launches"] > 0]
    traffic = _pmc_traffic()
    conv_ms_layers = sum(ms for _, ms in conv_rows)                          # untimed per-launch pass
    other_ms = float(sum(layer_ms)) - conv_ms_layers                          # non-conv kernels (softmax / glob branch)
    conv_ms = forward_ms - other_ms                                          # timed region: the conv launches incl. their boundaries
    conv_flops = sum(r["flops"] for r, _ in conv_rows) * nb                 # algorithmic, per launch-set
    achieved_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # operand-split precisions: every algorithmic FLOP costs 3 (6) bf16 MFMA FLOPs, so the bound is the bf16 peak / products
    products = {"bf16x3": 3, "bf16x6": 6, "fp16x3": 3}.get(args.precision, 1)     # (fp16x3: three fp16 MFMA products, same dense rate as bf16)
    peak = (PEAK_BF16_DENSE_TFLOPS / products) if args.precision != "fp32" else PEAK_FP32_MFMA_TFLOPS
    ms_per_step = elapsed / args.steps * 1e3
    value = world * nb * args.steps / elapsed
    worst = sorted(conv_rows, key=lambda x: -x[1])[:6]
    result = {
        "metric": "256x256 images/sec",
        "value": round(value, 2),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "setup_forwards": max(args.setup_forwards, 0),
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: batch %d random 256x256 L-channels + random sparse ab "
                               "hint masks per GPU, %s MFMA conv path, Local-Hints SIGGRAPHGenerator forward "
                               "(dist=False), seeded random-init weights (%s)" % (
                                   nb, args.precision, "torch-default init + randomised BN buffers: SURVEY.md 8(d) config 3" if args.weights == "torch"
                                   else "he-style full-range"),
                   "weights": args.weights, "options": args.option,
                   "global_batch": world * nb, "per_gpu_batch": nb, "height": H, "width": W,
                   "parallelism": "independent images sharded over %d GPU(s); one RCCL weight broadcast (transport: %s%s)" % (
                       world, sc.transport_used or args.transport, ", c_abi fell back: " + sc.transport_fallback_reason if sc.transport_fallback_reason else ""),
                   "weights_blob_bytes": int(e.blob_bytes()),
                   "weights_broadcast_ms": sc.weights_broadcast_ms,
                   "weights_broadcast_ms_per_rank": [None if x is None else round(float(x), 3) for x in bcast_ms]},
        "roofline": {"bound": "mfma", "achieved": round(achieved_tflops, 2), "peak": round(peak, 2), "unit": "TFLOP/s",
                     "peak_is": ("dense bf16 MFMA peak %.0f TFLOP/s / %d bf16 products per fp32 product" % (PEAK_BF16_DENSE_TFLOPS, products)) if products > 1
                                else ("dense bf16 MFMA peak" if args.precision == "bf16" else "exact-fp32 MFMA peak"),
                     "frac": round(achieved_tflops / peak, 4), "traffic": traffic.get("conv_family_bytes_per_forward"),
                     "traffic_source": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command run by "
                                       "the builder (%s), replayed here -- not measured in this process" % traffic.get("tag", "r01j"),
                     "kernel": "conv kernel family (conv_igemm_v2p / conv_igemm_v2m / conv_igemm_v2 / conv_ds_fused_m / conv1_block_fused / conv_igemm, %s): the %d conv/deconv launches of one "
                               "forward taken together" % (args.precision, len(conv_rows)),
                     "launches_per_forward": len(conv_rows),
                     "algorithmic_flop_per_forward": conv_flops,
                     "avg_launch_us": round(conv_ms / max(len(conv_rows), 1) * 1e3, 2),
                     "traffic_per_launch": traffic.get("conv_family_bytes_per_launch"),
                     "traffic_all_kernels_per_forward": traffic.get("hbm_bytes_per_forward"),
                     "conv_ms_per_forward": round(conv_ms, 4), "other_kernels_ms_per_forward": round(other_ms, 4),
                     "conv_ms_per_forward_per_launch_pass": round(conv_ms_layers, 4),
                     "slowest_layers_ms": {r["name"]: round(ms, 4) for r, ms in worst},
                     "whole_forward_frac": round(FLOP_PER_IMAGE_256 * nb / (ms_per_step * 1e-3) / 1e12 / peak, 4)},
        "repeats": _spread([world * nb * args.steps / t_ for t_ in [elapsed] + extra], args.steps),
        "per_rank_images_per_sec": [round(nb * args.steps / t_, 2) for t_ in per_rank],
        "process_group": {"backend": (dist.get_backend() if world > 1 else None), "ranks": world,
                          "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0),
                          "launched_by": os.environ.get("IDC_BENCH_LAUNCHED_BY", "external launcher" if world > 1 else "single process")},
        "scaling_note": "no hardware scaling curve exists from the builder (1-GPU boxes only): per-N values are whatever the "
                        "driver's 8-GPU run of this command measures; the path has no data-path collective",
        "layers_ms": {r["name"]: round(float(layer_ms[r["index"]]), 4) for r in table},
        "layers_ms_min": {r["name"]: round(float(layer_min[r["index"]]), 4) for r in table if r["launches"] > 0},
        # same-shape layers that differ (conv5_1 / conv3_2 on the driver's boxes): median vs min over the
        # pass says whether a layer is slow on every forward or was slow on a few of them
        "layers_unsteady": {r["name"]: {"min": round(float(layer_min[r["index"]]), 4), "median": round(float(layer_ms[r["index"]]), 4),
                                         "max": round(float(layer_max[r["index"]]), 4)}
                            for r in table if r["launches"] > 0 and layer_min[r["index"]] > 0.02
                            and layer_max[r["index"]] > 1.10 * layer_min[r["index"]]},
        "layers_ms_note": "separate untimed pass of %d forwards with an event pair around every launch (these pairs cost ~4 %%): "
                          "`layers_ms` is the per-layer MEDIAN, `layers_ms_min` the minimum, `layers_unsteady` lists launches whose "
                          "max exceeds their min by more than 10 %%" % LAYER_PASS_FORWARDS,
    }
    if world == 1 and not args.no_end_to_end:
        result["end_to_end"] = measure_end_to_end(e, nb, args.steps, args.warmup, value)
    if world == 1:
        # the other weight style on the same engine, inputs and launches: `value` is on SURVEY.md 8d config 3's torch-default
        # init (+ randomised BN buffers); he-style full-range weights are the harder data for a power-capped chip
        other = "he" if args.weights == "torch" else "torch"
        key = "he_style_weights" if other == "he" else "torch_init_weights"
        try:
            e.load_state_dict(workloads.random_state_dict(0, other))
            for _ in range(args.warmup):
                e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e.sync()
            tt = time.perf_counter()
            for _ in range(args.steps):
                e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e.sync()
            tms = (time.perf_counter() - tt) / args.steps * 1e3
            result[key] = {
                "value": round(nb / (tms * 1e-3), 2), "unit": "images/sec", "ms_per_step": round(tms, 4),
                "whole_forward_frac": round(FLOP_PER_IMAGE_256 * nb / (tms * 1e-3) / 1e12 / peak, 4),
                "how": "the same %d steps with workloads.random_state_dict(0, '%s') instead of the '%s' weights `value` is quoted on"
                       % (args.steps, other, args.weights)}
            e.load_state_dict(sd)
        except Exception as ex:
            result[key] = {"error": str(ex)[:200]}
        # both styles under FIXED keys whatever --weights says: cross-round comparisons use the same key, never `value`
        result["he_style_weights" if args.weights == "he" else "torch_init_weights"] = {
            "value": round(value, 2), "unit": "images/sec", "ms_per_step": round(ms_per_step, 4),
            "whole_forward_frac": round(FLOP_PER_IMAGE_256 * nb / (ms_per_step * 1e-3) / 1e12 / peak, 4), "how": "= `value` (--weights %s)" % args.weights}
    if world == 1 and not args.no_peak_probe and args.precision == "bf16":
        # the SAME launches on all-zero operands (weights and inputs): no kernel branches on data, so the instruction
        # streams are identical and the time difference is the clock the power management grants -- separates the code's
        # cycle efficiency from the DVFS response to switching activity (tools/power_probe.py)
        try:
            e.load_state_dict({k: np.zeros_like(v) for k, v in sd.items()})
            for t_ in (dL, dab, dm):
                t_.zero_()
            torch.cuda.synchronize(dev)
            for _ in range(args.warmup):
                e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e.sync()
            tz = time.perf_counter()
            for _ in range(args.steps):
                e.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e.sync()
            zms = (time.perf_counter() - tz) / args.steps * 1e3
            result["roofline"]["zero_operand_probe"] = {
                "ms_per_step": round(zms, 4), "whole_forward_frac_of_nominal": round(conv_flops / (zms * 1e-3) / 1e12 / peak, 4),
                "bench_operands_ms_per_step": round(ms_per_step, 4), "dvfs_factor": round(zms / ms_per_step, 4),
                "how": "same engine, same %d launches, all-zero weights and inputs, %d steps after the timed region: what the "
                       "instruction streams deliver at the clock an idle data path is granted" % (len(conv_rows), args.steps)}
        except Exception as ex:                                   # never let a diagnostic leg take the bench line down
            result["roofline"]["zero_operand_probe"] = {"error": str(ex)[:200]}
    e.close()
    if args.dryrun_single_gpu:
        result["dryrun_single_gpu"] = ("%d ranks shared ONE GPU over a gloo group: control-flow check only, `value` is not a "
                                       "scaling measurement" % world)
    if world == 1 and not args.no_peak_probe and args.precision == "bf16":
        probe = mfma_peak_probe()
        if probe:
            r = result["roofline"]
            r["attainable_peak"] = probe
            r["frac_of_attainable"] = round(r["achieved"] / probe["tflops_random_operands"], 4)
    if world == 1 and not args.no_latency:
        result["latency"] = measure_latency(sd, local_rank)
    contract_out = None
    if world == 1 and not args.no_contract_leg and args.precision == "bf16":
        # The headline dtype (bf16, BASELINE configs[2]) cannot meet the reference's 1e-3 contract through 30 layers; the path that does at the highest
        # rate is precision fp16x3 (three fp16 MFMA products per fp32 product, fp32 accumulation).  Same workload, same weights,
        # same step count, its own engine; NEVER `value`.
        try:
            for t_, h_ in ((dL, Lh), (dab, abh), (dm, mh)):          # (the zero-operand probe above cleared the resident inputs)
                t_.copy_(torch.from_numpy(h_))
            torch.cuda.synchronize(dev)
            e3 = engine.HipColorizer(H, W, max_batch=nb, precision="fp16x3", device=local_rank)
            e3.load_state_dict(sd)
            for _ in range(args.warmup):
                e3.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e3.sync()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                e3.forward_device(nb, dL, dab, dm, dout, 0.0, sync=False)
            e3.sync()
            ms3 = (time.perf_counter() - t3) / args.steps * 1e3
            contract_out = dout[:2].cpu().numpy().copy()
            e3.close()
            result["fp32_contract_path"] = {
                "precision": "fp16x3", "value": round(nb / (ms3 * 1e-3), 2), "unit": "images/sec", "ms_per_step": round(ms3, 4),
                "whole_forward_frac_of_bf16_peak_over_3": round(FLOP_PER_IMAGE_256 * nb / (ms3 * 1e-3) / 1e12 / (PEAK_BF16_DENSE_TFLOPS / 3), 4),
                "how": "the same %d images, weights and %d steps (after %d warm-ups) on an fp16x3 engine: fp32 operands as two fp16 parts (weights "
                       "scaled per layer by a power of two), three v_mfma_f32_16x16x32_f16 products per fp32 product, fp32 accumulation / bias / BN / "
                       "tanh head; device-resident like `value`" % (nb, args.steps, args.warmup)}
        except Exception as ex:                                   # a secondary leg: never sinks the line
            result["fp32_contract_path"] = {"error": str(ex)[:200]}
    if world == 1 and not args.no_cpu_baseline:
        keep = []
        result["cpu_baseline"] = cpu_baseline(sd, style=args.weights, keep_outputs=keep)
        if out_head is not None and len(keep) >= 2 and "fp32_contract_path" in result and contract_out is not None:
            # the CPU baseline's own fp32 outputs for images 0 and 1 of the batch (the reference module where its tree exists, else the bit-identical
            # port) are the reference's answer: the distance of the GPU outputs from them IS the contract of colorize_image.py:263
            ref01 = np.stack([k_[0] if k_.ndim == 4 else k_ for k_ in keep[:2]])
            result["fp32_contract_path"]["parity"] = {
                "against": "cpu_baseline's fp32 forward (kind: %s) of images 0 and 1 of this batch" % result["cpu_baseline"]["kind"],
                "tolerance": 1e-3,
                "max_abs_err": {"bf16": round(float(np.abs(out_head - ref01).max()), 6), "fp16x3": round(float(np.abs(contract_out - ref01).max()), 8)},
                "weights": args.weights}
    elif world == 1:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()


def measure_end_to_end(e, nb, steps, warmup, device_resident_value):
    """SURVEY.md 8d config 3 "end-to-end": the same batches with HOST pointers in and out (33.5 MB in, 16.8 MB out per
    batch over PCIe) through the two-slot pipeline (idc_forward_async / idc_wait: copy-in, compute and copy-out on
    three streams), pinned host buffers (idc_alloc_host), two batches in flight.  Never `value`."""
    bufs = []
    for k in range(2):
        Lh, abh, mh = __import__("interactive_deep_colorization_amd.workloads", fromlist=["x"]).random_batch(nb, H, seed=0, start=k * nb)
        arrs = [e.pinned_empty(x.shape) for x in (Lh, abh, mh)] + [e.pinned_empty((nb, 2, H, W))]
        for dst, src in zip(arrs[:3], (Lh, abh, mh)):
            dst[...] = src
        bufs.append(arrs)

    def run(n_steps):
        for i in range(n_steps):
            k = i & 1
            if i >= 2:
                e.wait(k)
            e.forward_async(k, bufs[k][0], bufs[k][1], bufs[k][2], bufs[k][3], 0.0)
        e.wait(0); e.wait(1)

    e.set_profiling(False)
    run(max(warmup, 2))
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    v = nb * steps / dt
    res = {"value": round(v, 2), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 4),
           "frac_of_device_resident": round(v / device_resident_value, 4),
           "how": "host pointers, pinned buffers, idc_forward_async/idc_wait two-slot pipeline (H2D / compute / D2H on three "
                  "streams), %d steps of %d images; 50.3 MB over PCIe per step" % (steps, nb)}
    # where the stages of the last two batches sat on the device clock (HIP events on the three streams): says whether the
    # copies ran UNDER the other slot's kernels or the box serialised them (round-2 driver box: 0.66 of device-resident)
    try:
        last, prev = (steps - 1) & 1, steps & 1
        tl, tp = e.pipeline_times(last), e.pipeline_times(prev)
        base = float(min(tp[0], tl[0]))
        names = ("h2d_start", "h2d_end", "compute_start", "compute_end", "d2h_start", "d2h_end")
        res["stages_ms"] = {
            "batch_k_minus_1": {n_: round(float(x) - base, 3) for n_, x in zip(names, tp)},
            "batch_k": {n_: round(float(x) - base, 3) for n_, x in zip(names, tl)},
            "h2d_ms": round(float(tl[1] - tl[0]), 3), "compute_ms": round(float(tl[3] - tl[2]), 3), "d2h_ms": round(float(tl[5] - tl[4]), 3),
            "period_ms": round(float(tl[3] - tp[3]), 3),
            "h2d_of_k_under_compute_of_k_minus_1_ms": round(max(0.0, float(min(tl[1], tp[3]) - max(tl[0], tp[2]))), 3),
            "d2h_of_k_minus_1_under_compute_of_k_ms": round(max(0.0, float(min(tp[5], tl[3]) - max(tp[4], tl[2]))), 3),
            "legend": "compute_ms well above the device-resident ms_per_step would mean the copies slowed the kernels they ran "
                      "beside; period_ms ~ h2d + compute + d2h would mean the box ran the stages one after the other"}
        st = res["stages_ms"]
        serial = st["h2d_ms"] + st["compute_ms"] + st["d2h_ms"]
        dev_ms = nb / device_resident_value * 1e3
        st["diagnosis"] = ("stages serialised (period %.2f ms ~ h2d + compute + d2h %.2f ms)" % (st["period_ms"], serial)
                           if st["period_ms"] > 0.95 * serial else
                           "copies slowed the kernels (compute %.2f ms vs %.2f ms device-resident)" % (st["compute_ms"], dev_ms)
                           if st["compute_ms"] > 1.05 * dev_ms else
                           "copies ran under the other slot's kernels (period %.2f ms, compute %.2f ms, device-resident %.2f ms)"
                           % (st["period_ms"], st["compute_ms"], dev_ms))
    except Exception as ex:
        res["stages_ms"] = {"error": str(ex)[:200]}
    ref_out = np.array(bufs[(steps - 1) & 1][3], copy=True)
    # zero-copy leg: idc_forward_device on the PINNED HOST pointers themselves (mapped into the device's address space):
    # conv1 reads L / ab / mask over PCIe, the tanh head writes out_ab into host memory -- no copy engine, no blit kernel
    try:
        def run_mapped(n_steps):
            for i in range(n_steps):
                k = i & 1
                e.forward_device(nb, bufs[k][0], bufs[k][1], bufs[k][2], bufs[k][3], 0.0, sync=False)
            e.sync()
        bufs[(steps - 1) & 1][3][...] = 0
        run_mapped(max(warmup, 2))
        t0 = time.perf_counter()
        run_mapped(steps)
        dm_ = time.perf_counter() - t0
        vm = nb * steps / dm_
        res["mapped_io"] = {"value": round(vm, 2), "ms_per_step": round(dm_ / steps * 1e3, 4),
                            "frac_of_device_resident": round(vm / device_resident_value, 4),
                            "bit_identical_to_pipeline": bool(np.array_equal(ref_out, bufs[(steps - 1) & 1][3])),
                            "how": "idc_forward_device given the pinned host buffers directly (zero-copy), %d steps back to back" % steps}
        if vm > v and res["mapped_io"]["bit_identical_to_pipeline"]:
            res["best"] = "mapped_io"
            res["best_frac_of_device_resident"] = res["mapped_io"]["frac_of_device_resident"]
        else:
            res["best"] = "copy_pipeline"
            res["best_frac_of_device_resident"] = res["frac_of_device_resident"]
    except Exception as ex:
        res["mapped_io"] = {"error": str(ex)[:200]}
    # the blocking single-slot call for comparison (serial H2D -> run -> D2H through the handle's own staging)
    t0 = time.perf_counter()
    for _ in range(max(3, steps // 4)):
        e.forward(bufs[0][0], bufs[0][1], bufs[0][2], 0.0)
    res["blocking_idc_forward_images_per_sec"] = round(nb * max(3, steps // 4) / (time.perf_counter() - t0), 2)
    return res


def _spread(values, steps):
    vs = sorted(values)
    return {"n": len(vs), "steps_each": steps, "values": [round(x, 2) for x in values], "min": round(vs[0], 2),
            "median": round(statistics.median(vs), 2), "max": round(vs[-1], 2),
            "note": "`value` is values[0] (the contract's timed region); boxes of the pool differ by up to 7 % on one binary"}


def mfma_peak_probe(seconds=2.0):
    """What the matrix pipes of THIS box sustain on full-range random bf16 operands (no memory traffic at all): the
    chip clocks to its power budget, so the nominal 2.5 PFLOP/s (2.4 GHz) is not reachable on real data
    (profiles/r02_mfma_peak.txt: zeros 2484, uniform random 1813 TFLOP/s at 1.78 GHz).  Runs tools/ubench/mfma_peak
    (built by __graft_entry__.build()) for `seconds`; None when the binary is missing."""
    import subprocess
    exe = os.path.join(REPO, "tools", "ubench", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        def one(shape):
            out = subprocess.run([exe, str(seconds), "random", shape], capture_output=True, text=True, timeout=60).stdout
            tf = ghz = None
            for line in out.splitlines():
                if line.strip().startswith("mean"):
                    tf = float(line.split()[1])
                if "shader clock" in line:
                    ghz = float(line.split("shader clock")[1].split()[0])
            return tf, ghz
        tf32, ghz32 = one("32")
        tf16, ghz16 = one("16")
        if tf32 is None:
            return None
        best = max(tf32, tf16 or 0.0)
        return {"tflops_random_operands": round(best, 1), "tflops_32x32x16": round(tf32, 1),
                "tflops_16x16x32": round(tf16, 1) if tf16 else None, "shader_clock_ghz": ghz16 if (tf16 or 0) >= tf32 else ghz32,
                "seconds": seconds,
                "how": "tools/ubench/mfma_peak: 256 CUs x 8 waves x 8 independent accumulators, uniform random [-1,1) bf16 operands, "
                       "back-to-back launches, once per MFMA shape (v_mfma_f32_32x32x16_bf16: conv_ds_fused / conv1_block_fused; "
                       "v_mfma_f32_16x16x32_bf16: conv_igemm_v2m); `tflops_random_operands` is the better of the two; measured in this bench run"}
    except Exception:
        return None


def _pmc_traffic():
    """HBM bytes from the rocprofv3 PMC passes of this same command (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of
    MI355X_MICROARCH.md); empty when no summary is committed."""
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        d["conv_family_bytes_per_forward"] = d["conv_family_bytes_per_launch"] * d["conv_family_launches_per_forward"]
        return d
    except Exception:
        return {}


if __name__ == "__main__":
    main()

"""Batched path across the GPUs of one node: one process per GPU, independent images.

The reference has no multi-device code at all; the path shards trivially because
eval-mode BatchNorm uses fixed statistics (``data/colorize_image.py:232``) and no op mixes images.
So:

* images are split into contiguous shards, one per rank -- no data-path collective;
* the ONLY collective is a one-time broadcast of the packed weight blob from rank 0 (bf16: 68 MB -- one MFMA-tiled
  image per layer, read by the throughput and the batch-1 click kernels alike; fp32: 136 MB for a throughput handle --
  ``throughput_blob=True`` -- or 384 MB with the Winograd images; ``engine.blob_bytes()`` is the exact figure): RCCL over xGMI when the process group is ``nccl``, ``gloo`` in the CPU tests.
  Rank 0 packs once on the host; every other rank receives device-ready bytes straight into the
  memory its engine then adopts (no re-packing, no host copy on the receivers);
* results stay on the rank that produced them unless ``gather_to_rank0`` is asked for.

torch / torch.distributed are plumbing here (device memory for the blob, the process group);
the arithmetic is the HIP engine's.
"""
import os

import numpy as np

from .workloads import shard_bounds


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Join the job's process group (idempotent).  ``nccl`` (= RCCL on ROCm) when a GPU is visible,
    else ``gloo``.  Rendezvous comes from MASTER_ADDR/MASTER_PORT (use 127.0.0.1 on one node)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world == 1 and not dist.is_initialized():
        return rank, local_rank, world
    if backend is None:
        backend = dist.get_backend() if dist.is_initialized() else ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)          # also when the caller created the group: torch's "current device" must be ours
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class ShardedColorizer(object):
    """Per-rank engine + shard bookkeeping.

    ``engine`` is a :class:`~interactive_deep_colorization_amd.engine.HipColorizer` in production.
    The CPU (gloo) tests inject a stand-in with the same four methods (``blob_bytes``,
    ``set_weights_blob``, ``set_weights_device``, ``forward``) to exercise the sharding and the
    broadcast without a GPU; nothing in this module computes anything itself.
    """

    def __init__(self, engine, rank=None, world_size=None):
        env_rank, _, env_world = dist_env()
        self.engine = engine
        self.rank = env_rank if rank is None else int(rank)
        self.world_size = env_world if world_size is None else int(world_size)
        self.weights_broadcast_ms = None
        self.transport_used = None                 # 'torch' | 'c_abi' after broadcast_weights (c_abi falls back to torch, logged)
        self.transport_fallback_reason = None

    # ---- the one collective -------------------------------------------------------------------
    def broadcast_weights(self, packed_blob=None, src=0, transport="torch"):
        """Rank ``src`` passes the packed blob (uint8 ndarray from ``engine.pack_weights``); the
        others pass None.  After the call every rank's engine holds the weights.

        ``transport='torch'``: ``torch.distributed.broadcast`` on a uint8 tensor (RCCL when the group is ``nccl``; the
        receivers' engines adopt the device memory the broadcast landed in; ``gloo`` groups move host bytes -- the CPU
        tests and the single-GPU dry run of bench.py).  ``transport='c_abi'``: the library's own
        ``idc_broadcast_weights`` (``ncclBroadcast`` from librccl, communicator created from a unique id that travels
        through the existing process group) -- the path a non-Python host would use."""
        import time

        import torch
        import torch.distributed as dist
        nbytes = int(self.engine.blob_bytes())
        if self.world_size == 1 or not dist.is_initialized():
            if packed_blob is None:
                raise ValueError("single process: the packed blob must be given")
            self.engine.set_weights_blob(packed_blob)
            self.weights_broadcast_ms = 0.0
            return
        self.transport_used = transport
        if transport == "c_abi":
            # idc_broadcast_weights has only ever run with one rank on hardware: every way it can fail
            # without hanging -- librccl not found beside the HIP runtime, communicator init, the broadcast itself -- is agreed on
            # by ALL ranks through the existing process group and answered by the torch transport below, with the reason logged,
            # instead of sinking the job.
            if self.rank == src:
                if packed_blob is None or int(packed_blob.size) != nbytes:
                    raise ValueError("rank %d must provide a %d-byte packed blob" % (src, nbytes))
                self.engine.set_weights_blob(packed_blob)
            # pre-flight: EVERY rank probes its own librccl (dlopen + ncclGetUniqueId, result discarded) and the
            # answers are gathered BEFORE anyone enters a collective of the new communicator -- a rank whose librccl cannot be
            # opened (different LD_LIBRARY_PATH, heterogeneous nodes) would otherwise raise while the healthy ranks sit inside
            # ncclCommInitRank waiting for it.
            mine = None
            uid = None
            try:
                uid = self.engine.comm_unique_id()
            except Exception as ex:
                mine = "idc_comm_unique_id failed on rank %d: %s" % (self.rank, str(ex)[:200])
            probes = [None] * self.world_size
            dist.all_gather_object(probes, mine)
            bad = [r for r in probes if r]
            why = bad[0] if bad else None
            if why is None:
                box = [uid if self.rank == src else None]
                dist.broadcast_object_list(box, src=src)
                dist.barrier()
                t0 = time.perf_counter()
                mine = None
                try:
                    self.engine.broadcast_weights(box[0], self.rank, self.world_size, src)
                except Exception as ex:
                    mine = "idc_broadcast_weights failed on rank %d: %s" % (self.rank, str(ex)[:200])
                reasons = [None] * self.world_size
                dist.all_gather_object(reasons, mine)
                bad = [r for r in reasons if r]
                if not bad:
                    self.weights_broadcast_ms = (time.perf_counter() - t0) * 1e3
                    return
                why = bad[0]
            self.transport_used = "torch"
            self.transport_fallback_reason = why
            if self.rank == src:
                import sys
                print("sharded.broadcast_weights: transport 'c_abi' unavailable (%s) -- falling back to torch.distributed.broadcast"
                      % why, file=sys.stderr)
        on_gpu = dist.get_backend() == "nccl"
        gpu_dev = torch.device("cuda", int(getattr(self.engine, "device", 0))) if on_gpu else None
        if self.rank == src:
            if packed_blob is None or int(packed_blob.size) != nbytes:
                raise ValueError("rank %d must provide a %d-byte packed blob" % (src, nbytes))
            t = torch.from_numpy(np.ascontiguousarray(packed_blob, dtype=np.uint8))
            if on_gpu:
                t = t.to(gpu_dev, non_blocking=False)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=gpu_dev if on_gpu else "cpu")      # the ENGINE's device, explicitly
        if on_gpu:
            torch.cuda.synchronize(gpu_dev)
        t0 = time.perf_counter()
        dist.broadcast(t, src=src)
        if on_gpu:
            torch.cuda.synchronize(gpu_dev)
        self.weights_broadcast_ms = (time.perf_counter() - t0) * 1e3
        if on_gpu:
            # adopt the device memory the broadcast landed in; the tensor is kept alive by the engine
            self.engine.set_weights_device(t.data_ptr(), nbytes, copy=False, keepalive=t)
        else:
            self.engine.set_weights_blob(t.numpy())

    # ---- sharding -----------------------------------------------------------------------------
    def my_bounds(self, n_images):
        return shard_bounds(n_images, self.world_size, self.rank)

    def forward_shard(self, L_mc, ab, mask, maskcent=0.0, global_batch=True):
        """Run this rank's images.  With ``global_batch`` the arrays hold the WHOLE batch and the
        rank slices its contiguous shard; otherwise they are already the local shard.
        Returns (start, stop, ab_out[stop-start, 2, H, W])."""
        n = L_mc.shape[0]
        lo, hi = self.my_bounds(n) if global_batch else (0, n)
        if hi <= lo:
            return lo, hi, np.zeros((0, 2) + tuple(L_mc.shape[2:]), np.float32)
        out = self.engine.forward(L_mc[lo:hi], ab[lo:hi], mask[lo:hi], maskcent)
        return lo, hi, out

    def gather_to_rank0(self, n_images, lo, hi, local_out):
        """Optional: collect every shard on rank 0 (host gather; the path itself needs no exchange)."""
        import torch.distributed as dist
        if self.world_size == 1 or not dist.is_initialized():
            return local_out
        parts = [None] * self.world_size
        dist.all_gather_object(parts, (lo, hi, local_out if hi > lo else None))
        if self.rank != 0:
            return None
        full = np.zeros((n_images,) + tuple(local_out.shape[1:]), np.float32)
        for plo, phi, arr in parts:
            if phi > plo:
                full[plo:phi] = arr
        return full

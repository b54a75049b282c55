"""Worker for tests/test_sharded_gloo.py: world_size-2 run of the N>1 path on CPU (gloo).

The HIP engine cannot run here, so a stand-in engine with the same four methods is injected; it
"computes" with the oracle, which is allowed only because this is a test.  What is under test is
the product's sharding + weight-broadcast logic in interactive_deep_colorization_amd/sharded.py."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from interactive_deep_colorization_amd import engine, sharded, workloads  # noqa: E402
from oracle import siggraph_torch, weights  # noqa: E402


class StandInEngine(object):
    """Same surface as HipColorizer for what ShardedColorizer touches."""

    def __init__(self, precision, sd_for_compute, flags=0):
        self.precision = precision
        self.blob = None
        self.sd = sd_for_compute
        self.flags = flags

    def blob_bytes(self):
        return int(engine.N.load().idc_weights_blob_bytes(1 if self.precision == "bf16" else 0, self.flags))

    def comm_unique_id(self):                     # a host whose librccl cannot be opened (idc_comm_unique_id -> IDC_ERR_UNSUPPORTED)
        raise RuntimeError("librccl.so not found beside libamdhip64 (stand-in)")

    def broadcast_weights(self, *a):
        raise AssertionError("must not be reached when the unique id could not be made")

    def set_weights_blob(self, blob):
        self.blob = np.array(blob, copy=True)

    def set_weights_device(self, ptr, nbytes, copy=False, keepalive=None):
        raise AssertionError("gloo path must not hand out device pointers")

    def forward(self, L, ab, m, maskcent=0.0):
        assert self.blob is not None, "forward before the weights arrived"
        return siggraph_torch.forward(self.sd, L, ab, m, maskcent, num_threads=1)


def main():
    out_dir = sys.argv[1]
    n_images = int(sys.argv[2])
    rank, local_rank, world = sharded.init_process_group(backend="gloo")
    sd = weights.make_state_dict(1, "torch")             # every rank can draw them; only rank 0 PACKS
    mode = sys.argv[3] if len(sys.argv) > 3 else "torch"
    tb = mode == "c_abi_fallback"               # that run also uses the throughput blob (IDC_FLAG_THROUGHPUT_BLOB = 0x10)
    eng = StandInEngine("bf16", sd, flags=0x10 if tb else 0)
    sc = sharded.ShardedColorizer(eng, rank=rank, world_size=world)
    blob = engine.pack_weights(sd, "bf16", throughput_blob=tb) if rank == 0 else None
    sc.broadcast_weights(blob, transport="c_abi" if tb else "torch")
    L, ab, m = workloads.random_batch(n_images, 32, seed=4, max_points=4, max_p=2)
    lo, hi, out = sc.forward_shard(L, ab, m, 0.5)
    full = sc.gather_to_rank0(n_images, lo, hi, out)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, out=out,
             blob_sum=np.uint64(eng.blob.astype(np.uint64).sum()), blob_head=eng.blob[:64],
             bcast_ms=np.float64(sc.weights_broadcast_ms), transport=np.array(str(sc.transport_used)),
             why=np.array(str(sc.transport_fallback_reason)), blob_size=np.int64(eng.blob.size), full=full if full is not None else np.zeros(0))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The N>1 path on CPU: two gloo ranks launched the way the driver launches bench.py."""
import os
import subprocess
import sys

import numpy as np

from interactive_deep_colorization_amd import engine, workloads
from oracle import siggraph_torch, weights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, world, n_images, port, mode="torch"):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "_gloo_worker.py"), str(tmp_path), str(n_images), mode]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=REPO)
    return [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]


def test_two_ranks_shard_and_broadcast(tmp_path):
    n_images = 3                                            # ragged: shards of 2 and 1
    res = _run(tmp_path, 2, n_images, 29611)
    sd = weights.make_state_dict(1, "torch")
    blob = engine.pack_weights(sd, "bf16")
    # every rank received exactly rank 0's packed bytes
    for r in res:
        assert int(r["blob_sum"]) == int(blob.astype(np.uint64).sum())
        assert np.array_equal(r["blob_head"], blob[:64])
    assert (int(res[0]["lo"]), int(res[0]["hi"])) == (0, 2) and (int(res[1]["lo"]), int(res[1]["hi"])) == (2, 3)
    # shard results == the single-process result on the same images (no cross-image op exists)
    L, ab, m = workloads.random_batch(n_images, 32, seed=4, max_points=4, max_p=2)
    single = np.concatenate([siggraph_torch.forward(sd, L[i:i + 1], ab[i:i + 1], m[i:i + 1], 0.5, num_threads=1)
                             for i in range(n_images)])
    got = np.concatenate([res[0]["out"], res[1]["out"]])
    assert got.shape == single.shape
    assert np.abs(got - single).max() <= 2e-4               # oneDNN blocking differs with batch size
    assert np.abs(res[0]["full"] - got).max() == 0          # gather_to_rank0 reassembles in order
    assert res[1]["full"].size == 0


def test_c_abi_transport_falls_back_to_torch_with_a_reason(tmp_path):
    """`--transport c_abi` must not sink a multi-GPU job when librccl cannot be opened -- every rank
    agrees on the failure through the existing group and the torch transport carries the blob; here with the throughput
    blob (no Winograd images: about half the bytes)."""
    res = _run(tmp_path, 2, 2, 29613, mode="c_abi_fallback")
    sd = weights.make_state_dict(1, "torch")
    blob = engine.pack_weights(sd, "bf16", throughput_blob=True)
    full_blob_bytes = int(engine.N.load().idc_weights_blob_bytes(1, 0))
    assert blob.size == full_blob_bytes <= 140e6              # 68 MB (136 MB in the -DIDC_AB_PARTNERS build): since round 5 a bf16 blob carries no Winograd images, flag or not
    for r in res:
        assert str(r["transport"]) == "torch" and "librccl" in str(r["why"])
        assert int(r["blob_size"]) == blob.size
        assert int(r["blob_sum"]) == int(blob.astype(np.uint64).sum())

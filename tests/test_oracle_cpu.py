"""CPU tests: the oracle against the reference-generated golden vectors, and the host helpers."""
import hashlib

import numpy as np
import pytest

from interactive_deep_colorization_amd import workloads
from oracle import siggraph_numpy, siggraph_torch, weights

# The reference's own output moves by ~5e-4 (he-style weights, outputs spanning +-110) between
# oneDNN blockings (batch size / thread count): tests/golden/*.npz field batched_vs_single_f32.
# So a CPU with another core count reproduces the golden vectors to the fp32 summation noise
# floor, not bit for bit.
NOISE = {"he": 3e-3, "torch": 2e-4}


@pytest.mark.parametrize("name", ["net64_he_s0_mc05", "net64_torch_s1_mc0", "net32x48_he_s2"])
def test_oracle_reproduces_reference_golden(golden, make_sd, name):
    g = golden(name)
    style = str(g["weight_style"])
    sd = make_sd(int(g["weight_seed"]), style)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode()); h.update(np.ascontiguousarray(sd[k]).tobytes())
    assert h.hexdigest() == str(g["weights_sha256"]), "seeded weights are not byte-stable"
    out, _, acts = siggraph_torch.forward(sd, g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]), return_acts=True)
    assert out.shape == g["out_ab"].shape
    assert np.abs(out - g["out_ab"]).max() <= NOISE[style]
    # per-activation pins (sampled values) -- localises a drift to a layer
    for k, v in acts.items():
        pos = np.random.RandomState(12345 + int(np.prod(v.shape)) % 9973).randint(0, int(np.prod(v.shape)), 64)
        np.testing.assert_allclose(v.ravel()[pos], g["act_samples/" + k], rtol=2e-3, atol=2e-3, err_msg=k)
    # float64 restatement stored by the generator agrees with the fp32 reference to its noise floor
    assert np.abs(g["out_ab_f64"] - g["out_ab"]).max() <= NOISE[style]


def test_numpy_restatement_matches_torch_restatement(make_sd):
    """Independent float64 numpy graph vs the torch restatement in float64: pure algebra check."""
    import torch
    sd = make_sd(2, "he")
    L, ab, m = workloads.random_batch(1, 16, 24, seed=9, max_points=3, max_p=2)
    o_np, cl_np, acts_np = siggraph_numpy.forward(sd, L, ab, m, 0.5, dist=True, return_acts=True)
    o_t, cl_t, acts_t = siggraph_torch.forward(sd, L, ab, m, 0.5, dist=True, dtype=torch.float64, return_acts=True)
    assert np.abs(o_np - o_t).max() < 1e-8
    assert np.abs(cl_np - cl_t).max() < 1e-10
    for k in ("conv1_2", "conv4_3", "conv7_3", "conv8_1", "conv9_1", "conv10_2"):
        assert np.abs(acts_np[k] - acts_t[k]).max() < 1e-9, k


def test_dist_head_golden(golden, make_sd):
    g = golden("dist64_he_s0")
    sd = make_sd(int(g["weight_seed"]), str(g["weight_style"]))
    out, cl = siggraph_torch.forward(sd, g["L_mc"], g["ab"], g["mask"], float(g["maskcent"]), dist=True)
    assert cl.shape == (1, 529, 64, 64)
    np.testing.assert_allclose(cl.sum(axis=1), 1.0, atol=1e-5)
    assert np.abs(cl[:, :, ::4, ::4] - g["class_probs_lowres"]).max() < 1e-4
    assert np.array_equal(cl[:, :, 1::4, 2::4], cl[:, :, ::4, ::4])          # nearest x4
    assert np.abs(out - g["out_ab"]).max() <= NOISE["he"]


def test_param_count_matches_survey(make_sd):
    sd = make_sd(0, "he")
    # 34,187,027 trainable+BN-affine incl. model_class; here running stats count too
    conv = sum(int(np.prod(v.shape)) for k, v in sd.items() if k.endswith(".weight") or k.endswith(".bias"))
    assert conv == 34187027
    assert set(k for k in sd if k.startswith("model_out")) == {"model_out.0.weight", "model_out.0.bias"}
    assert sd["model8up.0.weight"].shape == (512, 256, 4, 4)       # ConvTranspose is (Cin, Cout, 4, 4)


def test_put_point_matches_notebook_semantics():
    ab = np.zeros((2, 256, 256)); mask = np.zeros((1, 256, 256))
    r_ab, r_m = workloads.put_point(ab, mask, [135, 160], 3, [23, -69])
    assert r_ab is ab and r_m is mask                                # in place, returns the same arrays
    assert mask.sum() == 49 and mask[0, 132:139, 157:164].all()
    assert (ab[0, 132:139, 157:164] == 23).all() and (ab[1, 132:139, 157:164] == -69).all()
    assert ab[:, 131, 160].tolist() == [0, 0]


def test_workloads_are_shard_stable():
    L, ab, m = workloads.random_batch(6, 32, seed=1)
    for world in (1, 2, 4):
        got = []
        for r in range(world):
            lo, hi = workloads.shard_bounds(6, world, r)
            got.extend(range(lo, hi))
        assert got == list(range(6))
    L2, ab2, m2 = workloads.random_batch(3, 32, seed=1)
    assert np.array_equal(L[:3], L2) and np.array_equal(ab[:3], ab2) and np.array_equal(m[:3], m2)
    assert workloads.shard_bounds(5, 8, 7) == (5, 5)                 # ragged: empty tail shard
    hab, hm = workloads.hints_config2()
    assert hm.sum() > 0 and hab.shape == (2, 256, 256)

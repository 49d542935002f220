"""The oracle's FREE-running sampler (its own splitmix64 / Box-Muller / exponential-race randomness, oracle/dfm_oracle.c) against
the REFERENCE's free runs (tests/golden/freerun_syn_24_16.npz: 512 runs of src/inference_base.py:390-468 under torch.randn /
torch.multinomial / scipy Rotation.random, tests/golden/make_golden_freerun.py).  No draw is injected on either side, so the
comparison is in distribution: two-sample Kolmogorov-Smirnov on |tr_update| and the rotation angle |rot_update|, a two-proportion
z test on P(final energy == 0) and KS on the non-degenerate outcomes.  96 oracle trajectories (CPU budget); the GPU engine is held
to the same fixtures with 2 048 trajectories per engine in tests/test_gpu_rng_stats.py.  CPU only.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

from conftest import ROOT, load_golden


def _runs(args):
    lo, hi = args
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob
    from oracle import oracle as ora
    ora.lib().ora_set_num_threads(1)
    o = ora.Oracle(pack_blob(make_random_weights(0)), make_complex(24, 16, seed=5))
    out = np.zeros((hi - lo, 8))
    for i, s in enumerate(range(lo, hi)):
        r = o.sample(num_steps=40, seed=70000 + s)
        out[i] = np.concatenate([r["tr_update"].reshape(3), r["rot_update"].reshape(3), [float(r["energy"]), float(r["num_clashes"])]])
    return out


def test_oracle_free_runs_match_the_reference_distribution():
    from scipy import stats
    g = load_golden("freerun_syn_24_16.npz")
    assert g["tr_update"].shape[0] >= 512 and int(g["num_steps"]) == 40
    n, workers = 96, min(8, os.cpu_count() or 1)
    chunk = n // workers
    with mp.get_context("spawn").Pool(workers) as pool:
        got = np.concatenate(pool.map(_runs, [(k * chunk, (k + 1) * chunk) for k in range(workers)]))
    ref = {"tr": np.linalg.norm(g["tr_update"], axis=1), "rot": np.linalg.norm(g["rot_update"], axis=1),
           "energy": g["energy"].astype(np.float64), "clashes": g["num_clashes"].astype(np.float64)}
    mine = {"tr": np.linalg.norm(got[:, 0:3], axis=1), "rot": np.linalg.norm(got[:, 3:6], axis=1), "energy": got[:, 6], "clashes": got[:, 7]}
    report = {k: stats.ks_2samp(mine[k], ref[k]) for k in ("tr", "rot", "energy", "clashes")}
    for k, ks in report.items():
        assert ks.pvalue > 1e-3, (k, ks, np.median(mine[k]), np.median(ref[k]))
    a, b = (mine["energy"] == 0).mean(), (ref["energy"] == 0).mean()
    pp = ((mine["energy"] == 0).sum() + (ref["energy"] == 0).sum()) / (mine["energy"].size + ref["energy"].size)
    z = (a - b) / max(np.sqrt(pp * (1 - pp) * (1 / mine["energy"].size + 1 / ref["energy"].size)), 1e-12)
    assert abs(z) < 4.0, (a, b, z)


def _sticky_runs(args):
    lo, hi = args
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from dfmdock_amd.metrics import compute_metrics
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_sticky_weights, pack_blob
    from oracle import oracle as ora
    ora.lib().ora_set_num_threads(1)
    cx = make_complex(24, 16, seed=5)
    o = ora.Oracle(pack_blob(make_sticky_weights()), cx)
    out = np.zeros((hi - lo, 9))
    for i, s in enumerate(range(lo, hi)):
        r = o.sample(num_steps=40, seed=81000 + s)
        l_rmsd = compute_metrics((cx["rec_pos"], r["lig_pos"]), (cx["rec_pos"], cx["lig_pos"]))["l_rmsd"]
        out[i] = np.concatenate([r["tr_update"].reshape(3), r["rot_update"].reshape(3), [float(r["energy"]), float(r["num_clashes"]), l_rmsd]])
    return out


def test_oracle_free_runs_on_the_sticky_draw_match_the_reference():
    """The same comparison on the weight draw whose free runs END IN CONTACT (weights.make_sticky_weights): the final
    energy and clash count are not degenerate here (P(energy == 0) < 0.2 on both sides), so the oracle's sampler, graph sampling and
    energy head are held to the reference's own free runs (freerun_sticky_syn_24_16.npz, 512 runs) where the outcome depends on them:
    KS on |tr_update|, rotation angle, energy, clash count and l_rmsd of the final pose."""
    from scipy import stats
    g = load_golden("freerun_sticky_syn_24_16.npz")
    assert g["energy"].shape[0] >= 512 and (g["energy"] == 0).mean() < 0.2
    n, workers = 64, min(8, os.cpu_count() or 1)
    chunk = n // workers
    with mp.get_context("spawn").Pool(workers) as pool:
        got = np.concatenate(pool.map(_sticky_runs, [(k * chunk, (k + 1) * chunk) for k in range(workers)]))
    ref = {"tr": np.linalg.norm(g["tr_update"], axis=1), "rot": np.linalg.norm(g["rot_update"], axis=1), "energy": g["energy"].astype(np.float64),
           "clashes": g["num_clashes"].astype(np.float64), "l_rmsd": g["l_rmsd"].astype(np.float64)}
    mine = {"tr": np.linalg.norm(got[:, 0:3], axis=1), "rot": np.linalg.norm(got[:, 3:6], axis=1), "energy": got[:, 6], "clashes": got[:, 7],
            "l_rmsd": got[:, 8]}
    assert (mine["energy"] == 0).mean() < 0.2
    for k in ref:
        ks = stats.ks_2samp(mine[k], ref[k])
        assert ks.pvalue > 1e-3, (k, ks, np.median(mine[k]), np.median(ref[k]))


def test_sample_many_equals_sequential_trajectories():
    """ora_sample_many (bench.py's trajectory-parallel CPU baseline: one single-threaded trajectory per OpenMP thread) is the same
    computation as ora_sample called once per seed (inference_base.py:644-657 runs the trajectories one after the other)."""
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob
    from oracle import oracle as ora
    o = ora.Oracle(pack_blob(make_random_weights(0)), make_complex(24, 16, seed=5))
    many = o.sample_many(6, num_steps=5, seed=900, n_threads=3)
    assert many["total_forwards"] == 6 * 6 and (many["forwards"] == 6).all()
    for k in range(6):
        one = o.sample(num_steps=5, seed=900 + k)
        assert np.array_equal(many["tr_update"][k], one["tr_update"][0]) and np.array_equal(many["rot_update"][k], one["rot_update"][0])
        assert many["energy"][k] == one["energy"] and many["num_clashes"][k] == one["num_clashes"]
    assert np.linalg.norm(many["tr_update"][0] - many["tr_update"][1]) > 1.0      # different seeds: different trajectories
    part = o.sample_many(4, num_steps=5, max_forwards=2, seed=900)                # the bounded sample bench.py times
    assert (part["forwards"] == 2).all()

"""Statistical gates of the NATIVE random path (SURVEY 8(d) gate 4).  Bit parity with torch.multinomial /
torch.randn / scipy Rotation.random is impossible (different generators); what must hold is equality IN DISTRIBUTION.

(a) Edge sampling (src/models/score_net_mlsb.py:85-131): per node, 40 of the N - 20 non-kNN residues by torch.multinomial without
    replacement, p ~ 1 / d^3 = successive sampling.  The engine runs an exponential race with Philox uniforms, hardware log2 and
    float32 d^3 (kernels_geom.hip: k_knn_sample).  Test: for 8 rows at N = 600 and N = 2000, the per-candidate inclusion counts
    of 1 024 / 512 engine graphs against a float64 numpy Monte-Carlo of successive sampling (16 384 draws per row): cells =
    candidates with an expected count >= 10 in the smaller sample (successes AND failures); the remaining far candidates -
    the tail where an approximate log or a float32 cube would show - are pooled, in order of distance, into groups of expected
    count >= 10.  Per cell z = (p_gpu - p_mc) / sqrt(var_cell (1/n_gpu + 1/n_mc)) with the cell's per-draw variance taken from
    the Monte-Carlo sample (a group's count per draw is not Bernoulli); sum z^2 against chi-square(cells) - conservative, the
    fixed sample size makes cells negatively correlated - gate p > 1e-3 per row and on the pooled statistic.
(b) Whole free-running trajectories against the REFERENCE's own free runs (tests/golden/make_golden_freerun.py: 512 runs of
    inference_base.Euler_Maruyama_sampler with its own torch / numpy / scipy randomness, 40 steps, on syn_24_16 and syn_64_48):
    two-sample Kolmogorov-Smirnov on |tr_update|, the rotation angle |rot_update|, the final energy and num_clashes, for the fp32
    and the 16-bit engine, 2 048 engine trajectories each; gate p > 1e-3 on every one of the 16 tests (seeds are fixed: the
    outcome is deterministic).  The oracle's free runs (its own RNG) are held to the same fixtures in tests/test_oracle_freerun.py.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, complex_for, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(blob):
    from dfmdock_amd import engine
    engine.set_device(0)
    m = engine.Model(blob)
    yield m
    m.close()


# ------------------------------------------------------------------------------------------------------------------------
def mc_successive_sampling(w, n_draws, n_pick, rng, chunk=2048):
    """Inclusion indicators [n_draws, len(w)] (uint8) of successive sampling without replacement with probabilities ~ w:
    the n_pick smallest of Exp(1) / w (exactly the distribution of torch.multinomial(replacement=False))."""
    out = np.zeros((n_draws, w.size), np.uint8)
    for lo in range(0, n_draws, chunk):
        m = min(chunk, n_draws - lo)
        keys = rng.standard_exponential((m, w.size)) / w[None]
        idx = np.argpartition(keys, n_pick - 1, axis=1)[:, :n_pick]
        np.put_along_axis(out[lo:lo + m], idx, 1, axis=1)
    return out


def inclusion_chi2(gpu_inc, mc_inc, order, floor=10.0):
    """gpu_inc [n_g, M], mc_inc [n_m, M] inclusion indicators over the same M candidates; `order` = candidates by increasing
    distance.  Returns (chi2, cells, p, worst |z|)."""
    from scipy import stats
    n_g, n_m = gpu_inc.shape[0], mc_inc.shape[0]
    n_small = min(n_g, n_m)
    p_mc = mc_inc.mean(0)
    cells, cur = [], []
    for j in order:      # near candidates stand alone; far ones are pooled, in order of distance, until the group's expected count reaches the floor
        e1, e0 = n_small * p_mc[j], n_small * (1.0 - p_mc[j])
        if e0 < floor:
            continue      # (almost) always included: no information
        if e1 >= floor and not cur:
            cells.append([j])
            continue
        cur.append(j)
        if n_small * p_mc[cur].sum() >= floor:
            cells.append(cur)
            cur = []
    if cur and cells:
        cells[-1] = cells[-1] + cur
    z2, worst = 0.0, 0.0
    for c in cells:
        g = gpu_inc[:, c].sum(1).astype(np.float64)
        m = mc_inc[:, c].sum(1).astype(np.float64)
        var = m.var(ddof=1)
        if var <= 0:
            continue
        z = (g.mean() - m.mean()) / np.sqrt(var * (1.0 / n_g + 1.0 / n_m))
        z2 += z * z
        worst = max(worst, abs(z))
    k = len(cells)
    return z2, k, float(stats.chi2.sf(z2, k)), worst


@pytest.mark.parametrize("case,B,n_seeds,rows", [
    ("c3_300_300", 256, 4, (0, 77, 150, 299, 300, 371, 512, 599)),
    ("c5_1000_1000", 32, 16, (0, 333, 700, 999, 1000, 1234, 1600, 1999)),
])
def test_edge_sampling_inclusion_chi_square(case, B, n_seeds, rows, model):
    from dfmdock_amd import engine
    from scipy import stats
    cx = complex_for(case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    N = gx.N
    poses = np.repeat(cx["lig_pos"][None], B, 0)
    edges = np.concatenate([gx.score(poses, 0.5, seed=4000 + s, mfma16=True, energy=False, return_edges=True)["edges"]
                            for s in range(n_seeds)])      # [n_g, N, 60]
    n_g = edges.shape[0]
    assert n_g == B * n_seeds and len(np.unique(edges[:, rows[0], 20:], axis=0)) > n_g // 2
    # the engine centres on the ligand CA centroid; distances do not depend on that
    ca = np.concatenate([cx["rec_pos"][:, 1], cx["lig_pos"][:, 1]]).astype(np.float64)
    rng = np.random.default_rng(2024)
    tot_z2, tot_k, report = 0.0, 0, []
    for i in rows:
        d = np.linalg.norm(ca - ca[i], axis=1)
        knn = edges[0, i, :20]
        assert (edges[:, i, :20] == knn).all()                       # the kNN slots do not depend on the seed
        assert d[knn].max() <= np.sort(d)[19] * (1 + 1e-6)               # (exactness of the kNN slots: tests/test_gpu_configs.py)
        cand = np.setdiff1d(np.arange(N), knn)
        w = 1.0 / np.maximum(d[cand], 1e-10) ** 3
        mc = mc_successive_sampling(w, 16384, 40, rng)
        pos = np.full(N, -1)
        pos[cand] = np.arange(cand.size)
        samp = pos[edges[:, i, 20:]]
        assert (samp >= 0).all()                                     # never a kNN member, never out of range
        gi = np.zeros((n_g, cand.size), np.uint8)
        np.put_along_axis(gi, samp, 1, axis=1)
        assert (gi.sum(1) == 40).all()                               # 40 distinct candidates per draw
        z2, k, p, worst = inclusion_chi2(gi, mc, np.argsort(d[cand], kind="stable"))
        report.append((i, k, round(z2, 1), p, round(worst, 2)))
        assert p > 1e-3, report
        assert k >= 40
        tot_z2 += z2
        tot_k += k
    p_all = float(stats.chi2.sf(tot_z2, tot_k))
    print(f"\n{case}: inclusion chi-square over {len(rows)} rows, {n_g} engine graphs vs 16384 Monte-Carlo draws: "
          f"sum z^2 = {tot_z2:.1f} on {tot_k} cells, p = {p_all:.3f}; per row (row, cells, z^2, p, worst |z|): {report}")
    assert p_all > 1e-3, (tot_z2, tot_k, report)
    gx.close()


# ------------------------------------------------------------------------------------------------------------------------
def _outcomes(r):
    return {"tr_norm": np.linalg.norm(r["tr_update"], axis=1), "rot_angle": np.linalg.norm(r["rot_update"], axis=1),
            "energy": r["energy"].astype(np.float64), "num_clashes": r["num_clashes"].astype(np.float64)}


@pytest.mark.parametrize("case", ["syn_24_16", "syn_64_48"])
def test_free_running_sampler_vs_reference_distribution(case, model):
    from dfmdock_amd import engine
    from scipy import stats
    path = os.path.join(GOLDEN, f"freerun_{case}.npz")
    g = load_golden(f"freerun_{case}.npz")
    ref = _outcomes(g)
    assert ref["tr_norm"].size >= 512 and int(g["num_steps"]) == 40
    cx = complex_for("fwd_" + case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    worst = 1.0
    lines = []
    for prec in ("fp32", "mfma16"):
        parts = [gx.sample(B=256, num_steps=40, seed=9000 + k, **engine.precision_kwargs(prec)) for k in range(8)]
        got = _outcomes({k: np.concatenate([p[k] for p in parts]) for k in ("tr_update", "rot_update", "energy", "num_clashes")})
        for k in ("tr_norm", "rot_angle", "energy", "num_clashes"):
            ks = stats.ks_2samp(got[k], ref[k])
            lines.append(f"{prec} {k}: KS D = {ks.statistic:.4f} p = {ks.pvalue:.3f} (engine median {np.median(got[k]):.4g}, "
                         f"reference median {np.median(ref[k]):.4g})")
            worst = min(worst, ks.pvalue)
            assert ks.pvalue > 1e-3, lines
        # the point mass at energy == 0 (no residue pair within the cut-off at the end): a two-proportion z test
        a, b = (got["energy"] == 0).mean(), (ref["energy"] == 0).mean()
        pp = ((got["energy"] == 0).sum() + (ref["energy"] == 0).sum()) / (got["energy"].size + ref["energy"].size)
        z = (a - b) / max(np.sqrt(pp * (1 - pp) * (1 / got["energy"].size + 1 / ref["energy"].size)), 1e-12)
        lines.append(f"{prec} P(energy == 0): engine {a:.3f} reference {b:.3f} z = {z:.2f}")
        assert abs(z) < 4.0, lines
    print("\n" + "\n".join(lines))
    assert worst > 1e-3
    gx.close()


# ------------------------------------------------------------------------------------------------------------------------
def argmin_rank(energy, l_rmsd, group):
    """inference() keeps the minimum-energy trajectory of a complex (src/inference_base.py:654-657).  Per group of `group` consecutive
    free runs: the rank (0 = best) of the selected trajectory's l_rmsd among the group's l_rmsd values, as a fraction of the group."""
    n = (energy.size // group) * group
    e, l = energy[:n].reshape(-1, group), l_rmsd[:n].reshape(-1, group)
    pick = e.argmin(1)
    chosen = l[np.arange(l.shape[0]), pick]
    return (l < chosen[:, None]).sum(1) / float(group)


@pytest.mark.parametrize("case", ["sticky_syn_64_48", "sticky_7CEI"])
def test_free_running_sampler_on_the_sticky_draw(case):
    """the free-run gate on a weight draw whose trajectories END IN CONTACT (weights.make_sticky_weights: shrunk
    scale heads + an attractive coordinate head), so that the final energy / clash count are NOT degenerate and the arg-min over
    energies has something to select on.  512 free runs of the REFERENCE's Euler_Maruyama_sampler (tests/golden/make_golden_freerun.py:
    scipy / torch.normal / torch.multinomial streams, nothing injected) on syn_64_48 and on 7CEI with its real ESM-2 block, against 2 048
    native (Philox) trajectories per engine: two-sample KS at p > 1e-3 on |tr_update|, the rotation angle, the final energy, the clash
    count and l_rmsd of the final pose; P(energy == 0) < 0.2 on both sides; and a two-sample test on the RANK of the arg-min
    trajectory's l_rmsd inside groups of 16 runs - what selection (a-15) delivers."""
    from dfmdock_amd import engine
    from dfmdock_amd.metrics import NativeContext, compute_metrics
    from dfmdock_amd.weights import make_sticky_weights, pack_blob
    from scipy import stats
    g = load_golden(f"freerun_{case}.npz")
    ref = _outcomes(g)
    ref["l_rmsd"] = g["l_rmsd"].astype(np.float64)
    assert ref["energy"].size >= 512 and int(g["num_steps"]) == 40
    assert (ref["energy"] == 0).mean() < 0.2 and np.unique(ref["energy"]).size > 400 and np.unique(ref["num_clashes"]).size > 10
    engine.set_device(0)
    model = engine.Model(pack_blob(make_sticky_weights()))
    cx = complex_for("7CEI" if "7CEI" in case else "fwd_syn_64_48")
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    native = NativeContext((cx["rec_pos"], cx["lig_pos"]))
    ref_rank = argmin_rank(ref["energy"], ref["l_rmsd"], 16)
    lines = []
    for prec in ("fp32", "mfma16"):
        parts = [gx.sample(B=256, num_steps=40, seed=7000 + k, **engine.precision_kwargs(prec)) for k in range(8)]
        r = {k: np.concatenate([p[k] for p in parts]) for k in ("tr_update", "rot_update", "energy", "num_clashes", "lig_pos")}
        got = _outcomes(r)
        got["l_rmsd"] = np.array([compute_metrics((cx["rec_pos"], lp), (cx["rec_pos"], cx["lig_pos"]), native)["l_rmsd"] for lp in r["lig_pos"]])
        assert (got["energy"] == 0).mean() < 0.2, (prec, (got["energy"] == 0).mean())
        for k in ("tr_norm", "rot_angle", "energy", "num_clashes", "l_rmsd"):
            ks = stats.ks_2samp(got[k], ref[k])
            lines.append(f"{prec} {k}: KS D = {ks.statistic:.4f} p = {ks.pvalue:.3f} (engine median {np.median(got[k]):.4g}, reference median {np.median(ref[k]):.4g})")
            assert ks.pvalue > 1e-3, lines
        rk = argmin_rank(got["energy"], got["l_rmsd"], 16)
        ks = stats.ks_2samp(rk, ref_rank)
        lines.append(f"{prec} rank of the arg-min trajectory's l_rmsd in groups of 16: KS D = {ks.statistic:.4f} p = {ks.pvalue:.3f} "
                     f"(engine mean rank {rk.mean():.3f} over {rk.size} groups, reference {ref_rank.mean():.3f} over {ref_rank.size})")
        assert ks.pvalue > 1e-3, lines
        lines.append(f"{prec} P(energy == 0): engine {(got['energy'] == 0).mean():.3f} reference {(ref['energy'] == 0).mean():.3f}")
    print("\n" + "\n".join(lines))
    gx.close()
    model.close()

"""Layer 0 behind the per-complex message table (DFM_F_L0_TABLE, kernels_edge.hip: k_l0_gather / k_edge_msg<1,1,1>; reference:
src/models/egnn.py:95-104 with the pose-independent embedding of score_net_mlsb.py:365-366 as node features).

  * against the direct evaluation on the same graphs the table changes nothing but the fp16 rounding of each stored message before
    the K-row sum: f / scores / energy within 2e-3 of the direct 16-bit result (measured 1.0e-3), and within SURVEY 8(d)'s 16-bit gates of the
    reference golden and of the oracle
  * the edges the edge model still evaluates are exactly the inter-chain ones plus the intra-chain ones whose bins differ between
    the pose at hand and the table's pose (counted on the host from the engine's own codes)
  * sampling: the reference's rollouts with every draw injected stay within the 16-bit rollout gate; a trajectory does not depend on
    the batch it is sampled in, nor on the run; set_pose rebuilds the table
"""
import numpy as np
import pytest

from conftest import complex_for, load_golden, pair_hparams

pytestmark = pytest.mark.gpu


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def ca_rmsd(a, b):
    return np.sqrt(((a[:, :, 1, :] - b[:, :, 1, :]) ** 2).sum(-1).mean(-1))


@pytest.fixture(scope="module")
def model(blob):
    from dfmdock_amd import engine
    engine.set_device(0)
    m = engine.Model(blob)
    yield m
    m.close()


@pytest.fixture(scope="module")
def model_pair(blob_pair):
    from dfmdock_amd import engine
    engine.set_device(0)
    m = engine.Model(blob_pair, pair_hparams())
    yield m
    m.close()


def _complex(model, case):
    from dfmdock_amd import engine
    cx = complex_for(case)
    return engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"]), cx


@pytest.mark.parametrize("case", ["fwd_syn_24_16", "fwd_syn_64_48_p1", "fwd_7CEI_p0", "fwd_7CEI_p2", "fwd_c3_300_300"])
def test_table_vs_direct_and_reference(case, model):
    g = load_golden(case + ".npz")
    gx, _ = _complex(model, case)
    kw = dict(edges=g["edges"], mfma16=True, energy=True)
    d = gx.score(g["lig_pos"], float(g["t"]), **kw)
    t = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, profile=True, **kw)
    p = gx.profile()
    assert p["l0_evals"] == 1 and p["l0_edges"] == g["edges"].size and p["l0_build_ms"] > 0
    for k in ("f", "tr_score", "rot_score"):
        assert rel_inf(t[k], d[k]) < 2e-3, (k, rel_inf(t[k], d[k]))
        assert rel_inf(t[k][0], g[k].reshape(t[k][0].shape)) < 1e-2, (k, "vs the reference")
    assert abs(float(t["energy"][0]) - float(d["energy"][0])) < 1e-3 * max(1.0, abs(float(d["energy"][0])))
    # a second call re-uses the table and gives bitwise the same result
    t2 = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, profile=True, **kw)
    assert gx.profile()["l0_build_ms"] == 0
    assert all((t2[k] == t[k]).all() for k in ("f", "tr_score", "rot_score", "energy"))
    gx.close()


def test_row_list_is_inter_chain_plus_bin_mismatches(model):
    """Engine-drawn graphs on moved ligands: the rows the edge model evaluates = inter-chain edges + intra-chain edges whose per-pose
    code differs from the code of the same pair in the stored pose (rare: last-bit effects at a bin boundary)."""
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    cx = make_complex(120, 90, seed=3)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    R, L = 120, 90
    rng = np.random.default_rng(0)
    B = 5
    poses = np.repeat(cx["lig_pos"][None], B, 0).astype(np.float32)
    for b in range(1, B):      # rigid moves: random rotation about the centroid + a shift
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        w, x, y, z = q
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)
        c = poses[b][:, 1].mean(0)
        poses[b] = (poses[b] - c) @ Rm.T + c + rng.standard_normal(3).astype(np.float32) * 4.0
    # codes of every intra-chain pair in the stored pose, through the engine itself (one evaluation per block of K neighbours would be
    # clumsy: use the debug tap of a direct evaluation on hand-made edge lists covering all pairs of a chain)
    N, K = gx.N, gx.K
    base = {}
    for start in range(0, N, K):
        e = np.zeros((1, N, K), np.int32)
        for i in range(N):
            lo, n = (0, R) if i < R else (R, L)
            e[0, i] = lo + (np.arange(start, start + K) % n)
        r = gx.score(cx["lig_pos"], 0.5, edges=e, mfma16=True, energy=False, debug=True)
        for i in range(N):
            for s in range(K):
                base[(i, int(e[0, i, s]))] = int(r["edge_codes"][0, i, s])
    d = gx.score(poses, 0.5, seed=11, mfma16=True, energy=False, debug=True)
    t = gx.score(poses, 0.5, edges=d["edges"], mfma16=True, energy=False, l0_table=True, profile=True)
    p = gx.profile()
    same = (np.arange(N)[None, :, None] < R) == (d["edges"] < R)
    inter = int((~same).sum())
    mism = sum(1 for b in range(B) for i in range(N) for s in range(K)
               if same[b, i, s] and base[(i, int(d["edges"][b, i, s]))] != int(d["edge_codes"][b, i, s]))
    assert p["l0_miss_rows"] == inter + mism, (p["l0_miss_rows"], inter, mism)
    assert inter > 0 and mism < 0.01 * same.sum()
    assert rel_inf(t["f"], d["f"]) < 2e-3 and rel_inf(t["tr_score"], d["tr_score"]) < 2e-3
    gx.close()


@pytest.mark.parametrize("case,steps", [("rollout_syn_24_16", 40), ("rollout_7CEI", 40)])
def test_rollout_with_table_vs_reference(case, steps, model):
    g = load_golden(case + ".npz")
    gx, _ = _complex(model, case)
    steps = min(steps, g["poses"].shape[0])
    inj = dict(R0=g["R0"].astype(np.float32).reshape(1, 9), tr_draw=g["tr_draw"].reshape(1, 3), edges=g["edges"][None],
               z_rot=g["z_rot"][None], z_tr=g["z_tr"][None])
    on = gx.sample(B=1, num_steps=g["z_rot"].shape[0], inject=inj, trace=True, mfma16=True, profile=True)
    p = gx.profile()
    assert p["l0_evals"] == g["z_rot"].shape[0] + 1, "dfm_sample uses the table by default"
    off = gx.sample(B=1, num_steps=g["z_rot"].shape[0], inject=inj, trace=True, mfma16=True, l0_table=False, profile=True)
    assert gx.profile()["l0_evals"] == 0
    r_on, r_off = ca_rmsd(on["trace_pose"][0], g["poses"]), ca_rmsd(off["trace_pose"][0], g["poses"])
    assert r_on[:steps].max() < 0.5, r_on
    assert r_on.max() < r_off.max() + 0.05, (r_on.max(), r_off.max())
    gx.close()


def test_sampling_is_batch_invariant_and_reproducible(model):
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    cx = make_complex(70, 50, seed=9)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    a = gx.sample(B=7, num_steps=6, seed=5, mfma16=True)
    b = gx.sample(B=7, num_steps=6, seed=5, mfma16=True)
    one = gx.sample(B=1, num_steps=6, seed=5, mfma16=True)
    assert (a["lig_pos"] == b["lig_pos"]).all() and (a["energy"] == b["energy"]).all()
    assert (a["lig_pos"][0] == one["lig_pos"][0]).all() and a["energy"][0] == one["energy"][0]
    assert np.isfinite(a["lig_pos"]).all()
    gx.close()


def test_set_pose_rebuilds_the_table(model):
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    c1, c2 = make_complex(40, 30, seed=1), make_complex(40, 30, seed=2)
    gx = engine.Complex(model, c1["rec_x"], c1["lig_x"], c1["rec_pos"], c1["lig_pos"])
    gx.score(c1["lig_pos"], 0.4, seed=1, mfma16=True, l0_table=True)
    gx.set_pose(c2["rec_pos"], c2["lig_pos"])          # another conformation of the same sequences
    r = gx.score(c2["lig_pos"], 0.4, seed=1, mfma16=True, l0_table=True, profile=True, return_edges=True)
    assert gx.profile()["l0_build_ms"] > 0
    fresh = engine.Complex(model, c1["rec_x"], c1["lig_x"], c2["rec_pos"], c2["lig_pos"])
    f = fresh.score(c2["lig_pos"], 0.4, edges=r["edges"], mfma16=True, l0_table=True)
    assert (f["f"] == r["f"]).all() and (f["tr_score"] == r["tr_score"]).all()
    fresh.close(); gx.close()


def test_pair_family_and_small_degree(model_pair, model):
    """family 1 (EGNN_Net trunk, mask_dist 20) and a complex smaller than the degree (K = N - 1 < 60)."""
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    for m, (R, L) in ((model_pair, (64, 48)), (model, (9, 7)), (model, (24, 16))):
        cx = make_complex(R, L, seed=4)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        d = gx.score(cx["lig_pos"], 0.6, seed=2, mfma16=True, return_edges=True)
        t = gx.score(cx["lig_pos"], 0.6, edges=d["edges"], mfma16=True, l0_table=True)
        for k in ("f", "tr_score", "rot_score"):
            assert rel_inf(t[k], d[k]) < 2e-3, (R, L, k, rel_inf(t[k], d[k]))
        gx.close()


def test_flag_is_refused_where_the_table_does_not_apply(model):
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    cx = make_complex(24, 16, seed=5)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    with pytest.raises(ValueError):
        gx.score(cx["lig_pos"], 0.5, f16=True, l0_table=True)         # fp32 A_i variant of the 16-bit engine (the fp32 engine has its own table)
    gx.close()


@pytest.mark.parametrize("case", ["fwd_syn_24_16", "fwd_syn_64_48_p1", "fwd_7CEI_p0", "fwd_c3_300_300"])
def test_fp32_table_vs_direct_and_reference(case, model):
    """The fp32 engine's own table (k_edge_f32m<1>, fp32 rows of 1 KiB): exact fp32 products and sums as in the direct evaluation, only the
    order of the K-row sum differs (slot order instead of tile partials): <= 2e-5 against the direct fp32 result, the reference's 1e-4 gates
    unchanged, batch invariance bitwise."""
    g = load_golden(case + ".npz")
    gx, _ = _complex(model, case)
    kw = dict(edges=g["edges"], energy=True)
    d = gx.score(g["lig_pos"], float(g["t"]), **kw)
    t = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, profile=True, **kw)
    p = gx.profile()
    assert p["l0_evals"] == 1 and p["l0_edges"] == g["edges"].size and p["l0_build_ms"] > 0
    for k in ("f", "tr_score", "rot_score"):
        assert rel_inf(t[k], d[k]) < 2e-5, (k, rel_inf(t[k], d[k]))
        assert rel_inf(t[k][0], g[k].reshape(t[k][0].shape)) < 1e-4, (k, "vs the reference")
    assert abs(float(t["energy"][0]) - float(g["energy"])) < 1e-4 * max(1.0, abs(float(g["energy"])))
    t2 = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, profile=True, **kw)
    assert gx.profile()["l0_build_ms"] == 0
    assert all((t2[k] == t[k]).all() for k in ("f", "tr_score", "rot_score", "energy"))
    # both tables of a complex live side by side
    m = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, mfma16=True, **kw)
    assert rel_inf(m["f"], d["f"]) < 1e-2
    t3 = gx.score(g["lig_pos"], float(g["t"]), l0_table=True, **kw)
    assert all((t3[k] == t[k]).all() for k in ("f", "tr_score", "rot_score", "energy"))
    gx.close()


def test_fp32_sampler_uses_the_table_and_is_batch_invariant(model):
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd import engine
    cx = make_complex(70, 50, seed=9)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    a = gx.sample(B=3, num_steps=6, seed=4, profile=True)
    p = gx.profile()
    assert p["l0_evals"] == 7 and 0 < p["l0_miss_rows"] < p["l0_edges"]
    b = gx.sample(B=1, num_steps=6, seed=4)
    np.testing.assert_array_equal(a["lig_pos"][0], b["lig_pos"][0])
    np.testing.assert_array_equal(a["energy"][0], b["energy"][0])
    off = gx.sample(B=3, num_steps=6, seed=4, l0_table=False, profile=True)
    assert gx.profile()["l0_evals"] == 0
    assert np.abs(off["lig_pos"] - a["lig_pos"]).max() < 1e-2      # same draws, fp32 either way: only the order of layer 0's row sums differs
    gx.close()


def test_table_is_safe_for_a_pose_that_is_not_a_rigid_image(model):
    """a hit used to be decided by the bin code alone, so a DIFFERENT conformer handed to dfm_score(DFM_F_L0_TABLE) whose
    bins still matched silently got the stored pose's radial term.  A table entry now carries its squared distance and a hit needs it
    to match up to the rounding of a rigid motion: on a ligand whose residues are jittered by 0.3 A (not a rigid image) the intra-
    ligand edges become misses - evaluated by the edge model on the pose at hand - and the result equals the direct evaluation at
    the table-vs-direct tolerance; on a rigid image of the same size nothing but inter-chain edges and bin mismatches misses."""
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    cx = make_complex(120, 90, seed=3)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    R, N = 120, 210
    rng = np.random.default_rng(5)
    rigid = (cx["lig_pos"] + np.float32([3.0, -2.0, 1.5])).astype(np.float32)
    bent = (cx["lig_pos"] + 0.3 * rng.standard_normal((90, 1, 3)).astype(np.float32)).astype(np.float32)      # per-residue jitter
    res = {}
    for name, pose in (("rigid", rigid), ("bent", bent)):
        d = gx.score(pose, 0.5, seed=11, mfma16=True, energy=True, debug=True)
        t = gx.score(pose, 0.5, edges=d["edges"], mfma16=True, energy=True, l0_table=True, profile=True)
        p = gx.profile()
        same = (np.arange(N)[None, :, None] < R) == (d["edges"] < R)
        intra_lig = int((same & (np.arange(N)[None, :, None] >= R)).sum())
        res[name] = (p["l0_miss_rows"], int((~same).sum()), intra_lig)
        for k in ("f", "tr_score", "rot_score"):
            assert rel_inf(t[k], d[k]) < 2e-3, (name, k, rel_inf(t[k], d[k]))
        assert abs(float(t["energy"][0]) - float(d["energy"][0])) < 1e-3 * max(1.0, abs(float(d["energy"][0])))
    miss, inter, intra_lig = res["rigid"]
    assert inter <= miss < inter + 0.02 * intra_lig, res["rigid"]                 # inter-chain edges + a few bin mismatches
    miss, inter, intra_lig = res["bent"]
    assert miss > inter + 0.9 * (intra_lig - 90), res["bent"]                    # (all but the 90 self edges of) the ligand's own edges miss
    gx.close()


def test_table_eligibility_does_not_depend_on_the_batch(model):
    """r04 left the table path above 32 M edges per batched evaluation, so a trajectory's bits depended on B there.
    Eligibility is now a property of the complex: a batch of 40 M edges (B = 1100 x N = 600 x K = 60) runs layer 0 through the
    table and its first trajectories equal those of a small batch bit for bit."""
    gx, _ = _complex(model, "fwd_c3_300_300")
    big = gx.sample(B=1100, num_steps=2, seed=21, mfma16=True, profile=True)
    assert gx.profile()["l0_evals"] == 3
    small = gx.sample(B=4, num_steps=2, seed=21, mfma16=True)
    for k in ("lig_pos", "energy", "tr_update", "rot_update"):
        np.testing.assert_array_equal(big[k][:4], small[k], err_msg=k)
    gx.close()

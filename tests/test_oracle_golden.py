"""Pin the CPU oracle (oracle/dfm_oracle.c) to golden vectors captured from the reference.

Every check here compares the plain-C restatement with numbers produced by running the
reference's own functions (tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest

from conftest import complex_for, load_golden, pair_hparams
from oracle import oracle as ora


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ---- a-3 / a-4 -------------------------------------------------------------------------
def test_diffusion_coefficients():
    k = load_golden("scalar_kats.npz")
    g3, gso, s3, sso = ora.diffusion_coefs(k["ts"])
    np.testing.assert_allclose(g3, k["g_r3"], rtol=1e-13)
    np.testing.assert_allclose(gso, k["g_so3"], rtol=1e-13)
    np.testing.assert_allclose(s3, k["sigma_r3"], rtol=1e-13)
    np.testing.assert_allclose(sso, k["sigma_so3"], rtol=1e-13)
    # SURVEY section 8 known answers
    assert abs(g3[0] - 101.325260692) < 1e-8 and abs(gso[0] - 1.503399185) < 1e-8
    assert abs(g3[-1] - 0.339682831) < 1e-8 and abs(gso[-1] - 0.792314388) < 1e-8
    # so3 sigma raises ValueError outside [0,1] in the reference -> NaN in the oracle
    assert np.isnan(ora.diffusion_coefs([1.5])[3][0])


def test_torch_reverse_zero_noise():
    k = load_golden("scalar_kats.npz")
    g3, gso, _, _ = ora.diffusion_coefs(k["ts"])
    for i in range(len(k["ts"])):
        a = ora.torch_reverse(g3[i], k["rev_score"], k["dt"], 0.0, np.zeros(3))
        b = ora.torch_reverse(gso[i], k["rev_score"], k["dt"], 0.0, np.zeros(3))
        np.testing.assert_array_equal(a, k["rev_r3"][i, 0])   # bit-exact float32
        np.testing.assert_array_equal(b, k["rev_so3"][i, 0])


# ---- a-14 ------------------------------------------------------------------------------
def test_rotation_conversions():
    k = load_golden("scalar_kats.npz")
    m = ora.axis_angle_to_matrix(k["axis_angle"]).reshape(-1, 3, 3)
    np.testing.assert_allclose(m, k["matrices"], atol=2e-7)
    np.testing.assert_allclose(m[0, 0], [0.3065077, -0.9414502, -0.1404437], atol=1e-6)
    back = ora.matrix_to_axis_angle(k["matrices"].reshape(-1, 9))
    np.testing.assert_allclose(back, k["axis_angle_back"], atol=3e-6)
    q = ora.matrix_to_quaternion(k["matrices"].reshape(-1, 9))
    np.testing.assert_allclose(q, k["quaternions"], atol=2e-7)
    c = ora.rot_compose(k["compose_r1"], k["compose_r2"])
    np.testing.assert_allclose(c, k["compose_out"], atol=5e-6)


def test_modify_coords_and_clash_force():
    k = load_golden("scalar_kats.npz")
    out = ora.modify_coords(k["mc_x"], k["mc_rot"], k["mc_tr"])
    np.testing.assert_allclose(out, k["mc_out"], atol=1e-5)
    cf = ora.clash_force(k["cf_rec"], k["cf_lig"])
    np.testing.assert_allclose(cf, k["cf_out"], rtol=2e-4, atol=1e-6)


# ---- a-6 / a-7 / a-8 / a-10 ------------------------------------------------------------
def test_coords6d_and_bins_full_matrix():
    g = load_golden("geometry_small.npz")
    pos = g["pos_centered"]
    N = pos.shape[0]
    dist, omega, theta, phi = ora.coords6d_full(pos)
    off = ~np.eye(N, dtype=bool)
    np.testing.assert_allclose(dist, g["dist"], atol=2e-5)
    for mine, ref in ((omega, g["omega"]), (theta, g["theta"]), (phi, g["phi"])):
        assert np.isnan(ref[~off]).all() and np.isnan(mine[~off]).all()   # NaN diagonal (-> bin 0)
        d = np.abs(mine[off] - ref[off])
        d = np.minimum(d, 360.0 - d)
        assert d.max() < 2e-2, d.max()          # degrees; ill-conditioned near +-180 only
        assert np.median(d) < 1e-5
    bins = ora.bins_full(pos)
    mism = int((bins != g["bins"]).sum())
    assert mism <= 2, f"{mism} bin flips out of {bins.size}"
    assert (bins[~off][:, 1:] == 0).all()
    rel = ora.relpos_full(40, 30)
    np.testing.assert_array_equal(rel, g["relpos"])


def test_knn_matches_reference_topk():
    g = load_golden("geometry_small.npz")
    ca = g["pos_centered"][:, 1, :]
    e = ora.knn_sample(ca, seed=1)
    assert e.shape == (70, 60)
    np.testing.assert_array_equal(e[:, :20], g["knn"])      # sorted ascending, slot 0 = self
    assert (e[:, 0] == np.arange(70)).all()
    for i in range(70):                                     # sampled slots: no repeats, disjoint from kNN
        assert len(set(e[i])) == 60


def test_sampling_distribution_inverse_cubic():
    """Sampled-slot inclusion frequencies follow successive sampling with p ~ 1/d^3."""
    g = load_golden("geometry_small.npz")
    ca = g["pos_centered"][:, 1, :]
    N = ca.shape[0]
    cnt = np.zeros((N, N))
    T = 400
    for s in range(T):
        e = ora.knn_sample(ca, seed=1000 + s)
        for i in (0, 17, 55):
            cnt[i, e[i, 20:]] += 1
    rng = np.random.default_rng(0)
    for i in (0, 17, 55):
        d = np.linalg.norm(ca - ca[i], axis=1)
        knn = set(g["knn"][i].tolist())
        pool = np.array([j for j in range(N) if j not in knn])
        w = 1.0 / np.maximum(d[pool], 1e-10) ** 3
        exp = np.zeros(N)
        M = 4000
        for _ in range(M):   # Monte-Carlo expectation of inclusion under the same scheme
            keys = rng.exponential(size=pool.size) / w
            exp[pool[np.argsort(keys)[:40]]] += 1
        exp /= M
        got = cnt[i] / T
        assert np.abs(got - exp).max() < 0.09, np.abs(got - exp).max()


# ---- a-5 (+ a-9, a-11, a-12, a-13): one score evaluation -------------------------------
FWD_CASES = ["fwd_syn_9_7", "fwd_syn_24_16", "fwd_syn_64_48_p0", "fwd_syn_64_48_p1", "fwd_syn_64_48_p2",
             "fwd_7CEI_p0", "fwd_7CEI_p1", "fwd_7CEI_p2", "fwd_7CEI_p3"]


@pytest.mark.parametrize("case", FWD_CASES)
def test_score_matches_reference(case, blob):
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"])
    flips = int((r["bins"] != g["bins"]).sum())
    assert flips == 0, f"{flips} feature-bin flips"
    np.testing.assert_array_equal(r["relpos"], g["relpos"])
    assert r["num_clashes"] == int(g["num_clashes"])
    habs = np.abs(r["h_layers"]).reshape(o.hp.depth, -1)
    np.testing.assert_allclose(habs.mean(1), g["h_absmean"], rtol=1e-5)
    assert rel_inf(r["h_layers"][0], g["h_first"]) < 2e-5
    assert rel_inf(r["h_layers"][-1], g["h_last"]) < 5e-5
    assert rel_inf(r["pos_out"], g["pos_out"]) < 1e-6
    # SURVEY 8(d) gate 1: <= 1e-4 rel (L-inf / |.|-inf) on scores and f, <= 1e-4 abs on energy
    assert rel_inf(r["f"], g["f"]) < 1e-4
    assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4
    assert rel_inf(r["ires"], g["ires"][:, 0]) < 1e-4


# ---- f-2: second model family, DFMDock.forward = move_to_lig_center + EGNN_Net(predict=True) ------------
FWD2_CASES = ["fwd2_syn_9_7", "fwd2_syn_24_16", "fwd2_syn_64_48_p0", "fwd2_syn_64_48_p1", "fwd2_syn_64_48_p2",
              "fwd2_7CEI_p0", "fwd2_7CEI_p1", "fwd2_7CEI_p2", "fwd2_sum_syn_24_16"]


@pytest.mark.parametrize("case", FWD2_CASES)
def test_pair_family_score_matches_reference(case, blob_pair):
    g = load_golden(case + ".npz")
    hp = pair_hparams(agg_mean="sum" not in case)
    o = ora.Oracle(blob_pair, complex_for(case), hp)       # the blob layout does not depend on `agg`
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"])
    assert r["num_clashes"] == int(g["num_clashes"])
    assert rel_inf(r["h_layers"][0], g["h_first"]) < 2e-5
    assert rel_inf(r["h_layers"][-1], g["h_last"]) < 5e-5
    assert rel_inf(r["f"], g["f"]) < 1e-4
    assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4 * max(1.0, abs(float(g["energy"])))
    assert abs(float(r["confidence"]) - float(g["confidence_logits"])) < 1e-4
    assert rel_inf(r["ires"], g["ires_logits"]) < 1e-4


# ---- a-1 / a-2: the sampler ------------------------------------------------------------
@pytest.mark.parametrize("case,steps", [("rollout_syn_24_16", 40), ("rollout_syn_64_48", 40), ("rollout_7CEI", 6),
                                        ("rollout_esm_1QA9", 5), ("rollout_esm_1AVX", 5), ("rollout_esm_1H1V", 5),
                                        ("rollout_esmq_1JPS", 5), ("rollout_esmq_2SNI", 5)])
def test_sampler_rollout_injected(case, steps, blob):
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    inj = dict(R0=g["R0"], tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"].astype(np.int32))
    r = o.sample(num_steps=steps, inject=inj, trace=True, seed=0)
    assert r["forwards"] == steps + 1
    np.testing.assert_allclose(r["init_pose"], g["init_pose"], atol=2e-5)
    ca = r["trace_pose"][:, :, 1, :]
    ref = g["poses"][:, :, 1, :]
    rmsd = np.sqrt(((ca - ref) ** 2).sum(-1).mean(-1))
    # gate 3: short injected rollouts stay within 0.05 A (fp32); whole run reported
    assert rmsd[:5].max() < 0.05, rmsd[:5]
    assert rmsd.max() < 0.5, rmsd.max()
    # free-running: pose drift feeds back into the scores, so this is looser than the teacher-forced gate
    np.testing.assert_allclose(r["trace_scores"][:5, 0:3], g["tr_score"][:5], rtol=0,
                               atol=1e-3 * np.abs(g["tr_score"]).max())
    if rmsd.max() < 1e-3:
        assert abs(float(r["energy"]) - float(g["final_energy"])) < 1e-3
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)
        np.testing.assert_allclose(r["rot_update"], g["rot_update"], atol=2e-4)


@pytest.mark.parametrize("case,steps", [("rollout_syn_64_48", 40)])
def test_sampler_teacher_forced_steps(case, steps, blob):
    """Per-step: feed the reference's pose, compare the scores and the Euler-Maruyama update."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    k = load_golden("scalar_kats.npz")
    ts, dt = k["time_steps"], k["dt"]
    g3, gso, _, _ = ora.diffusion_coefs(ts)
    for i in (0, 1, 7, 20, 38, 39):
        pose = g["init_pose"] if i == 0 else g["poses"][i - 1]
        r = o.score(pose, ts[i], edges=g["edges"][i], debug=False)
        assert rel_inf(r["tr_score"], g["tr_score"][i]) < 1e-4
        assert rel_inf(r["rot_score"], g["rot_score"][i]) < 1e-4
        assert abs(float(r["energy"]) - float(g["energy"][i])) < 1e-4
        ns = 0.0 if i == steps - 1 else 0.5
        rot = ora.torch_reverse(gso[i], g["rot_score"][i], dt, ns, g["z_rot"][i])
        tr = ora.torch_reverse(g3[i], g["tr_score"][i], dt, ns, g["z_tr"][i])
        np.testing.assert_allclose(rot, g["step_rot"][i], atol=1e-7, rtol=1e-6)
        np.testing.assert_allclose(tr, g["step_tr"][i], atol=1e-6, rtol=1e-6)
        nxt = ora.modify_coords(pose, g["step_rot"][i], g["step_tr"][i])
        assert np.abs(nxt - g["poses"][i]).max() < 1e-4


# ---- f-4: sampler variants pinned to reference runs (tests/golden/make_golden_r02.py) ---------------------------
def _ca_rmsd(a, b):
    return np.sqrt(((a[:, :, 1, :] - b[:, :, 1, :]) ** 2).sum(-1).mean(-1))


@pytest.mark.parametrize("case,steps", [("rollout_anneal_syn_24_16", 40), ("rollout_anneal_7CEI", 6)])
def test_sampler_noise_annealing_vs_reference(case, steps, blob):
    """inference_base.py:428-430: noise scale = the time step."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    inj = dict(R0=g["R0"], tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    r = o.sample(num_steps=steps, inject=inj, trace=True, noise_annealing=True)
    rmsd = _ca_rmsd(r["trace_pose"], g["poses"])
    assert rmsd[:5].max() < 0.05 and rmsd.max() < 0.5, rmsd
    np.testing.assert_allclose(r["trace_scores"][0, 0:3], g["tr_score"][0], rtol=0, atol=1e-4 * np.abs(g["tr_score"][0]).max())
    if rmsd.max() < 1e-3:
        assert abs(float(r["energy"]) - float(g["final_energy"])) < 1e-3
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)


@pytest.mark.parametrize("case,steps", [("rollout_ode_syn_24_16", 40), ("rollout_ode_7CEI", 6)])
def test_sampler_ode_vs_reference(case, steps, blob):
    """inference_mlsb.Sampler.Euler_Maruyama_sampler(ode=True) (src/inference_mlsb.py:264-350; so3_diffuser.py:367-368).
    That sampler centres both chains first, so its poses are inference_base's shifted by -c1 (SE(3) equivariance)."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    inj = dict(R0=g["R0"], tr_draw=g["tr_draw"], edges=g["edges"])
    r = o.sample(num_steps=steps, inject=inj, trace=True, ode=True)
    np.testing.assert_allclose(r["init_pose"] - g["c1"], g["init_pose"], atol=5e-5)
    rmsd = _ca_rmsd(r["trace_pose"] - g["c1"], g["poses"])
    assert rmsd[:5].max() < 0.05 and rmsd.max() < 0.5, rmsd
    np.testing.assert_allclose(r["trace_scores"][0, 0:3], g["tr_score"][0], rtol=0, atol=2e-4 * np.abs(g["tr_score"][0]).max())
    if rmsd.max() < 1e-3:
        assert abs(float(r["energy"]) - float(g["final_energy"])) < 1e-3
        assert r["num_clashes"] == int(g["final_num_clashes"])


@pytest.mark.parametrize("case,steps", [("rollout2_syn_24_16", 40), ("rollout2_7CEI", 6)])
def test_pair_family_sampler_vs_reference(case, steps, blob_pair):
    """Second family's sampler (src/inference.py:292-372): randomize_pose / modify_coords about the ALL-ATOM centroids
    (:220-254), DFMDock.forward as the model."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob_pair, complex_for(case), pair_hparams())
    inj = dict(R0=g["R0"], tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    r = o.sample(num_steps=steps, inject=inj, trace=True)
    np.testing.assert_allclose(r["init_pose"], g["init_pose"], atol=3e-5)
    rmsd = _ca_rmsd(r["trace_pose"], g["poses"])
    assert rmsd[:5].max() < 0.05 and rmsd.max() < 0.5, rmsd
    if rmsd.max() < 1e-3:
        assert abs(float(r["energy"]) - float(g["final_energy"])) < 1e-3 * max(1.0, abs(float(g["final_energy"])))
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)
        np.testing.assert_allclose(r["rot_update"], g["rot_update"], atol=2e-4)


@pytest.mark.parametrize("flag", [0, 1])
def test_pair_family_sym_channel(flag):
    """positional_embed_dim = 67 (configs/model/DFMDock.yaml:5): the 67th channel is the homomer flag."""
    from dfmdock_amd.weights import HParams, make_random_weights, pack_blob
    hp = HParams(family=1, mask_dist=20.0, positional_embed_dim=67)
    blob67 = pack_blob(make_random_weights(0, hp), hp)
    g = load_golden(f"fwd2_sym{flag}_syn_24_16.npz")
    o = ora.Oracle(blob67, complex_for("syn_24_16"), hp, homomer=bool(flag))
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"])
    assert rel_inf(r["f"], g["f"]) < 1e-4 and rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4 * max(1.0, abs(float(g["energy"])))
    assert abs(float(r["confidence"]) - float(g["confidence_logits"])) < 1e-4


# ---- BASELINE configurations C3 / C5 / DB5 backbones: one reference evaluation each -------------------------------
@pytest.mark.parametrize("case", ["fwd_c3_300_300", "fwd_db5_1AVX", "fwd_db5_4POU", "fwd_c5_1000_1000"])
def test_score_matches_reference_large(case, blob):
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32))
    assert int(r["bins"].astype(np.int64).sum()) == int(g["codes_sum"])
    np.testing.assert_array_equal(r["bins"][::37], g["bins_sample"])
    np.testing.assert_array_equal(r["relpos"][::37], g["relpos_sample"])
    assert r["num_clashes"] == int(g["num_clashes"])
    habs = np.abs(r["h_layers"]).reshape(o.hp.depth, -1)
    np.testing.assert_allclose(habs.mean(1), g["h_absmean"], rtol=2e-5)
    assert rel_inf(r["f"], g["f"]) < 1e-4
    assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4
    assert rel_inf(r["ires"], g["ires"][:, 0]) < 1e-4


# ---- DB5 complexes with the reference's REAL ESM-2 node features (tests/golden/make_golden_r05.py) ---------------------------
@pytest.mark.parametrize("case", ["fwd_esm_1QA9", "fwd_esm_1AVX", "fwd_esm_1H1V"])
def test_score_matches_reference_real_esm(case, blob):
    """x = cat[ESM-2 block, one-hot(seq)] as ppi_dataset.py:249-265 builds it (ESM rounded to fp16 on both sides).  The per-edge
    bins are compared element by element: at most TWO of the N*K*4 bins may sit on the other side of a bin boundary (torch's
    kernels and libm differ in the last bit of atan2 / acos / a 3-term sum; on real backbones that happens about once per 10^5
    bins), by one bin.  A flipped bin swaps one edge's embedding row, which moves that node's force by ~1e-3: the evaluation is
    then repeated with the reference's bins handed in, so that everything downstream is still held to 1e-4."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob, complex_for(case))
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32))
    flips = r["bins"] != g["bins"]
    assert flips.sum() <= 2, int(flips.sum())
    if flips.any():
        d = np.abs(r["bins"].astype(int) - g["bins"].astype(int))[flips]
        assert set(d.tolist()) <= {1, 23}, d      # the neighbouring bin (24-bin dihedrals wrap)
        r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), bins=g["bins"])
        np.testing.assert_array_equal(r["bins"], g["bins"])
    np.testing.assert_array_equal(r["relpos"], g["relpos"])
    assert r["num_clashes"] == int(g["num_clashes"])
    habs = np.abs(r["h_layers"]).reshape(o.hp.depth, -1)
    np.testing.assert_allclose(habs.mean(1), g["h_absmean"], rtol=2e-5)
    assert rel_inf(r["f"], g["f"]) < 1e-4
    assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4
    assert rel_inf(r["ires"], g["ires"][:, 0]) < 1e-4


def test_score_matches_reference_on_the_other_twenty_db5_complexes(blob):
    """The 20 DB5 test complexes without a committed fp16 block, on their ESM-2 features quantised to int8 per residue
    (tests/golden/make_golden_r06.py: both sides use the dequantised values): one reference evaluation each at a noised pose, the same
    checks as above (bins element by element, <= 2 boundary flips re-evaluated on the reference's bins, everything else at 1e-4)."""
    from conftest import Q8_ESM_IDS, q8_golden, real_db5_complex
    flips_total = 0
    for cid in Q8_ESM_IDS:
        g = q8_golden(cid)
        o = ora.Oracle(blob, real_db5_complex(cid))
        e = g["edges"].astype(np.int32)
        r = o.score(g["lig_pos"], float(g["t"]), edges=e)
        flips = r["bins"] != g["bins"]
        assert flips.sum() <= 2, (cid, int(flips.sum()))
        flips_total += int(flips.sum())
        if flips.any():
            r = o.score(g["lig_pos"], float(g["t"]), edges=e, bins=g["bins"])
        np.testing.assert_array_equal(r["relpos"], g["relpos"])
        assert r["num_clashes"] == int(g["num_clashes"]), cid
        assert rel_inf(r["f"], g["f"]) < 1e-4, (cid, rel_inf(r["f"], g["f"]))
        assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4 and rel_inf(r["rot_score"], g["rot_score"]) < 1e-4, cid
        assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4, cid
        np.testing.assert_allclose(np.abs(r["h_layers"]).reshape(o.hp.depth, -1).max(1), g["h_absmax"], rtol=1e-4)
    assert flips_total <= 6


@pytest.mark.parametrize("case", ["fwd2_esm_1QA9", "fwd2_esm_1AVX", "fwd2_esm_1H1V"])
def test_pair_family_matches_reference_real_esm(case, blob_pair):
    """Second model family on the REAL ESM-2 feature blocks (tests/golden/make_golden_r05.py pair).  No per-edge bins in these
    fixtures: a bin-boundary flip (see test_score_matches_reference_real_esm) would show as ~1e-3 on f - gates 1e-4, widened to 2e-3
    only for a quantity that misses 1e-4 while |h| after the last layer still agrees to 1e-4 (none does today)."""
    g = load_golden(case + ".npz")
    o = ora.Oracle(blob_pair, complex_for(case), pair_hparams())
    r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32))
    assert r["num_clashes"] == int(g["num_clashes"])
    assert float(np.abs(r["h_layers"][0]).max()) == pytest.approx(float(g["h_absmax"][0]), rel=1e-4)
    assert float(np.abs(r["h_layers"][-1]).max()) == pytest.approx(float(g["h_absmax"][1]), rel=1e-4)
    assert rel_inf(r["f"], g["f"]) < 1e-4
    assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4
    assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4
    assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4 * max(1.0, abs(float(g["energy"])))
    assert abs(float(r["confidence"]) - float(g["confidence_logits"])) < 1e-4
    assert rel_inf(r["ires"], g["ires_logits"]) < 1e-4


def test_pair_family_matches_reference_on_the_other_twenty_db5_complexes(blob_pair):
    """Second model family on the int8-quantised ESM blocks of the remaining 20 DB5 complexes (tests/golden/make_golden_r06.py pair): same gates
    as the three fp16-block fixtures above."""
    from conftest import Q8_ESM_IDS, q8_golden, real_db5_complex
    for cid in Q8_ESM_IDS:
        g = q8_golden(cid, family=1)
        o = ora.Oracle(blob_pair, real_db5_complex(cid), pair_hparams())
        r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32))
        assert r["num_clashes"] == int(g["num_clashes"]), cid
        assert float(np.abs(r["h_layers"][-1]).max()) == pytest.approx(float(g["h_absmax"][1]), rel=1e-4), cid
        tol = 1e-4      # (worst of the 20 today: 2.2e-6 on f - no bin-boundary flip among these poses)
        assert rel_inf(r["f"], g["f"]) < tol, (cid, rel_inf(r["f"], g["f"]))
        assert rel_inf(r["tr_score"], g["tr_score"]) < tol and rel_inf(r["rot_score"], g["rot_score"]) < tol, cid
        assert abs(float(r["energy"]) - float(g["energy"])) < tol * max(1.0, abs(float(g["energy"]))), cid
        assert abs(float(r["confidence"]) - float(g["confidence_logits"])) < tol, cid
        assert rel_inf(r["ires"], g["ires_logits"]) < tol, cid


def test_phi_zero_pair_known_answer():
    """tests/golden/phi0_pair.npz: two residues of 1H1V (one rigid pose) whose planar angle is 0 to the last bit.  torch
    evaluates cos(phi) = 1.0 exactly -> phi = 0 -> bin 0; a left-to-right float32 evaluation gives 0.99999994 -> 0.02 degrees ->
    bin 1.  Pinned here: the oracle's angle is within 0.05 degrees of the reference's 0 and every other feature of the pair is
    identical - i.e. the only freedom is the documented one-ulp one."""
    g = load_golden("phi0_pair.npz")
    dist, omega, theta, phi = ora.coords6d_full(g["pos"])
    assert g["phi"][0, 1] == 0.0 and 0.0 <= phi[0, 1] < 0.05
    np.testing.assert_allclose(dist, g["dist"], rtol=2e-7)
    # CA_i, CB_i, CB_j are collinear here, so the two dihedrals through that axis are ill-conditioned (1e-3 degrees apart)
    np.testing.assert_allclose(omega[0, 1], g["omega"][0, 1], atol=5e-3)
    np.testing.assert_allclose(theta[[0, 1], [1, 0]], g["theta"][[0, 1], [1, 0]], atol=5e-3)
    np.testing.assert_allclose(phi[1, 0], g["phi"][1, 0], atol=2e-4)
    b = ora.bins_full(g["pos"])
    np.testing.assert_array_equal(b[..., :3], g["bins"][..., :3])
    assert b[1, 0, 3] == g["bins"][1, 0, 3] and b[0, 1, 3] in (0, 1)


# ---- further weight draws (two more seeds + one 3x-scaled draw), both families: the reference run on each -----------------
from conftest import DRAWS, DRAW_CASES, draw_blob, draw_golden, draw_hparams  # noqa: E402


@pytest.mark.parametrize("draw", DRAWS)
@pytest.mark.parametrize("family", [0, 1])
def test_score_matches_reference_on_other_weight_draws(family, draw):
    hp = draw_hparams(family)
    bl = draw_blob(family, draw)
    for case in DRAW_CASES[family]:
        g = draw_golden(family, draw, case)
        o = ora.Oracle(bl, complex_for(case), hp)
        r = o.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32))
        assert r["num_clashes"] == int(g["num_clashes"]), case
        habs = np.abs(r["h_layers"]).reshape(hp.depth, -1)
        np.testing.assert_allclose(habs.mean(1), g["h_absmean"], rtol=2e-5, err_msg=case)
        assert rel_inf(r["f"], g["f"]) < 1e-4, case
        assert rel_inf(r["tr_score"], g["tr_score"]) < 1e-4, case
        assert rel_inf(r["rot_score"], g["rot_score"]) < 1e-4, case
        assert abs(float(r["energy"]) - float(g["energy"])) < 1e-4 * max(1.0, abs(float(g["energy"]))), case
        assert rel_inf(r["ires"], g["ires"]) < 1e-4, case
        if family:
            assert abs(float(r["confidence"]) - float(g["confidence_logits"])) < 1e-4, case


@pytest.mark.parametrize("which,steps", [("rollout", 40), ("rollout7", 6)])
@pytest.mark.parametrize("draw", DRAWS)
@pytest.mark.parametrize("family", [0, 1])
def test_sampler_rollout_on_other_weight_draws(family, draw, which, steps):
    """Reference sampler runs per draw (40 steps on syn_24_16, 6 on the DB5 pair 7CEI) with R0 / the N(0,30^2) draw / z / edge
    lists of the seed-0 rollouts replayed."""
    g = draw_golden(family, draw, which)
    o = ora.Oracle(draw_blob(family, draw), complex_for("7CEI" if which == "rollout7" else "syn_24_16"), draw_hparams(family))
    inj = dict(R0=g["R0"], tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    r = o.sample(num_steps=steps, inject=inj, trace=True)
    rmsd = _ca_rmsd(r["trace_pose"], g["poses"])
    assert rmsd[:5].max() < 0.05 and rmsd.max() < 0.5, rmsd
    if rmsd.max() < 1e-3:
        assert abs(float(r["energy"]) - float(g["final_energy"])) < 1e-3 * max(1.0, abs(float(g["final_energy"])))
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)
        np.testing.assert_allclose(r["rot_update"], g["rot_update"], atol=2e-4)


def test_pair_family_dist_logits_match_reference(blob_pair):
    """dist_logits = to_dist(cat[h_r, h_l, D]) [R, L, 64] (egnn_net.py:347-352,:447,:500; tests/golden/make_golden_pair.py dist)."""
    d = load_golden("fwd2_dist.npz")
    for case, key, stride in (("fwd2_syn_24_16", "syn_24_16", 1), ("fwd2_7CEI_p1", "cei_p1_stride8", 8)):
        g = load_golden(case + ".npz")
        r = ora.Oracle(blob_pair, complex_for(case), pair_hparams()).score(g["lig_pos"], float(g["t"]), edges=g["edges"], dist=True)
        got = r["dist_logits"][::stride, ::stride]
        assert got.shape == d[key].shape and rel_inf(got, d[key]) < 1e-4, case
        np.testing.assert_allclose(r["dist_logits"][:4, :4], g["dist_logits_sample"], atol=1e-4 * np.abs(d[key]).max())

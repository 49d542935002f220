"""GPU parity of the sampler variants and of the optional outputs against goldens captured from the reference
(tests/golden/make_golden_r02.py), plus regression tests of the drop-in adapter:

  noise annealing            inference_base.py:428-430
  ODE sampler                inference_mlsb.py:264-350, so3_diffuser.py:367-368
  second family's sampler    src/inference.py:220-372 (all-atom centroids)
  sym channel                configs/model/DFMDock.yaml:5 (positional_embed_dim 67)
  ires                       score_net_mlsb.py:297-303,:383 / egnn_net.py:362-368,:462
  GraphNorm with |mean| >> std
  Score_Model.complex_for    a second ligand pose / a re-centred receptor must never be served from a stale complex
"""
import os

import numpy as np
import pytest

from conftest import ROOT, complex_for, load_golden, pair_hparams

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


def _inject(g, with_z=True):
    inj = dict(R0=g["R0"].astype(np.float32).reshape(1, 9), tr_draw=g["tr_draw"].reshape(1, 3), edges=g["edges"][None])
    if with_z:
        inj.update(z_rot=g["z_rot"][None], z_tr=g["z_tr"][None])
    return inj


ROLL_GATE = {"fp32": (0.05, 0.5), "mfma16": (0.5, 0.5), "f16": (0.5, 0.5)}


@pytest.mark.parametrize("prec", ["fp32", "mfma16", "f16"])
@pytest.mark.parametrize("case,steps", [("rollout_anneal_syn_24_16", 40), ("rollout_anneal_7CEI", 6)])
def test_noise_annealing_vs_reference(case, steps, prec, model):
    from dfmdock_amd import engine
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.sample(B=1, num_steps=steps, inject=_inject(g), trace=True, noise_annealing=True, mfma16=prec == "mfma16", f16=prec == "f16")
    rmsd = ca_rmsd(r["trace_pose"][0], g["poses"])
    g5, gall = ROLL_GATE[prec]
    assert rmsd[:5].max() < g5 and rmsd.max() < gall, rmsd
    if prec == "fp32" and rmsd.max() < 1e-3:
        assert abs(float(r["energy"][0]) - float(g["final_energy"])) < 1e-3
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)
    gx.close()


@pytest.mark.parametrize("prec", ["fp32", "mfma16", "f16"])
@pytest.mark.parametrize("case,steps", [("rollout_ode_syn_24_16", 40), ("rollout_ode_7CEI", 6)])
def test_ode_sampler_vs_reference(case, steps, prec, model):
    """The reference's ODE run comes from inference_mlsb.Sampler, which centres both chains first: its poses are this engine's
    (inference_base convention) shifted by the receptor CA centroid c1."""
    from dfmdock_amd import engine
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.sample(B=1, num_steps=steps, inject=_inject(g, with_z=False), trace=True, ode=True, mfma16=prec == "mfma16", f16=prec == "f16")
    np.testing.assert_allclose(r["init_pose"][0] - g["c1"], g["init_pose"], atol=1e-4)
    rmsd = ca_rmsd(r["trace_pose"][0] - g["c1"], g["poses"])
    g5, gall = ROLL_GATE[prec]
    assert rmsd[:5].max() < g5 and rmsd.max() < gall, rmsd
    if prec == "fp32" and rmsd.max() < 1e-3:
        assert abs(float(r["energy"][0]) - float(g["final_energy"])) < 1e-3
        assert int(r["num_clashes"][0]) == int(g["final_num_clashes"])
    gx.close()


@pytest.mark.parametrize("prec", ["fp32", "mfma16", "f16"])
@pytest.mark.parametrize("case,steps", [("rollout2_syn_24_16", 40), ("rollout2_7CEI", 6)])
def test_pair_family_sampler_vs_reference(case, steps, prec, model_pair):
    """src/inference.py's sampler: randomize_pose / modify_coords about the all-backbone-atom centroids."""
    from dfmdock_amd import engine
    from oracle import oracle as ora
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(model_pair, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.sample(B=1, num_steps=steps, inject=_inject(g), trace=True, mfma16=prec == "mfma16", f16=prec == "f16")
    np.testing.assert_allclose(r["init_pose"][0], g["init_pose"], atol=5e-5)
    rmsd = ca_rmsd(r["trace_pose"][0], g["poses"])
    g5, gall = ROLL_GATE[prec]
    assert rmsd[:5].max() < g5 and rmsd.max() < gall, rmsd
    if prec == "fp32" and rmsd.max() < 1e-3:
        assert abs(float(r["energy"][0]) - float(g["final_energy"])) < 1e-3 * max(1.0, abs(float(g["final_energy"])))
        np.testing.assert_allclose(r["tr_update"], g["tr_update"], atol=2e-3)
        np.testing.assert_allclose(r["rot_update"], g["rot_update"], atol=2e-4)
    # rigid-body bookkeeping of this family: final pose = input ligand moved by (rot_update, tr_update) about its ALL-ATOM centroid
    x = ora.modify_coords_all_atom(cx["lig_pos"], r["rot_update"][0], r["tr_update"][0])
    assert np.abs(x - r["lig_pos"][0]).max() < 2e-3
    gx.close()


@pytest.mark.parametrize("flag", [0, 1])
def test_sym_channel_vs_reference(flag):
    """A 67-channel checkpoint (the stock DFMDock.yaml) loads; the homomer flag adds the sym column's contribution."""
    from dfmdock_amd import engine
    from dfmdock_amd.weights import HParams, make_random_weights, pack_blob
    hp = HParams(family=1, mask_dist=20.0, positional_embed_dim=67)
    engine.set_device(0)
    m = engine.Model(pack_blob(make_random_weights(0, hp), hp), hp)
    g = load_golden(f"fwd2_sym{flag}_syn_24_16.npz")
    cx = complex_for("syn_24_16")
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    gx.set_homomer(bool(flag))
    for prec, tol in (("fp32", 1e-4), ("mfma16", 1e-2), ("f16", 1e-2)):
        r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, mfma16=prec == "mfma16", f16=prec == "f16")
        assert rel_inf(r["f"][0], g["f"]) < tol and rel_inf(r["tr_score"][0], g["tr_score"].reshape(3)) < tol, prec
        assert abs(float(r["energy"][0]) - float(g["energy"])) < max(tol, 3e-2 if prec != "fp32" else tol) * max(1.0, abs(float(g["energy"])))
        assert abs(float(r["confidence"][0]) - float(g["confidence_logits"])) < max(tol, 3e-2 if prec != "fp32" else tol)
    if flag:     # switching the flag back reproduces the 66-channel behaviour exactly
        g0 = load_golden("fwd2_sym0_syn_24_16.npz")
        gx.set_homomer(False)
        r = gx.score(g0["lig_pos"], float(g0["t"]), edges=g0["edges"], energy=True)
        assert rel_inf(r["f"][0], g0["f"]) < 1e-4
    # a 66-channel model refuses the flag
    m66 = engine.Model(pack_blob(make_random_weights(0, pair_hparams()), pair_hparams()), pair_hparams())
    gx66 = engine.Complex(m66, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    with pytest.raises(ValueError):
        gx66.set_homomer(True)
    gx66.set_homomer(False)
    gx.close(); gx66.close(); m.close(); m66.close()


@pytest.mark.parametrize("case", ["fwd_syn_24_16", "fwd_7CEI_p1", "fwd_c3_300_300"])
def test_ires_vs_reference(case, model):
    from dfmdock_amd import engine
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    e = g["edges"].astype(np.int32)
    r = gx.score(g["lig_pos"], float(g["t"]), edges=e, energy=True, ires=True)
    assert r["ires"].shape == (1, gx.N)
    assert rel_inf(r["ires"][0], g["ires"][:, 0]) < 1e-4
    r16 = gx.score(np.stack([g["lig_pos"]] * 3), float(g["t"]), edges=np.stack([e] * 3), energy=False, ires=True, mfma16=True)
    assert rel_inf(r16["ires"][2], g["ires"][:, 0]) < 3e-2
    gx.close()


def test_pair_family_ires_vs_reference(model_pair):
    from dfmdock_amd import engine
    g = load_golden("fwd2_7CEI_p1.npz")
    cx = complex_for("7CEI")
    gx = engine.Complex(model_pair, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, ires=True)
    assert rel_inf(r["ires"][0], g["ires_logits"]) < 1e-4
    gx.close()


def test_graphnorm_large_mean_channels(blob):
    """GraphNorm channels whose mean is ~100x their standard deviation (a large node_mlp.0 bias): the fused column
    statistics of the 16-bit engines must not cancel (torch_geometric graph_norm.py subtracts the mean first)."""
    from dfmdock_amd import engine
    from dfmdock_amd.weights import HParams, pack_blob, unpack_blob
    from oracle import oracle as ora
    w = {k: v.copy() for k, v in unpack_blob(blob, HParams()).items()}
    rng = np.random.default_rng(2)
    for l in range(6):
        b = w[f"network.EGNN_{l}.egcl.node_mlp.0.bias"]
        idx = rng.choice(256, 24, replace=False)
        b[idx] = rng.choice([-1.0, 1.0], 24) * rng.uniform(60.0, 250.0, 24)     # |mean| / std of u is then 100 .. 500
    blob2 = pack_blob(w)
    engine.set_device(0)
    m = engine.Model(blob2)
    cx = complex_for("syn_64_48")
    g = load_golden("fwd_syn_64_48_p1.npz")
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    o = ora.Oracle(blob2, cx).score(g["lig_pos"], float(g["t"]), edges=g["edges"])
    r32 = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, debug=True)
    assert rel_inf(r32["h_last"][0], o["h_layers"][-1]) < 1e-4 and rel_inf(r32["f"][0], o["f"]) < 1e-4
    for prec, th, tf in (("f16", 3e-2, 1e-2), ("mfma16", 3e-2, 1e-2)):
        r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, debug=True, mfma16=prec == "mfma16", f16=prec == "f16")
        assert rel_inf(r["h_last"][0], o["h_layers"][-1]) < th, prec
        assert rel_inf(r["f"][0], o["f"]) < tf and rel_inf(r["tr_score"][0], o["tr_score"].reshape(3)) < tf, prec
    # batched == single stays bit-exact with the new statistics
    rb = gx.score(np.stack([g["lig_pos"]] * 5), float(g["t"]), edges=np.stack([g["edges"]] * 5), energy=True, mfma16=True)
    r1 = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, mfma16=True)
    for k in ("f", "tr_score", "rot_score", "energy"):
        np.testing.assert_array_equal(rb[k][3], r1[k][0])
    gx.close(); m.close()


def test_score_model_never_serves_a_stale_complex(blob):
    """Two sampler calls with the same features and two ligand conformations: each returned (rot_update, tr_update) must move
    ITS input ligand onto the returned pose (inference_base.py:408-412,:354-364); forward() follows a re-centred receptor."""
    import torch
    from dfmdock_amd.score_model import Euler_Maruyama_sampler, Score_Model
    from oracle import oracle as ora
    cx = complex_for("7CEI")
    m = Score_Model(blob, precision="fp32")
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cx.items()}
    g = load_golden("fwd_7CEI_p2.npz")
    lig_a, lig_b = cx["lig_pos"], g["lig_pos"]                     # native and a rigidly displaced conformation
    assert np.abs(lig_a - lig_b).max() > 1.0
    for lig in (lig_a, lig_b, lig_a):
        batch["lig_pos"] = torch.from_numpy(lig.copy())
        _, lig_pos, rot_update, tr_update, _ = Euler_Maruyama_sampler(m, batch, num_steps=5, device="cuda", seed=7)
        moved = ora.modify_coords(lig, rot_update.numpy(), tr_update.numpy())
        assert np.abs(moved - lig_pos.numpy()).max() < 2e-3
    # in-place edit of the same tensor object must be seen as well
    with torch.no_grad():
        batch["lig_pos"] += 3.0
    _, lig_pos, rot_update, tr_update, _ = Euler_Maruyama_sampler(m, batch, num_steps=5, device="cuda", seed=7)
    moved = ora.modify_coords(lig_a + np.float32(3.0), rot_update.numpy(), tr_update.numpy())
    assert np.abs(moved - lig_pos.numpy()).max() < 2e-3
    # forward(): translating the whole complex leaves energy / scores unchanged, and the features are not re-uploaded
    batch["lig_pos"] = torch.from_numpy(g["lig_pos"].copy())
    batch["t"] = torch.tensor([float(g["t"])])
    m.seed, m._calls = 3, 0
    a = m(batch)
    handle = m._cx
    shift = torch.tensor([11.0, -4.0, 6.5])
    batch["rec_pos"] = batch["rec_pos"] + shift
    batch["lig_pos"] = batch["lig_pos"] + shift
    m._calls = 0                                                   # same Philox stream -> same graph
    b = m(batch)
    assert m._cx is handle
    assert rel_inf(b["tr_score"].numpy(), a["tr_score"].numpy()) < 1e-3 and abs(float(a["energy"]) - float(b["energy"])) < 1e-3
    assert a["ires"].shape == (214, 1)
    # new features -> a new complex
    batch["rec_x"] = batch["rec_x"] * 1.0001
    m(batch)
    assert m._cx is not handle


def test_tile_tasks_equal_node_tasks_bitwise(tmp_path):
    """k_edge_msg hands a wave whole nodes (segment sum in registers) or - small launches - single tiles (partial sums added
    atomically to the zeroed agg).  Two addends per element: both forms must give bitwise the same score, also across batch sizes
    (DFM_EDGE_SPLIT forces the form; it is read once per process, hence the subprocesses)."""
    import subprocess, sys, textwrap
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {ROOT!r})
        from dfmdock_amd import engine
        from dfmdock_amd.synthetic import make_complex
        from dfmdock_amd.weights import make_random_weights, pack_blob
        engine.set_device(0)
        model = engine.Model(pack_blob(make_random_weights(0)))
        cx = make_complex(120, 94, seed=3)
        gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        B = int(sys.argv[2])
        r = gx.score(np.repeat(cx["lig_pos"][None], B, 0), 0.5, seed=5, mfma16=True, energy=True, debug=True)
        np.savez(sys.argv[1], **{{k: r[k][0] for k in ("f", "tr_score", "rot_score", "energy", "edges", "h_first", "h_last")}})
    """))
    outs = {}
    for tag, env, B in (("node1", "0", 1), ("tile1", "1", 1), ("node9", "0", 9), ("tile9", "1", 9)):
        p = subprocess.run([sys.executable, str(script), str(tmp_path / f"{tag}.npz"), str(B)], env=dict(os.environ, DFM_EDGE_SPLIT=env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        outs[tag] = np.load(tmp_path / f"{tag}.npz")
    for tag in ("tile1", "node9", "tile9"):
        for k in outs["node1"].files:
            assert (outs[tag][k] == outs["node1"][k]).all(), (tag, k)


def test_engine_variants_stay_within_the_16bit_gate(tmp_path):
    """The 16-bit engine's plan against its variants - fp32 A_i (DFM_F_F16), bf16 operands in layers 0..4 (DFM_F_BF16_OPS): same
    graph, all within SURVEY's 16-bit gate of the fp32 engine (the bf16-operand plan at its own stated 2e-2).  Precision is selected
    by flags only; the one diagnostic switch left (DFM_EDGE_SPLIT: task granularity, bitwise-neutral) is named by the config string
    when set (its own process: switches are read once)."""
    import subprocess, sys, textwrap
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {ROOT!r})
        from dfmdock_amd import engine
        from dfmdock_amd.synthetic import make_complex
        from dfmdock_amd.weights import make_random_weights, pack_blob
        engine.set_device(0)
        model = engine.Model(pack_blob(make_random_weights(0)))
        cx = make_complex(150, 110, seed=4)
        gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        poses = np.repeat(cx["lig_pos"][None], 3, 0)
        r = gx.score(poses, 0.4, seed=7, mfma16=True, energy=True, return_edges=True)
        out = {{"cfg": np.array(engine.config_string())}}
        for tag, kw in (("", dict(mfma16=True)), ("32", {{}}), ("_a32", dict(f16=True)), ("_bf", dict(mfma16=True, bf16_ops=True))):
            x = gx.score(poses, 0.4, edges=r["edges"], energy=True, **kw)
            out.update({{k + tag: x[k] for k in ("f", "tr_score", "rot_score", "energy")}})
        np.savez(sys.argv[1], **out)
    """))
    outs = {}
    for tag, env in (("default", {}), ("split", {"DFM_EDGE_SPLIT": "1"})):
        e = {k: v for k, v in os.environ.items() if not k.startswith("DFM_")}
        p = subprocess.run([sys.executable, str(script), str(tmp_path / f"{tag}.npz")], env=dict(e, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        outs[tag] = np.load(tmp_path / f"{tag}.npz")
    ref = outs["default"]
    assert "DFM_EDGE_SPLIT=1" in str(outs["split"]["cfg"]) and "env: none" in str(ref["cfg"])
    for k in ("f", "tr_score", "rot_score", "energy", "f32"):
        assert (outs["split"][k] == ref[k]).all(), k                        # task granularity never changes a bit
    for sfx, gate in (("", 1.0), ("_a32", 1.0), ("_bf", 2.0)):
        for k, tol in (("f", 1e-2), ("tr_score", 1e-2), ("rot_score", 1e-2), ("energy", 3e-2)):
            scale = np.abs(ref[k + "32"]).max() + 1e-12
            if k == "energy": scale = max(scale, 0.1)              # the energy gate's convention (test_gpu_configs.check_vs)
            assert np.abs(ref[k + sfx] - ref[k + "32"]).max() / scale < tol * gate, (sfx, k)
    for other in (ref["f_a32"], ref["f_bf"]):                            # the flags do select something else
        assert (other != ref["f"]).any()


def test_guards_and_optional_heads(blob):
    """(1) The message kernel addresses the [B][N][K] edge arrays through 32-bit buffer offsets: B * N * K * 4 >= 2^31 is refused up
    front (ValueError, nothing allocated) instead of silently reading zeros.  (2) Score_Model(with_ires=False) skips the
    interface-residue head and leaves the key out; the other outputs do not change."""
    import torch
    from dfmdock_amd import engine
    from dfmdock_amd.score_model import Score_Model
    from dfmdock_amd.synthetic import make_complex
    engine.set_device(0)
    cx = make_complex(40, 20, seed=9)                      # N = 60 -> K = 60
    gx = engine.Complex(engine.Model(blob), cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    B = (1 << 31) // (60 * 60 * 4) + 1                     # 149 131 trajectories of 60 nodes
    with pytest.raises(ValueError, match="split the batch"):
        gx.sample(B=B, num_steps=2, seed=1, mfma16=True)
    gx.close()
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cx.items()}
    batch["t"] = torch.tensor([0.4])
    a = Score_Model(blob, precision="fp32", seed=3)(batch)
    b = Score_Model(blob, precision="fp32", seed=3, with_ires=False)(batch)
    assert "ires" in a and "ires" not in b and a["ires"].shape == (60, 1)
    for k in ("tr_score", "rot_score", "f", "energy"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("prec", ["fp32", "mfma16", "f16"])
def test_ligand_only_last_layer_changes_nothing(prec, model):
    """When nobody reads the final node features (no energy / ires / debug tap) the last layer computes the messages of the ligand
    nodes only and skips its node model (score_net_mlsb.py:383-398: the force needs pos_out of the ligand nodes alone).  f and both
    scores must be bitwise what the full evaluation gives - on a 64+48 complex (node tasks) and at B = 1 (tile tasks), and a sampler
    run without traces (ligand-only in its 40 step evaluations) must end on the bitwise same pose as one with traces (full)."""
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    kw = dict(mfma16=prec == "mfma16", f16=prec == "f16")
    for (R, L, B) in [(64, 48, 3), (24, 16, 1), (129, 67, 2)]:
        cx = make_complex(R, L, seed=11)
        gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        rng = np.random.default_rng(R)
        poses = np.stack([cx["lig_pos"] + rng.normal(0, 0.7, 3).astype(np.float32) for _ in range(B)])
        full = gx.score(poses, 0.4, seed=5, energy=True, debug=True)
        lean = gx.score(poses, 0.4, edges=full["edges"], **kw)
        ref = gx.score(poses, 0.4, edges=full["edges"], energy=True, **kw)
        for k in ("f", "tr_score", "rot_score"):
            assert (lean[k] == ref[k]).all(), (R, L, B, k)
        s_full = gx.sample(B=B, num_steps=6, seed=9, trace=True, **kw)
        s_lean = gx.sample(B=B, num_steps=6, seed=9, **kw)
        assert (s_full["lig_pos"] == s_lean["lig_pos"]).all() and (s_full["energy"] == s_lean["energy"]).all(), (R, L, B)
        gx.close()


@pytest.mark.parametrize("family", [0, 1])
@pytest.mark.parametrize("depth", [1, 2, 3, 5])
def test_other_depths_both_families_lean_equals_full(depth, family):
    """the final node features end in W.h or W.h2 depending on the parity of the depth, and the pair heads of the
    second family read W.h on EVERY evaluation - with an odd depth and no energy / debug request they were fed the penultimate
    layer's features (default depth 6 hid it).  For depth 1 / 2 / 3 / 5 and both families: the fp32 engine against the oracle on the
    same graph, and for every engine the lean call (no energy, no taps - what the sampler's step evaluations issue) bitwise equal to
    the full one, including the h_first tap of a depth-1 model (its first layer is its last)."""
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import HParams, make_random_weights, pack_blob
    from oracle import oracle as ora
    hp = HParams(depth=depth, **({"family": 1, "mask_dist": 20.0} if family else {}))
    blob = pack_blob(make_random_weights(3, hp), hp)
    engine.set_device(0)
    m = engine.Model(blob, hp)
    cx = make_complex(40, 33, seed=21)
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    poses = np.stack([cx["lig_pos"], cx["lig_pos"] + np.float32(0.8)])
    full = gx.score(poses, 0.35, seed=2, energy=True, debug=True)
    o = ora.Oracle(blob, cx, hp)
    for b in range(2):
        r = o.score(poses[b], 0.35, edges=full["edges"][b])
        assert rel_inf(full["f"][b], r["f"]) < 1e-4 and rel_inf(full["tr_score"][b], np.asarray(r["tr_score"]).reshape(3)) < 1e-4, (depth, family, b)
        assert rel_inf(full["rot_score"][b], np.asarray(r["rot_score"]).reshape(3)) < 1e-4
        assert abs(float(full["energy"][b]) - float(r["energy"])) < 1e-4 * max(1.0, abs(float(r["energy"])))
    for kw in ({}, dict(mfma16=True), dict(f16=True)):
        ref = gx.score(poses, 0.35, edges=full["edges"], energy=True, debug=True, **kw)
        lean = gx.score(poses, 0.35, edges=full["edges"], energy=False, **kw)
        for k in ("f", "tr_score", "rot_score"):
            assert (lean[k] == ref[k]).all(), (depth, family, kw, k)
        if kw:
            assert rel_inf(ref["f"], full["f"]) < 1e-2 and rel_inf(ref["tr_score"], full["tr_score"]) < 1e-2
        s_full = gx.sample(B=2, num_steps=4, seed=9, trace=True, **kw)
        s_lean = gx.sample(B=2, num_steps=4, seed=9, **kw)
        assert (s_full["lig_pos"] == s_lean["lig_pos"]).all() and (s_full["energy"] == s_lean["energy"]).all(), (depth, family, kw)
    gx.close()
    m.close()


def test_narrow_gemm_tiles_equal_wide_tiles_bitwise(tmp_path):
    """k_gemm_split runs 64 x 64 / 64 x 128 tiles for small launches and 64 x 256 tiles above two workgroups per CU: the same MFMA sequence per
    output element and the same row order in the GraphNorm column statistics, so the choice (a function of the batch size) must not
    show in a single bit - of a score evaluation (node features of the first and last layer included) or of a sampled pose.
    DFM_GEMM_NARROW_MAXWG forces the form (read once per process: subprocesses); B = 1 and B = 7 also cross the tile-task threshold."""
    import subprocess, sys, textwrap
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {ROOT!r})
        from dfmdock_amd import engine
        from dfmdock_amd.synthetic import make_complex
        from dfmdock_amd.weights import make_random_weights, pack_blob
        engine.set_device(0)
        model = engine.Model(pack_blob(make_random_weights(0)))
        cx = make_complex(120, 94, seed=3)
        gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        B = int(sys.argv[2])
        r = gx.score(np.repeat(cx["lig_pos"][None], B, 0), 0.5, seed=5, mfma16=True, energy=True, debug=True)
        s = gx.sample(B=B, num_steps=5, seed=4, mfma16=True)
        out = {{k: r[k][0] for k in ("f", "tr_score", "rot_score", "energy", "edges", "h_first", "h_last")}}
        out.update(pose=s["lig_pos"][0], final_energy=s["energy"][0])
        np.savez(sys.argv[1], **out)
    """))
    outs = {}
    # quarter1 / quarter7: 64 x 64 tiles (waves 2 x 2; what launches with fewer 64 x 128 workgroups than half the CUs run), forced at
    # B = 7 too.  The column statistics go out per 32-row half of a tile in every shape.
    # (quarter tiles run a different K loop - four stages in flight, double-buffered operand tiles - with the same MFMA sequence)
    for tag, env, B in (("wide1", "0", 1), ("narrow1", "1000000", 1), ("wide7", "0", 7), ("narrow7", "1000000", 7), ("quarter1", "1000000", 1),
                        ("quarter7", "1000000", 7)):
        p = subprocess.run([sys.executable, str(script), str(tmp_path / f"{tag}.npz"), str(B)],
                           env=dict(os.environ, DFM_GEMM_NARROW_MAXWG=env, DFM_GEMM_QUARTER_MAXWG="1000000" if tag.startswith("quarter") else "0"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        outs[tag] = np.load(tmp_path / f"{tag}.npz")
    for tag in ("narrow1", "wide7", "narrow7", "quarter1", "quarter7"):
        for k in outs["wide1"].files:
            assert (outs[tag][k] == outs["wide1"][k]).all(), (tag, k)


def test_long_schedule_beyond_the_default_time_grid(model):
    """The reference's sampler takes any num_steps (inference_base.py:403-404); the workspace's time grid (t, the time-dependent half
    of the two scale MLPs, the replayed step parameters) is sized for 4096 steps and grows on demand (4097+ steps were
    rejected).  A 4500-step run, then a short one on the grown grid equal to a fresh handle's."""
    from dfmdock_amd import engine
    cx = complex_for("fwd_syn_9_7")
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    a = gx.sample(B=2, num_steps=40, seed=5, mfma16=True)
    r = gx.sample(B=2, num_steps=4500, seed=5, mfma16=True)
    assert np.isfinite(r["lig_pos"]).all() and np.isfinite(r["energy"]).all()
    b = gx.sample(B=2, num_steps=40, seed=5, mfma16=True)
    for k in ("lig_pos", "energy", "tr_update", "rot_update"):
        np.testing.assert_array_equal(a[k], b[k])
    gx.close()

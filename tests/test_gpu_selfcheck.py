"""dfm_complex_selfcheck - runtime parity evidence for weights the build has never seen (SURVEY 8(d) gate 5).

The reference computes in fp32 throughout (src/models/score_net_mlsb.py:343-425) and loads whatever checkpoint the user has
(src/inference_base.py:611-616); the 16-bit engine clamps fp16 silently.  The self-check runs the complex's own pose through the
fp32 engine and the 16-bit engine on the same engine-drawn graphs and reports deviations, the cancellation ratios that condition the
two scores, per-layer magnitudes of everything stored as fp16, and a verdict.  Tested here:

  * its numbers ARE the engines' numbers: deviations recomputed from separate score calls on the same graphs, max|h| from the taps
  * the four weight draws of the parity suite pass on both families; a draw with every MLP Linear x30 MUST be flagged
    (fp16 range), and so must a model whose A_i exceeds 65504 by construction
  * the relation behind the score gates: |d score| <= 2 * score_bound (= sqrt(3) dev_f max|f| / |pooled vector|) on perturbed
    poses of random complexes (tools/stress_sizes.py's population, where r03 found tr_score off by 1.75e-2 with f at 1.8e-3)
  * the drivers print the line and fall back to the fp32 ENGINE when the check fails
"""
import numpy as np
import pytest

from conftest import complex_for, draw_blob, draw_hparams

pytestmark = pytest.mark.gpu


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(family, draw):
    from dfmdock_amd import engine
    from dfmdock_amd.weights import make_weight_draw, pack_blob
    engine.set_device(0)
    hp = draw_hparams(family)
    blob = draw_blob(family, draw) if draw != "s0" else pack_blob(make_weight_draw("s0", hp), hp)
    return engine.Model(blob, hp), hp


@pytest.mark.parametrize("family", [0, 1])
def test_selfcheck_reports_the_engines_own_numbers(family):
    from dfmdock_amd import engine
    m, hp = _model(family, "s1")
    cx = complex_for("fwd_syn_64_48_p0")
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    ts = np.array([1.0, 0.4, 0.001], np.float32)
    for prec in ("mfma16", "f16"):
        r = gx.selfcheck(t=ts, seed=11, precision=prec)
        assert r["n_eval"] == 3 and r["depth"] == hp.depth and len(r["max_h"]) == hp.depth + 1 and len(r["max_A"]) == hp.depth
        # the same three graphs: evaluation 0 of a call with this seed draws them again
        poses = np.repeat(cx["lig_pos"][None], 3, 0)
        a = gx.score(poses, ts, seed=11, energy=True, debug=True)
        # (the 16-bit pass of the check is the engine as dfm_sample runs it: layer 0 through the message table for "mfma16")
        b = gx.score(poses, ts, edges=a["edges"], energy=True, l0_table=prec == "mfma16", **engine.precision_kwargs(prec))
        assert r["dev_f"] == pytest.approx(max(rel_inf(b["f"][k], a["f"][k]) for k in range(3)), rel=1e-5)
        assert r["dev_tr_score"] == pytest.approx(max(rel_inf(b["tr_score"][k], a["tr_score"][k]) for k in range(3)), rel=1e-5)
        assert r["dev_rot_score"] == pytest.approx(max(rel_inf(b["rot_score"][k], a["rot_score"][k]) for k in range(3)), rel=1e-5)
        assert r["dev_energy"] == pytest.approx(max(abs(float(b["energy"][k]) - float(a["energy"][k])) / max(abs(float(a["energy"][k])), 0.1)
                                                    for k in range(3)), rel=1e-4, abs=1e-9)
        assert r["max_h"][-1] == pytest.approx(float(np.abs(a["h_last"]).max()), rel=1e-6)
        assert r["max_h"][1] == pytest.approx(float(np.abs(a["h_first"]).max()), rel=1e-6)
        f = a["f"].astype(np.float64)
        ratio = min(np.linalg.norm(f[k].mean(0)) / np.linalg.norm(f[k], axis=1).mean() for k in range(3))
        assert r["cancel_ratio"][0] == pytest.approx(ratio, rel=1e-4)
        assert 0 < r["cancel_ratio"][1] <= 1 and all(x > 0 for x in r["max_pre"] + r["max_acc"] + r["max_A"] + r["max_Bm"] + r["max_tab"])
        assert all(s >= b_ for s, b_ in zip(r["max_sum16"], r["max_Bm"]))
        assert r["ok"] and r["range_ok"] and r["dev_ok"] and r["saturated"] == 0 and r["headroom"] > 1
        assert "OK" in engine.format_selfcheck(r, "syn")
    with pytest.raises(ValueError):
        gx.selfcheck(precision="fp32")
    with pytest.raises(ValueError):
        gx.selfcheck(t=[1.5])
    gx.close()
    m.close()


@pytest.mark.parametrize("family", [0, 1])
@pytest.mark.parametrize("draw", ["s0", "s1", "s2", "x3"])
def test_four_weight_draws_pass_the_selfcheck(family, draw):
    """The draws every tolerance of the parity suite rests on (tests/golden/make_golden_draws.py) must not be flagged - on the
    7CEI native pose and on a synthetic complex, for the shipped 16-bit engine."""
    from dfmdock_amd import engine
    m, _ = _model(family, draw)
    for case in ("fwd_7CEI_p0", "fwd_syn_24_16"):
        cx = complex_for(case)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        r = gx.selfcheck(n_eval=4, seed=3)
        assert r["ok"], engine.format_selfcheck(r, f"{case} f{family} {draw}")
        assert r["dev_f"] < 1e-2 and r["dev_energy"] < 3e-2 and r["headroom"] > 10
        gx.close()
    m.close()


def test_out_of_range_models_are_flagged():
    """(1) every edge / node / coordinate MLP Linear x30: pre-activations leave the fp16 range - the check MUST fail on range (and the
    deviations show it); (2) a model that is fine except for ONE huge bias in layer 2's edge_mlp.0 (A_i = Wa h_i + b1 stored as fp16):
    saturation is counted in the 16-bit pass itself."""
    from dfmdock_amd import engine
    from dfmdock_amd.weights import HParams, make_random_weights, pack_blob
    engine.set_device(0)
    cx = complex_for("fwd_syn_64_48_p0")
    w = make_random_weights(3)
    for name in w:
        if any(s in name for s in ("edge_mlp.0.", "edge_mlp.2.", "node_mlp.0.", "node_mlp.3.", "coord_mlp.0.")):
            w[name] = (w[name] * np.float32(30.0)).astype(np.float32)
    m = engine.Model(pack_blob(w))
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.selfcheck(n_eval=2, seed=1)
    line = engine.format_selfcheck(r, "x30")
    assert not r["ok"] and not r["range_ok"] and r["headroom"] < 1 and "FAILED" in line, line
    gx.close()
    m.close()
    w = make_random_weights(0)
    b = w["network.EGNN_2.egcl.edge_mlp.0.bias"].copy()
    b[17] = 9.0e4
    w["network.EGNN_2.egcl.edge_mlp.0.bias"] = b
    m = engine.Model(pack_blob(w))
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.selfcheck(n_eval=2, seed=1)
    assert not r["range_ok"] and r["saturated"] > 0 and r["max_A"][2] > 6.0e4 and max(r["max_A"][:2]) < 6.0e4, engine.format_selfcheck(r, "bias")
    rf = gx.selfcheck(n_eval=2, seed=1, precision="f16")      # fp32 A_i: the same bias is no range problem for A, but the SiLU output it feeds is
    assert rf["max_pre"][2] > 6.0e4 and not rf["range_ok"]
    gx.close()
    m.close()


def test_score_deviation_is_bounded_by_force_deviation_over_cancellation():
    """the scores are unit vectors of the pooled force / torque times a learned scale, so their error is the
    force error over a cancellation ratio and nothing else bounds it.  Pinned on tools/stress_sizes.py's population (random sizes,
    perturbed poses): |d tr_score| <= 2 score_bound[0], |d rot_score| <= 2 score_bound[1] wherever the force itself is within its gate,
    and the bound is not vacuous - on well-conditioned poses (cancellation ratio > 0.2) both scores sit inside the plain 1e-2 gate."""
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob
    engine.set_device(0)
    m = engine.Model(pack_blob(make_random_weights(0)))
    rng = np.random.default_rng(0)
    n_ill = 0
    for it in range(20):
        R, L = int(rng.integers(3, 300)), int(rng.integers(2, 200))
        cx = make_complex(R, L, seed=100 + it)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        for rep in range(2):
            gx.set_pose(lig_pos=cx["lig_pos"] + rng.normal(0, 1.0, 3).astype(np.float32))
            r = gx.selfcheck(n_eval=2, seed=it * 7 + rep)
            line = engine.format_selfcheck(r, f"{R}+{L}")
            assert r["range_ok"] and r["dev_f"] < 1e-2, line
            assert r["dev_tr_score"] <= max(2 * r["score_bound"][0], 1e-4), line
            assert r["dev_rot_score"] <= max(2 * r["score_bound"][1], 1e-4), line
            assert r["dev_ok"], line
            if min(r["cancel_ratio"]) > 0.2:
                assert r["dev_tr_score"] < 1e-2 and r["dev_rot_score"] < 1e-2, line
            else:
                n_ill += 1
        gx.close()
    m.close()
    print("ill-conditioned poses met:", n_ill)


def test_drivers_selfcheck_and_fall_back_to_the_fp32_engine(tmp_path, capsys):
    from dfmdock_amd import driver, engine
    from dfmdock_amd.weights import make_random_weights, pack_blob
    engine.set_device(0)
    cx = dict(complex_for("fwd_syn_24_16"), id="syn")
    good = engine.Model(pack_blob(make_random_weights(0)))
    checks = []
    rows, _ = driver.run_set(good, [cx], num_samples=3, num_steps=3, seed=1, checks_out=checks)
    assert len(rows) == 3 and checks[0]["precision"] == "mfma16" and checks[0]["selfcheck"]["ok"]
    assert "selfcheck syn" in capsys.readouterr().err
    w = make_random_weights(0)
    b = w["network.EGNN_1.egcl.edge_mlp.0.bias"].copy()
    b[3] = -9.0e4
    w["network.EGNN_1.egcl.edge_mlp.0.bias"] = b
    bad = engine.Model(pack_blob(w))
    checks = []
    rows, _ = driver.run_set(bad, [cx], num_samples=3, num_steps=3, seed=1, checks_out=checks)
    err = capsys.readouterr().err
    assert checks[0]["precision"] == "fp32" and not checks[0]["selfcheck"]["ok"] and "FAILED" in err and "fp32 engine" in err
    ref_rows, _ = driver.run_set(bad, [cx], num_samples=3, num_steps=3, seed=1, precision="fp32")
    assert [r["energy"] for r in rows] == [r["energy"] for r in ref_rows]      # it really ran the fp32 engine
    with pytest.raises(RuntimeError):
        driver.run_set(bad, [cx], num_samples=1, num_steps=3, on_selfcheck_fail="raise")
    good.close()
    bad.close()


def test_score_model_selfcheck(blob):
    from dfmdock_amd.score_model import Score_Model
    cx = complex_for("fwd_syn_24_16")
    sm = Score_Model(blob)
    r = sm.selfcheck(cx)
    assert r["ok"] and r["precision"] == "mfma16"

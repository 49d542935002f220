"""GPU parity on further weight draws: two more seeds and one 3x-scaled draw (dfmdock_amd/weights.py:
WEIGHT_DRAWS), both model families, all three engines, through the C ABI against outputs of the REFERENCE on those weights
(tests/golden/make_golden_draws.py; reference src/models/score_net_mlsb.py:343-425, src/models/egnn_net.py:408-505).

Gates are SURVEY 8(d)'s, unchanged: fp32 <= 1e-4 rel (L-inf / |.|-inf) on tr_score / rot_score / f, 1e-4 abs on energy;
16-bit (fp16 MFMA operands) <= 1e-2 on scores and f, 3e-2 on energy; the fp32-A_i variant (f16) the same.  40-step rollouts with every draw replayed:
ligand CA-RMSD <= 0.05 A over the first 5 steps and 0.5 A over all 40 (fp32), 0.5 A over all 40 steps (16-bit engines).
tools/tol_report.py prints the per-draw worst table (profiles/r03_tol_report.txt).
"""
import numpy as np
import pytest

from conftest import DRAWS, DRAW_CASES, complex_for, draw_blob, draw_golden, draw_hparams

pytestmark = pytest.mark.gpu

# (f, tr_score, rot_score, energy)
# "mfma16" = the 16-bit MFMA engine as shipped (DFM_F_MFMA16: fp16 operands in every layer, fp16 A_i, three-term node GEMMs): measured
# worst over the four draws x two families 5.5e-3 / 3.2e-3 / 3.3e-3 / 2.4e-3 (profiles/r03_tol_report.txt).  "f16" = fp32 A_i:
# 5.7e-3 / 3.4e-3 / 1.6e-3 / 2.8e-3 - its r02 gates of 3e-3 / 5e-3 were read off ONE draw and are replaced by SURVEY's 16-bit gates.
TOL = {"fp32": (1e-4, 1e-4, 1e-4, 1e-4), "mfma16": (1e-2, 1e-2, 1e-2, 3e-2), "f16": (8e-3, 8e-3, 8e-3, 1e-2)}      # measured worst 5.7e-3 / 3.4e-3 / 1.6e-3 / 2.8e-3
KW = {"fp32": {}, "mfma16": dict(mfma16=True), "f16": dict(f16=True)}


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


_models = {}


def gpu_model(family, draw):
    from dfmdock_amd import engine
    if (family, draw) not in _models:
        engine.set_device(0)
        _models[(family, draw)] = engine.Model(draw_blob(family, draw), draw_hparams(family))
    return _models[(family, draw)]


@pytest.mark.parametrize("draw", DRAWS)
@pytest.mark.parametrize("family", [0, 1])
@pytest.mark.parametrize("case_i", range(11))
def test_score_on_other_weight_draws(case_i, family, draw):
    from dfmdock_amd import engine
    if case_i >= len(DRAW_CASES[family]):
        pytest.skip("no such case in this family")
    case = DRAW_CASES[family][case_i]
    g = draw_golden(family, draw, case)
    cx = complex_for(case)
    gx = engine.Complex(gpu_model(family, draw), cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    try:
        for prec, (tf, ttr, trot, te) in TOL.items():
            r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, **KW[prec])
            msg = f"{case} family {family} draw {draw} {prec}"
            assert rel_inf(r["f"][0], g["f"]) < tf, msg
            assert rel_inf(r["tr_score"][0], g["tr_score"]) < ttr, msg
            assert rel_inf(r["rot_score"][0], g["rot_score"]) < trot, msg
            escale = 1.0 if prec == "fp32" else max(abs(float(g["energy"])), 0.1)
            assert abs(float(r["energy"][0]) - float(g["energy"])) < te * escale, msg
            assert int(r["num_clashes"][0]) == int(g["num_clashes"]), msg
            if family:
                c = float(g["confidence_logits"])
                assert abs(float(r["confidence"][0]) - c) < te * (1.0 if prec == "fp32" else max(abs(c), 0.1)), msg
    finally:
        gx.close()


@pytest.mark.parametrize("which,steps", [("rollout", 40), ("rollout7", 6)])
@pytest.mark.parametrize("prec", ["fp32", "mfma16", "f16"])
@pytest.mark.parametrize("draw", DRAWS)
@pytest.mark.parametrize("family", [0, 1])
def test_rollout_on_other_weight_draws(family, draw, prec, which, steps):
    """40 steps on syn_24_16 and 6 steps on the DB5 pair 7CEI (87 + 127 residues, ESM features) per draw, every draw replayed."""
    from dfmdock_amd import engine
    g = draw_golden(family, draw, which)
    cx = complex_for("7CEI" if which == "rollout7" else "syn_24_16")
    gx = engine.Complex(gpu_model(family, draw), cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    inj = dict(R0=g["R0"].astype(np.float32), tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    r = gx.sample(B=1, num_steps=steps, inject=inj, trace=True, **KW[prec])
    rmsd = np.sqrt(((r["trace_pose"][0][:, :, 1, :] - g["poses"][:, :, 1, :]) ** 2).sum(-1).mean(-1))
    assert rmsd[:5].max() < (0.05 if prec == "fp32" else 0.5), rmsd[:5]
    assert rmsd.max() < 0.5, rmsd.max()
    gx.close()


@pytest.mark.parametrize("draw", ["s0", "s1", "s2", "x3"])
@pytest.mark.parametrize("family", [0, 1])
def test_bf16_operand_plan_is_opt_in_and_within_its_stated_bound(family, draw):
    """DFM_F_BF16_OPS (bf16 MFMA operands in layers 0..depth-2, the plan of rounds 1-2) is no longer the default: it does NOT meet
    SURVEY's 1e-2 gate on every draw (second family, seed 1: 1.5e-2 on tr_score; 3x-scaled draw: 1.5e-2 on f, 1.4e-2 on rot_score).
    What include/dfmdock_amd.h states for it - deviations up to 1.5e-2, tested at 2e-2 (3e-2 energy) - is checked here."""
    from dfmdock_amd import engine
    from conftest import load_golden
    from dfmdock_amd.weights import make_random_weights, pack_blob
    hp = draw_hparams(family)
    m = gpu_model(family, draw) if draw != "s0" else engine.Model(pack_blob(make_random_weights(0, hp), hp), hp)
    for case in DRAW_CASES[family]:
        if draw == "s0":
            if family == 1 and case.startswith("fwd_"):
                continue
            g = load_golden(case + ".npz")
        else:
            g = draw_golden(family, draw, case)
        cx = complex_for(case)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, mfma16=True, bf16_ops=True)
        d = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, mfma16=True)
        gx.close()
        assert rel_inf(r["f"][0], g["f"]) < 2e-2 and rel_inf(r["tr_score"][0], np.asarray(g["tr_score"]).reshape(3)) < 2e-2, case
        assert rel_inf(r["rot_score"][0], np.asarray(g["rot_score"]).reshape(3)) < 2e-2, case
        assert abs(float(r["energy"][0]) - float(g["energy"])) < 3e-2 * max(abs(float(g["energy"])), 0.1), case
        assert (r["f"] != d["f"]).any(), "the flag selects a different kernel"

"""The headline call itself against the oracle: what bench.py times is

    dfm_sample(B = 256, 300+300 residues, DFM_F_MFMA16, layer 0 through the per-complex message table, the last layer of the
               40 step evaluations over the ligand nodes only)

and every piece of that composition has its own parity test - this file checks the COMPOSITION, at the batch size and shape of
the bench line: every random draw of every trajectory is injected (initial rotation, the N(0,30^2) draw, per-step z, the edge
lists of every evaluation - drawn by the engine itself on near-native poses, one graph per trajectory and evaluation), four
spread-out trajectories are replayed step by step through the CPU oracle (which is pinned to the reference,
tests/test_oracle_golden.py) and compared at SURVEY 8(d)'s rollout gates: CA-RMSD <= 0.5 A for the 16-bit engine, <= 0.05 A for
the fp32 engine, final energies at the evaluation gates.  The profile counters prove which path ran: layer 0 through the table
in all S + 1 evaluations, S ligand-only launches of the last layer.  The same for C5 (1000+1000, B = 32).

Reference: src/inference_base.py:416-466 (the step loop), src/models/score_net_mlsb.py:343-425 (the evaluation).

Two step sizes: eps = 1e-3 (the call's default: five steps of dt = 0.2 - with the build's seeded weights the ligand leaves the
receptor after the first step, so later evaluations see inter-chain edges only through the sampled slots) and eps = 0.999 (five
steps of dt = 2.5e-4 at t ~ 1: the ligand stays at the interface, the energy head is live at the end and every evaluation
re-bins the same contacts from a slightly different pose - the regime where a table hit / miss decision matters).
"""
import numpy as np
import pytest

from conftest import complex_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(blob):
    from dfmdock_amd import engine
    engine.set_device(0)
    m = engine.Model(blob)
    yield m
    m.close()


def _near_native_injection(gx, cx, B, S, rng, graph_seed):
    """Draws of B trajectories: identity rotation, an N(0,30^2) 'draw' that leaves the ligand within a few A of its native place
    (randomize_pose adds c1 - c2 to it, inference_base.py:330-333), N(0,1) z, and per (trajectory, evaluation) an edge list the
    engine drew itself on a rigidly jittered native pose (kNN slots of the native neighbourhood, 40 sampled slots)."""
    c1, c2 = cx["rec_pos"][:, 1].mean(0), cx["lig_pos"][:, 1].mean(0)
    inj = dict(R0=np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (B, 1)),
               tr_draw=(c2 - c1)[None].astype(np.float32) + 1.5 * rng.standard_normal((B, 3)).astype(np.float32),
               z_rot=rng.standard_normal((B, S, 3)).astype(np.float32), z_tr=rng.standard_normal((B, S, 3)).astype(np.float32))
    edges = np.empty((B, S + 1, gx.N, gx.K), np.int32)
    for s in range(S + 1):
        poses = (cx["lig_pos"][None] + rng.standard_normal((B, 1, 1, 3)).astype(np.float32)).astype(np.float32)
        edges[:, s] = gx.score(poses, 0.5, seed=graph_seed + s, mfma16=True, energy=False, return_edges=True)["edges"]
    inj["edges"] = edges
    return inj


def _replay(o, inj, b, S, eps):
    one = {k: (v[b].astype(np.float64) if k == "R0" else v[b]) for k, v in inj.items()}
    return o.sample(num_steps=S, eps=eps, inject=one, trace=True)


def _ca_rmsd(a, b):
    return np.sqrt(((a[:, :, 1] - b[:, :, 1]) ** 2).sum(-1).mean(-1))


@pytest.mark.parametrize("eps", [1e-3, 0.999])
def test_c3_headline_call_vs_oracle(eps, model, blob):
    from dfmdock_amd import engine
    from oracle import oracle as ora
    cx = complex_for("c3_300_300")
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    B, S = 256, 5
    inj = _near_native_injection(gx, cx, B, S, np.random.default_rng(11), graph_seed=700)
    assert (inj["edges"][0, 0] != inj["edges"][1, 0]).any() and (inj["edges"][0, 0] != inj["edges"][0, 1]).any()
    # the bench's call: mfma16, default flags (table on, ligand-only last layer in the step evaluations), B = 256
    r16 = gx.sample(B=B, num_steps=S, eps=eps, inject=inj, trace=True, step_energy=False, profile=True, mfma16=True)
    p = gx.profile()
    assert p["l0_evals"] == S + 1, p                   # layer 0 through the message table in every evaluation
    assert p["edge_lig_launches"] == S, p              # the step evaluations' last layer ran over the ligand nodes only
    assert p["edge_kernel_launches"] == 5 * (S + 1), p  # layers 1-5 of S + 1 evaluations; layer 0 is not a message launch
    assert 0 < p["l0_miss_rows"] < 0.5 * p["l0_edges"]
    # the message kernel's own clock stamps (s_memtime against the 100 MHz s_memrealtime, first wave of workgroup 0, summed over the launches):
    # a plausible shader clock, and the stamps cover the launches' time (one wave's lifetime <= its launch)
    assert 500.0 < p["edge_sclk_mhz"] < 2600.0, p["edge_sclk_mhz"]
    assert 0.5 * p["edge_kernel_ms"] < p["edge_ref_ticks"] / 1e5 <= 1.02 * p["edge_kernel_ms"], (p["edge_ref_ticks"], p["edge_kernel_ms"])
    r32 = gx.sample(B=B, num_steps=S, eps=eps, inject=inj, trace=True, step_energy=False, profile=True)      # fp32 engine, its own table
    assert gx.profile()["l0_evals"] == S + 1
    assert np.isfinite(r16["lig_pos"]).all() and np.isfinite(r32["lig_pos"]).all()
    o = ora.Oracle(blob, cx)
    live = 0
    for b in (0, 85, 170, 255):
        ob = _replay(o, inj, b, S, eps)
        rm32, rm16 = _ca_rmsd(r32["trace_pose"][b], ob["trace_pose"]), _ca_rmsd(r16["trace_pose"][b], ob["trace_pose"])
        assert rm32.max() < 0.05, (eps, b, rm32)
        assert rm16.max() < 0.5, (eps, b, rm16)
        e = float(ob["energy"])
        live += abs(e) > 1e-3
        assert abs(float(r32["energy"][b]) - e) < 1e-3 * max(1.0, abs(e)), (eps, b, "fp32 energy")
        assert abs(float(r16["energy"][b]) - e) < 3e-2 * max(abs(e), 0.1) + (0.05 if rm16.max() > 1e-2 else 0.0), (eps, b, "16-bit energy")
        assert int(r32["num_clashes"][b]) == int(ob["num_clashes"])
        # first evaluation: same pose on both sides -> the evaluation gates themselves
        for r, tol in ((r32, 1e-4), (r16, 1e-2)):
            for lo in (0, 3):
                ref = ob["trace_scores"][0, lo:lo + 3]
                assert np.abs(r["trace_scores"][b][0, lo:lo + 3] - ref).max() < tol * np.abs(ref).max(), (eps, b, lo, tol)
    if eps > 0.5:
        assert live == 4      # small steps: the ligand is still at the interface, the energy head was checked on live values
    # batch invariance at the headline batch: trajectory 170 alone equals row 170 of the batch bit for bit
    one = {k: np.ascontiguousarray(v[170:171]) for k, v in inj.items()}
    r1 = gx.sample(B=1, num_steps=S, eps=eps, inject=one, trace=True, step_energy=False, mfma16=True)
    for k in ("lig_pos", "trace_pose", "energy", "tr_update", "rot_update"):
        np.testing.assert_array_equal(r1[k][0], r16[k][170], err_msg=k)
    gx.close()


def test_c5_headline_call_vs_oracle(model, blob):
    """C5 (1000+1000, B = 32): two trajectories replayed through the oracle over two steps."""
    from dfmdock_amd import engine
    from oracle import oracle as ora
    cx = complex_for("c5_1000_1000")
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    B, S, eps = 32, 2, 0.999
    inj = _near_native_injection(gx, cx, B, S, np.random.default_rng(12), graph_seed=800)
    r16 = gx.sample(B=B, num_steps=S, eps=eps, inject=inj, trace=True, step_energy=False, profile=True, mfma16=True)
    p = gx.profile()
    assert p["l0_evals"] == S + 1 and p["edge_lig_launches"] == S and p["edge_kernel_launches"] == 5 * (S + 1), p
    r32 = gx.sample(B=B, num_steps=S, eps=eps, inject=inj, trace=True, step_energy=False)
    o = ora.Oracle(blob, cx)
    for b in (3, 29):
        ob = _replay(o, inj, b, S, eps)
        rm32, rm16 = _ca_rmsd(r32["trace_pose"][b], ob["trace_pose"]), _ca_rmsd(r16["trace_pose"][b], ob["trace_pose"])
        assert rm32.max() < 0.05 and rm16.max() < 0.5, (b, rm32, rm16)
        e = float(ob["energy"])
        assert abs(e) > 1e-3
        assert abs(float(r32["energy"][b]) - e) < 1e-3 * max(1.0, abs(e))
        assert abs(float(r16["energy"][b]) - e) < 3e-2 * max(abs(e), 0.1)
        assert int(r32["num_clashes"][b]) == int(ob["num_clashes"])
    gx.close()

"""dfm_sample replays ONE captured step (score evaluation + heads + Euler-Maruyama update) as a hipGraph (reference loop:
src/inference_base.py:416-466).  The captured launches are the plain path's launches with the per-step / per-call scalars read from
device memory, so the trajectories must be bitwise those of the plain path (the default) - for every engine, every sampler variant,
across calls that re-use the executable graph with another seed / time grid, and after the buffers it points into have moved.
"""
import numpy as np
import pytest

from conftest import pair_hparams

pytestmark = pytest.mark.gpu

KEYS = ("lig_pos", "energy", "rot_update", "tr_update", "final_scores", "num_clashes")


def same(a, b):
    return all((a[k] == b[k]).all() for k in KEYS)


@pytest.fixture(scope="module")
def gx(blob):
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    engine.set_device(0)
    m = engine.Model(blob)
    cx = make_complex(70, 50, seed=9)
    g = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    yield g
    g.close(); m.close()


@pytest.mark.parametrize("kw", [dict(mfma16=True), dict(), dict(f16=True), dict(mfma16=True, l0_table=False), dict(mfma16=True, ode=True),
                                dict(mfma16=True, noise_annealing=True), dict(mfma16=True, use_clash_force=True)],
                         ids=["mfma16", "fp32", "f16", "mfma16-direct-l0", "ode", "annealing", "clash-force"])
def test_graph_equals_plain_launches(kw, gx):
    for B, steps, seed in ((1, 6, 3), (7, 5, 4)):
        a = gx.sample(B=B, num_steps=steps, seed=seed, graph=True, **kw)
        b = gx.sample(B=B, num_steps=steps, seed=seed, **kw)
        assert same(a, b), (kw, B)
        assert np.isfinite(a["lig_pos"]).all()


def test_graph_is_reused_across_seeds_and_time_grids(gx):
    """Same (B, engine): the second and third call replay the graph of the first with another seed, another number of steps and other
    noise scales - all of which live in device memory, not in the captured arguments."""
    first = gx.sample(B=3, num_steps=4, seed=1, mfma16=True, graph=True)
    for kw in (dict(num_steps=4, seed=2), dict(num_steps=7, seed=2, eps=1e-2), dict(num_steps=4, seed=1, tr_noise_scale=0.1, rot_noise_scale=0.9),
               dict(num_steps=4, seed=1)):
        a = gx.sample(B=3, mfma16=True, graph=True, **kw)
        b = gx.sample(B=3, mfma16=True, **kw)
        assert same(a, b), kw
    assert same(a, first)                       # (the last setting is the first call's)
    assert not same(gx.sample(B=3, num_steps=4, seed=2, mfma16=True, graph=True), first)


def test_graph_survives_moved_buffers(gx):
    """A larger batch re-allocates the workspace, set_pose rebuilds the message table: the graph is captured again."""
    a1 = gx.sample(B=2, num_steps=4, seed=5, mfma16=True, graph=True)
    gx.sample(B=9, num_steps=3, seed=5, mfma16=True, graph=True)       # workspace grows
    assert same(gx.sample(B=2, num_steps=4, seed=5, mfma16=True, graph=True), a1)
    gx.set_pose(None, gx.lig_pos0)                                     # table invalidated (same geometry: same numbers)
    assert same(gx.sample(B=2, num_steps=4, seed=5, mfma16=True, graph=True), a1)
    assert same(gx.sample(B=2, num_steps=4, seed=5, mfma16=True), a1)


def test_graph_pair_family(blob_pair):
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    m = engine.Model(blob_pair, pair_hparams())
    cx = make_complex(40, 30, seed=2)
    g = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    for kw in (dict(mfma16=True), dict()):
        assert same(g.sample(B=4, num_steps=5, seed=8, graph=True, **kw), g.sample(B=4, num_steps=5, seed=8, **kw))
    g.close(); m.close()


DYN_CASES = [(24, 150, 110, 5), (9, 301, 160, 3), (17, 129, 131, 3), (8, 380, 213, 2)]      # (B, R, L, steps): >= 2 node tasks per wave, ragged sizes


def _dyn_case(g, B, steps):
    return g.sample(B=B, num_steps=steps, seed=21, mfma16=True)


def test_graph_and_dynamic_tasks_on_a_large_launch(blob):
    """r06: the message kernel's waves take their tasks from per-workgroup counters that reset themselves at the end of every launch
    (no memset node, no host-side state), so a REPLAYED step graph and plain launches run the same kernel the same way.  At sizes where
    dynamic tasks engage (B >= 8 and at least two node tasks per wave; ragged B / R / L included): graph == plain launches == the fixed
    stride of r01-r05 (DFM_EDGE_DYNAMIC=0 in a child process), bit for bit - a node's rows do not depend on which wave computes them
    (reference: the trajectories of Euler_Maruyama_sampler are independent, src/inference_base.py:390-468)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    engine.set_device(0)
    m = engine.Model(blob)
    mine = []
    for k, (B, R, L, steps) in enumerate(DYN_CASES):
        cx = make_complex(R, L, seed=12 + k)
        g = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        a = g.sample(B=B, num_steps=steps, seed=21, mfma16=True, graph=True)
        b = _dyn_case(g, B, steps)
        c = g.sample(B=B, num_steps=steps, seed=21, mfma16=True, graph=True)      # the replayed graph again: the counters were left at zero
        assert same(a, b) and same(a, c) and np.isfinite(a["lig_pos"]).all(), (B, R, L)
        mine.append(b)
        g.close()
    m.close()
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from dfmdock_amd import engine\nfrom dfmdock_amd.synthetic import make_complex\n"
            "from dfmdock_amd.weights import make_random_weights, pack_blob\n"
            "engine.set_device(0); m = engine.Model(pack_blob(make_random_weights(0))); out = []\n"
            "for k, (B, R, L, steps) in enumerate(%r):\n"
            "    cx = make_complex(R, L, seed=12 + k)\n"
            "    g = engine.Complex(m, cx['rec_x'], cx['lig_x'], cx['rec_pos'], cx['lig_pos'])\n"
            "    r = g.sample(B=B, num_steps=steps, seed=21, mfma16=True)\n"
            "    out.append({'lig': r['lig_pos'].astype(np.float64).tolist(), 'e': r['energy'].astype(np.float64).tolist()}); g.close()\n"
            "print(json.dumps(out))\n" % (ROOT, os.path.join(ROOT, "tests"), DYN_CASES))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DFM_EDGE_DYNAMIC="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    st = json.loads(p.stdout.decode().strip().splitlines()[-1])
    for k, r in enumerate(mine):
        assert np.array_equal(np.asarray(st[k]["lig"], np.float32), r["lig_pos"]) and np.array_equal(np.asarray(st[k]["e"], np.float32), r["energy"]), DYN_CASES[k]

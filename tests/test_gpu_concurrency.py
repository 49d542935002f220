"""Two complex handles driven at once from two host threads must not change each other's numbers (include/dfmdock_amd.h: "one stream
per handle"; driver.run_set relies on it).  r05 found a silent miscompute here - waves of k_edge_feat<0> computing wrong theta bins
while another handle's 160 KiB message-kernel workgroups were resident (profiles/r05_concurrency.txt) - and contained it (token LDS
allocation on LDS-free kernels, no SLP vectorisation); r06 traced it to one packed-fp32 instruction form (op_sel = [0,1]) that
miscomputes under that co-residency even in a stand-alone kernel (profiles/r06_concurrency.txt) and removed the form from the library
(static audit: tests/test_abi_cpu.py).  This file is the victim x aggressor matrix of that investigation as a test of
the SHIPPED build: every cell must show 0 deviating calls.  What a victim call covers, kernel by kernel (reference path in brackets):

  sample mfma16 direct : k_prep_pose, k_knn_sample, k_edge_feat<0>, k_gemm_split, k_edge_msg<1,1,0>, k_edge_coord, k_heads
                         (Euler_Maruyama_sampler, src/inference_base.py:390-468; get_coords6d, src/utils/coords6d.py:25-43)
  sample mfma16 table  : + k_edge_feat<1>, k_edge_msg<1,1,1> (row list), k_l0_gather
  sample fp32          : k_edge_f32m, k_gemm_f32v, k_gn_stats, k_l0_gather32
  score mfma16 debug   : the same evaluation with its intermediate taps (edges <- k_knn_sample, edge_codes <- k_edge_feat,
                         h_first / h_last <- GEMMs + message kernels, f <- coordinate kernel + heads): says WHICH stage deviates first
  pair-family sample   : k_pair_head_m, k_pair_finish_s (src/models/egnn_net.py:430-482)

tools/concurrency_probe*.py (r05) are folded into this file; tools/concurrency_repro.py + tools/asm_variant.py reproduce the effect on purpose
(DFM_TOKEN_LDS=0 with an SLP-vectorised kernels_geom.hip), tools/pkmul_probe.py outside the engine.
"""
import threading

import numpy as np
import pytest

from conftest import pair_hparams

pytestmark = pytest.mark.gpu

CALLS = int(__import__("os").environ.get("DFM_MATRIX_CALLS", "8"))      # victim calls per cell; DFM_MATRIX_CALLS=48: the soak of profiles/r06_concurrency.txt (r05: the un-fenced build deviated in 11-12 of 12)


@pytest.fixture(scope="module")
def handles(blob, blob_pair):
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    engine.set_device(0)
    model = engine.Model(blob)
    model2 = engine.Model(blob_pair, pair_hparams())
    ca, cb = make_complex(223, 172, seed=5), make_complex(120, 90, seed=6)
    A = engine.Complex(model, ca["rec_x"], ca["lig_x"], ca["rec_pos"], ca["lig_pos"])
    A2 = engine.Complex(model2, ca["rec_x"], ca["lig_x"], ca["rec_pos"], ca["lig_pos"])
    Bc = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])
    yield dict(A=A, A2=A2, B=Bc, ca=ca, cb=cb, model=model)
    for h in (A, A2, Bc):
        h.close()


def _victims(h):
    A, A2, ca = h["A"], h["A2"], h["ca"]
    rng = np.random.default_rng(0)
    poses = (ca["lig_pos"][None] + rng.standard_normal((24, 1, 1, 3)).astype(np.float32) * 3).astype(np.float32)
    return {
        "sample mfma16 direct": lambda: A.sample(B=24, num_steps=5, seed=11, mfma16=True, l0_table=False),
        "sample mfma16 table": lambda: A.sample(B=24, num_steps=5, seed=11, mfma16=True, l0_table=True),
        "sample fp32": lambda: A.sample(B=6, num_steps=2, seed=11),
        "score mfma16 debug": lambda: A.score(poses, 0.5, seed=3, energy=True, debug=True, mfma16=True),
        "pair family sample": lambda: A2.sample(B=16, num_steps=4, seed=11, mfma16=True),
    }


def _aggressors(h):
    from dfmdock_amd import engine
    Bc, cb, model = h["B"], h["cb"], h["model"]
    pb = np.repeat(cb["lig_pos"][None], 40, 0)

    def create_close():
        g = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])
        g.sample(B=8, num_steps=2, seed=1, mfma16=True)      # builds the layer-0 table on the new handle's stream
        g.close()

    return {
        "sample mfma16 direct": lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False),
        "sample mfma16 table": lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=True),
        "sample fp32": lambda: Bc.sample(B=8, num_steps=2, seed=2),
        "score mfma16": lambda: Bc.score(pb, 0.5, seed=2, mfma16=True, energy=True),
        "create + table + close": create_close,
        "selfcheck": lambda: Bc.selfcheck(precision="mfma16", seed=0),
    }


def _first_difference(a, b):
    """Names of the outputs that differ, in pipeline order."""
    order = ["edges", "edge_codes", "h_first", "h_last", "f", "tr_score", "rot_score", "energy", "num_clashes",
             "lig_pos", "rot_update", "tr_update"]
    keys = [k for k in order if k in a] + sorted(k for k in a if k not in order)
    return [k for k in keys if isinstance(a[k], np.ndarray) and not np.array_equal(a[k], b[k], equal_nan=True)]


def test_victim_aggressor_matrix_is_clean(handles):
    victims, aggressors = _victims(handles), _aggressors(handles)
    solo = {name: fn() for name, fn in victims.items()}
    for name, fn in victims.items():      # the solo call is reproducible to begin with
        assert not _first_difference(solo[name], fn()), name
    table, bad = [], []
    for an, afn in aggressors.items():
        stop, err = [False], []

        def loop():
            try:
                while not stop[0]:
                    afn()
            except BaseException as e:      # an aggressor failure is a test failure, not a hang
                err.append(e)

        t = threading.Thread(target=loop)
        t.start()
        try:
            for vn, vfn in victims.items():
                deviating, fields = 0, set()
                for _ in range(CALLS):
                    d = _first_difference(solo[vn], vfn())
                    if d:
                        deviating += 1
                        fields.add(d[0])
                table.append((vn, an, deviating))
                if deviating:
                    bad.append(f"victim '{vn}' next to '{an}': {deviating} of {CALLS} calls deviate, first in {sorted(fields)}")
        finally:
            stop[0] = True
            t.join()
        assert not err, (an, err)
    print("\nvictim x aggressor (deviating calls of %d):" % CALLS)
    for vn, an, d in table:
        print(f"  {vn:24s} | {an:26s} | {d}")
    assert not bad, "\n".join(bad)


def test_two_samplers_on_two_threads_equal_solo(handles):
    """Both handles sampling the headline way (table on, ligand-only last layer) at once, each checked against its own solo result."""
    A, Bc = handles["A"], handles["B"]
    fa = lambda: A.sample(B=40, num_steps=8, seed=5, mfma16=True)
    fb = lambda: Bc.sample(B=40, num_steps=8, seed=6, mfma16=True)
    sa, sb = fa(), fb()
    out = {}

    def run(key, fn, n):
        out[key] = [fn() for _ in range(n)]

    ta, tb = threading.Thread(target=run, args=("a", fa, 6)), threading.Thread(target=run, args=("b", fb, 10))
    ta.start(); tb.start(); ta.join(); tb.join()
    for r in out["a"]:
        assert not _first_difference(sa, r)
    for r in out["b"]:
        assert not _first_difference(sb, r)

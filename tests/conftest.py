import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """The engine's precision plan and any diagnostic switch in force (dfm_config_string): printed on top of every run."""
    try:
        from dfmdock_amd import engine
        return "dfmdock_amd engine: " + engine.config_string()
    except Exception as e:      # library not built: the ABI tests say so
        return f"dfmdock_amd engine: unavailable ({e})"


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The oracle's OpenMP loops stop scaling well before a 128-core host is full (bench.py cpu_baseline: 0.20 trajectories/s at 32
    threads, 0.06 at 128): the checker runs on at most 32 threads."""
    try:
        from oracle import oracle as ora
        L = ora.lib()
        L.ora_set_num_threads(min(32, int(L.ora_num_threads())))
    except Exception:
        pass
    yield


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def blob():
    from dfmdock_amd.weights import make_random_weights, pack_blob
    return pack_blob(make_random_weights(0))


PAIR_HP_KW = dict(family=1, mask_dist=20.0)      # configs/model/DFMDock.yaml (EGNN_Net behind DFMDock.forward)


def pair_hparams(agg_mean=True):
    from dfmdock_amd.weights import HParams
    return HParams(agg_mean=agg_mean, **PAIR_HP_KW)


@pytest.fixture(scope="session")
def blob_pair():
    from dfmdock_amd.weights import make_random_weights, pack_blob
    hp = pair_hparams()
    return pack_blob(make_random_weights(0, hp), hp)


# further weight draws (dfmdock_amd/weights.py: WEIGHT_DRAWS; fixtures tests/golden/draws_f<family>_<draw>.npz produced by
# tests/golden/make_golden_draws.py running the reference on the seed-0 goldens' poses and edge lists)
DRAWS = ("s1", "s2", "x3")
DRAW_CASES = {0: ["fwd_syn_9_7", "fwd_syn_24_16", "fwd_syn_64_48_p0", "fwd_syn_64_48_p1", "fwd_syn_64_48_p2", "fwd_7CEI_p0",
                  "fwd_7CEI_p1", "fwd_7CEI_p2", "fwd_7CEI_p3", "fwd_c3_300_300", "fwd_db5_1AVX"],
              1: ["fwd2_syn_9_7", "fwd2_syn_24_16", "fwd2_syn_64_48_p0", "fwd2_syn_64_48_p1", "fwd2_syn_64_48_p2",
                  "fwd2_7CEI_p0", "fwd2_7CEI_p1", "fwd2_7CEI_p2", "fwd_c3_300_300", "fwd_db5_1AVX"]}
_draw_blobs = {}


def draw_hparams(family):
    from dfmdock_amd.weights import HParams
    return pair_hparams() if family else HParams()


def draw_blob(family, draw):
    from dfmdock_amd.weights import make_weight_draw, pack_blob
    if (family, draw) not in _draw_blobs:
        hp = draw_hparams(family)
        _draw_blobs[(family, draw)] = pack_blob(make_weight_draw(draw, hp), hp)
    return _draw_blobs[(family, draw)]


def draw_golden(family, draw, case):
    """Outputs of the reference on weight draw `draw` for `case` (inputs = the seed-0 golden's pose, t and edge list)."""
    d = load_golden(f"draws_f{family}_{draw}.npz")
    g = load_golden(("rollout2_" if family else "rollout_") + ("7CEI" if case == "rollout7" else "syn_24_16") + ".npz") \
        if case in ("rollout", "rollout7") else \
        {k: v for k, v in load_golden(case + ".npz").items() if k in ("lig_pos", "t", "edges")}
    g = dict(g)
    g.update({k.split("/", 1)[1]: v for k, v in d.items() if k.startswith(case + "/")})
    return g


_db5 = None


def db5_ids():
    global _db5
    if _db5 is None:
        _db5 = load_golden("db5_backbones.npz")
    return [str(x) for x in _db5["ids"]]


def db5_complex(cid):
    """One of the 24 DB5 test complexes: the reference's backbone + sequence (tests/golden/db5_backbones.npz) with
    node features N(0,1) seeded by the id || one-hot(seq) - the ESM-2 blocks are too large to commit."""
    import zlib
    from dfmdock_amd.synthetic import seq_to_onehot
    db5_ids()
    out = {"id": cid}
    for side in ("rec", "lig"):
        seq = str(_db5[f"{cid}_{side}_seq"])
        rng = np.random.Generator(np.random.PCG64(zlib.crc32(f"{cid}:{side}".encode())))
        out[side + "_x"] = np.concatenate([rng.standard_normal((len(seq), 1280)).astype(np.float32), seq_to_onehot(seq)], 1)
        out[side + "_pos"] = _db5[f"{cid}_{side}_pos"]
        out[side + "_seq"] = seq
    return out


REAL_ESM_IDS = ("1QA9", "1AVX", "1H1V", "7CEI")      # DB5 complexes whose ESM-2 block is committed (fp16)
# the other 20: ESM-2 block committed as int8 + one fp16 scale per residue (tests/golden/make_golden_r06.py; the goldens were generated
# on the DEQUANTISED values, which both sides use)
Q8_ESM_IDS = ("1HCF", "1IRA", "1JIW", "1JPS", "1MLC", "1NW9", "1VFB", "1ZHI", "2A1A", "2A9K", "2AYO", "2SIC", "2SNI", "2VDB", "3SZK", "4POU",
              "5C7X", "5HGG", "5JMO", "6B0S")
_q8 = None


def q8_golden(cid, family=0):
    """The reference's evaluation of `cid` on its dequantised ESM block (fwd_esmq_db5.npz, second model family: fwd2_esmq_db5.npz;
    keys <id>/<name>)."""
    g = load_golden("fwd2_esmq_db5.npz" if family else "fwd_esmq_db5.npz")
    return {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(cid + "/")}


def real_db5_complex(cid):
    """A DB5 test complex with the reference's REAL node features (src/datasets/ppi_dataset.py:249-265: x = cat[ESM-2 block,
    one-hot(seq)]): backbone + sequence from db5_backbones.npz, ESM block from esm_<id>.npz (tests/golden/make_golden_r05.py;
    float16 - the goldens were generated on the rounded values).  7CEI: cx_7CEI.npz (make_golden.py)."""
    from dfmdock_amd.synthetic import seq_to_onehot
    if cid == "7CEI":
        cx = complex_for("7CEI")
        d = load_golden("cx_7CEI.npz")
        return dict(cx, id=cid, rec_seq=str(d["rec_seq"]), lig_seq=str(d["lig_seq"]))
    db5_ids()
    if cid in Q8_ESM_IDS:
        global _q8
        if _q8 is None:
            _q8 = load_golden("esm_db5_q8.npz")
        x = _q8[cid + "_q"].astype(np.float32) * _q8[cid + "_s"].astype(np.float32)[:, None]
        out, R = {"id": cid}, len(str(_db5[f"{cid}_rec_seq"]))
        for side, blk in (("rec", x[:R]), ("lig", x[R:])):
            seq = str(_db5[f"{cid}_{side}_seq"])
            assert blk.shape[0] == len(seq)
            out[side + "_x"] = np.concatenate([blk, seq_to_onehot(seq)], 1)
            out[side + "_pos"] = _db5[f"{cid}_{side}_pos"]
            out[side + "_seq"] = seq
        return out
    e = load_golden(f"esm_{cid}.npz")
    out = {"id": cid}
    for side in ("rec", "lig"):
        seq = str(_db5[f"{cid}_{side}_seq"])
        assert seq == str(e[side + "_seq"])
        out[side + "_x"] = np.concatenate([e[side + "_esm16"].astype(np.float32), seq_to_onehot(seq)], 1)
        out[side + "_pos"] = _db5[f"{cid}_{side}_pos"]
        out[side + "_seq"] = seq
    return out


def complex_for(case):
    """Rebuild the complex a golden file was generated on (tests/golden/make_golden.py)."""
    from dfmdock_amd.synthetic import make_complex, seq_to_onehot
    if "esm_" in case:       # DB5 backbone + the real ESM-2 block (tests/golden/make_golden_r05.py)
        return real_db5_complex(case.split("esm_")[1].split(".")[0])
    if "esmq_" in case:      # ... + the int8-quantised block (tests/golden/make_golden_r06.py)
        return real_db5_complex(case.split("esmq_")[1].split(".")[0])
    if "7CEI" in case:
        d = load_golden("cx_7CEI.npz")
        rx = np.concatenate([d["rec_esm16"].astype(np.float32), seq_to_onehot(str(d["rec_seq"]))], 1)
        lx = np.concatenate([d["lig_esm16"].astype(np.float32), seq_to_onehot(str(d["lig_seq"]))], 1)
        return {"rec_x": rx, "lig_x": lx, "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    if "db5_" in case:       # DB5 backbone + seeded features (tests/golden/make_golden_r02.py: seeded_features)
        return db5_complex(case.split("db5_")[1].split(".")[0])
    table = {"syn_24_16": (24, 16, 5), "syn_9_7": (9, 7, 6), "syn_64_48": (64, 48, 7), "c3_300_300": (300, 300, 1),
             "c5_1000_1000": (1000, 1000, 1)}
    for k, (R, L, seed) in table.items():
        if k in case:
            return make_complex(R, L, seed=seed)
    raise KeyError(case)

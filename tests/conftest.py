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


def complex_for(case):
    """Rebuild the complex a golden file was generated on (tests/golden/make_golden.py)."""
    from dfmdock_amd.synthetic import make_complex, seq_to_onehot
    if "7CEI" in case:
        d = load_golden("cx_7CEI.npz")
        rx = np.concatenate([d["rec_esm16"].astype(np.float32), seq_to_onehot(str(d["rec_seq"]))], 1)
        lx = np.concatenate([d["lig_esm16"].astype(np.float32), seq_to_onehot(str(d["lig_seq"]))], 1)
        return {"rec_x": rx, "lig_x": lx, "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    table = {"syn_24_16": (24, 16, 5), "syn_9_7": (9, 7, 6), "syn_64_48": (64, 48, 7)}
    for k, (R, L, seed) in table.items():
        if k in case:
            return make_complex(R, L, seed=seed)
    raise KeyError(case)

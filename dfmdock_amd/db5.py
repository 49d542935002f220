"""Reader for the reference's ``data/db5_test/*.pt`` complexes without torch_geometric.

Each file is a pickled ``torch_geometric.data.HeteroData`` holding
``receptor/ligand .x [n,1280] f32`` (pre-computed ESM-2 embeddings),
``.pos [n,3,3] f32`` (N, CA, C) and ``.seq`` (reference:
src/datasets/ppi_dataset.py:249-265).  Three attribute-bag stand-ins are enough
to unpickle them.
"""
from __future__ import annotations

import sys
import types

import numpy as np

from .synthetic import seq_to_onehot


class _Bag:
    def __setstate__(self, d):
        self.__dict__.update(d)


def _install_stubs():
    if "torch_geometric.data.hetero_data" in sys.modules:
        return
    names = ["torch_geometric", "torch_geometric.data", "torch_geometric.data.hetero_data",
             "torch_geometric.data.storage"]
    mods = {}
    for n in names:
        mods[n] = sys.modules.get(n) or types.ModuleType(n)
    mods["torch_geometric.data.hetero_data"].HeteroData = type("HeteroData", (_Bag,), {})
    mods["torch_geometric.data.storage"].BaseStorage = type("BaseStorage", (_Bag,), {})
    mods["torch_geometric.data.storage"].NodeStorage = type("NodeStorage", (_Bag,), {})
    for n in names:
        sys.modules.setdefault(n, mods[n])


def load_db5_pt(path: str):
    """-> dict(id, rec_x[R,1301], lig_x[L,1301], rec_pos, lig_pos, rec_seq, lig_seq)."""
    import torch
    _install_stubs()
    data = torch.load(path, weights_only=False, map_location="cpu")
    stores = data.__dict__["_node_store_dict"]
    out = {}
    for key, short in (("receptor", "rec"), ("ligand", "lig")):
        m = stores[key].__dict__["_mapping"]
        x = m["x"].float().numpy()
        seq = m["seq"]
        out[short + "_seq"] = seq
        out[short + "_esm"] = x
        out[short + "_x"] = np.concatenate([x, seq_to_onehot(seq)], axis=1).astype(np.float32)
        out[short + "_pos"] = m["pos"].float().numpy()
    g = data.__dict__["_global_store"].__dict__["_mapping"]
    out["id"] = g.get("name", "")
    return out

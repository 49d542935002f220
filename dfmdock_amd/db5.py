"""Reader for the reference's ``data/db5_test/*.pt`` complexes without torch_geometric.

Each file is a pickled ``torch_geometric.data.HeteroData`` holding
``receptor/ligand .x [n,1280] f32`` (pre-computed ESM-2 embeddings),
``.pos [n,3,3] f32`` (N, CA, C) and ``.seq`` (reference:
src/datasets/ppi_dataset.py:249-265).  Attribute-bag stand-ins, resolved by a private Unpickler (nothing is registered
in sys.modules), are enough to unpickle them.
"""
from __future__ import annotations

import numpy as np

from .synthetic import seq_to_onehot


class _Bag:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, d):
        if isinstance(d, dict):
            self.__dict__.update(d)


def _pickle_module():
    """A pickle module whose Unpickler resolves the torch_geometric container classes (and anything else it does not know)
    to inert attribute bags: nothing is registered in sys.modules - a real torch_geometric, if installed, stays untouched -
    and only the tensor-rebuild helpers of weights._ALLOWED_GLOBALS are ever looked up."""
    import pickle
    import types

    from .weights import _ALLOWED_BUILTINS, _ALLOWED_GLOBALS, _TORCH_DTYPES

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _ALLOWED_GLOBALS or (module == "builtins" and name in _ALLOWED_BUILTINS):
                return super().find_class(module, name)
            if module == "torch" and (name.endswith("Storage") or name in _TORCH_DTYPES):
                return super().find_class(module, name)
            return type(name, (_Bag,), {"__module__": module})

    mod = types.ModuleType("dfm_db5_pickle")
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = "pickle"
    for k in ("dump", "dumps", "loads", "HIGHEST_PROTOCOL", "PickleError", "UnpicklingError", "Pickler"):
        setattr(mod, k, getattr(pickle, k))
    return mod


def load_db5_pt(path: str):
    """-> dict(id, rec_x[R,1301], lig_x[L,1301], rec_pos, lig_pos, rec_seq, lig_seq)."""
    import torch
    data = torch.load(path, weights_only=False, map_location="cpu", pickle_module=_pickle_module())
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

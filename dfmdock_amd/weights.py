"""Parameter inventory of the score network and the blob layout of the C ABI.

The blob handed to ``dfm_model_create`` is the concatenation, in ``PARAM_SPECS``
order, of every tensor of the reference ``Score_Net.state_dict()`` flattened
row-major as float32 (reference: src/models/score_net_mlsb.py:249-341,
src/models/egnn.py:37-93; Lightning checkpoints prefix every key with ``net.``).

``make_random_weights`` is the build's own deterministic generator (numpy
PCG64).  It is NOT the reference initialiser: it draws O(1)-conditioned weights
so that parity tests exercise the non-linearities, the clamp and the GraphNorm
statistics; both the reference (through ``load_state_dict``) and this engine
load the very same numbers, which is all parity needs.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np


@dataclass(frozen=True)
class HParams:
    """configs/model/score_model_mlsb.yaml:3-27 + score_net_mlsb.py:85,33."""
    lm_embed_dim: int = 1301
    positional_embed_dim: int = 66
    spatial_embed_dim: int = 100
    node_dim: int = 256
    edge_dim: int = 128
    inner_dim: int = 128
    depth: int = 6
    knn: int = 20
    n_sample: int = 40
    cut_off: float = 20.0
    mask_dist: float = 22.0
    r3_min_sigma: float = 0.1
    r3_max_sigma: float = 30.0
    so3_min_sigma: float = 0.1
    so3_max_sigma: float = 1.5
    # model family: 0 = Score_Net (score_net_mlsb.py, the network inference_single.py loads),
    #               1 = EGNN_Net behind DFMDock.forward (egnn_net.py:408-505, DFMDock.py:68-75): no coordinate update,
    #                   pair-force / energy / confidence heads on [h_r, h_l, D]; configs/model/DFMDock.yaml
    family: int = 0
    agg_mean: bool = True      # EGNN_Net `agg`: 'mean' (True) or 'sum' pooling of pair energies / forces

    def as_dict(self):
        return asdict(self)


def param_specs(hp: HParams = HParams()):
    """(name, shape) in reference state_dict order."""
    H, He, Hi = hp.node_dim, hp.edge_dim, hp.inner_dim
    s = [
        ("single_embed.weight", (H, hp.lm_embed_dim)),
        ("spatial_embed.weight", (He, hp.spatial_embed_dim)),
        ("positional_embed.weight", (He, hp.positional_embed_dim)),
    ]
    for l in range(hp.depth):
        p = f"network.EGNN_{l}.egcl."
        s += [
            (p + "edge_mlp.0.weight", (H, 2 * H + 1 + He)),
            (p + "edge_mlp.0.bias", (H,)),
            (p + "edge_mlp.2.weight", (H, H)),
            (p + "edge_mlp.2.bias", (H,)),
            (p + "node_mlp.0.weight", (H, 2 * H)),
            (p + "node_mlp.0.bias", (H,)),
            (p + "node_mlp.1.weight", (H,)),
            (p + "node_mlp.1.bias", (H,)),
            (p + "node_mlp.1.mean_scale", (H,)),
            (p + "node_mlp.3.weight", (H, H)),
            (p + "node_mlp.3.bias", (H,)),
        ]
        if l == hp.depth - 1 and hp.family == 0:
            s += [
                (p + "coord_mlp.0.weight", (H, H)),
                (p + "coord_mlp.0.bias", (H,)),
                (p + "coord_mlp.2.weight", (1, H)),
            ]
        s += [
            (p + "att_mlp.0.weight", (1, H)),
            (p + "att_mlp.0.bias", (1,)),
        ]
    if hp.family == 1:      # egnn_net.py:329-360: four pair heads on cat[h_r, h_l, D]
        for head, nout in (("to_energy", 1), ("to_force", 1), ("to_dist", 64), ("to_confidence", 1)):
            s += [
                (head + ".0.weight", (H, 2 * H + 1)),
                (head + ".1.weight", (H,)),
                (head + ".1.bias", (H,)),
                (head + ".3.weight", (nout, H)),
            ]
    else:
        s += [
            ("to_energy.0.weight", (H, 2 * H)),
            ("to_energy.1.weight", (H,)),
            ("to_energy.1.bias", (H,)),
            ("to_energy.3.weight", (1, H)),
        ]
    s += [
        ("to_ires.0.weight", (2 * H, H)),
        ("to_ires.0.bias", (2 * H,)),
        ("to_ires.2.weight", (2 * H, 2 * H)),
        ("to_ires.2.bias", (2 * H,)),
        ("to_ires.4.weight", (1, 2 * H)),
        ("to_ires.4.bias", (1,)),
        ("t_embed.0.W", (Hi // 2,)),
        ("t_embed.1.weight", (Hi, Hi)),
        ("tr_scale.0.weight", (Hi, Hi + 1)),
        ("tr_scale.1.weight", (Hi,)),
        ("tr_scale.1.bias", (Hi,)),
        ("tr_scale.4.weight", (1, Hi)),
        ("rot_scale.0.weight", (Hi, Hi + 1)),
        ("rot_scale.1.weight", (Hi,)),
        ("rot_scale.1.bias", (Hi,)),
        ("rot_scale.4.weight", (1, Hi)),
    ]
    return s


def n_params(hp: HParams = HParams()) -> int:
    return sum(int(np.prod(sh)) for _, sh in param_specs(hp))


def make_random_weights(seed: int = 0, hp: HParams = HParams()) -> "OrderedDict[str, np.ndarray]":
    """Deterministic, well-conditioned float32 parameters (build-owned)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    H = hp.node_dim
    for name, shape in param_specs(hp):
        leaf = name.split(".")[-1]
        if name == "t_embed.0.W":
            w = rng.standard_normal(shape)
        elif leaf == "mean_scale":
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif ("node_mlp.1." in name or "to_energy.1." in name or "to_force.1." in name or "to_dist.1." in name
              or "to_confidence.1." in name or "tr_scale.1." in name or "rot_scale.1." in name):
            # GraphNorm / LayerNorm affine
            w = (1.0 + 0.1 * rng.standard_normal(shape)) if leaf == "weight" \
                else 0.1 * rng.standard_normal(shape)
        elif leaf == "bias":
            w = 0.1 * rng.standard_normal(shape)
        elif name.endswith("coord_mlp.2.weight"):
            w = rng.standard_normal(shape) * (2.0 / math.sqrt(shape[1]))
        elif name in ("tr_scale.4.weight", "rot_scale.4.weight"):
            # no bias in this layer: a negative mean keeps softplus(.) ~ 1e-2 so
            # that Euler-Maruyama rollouts move by Angstroms, not hundreds of them
            w = -0.15 + 0.02 * rng.standard_normal(shape)
        elif name.endswith("node_mlp.3.weight"):
            # residual branch: keep h from blowing up over 6 layers
            w = rng.standard_normal(shape) * (0.5 / math.sqrt(shape[1]))
        else:
            w = rng.standard_normal(shape) * (1.0 / math.sqrt(shape[-1]))
            if name.endswith("edge_mlp.0.weight"):
                # column 2H multiplies radial = |x_i-x_j|^2 (up to ~1e4 A^2)
                w[:, 2 * H] = rng.standard_normal(shape[0]) * 2e-3
            elif hp.family == 1 and name.endswith(".0.weight") and name.startswith("to_") and shape[-1] == 2 * H + 1:
                # column 2H multiplies the CA distance D (up to ~1e2 A)
                w[:, 2 * H] = rng.standard_normal(shape[0]) * 2e-2
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# Named weight draws of the parity suite (tests/golden/make_golden_draws.py runs the REFERENCE on each of them):
# three independent seeds at the generator's scale and one draw at trained-like scale - every Linear of the edge, node and
# coordinate MLPs (weights and biases) times 3, so pre-activations are O(10), SiLU runs well outside its linear range and the
# attention gates saturate.  The trained checkpoint is not in the reference tree (.MISSING_LARGE_BLOBS).
WEIGHT_DRAWS = {"s0": (0, 1.0), "s1": (1, 1.0), "s2": (2, 1.0), "x3": (3, 3.0)}
_SCALED_LINEARS = ("edge_mlp.0.", "edge_mlp.2.", "node_mlp.0.", "node_mlp.3.", "coord_mlp.0.")


def make_weight_draw(draw: str, hp: HParams = HParams()) -> "OrderedDict[str, np.ndarray]":
    seed, scale = WEIGHT_DRAWS[draw]
    w = make_random_weights(seed, hp)
    if scale != 1.0:
        for name in w:
            if any(s in name for s in _SCALED_LINEARS):
                w[name] = (w[name] * np.float32(scale)).astype(np.float32)
    return w


def make_sticky_weights(seed: int = 4, hp: HParams = HParams(), tr_scale: float = 0.1, rot_scale: float = 0.2,
                        coord_bias: float = 1.0, coord_w: float = -0.02) -> "OrderedDict[str, np.ndarray]":
    """A fifth weight draw for the FREE-running sampler tests.  With the seeded random-init draws the two scale
    heads return softplus(~0) = 0.69, i.e. a 180 A translation per step at t = 1: the ligand leaves for good, ~90 % of free runs end
    at energy == 0 / num_clashes == 0 and a distribution test of those outcomes has no power.  This draw keeps every other parameter
    of make_random_weights(seed) and
      * makes tr_scale / rot_scale constant heads (LayerNorm weight 0, bias 1 -> SiLU(1) in every channel; last Linear chosen so that
        Softplus gives `tr_scale` / `rot_scale`): at most 263 * 0.1 = 26 A per step at t = 1, falling with g(t)^2;
      * biases the last layer's coordinate MLP negative (coord_mlp.0.bias = 1, coord_mlp.2.weight = coord_w): every edge pulls its
        node towards the neighbour, so the pooled force points from the ligand to the receptor and the chains end in contact.
    Measured on syn_64_48 (48 CPU free runs, tests/): P(final energy != 0) = 1.00, energy quartiles 0.080 / 0.094 / 0.112, 26 clashes on average.
    (score_net_mlsb.py:396-411: tr_score = unit(mean f) * tr_scale(|mean f|, t); egnn.py:118-137: the coordinate update.)"""
    w = make_random_weights(seed, hp)

    def const_head(prefix, target):
        w[prefix + ".1.weight"] = np.zeros_like(w[prefix + ".1.weight"])
        w[prefix + ".1.bias"] = np.ones_like(w[prefix + ".1.bias"])
        silu1 = 1.0 / (1.0 + np.exp(-1.0))
        w[prefix + ".4.weight"] = np.full_like(w[prefix + ".4.weight"], np.log(np.expm1(target)) / (hp.inner_dim * silu1))

    const_head("tr_scale", tr_scale)
    const_head("rot_scale", rot_scale)
    last = f"network.EGNN_{hp.depth - 1}.egcl."
    w[last + "coord_mlp.0.bias"] = np.full_like(w[last + "coord_mlp.0.bias"], coord_bias)
    w[last + "coord_mlp.2.weight"] = np.full_like(w[last + "coord_mlp.2.weight"], coord_w)
    return w


def pack_blob(weights, hp: HParams = HParams()) -> np.ndarray:
    """state_dict-like mapping (optionally ``net.``-prefixed) -> flat float32 blob."""
    parts = []
    for name, shape in param_specs(hp):
        if name in weights:
            w = weights[name]
        elif "net." + name in weights:
            w = weights["net." + name]
        else:
            raise KeyError(f"missing parameter {name}")
        if hasattr(w, "detach"):
            w = w.detach().cpu().numpy()
        w = np.asarray(w, dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(w.shape)} != {tuple(shape)}")
        parts.append(w.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def unpack_blob(blob: np.ndarray, hp: HParams = HParams()):
    out = OrderedDict()
    off = 0
    for name, shape in param_specs(hp):
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError(f"blob has {blob.size} floats, expected {off}")
    return out


# ---------------------------------------------------------------------------------------------------
# Lightning checkpoint reader (reference: Score_Model.load_from_checkpoint, src/inference_base.py:611-614).
# A Lightning ckpt is a torch-pickled dict with `state_dict` (keys `net.<name>`) and `hyper_parameters`
# holding omegaconf DictConfig objects.  Neither pytorch_lightning nor omegaconf is needed to read it:
# only an explicit allow-list of globals is resolved, every other class is unpickled into an inert attribute bag.
class _Bag:
    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        return self


# Globals a Lightning / torch checkpoint legitimately needs: tensor rebuild helpers, storages, plain containers.  Anything
# else named by the pickle stream (omegaconf, pytorch_lightning, arbitrary builtins such as getattr / eval / __import__ ...)
# is NOT resolved: it becomes an inert attribute bag (_Bag), so a crafted file cannot reach code through this reader.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch", "Tensor"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("_codecs", "encode"),
}
_ALLOWED_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str", "bytes",
                     "bytearray", "slice", "range"}
_TORCH_DTYPES = {"float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool"}


def _stub_pickle_module():
    import pickle
    import types

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _ALLOWED_GLOBALS or (module == "builtins" and name in _ALLOWED_BUILTINS):
                return super().find_class(module, name)
            if module == "torch" and (name.endswith("Storage") or name in _TORCH_DTYPES):
                return super().find_class(module, name)
            return type(name, (_Bag,), {"__module__": module})      # inert: stores its arguments / state, runs nothing

    mod = types.ModuleType("dfm_stub_pickle")
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = "pickle"
    for k in ("dump", "dumps", "loads", "HIGHEST_PROTOCOL", "PickleError", "UnpicklingError", "Pickler"):
        setattr(mod, k, getattr(pickle, k))
    return mod


def _cfg_get(node, key, default=None):
    """Read `key` from a dict, an attribute bag or an unpickled omegaconf node."""
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(key, default)
    d = getattr(node, "__dict__", {})
    for holder in (d, d.get("_content", None), d.get("_state", None)):
        if isinstance(holder, dict) and key in holder:
            v = holder[key]
            vd = getattr(v, "__dict__", {})
            return vd.get("_val", v) if "_val" in vd else v
    return default


def load_lightning_checkpoint(path):
    """-> (state_dict name->np.float32 array without the `net.` prefix, HParams)."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_stub_pickle_module())
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    out = OrderedDict()
    for k, v in sd.items():
        k = k[4:] if k.startswith("net.") else k
        if hasattr(v, "detach"):
            out[k] = v.detach().cpu().float().numpy()
    hp = HParams()
    kw = {}
    # model family from the state_dict itself (a bare state_dict carries no hyper_parameters)
    if "to_force.0.weight" in out:      # EGNN_Net checkpoint (DFMDock.yaml: mask 20 A, agg)
        kw["family"] = 1
        kw["mask_dist"] = 20.0
    if "positional_embed.weight" in out:
        pd = int(out["positional_embed.weight"].shape[1])
        if pd not in (66, 67):
            raise ValueError(f"positional_embed_dim = {pd}: this engine supports 66 (relpos) and 67 (relpos + sym)")
        kw["positional_embed_dim"] = pd
    hyper = ck.get("hyper_parameters") if isinstance(ck, dict) and "state_dict" in ck else None
    model_cfg = _cfg_get(hyper, "model")
    if model_cfg is not None:
        for f in ("lm_embed_dim", "spatial_embed_dim", "node_dim", "edge_dim", "inner_dim", "depth", "cut_off"):
            v = _cfg_get(model_cfg, f)
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                kw[f] = type(getattr(hp, f))(v)
        v = _cfg_get(model_cfg, "positional_embed_dim")
        if isinstance(v, int) and "positional_embed_dim" in kw and v != kw["positional_embed_dim"]:
            raise ValueError(f"hyper_parameters say positional_embed_dim = {v}, the weights have {kw['positional_embed_dim']}")
        if kw.get("family") == 1:
            kw["agg_mean"] = (_cfg_get(model_cfg, "agg") != "sum")
    # diffusion schedules the reference reads from the checkpoint's config (score_model_mlsb.py:36-37, DFMDock.py:45-46)
    diff_cfg = _cfg_get(hyper, "diffuser")
    for sub, prefix in (("r3", "r3_"), ("so3", "so3_")):
        node = _cfg_get(diff_cfg, sub)
        for f in ("min_sigma", "max_sigma"):
            v = _cfg_get(node, f)
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                kw[prefix + f] = float(v)
    hp = HParams(**{**hp.as_dict(), **kw})
    return out, hp

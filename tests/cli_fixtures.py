"""Inputs of the command-line tests, written by the tests themselves: a seeded Lightning-style checkpoint, PDB files of a golden
complex's backbone, its features file and DB5-style `<id>.pt` pickles (same object graph as the reference's HeteroData files,
SURVEY.md Appendix A: `_global_store._mapping{name}`, `_node_store_dict{receptor, ligand}._mapping{x, pos, seq}`)."""
import os

import numpy as np


class BaseStorage:
    pass


class NodeStorage:
    pass


class HeteroData:
    pass


def write_ckpt(path, hp=None, seed=0, lightning=True):
    """torch.save of {"state_dict": {"net.<name>": tensor}, "hyper_parameters": {...}} (what Score_Model.load_from_checkpoint reads,
    src/inference_base.py:611-616) or of the bare state_dict."""
    import torch
    from dfmdock_amd.weights import HParams, make_random_weights
    hp = hp or HParams()
    w = make_random_weights(seed, hp)
    sd = {("net." + k if lightning else k): torch.from_numpy(v.copy()) for k, v in w.items()}
    if lightning:
        hyper = {"model": {"lm_embed_dim": hp.lm_embed_dim, "positional_embed_dim": hp.positional_embed_dim, "spatial_embed_dim": 100,
                           "node_dim": 256, "edge_dim": 128, "inner_dim": 128, "depth": hp.depth, "cut_off": hp.cut_off},
                 "diffuser": {"r3": {"min_sigma": 0.1, "max_sigma": 30.0}, "so3": {"min_sigma": 0.1, "max_sigma": 1.5}}}
        torch.save({"state_dict": sd, "hyper_parameters": hyper, "epoch": 3}, path)
    else:
        torch.save(sd, path)
    return w


def write_pair(tmp, cx, rec_seq, lig_seq, with_onehot=False):
    """rec.pdb / lig.pdb (N, CA, C, O, CB per residue in the reference's save_PDB format) + features.npz of a complex dict."""
    from dfmdock_amd import pdbio
    paths = {}
    for side, seq, chain_first in (("rec", rec_seq, True), ("lig", lig_seq, False)):
        p = os.path.join(tmp, side + ".pdb")
        pdbio.write_backbone_pdb(p, pdbio.full_backbone(cx[side + "_pos"]), seq, delim=len(seq) - 1 if chain_first else -1, mode="w")
        paths[side] = p
    feat = os.path.join(tmp, "features.npz")
    if with_onehot:
        np.savez(feat, rec_x=cx["rec_x"], lig_x=cx["lig_x"])
    else:
        np.savez(feat, rec_esm=cx["rec_x"][:, :1280], lig_esm=cx["lig_x"][:, :1280], rec_seq=rec_seq, lig_seq=lig_seq)
    return paths["rec"], paths["lig"], feat


def write_db5_pt(path, cid, cx, rec_seq, lig_seq):
    import torch
    d = HeteroData()
    g = BaseStorage()
    g.__dict__["_mapping"] = {"name": cid}
    stores = {}
    for side, key, seq in (("rec", "receptor", rec_seq), ("lig", "ligand", lig_seq)):
        s = NodeStorage()
        s.__dict__["_mapping"] = {"x": torch.from_numpy(np.ascontiguousarray(cx[side + "_x"][:, :1280])),
                                  "pos": torch.from_numpy(np.ascontiguousarray(cx[side + "_pos"], np.float32)), "seq": seq}
        stores[key] = s
    d.__dict__.update(_global_store=g, _node_store_dict=stores, _edge_store_dict={})
    torch.save(d, path)


def golden_7cei():
    from conftest import complex_for, load_golden
    d = load_golden("cx_7CEI.npz")
    return complex_for("fwd_7CEI_p0"), str(d["rec_seq"]), str(d["lig_seq"])

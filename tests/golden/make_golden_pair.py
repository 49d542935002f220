#!/usr/bin/env python3
"""Golden vectors of the second model family by RUNNING THE REFERENCE:
``DFMDock.forward`` (src/models/DFMDock.py:68-75: move_to_lig_center + EGNN_Net(batch, predict=True),
src/models/egnn_net.py:408-505) with the build's seeded weights, on the complexes the family-0 goldens use.

Same rules as make_golden.py (whose stand-ins for the absent third-party modules are reused by importing it):
runs only in the build container; only the .npz outputs travel.  positional_embed_dim is 66 here: the
reference's own ``get_position_matrix`` (utils/crop.py:193-207) produces 66 channels - the 67th ("sym") channel of
configs/model/DFMDock.yaml has no producer in the reference tree.

Usage:  python tests/golden/make_golden_pair.py        (rewrites fwd2_*.npz)
"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the stubs, puts the reference on sys.path)

import models.egnn_net as en  # noqa: E402
import models.DFMDock as dd  # noqa: E402
from utils.crop import get_position_matrix  # noqa: E402

from dfmdock_amd.weights import HParams, make_random_weights  # noqa: E402
from dfmdock_amd.synthetic import make_complex  # noqa: E402
from dfmdock_amd.db5 import load_db5_pt  # noqa: E402

HP1 = HParams(family=1, mask_dist=20.0)
_orig_knn = en.get_knn_and_sample


def build_net(seed=0, hp=HP1):
    conf = en.ModelConfig(lm_embed_dim=hp.lm_embed_dim, positional_embed_dim=hp.positional_embed_dim,
                          spatial_embed_dim=hp.spatial_embed_dim, node_dim=hp.node_dim, edge_dim=hp.edge_dim,
                          inner_dim=hp.inner_dim, depth=hp.depth, dropout=0.1, cut_off=hp.cut_off, normalize=True,
                          agg="mean" if hp.agg_mean else "sum")
    net = en.EGNN_Net(conf)
    w = make_random_weights(seed, hp)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    net.eval()
    return net


class Wrapper:
    """The two methods DFMDock.forward uses, bound to a bare object (the LightningModule needs hydra configs)."""
    move_to_lig_center = dd.DFMDock.move_to_lig_center
    forward = dd.DFMDock.forward

    def __init__(self, net):
        self.net = net


def forward_case(net, cx, lig_pos, t, seed):
    batch = {k: torch.from_numpy(np.ascontiguousarray(cx[k])).float() for k in ("rec_x", "lig_x", "rec_pos", "lig_pos")}
    batch["lig_pos"] = torch.from_numpy(lig_pos).float()
    batch = get_position_matrix(batch)
    batch["t"] = torch.tensor([t], dtype=torch.float32)
    recd = mg.EdgeRecorder()
    mg._orig_knn, keep = _orig_knn, mg._orig_knn      # the recorder calls mg._orig_knn
    en.get_knn_and_sample = recd
    hs = []
    hooks = [net.network._modules[f"EGNN_{l}"].register_forward_hook(lambda m, i, o: hs.append(o[0].detach().numpy().copy()))
             for l in range(HP1.depth)]
    torch.manual_seed(seed)
    out = Wrapper(net).forward(batch)
    for h in hooks:
        h.remove()
    en.get_knn_and_sample = _orig_knn
    mg._orig_knn = keep
    k, s = recd.rec[0]
    return {
        "lig_pos": lig_pos.astype(np.float32), "t": np.float32(t), "edges": mg.edges_of(k, s),
        "tr_score": out["tr_score"].detach().numpy(), "rot_score": out["rot_score"].detach().numpy(),
        "energy": out["energy"].detach().numpy(), "f": out["f"].detach().numpy(),
        "num_clashes": np.int64(out["num_clashes"].item()),
        "confidence_logits": out["confidence_logits"].detach().numpy(),
        "ires_logits": out["ires_logits"].detach().numpy()[:, 0],
        "dist_logits_sample": out["dist_logits"].detach().numpy()[:4, :4].copy(),
        "_dist_full": out["dist_logits"].detach().numpy().copy(),      # popped by the callers (only fwd2_dist.npz keeps it)
        "h_last": hs[-1].astype(np.float32), "h_first": hs[0].astype(np.float32),
        "centered_lig_ca": batch["lig_pos"][:, 1, :].detach().numpy().copy(),
    }


def slim(r):
    return {k: v for k, v in r.items() if not k.startswith("_")}


def gen_dist(net):
    """dist_logits = to_dist(interaction) [R, L, 64] (egnn_net.py:347-352,:447,:500): the whole tensor for the small synthetic
    complex, every 8th receptor / ligand residue for 7CEI (87 x 127 x 64 floats would be 2.8 MB)."""
    cx = make_complex(24, 16, seed=5)
    g = dict(np.load(os.path.join(HERE, "fwd2_syn_24_16.npz")))
    r = forward_case(net, cx, g["lig_pos"], float(g["t"]), seed=1)
    assert (r["edges"] == g["edges"]).all()
    d = load_db5_pt(os.path.join(mg.REF, "data/db5_test/7CEI.pt"))
    cx7 = {"rec_x": np.concatenate([d["rec_esm"].astype(np.float16).astype(np.float32), d["rec_x"][:, 1280:]], 1),
           "lig_x": np.concatenate([d["lig_esm"].astype(np.float16).astype(np.float32), d["lig_x"][:, 1280:]], 1),
           "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    g7 = dict(np.load(os.path.join(HERE, "fwd2_7CEI_p1.npz")))
    r7 = forward_case(net, cx7, g7["lig_pos"], float(g7["t"]), seed=41)
    assert (r7["edges"] == g7["edges"]).all(), "same seed as fwd2_7CEI_p1 -> same graph"
    mg.save("fwd2_dist.npz", syn_24_16=r["_dist_full"], cei_p1_stride8=r7["_dist_full"][::8, ::8].copy())


def main():
    net = build_net(0)
    rng = np.random.Generator(np.random.PCG64(29))
    cx = make_complex(24, 16, seed=5)
    mg.save("fwd2_syn_24_16.npz", R=24, L=16, cx_seed=5, **slim(forward_case(net, cx, cx["lig_pos"], 0.5, seed=1)))
    cx = make_complex(9, 7, seed=6)
    mg.save("fwd2_syn_9_7.npz", R=9, L=7, cx_seed=6, **slim(forward_case(net, cx, cx["lig_pos"], 0.3, seed=1)))
    cx = make_complex(64, 48, seed=7)
    for i, (t, rot, trs) in enumerate([(1.0, 40.0, 6.0), (0.49, 10.0, 2.0), (0.001, 0.0, 0.0)]):
        lp = mg.noised_pose(cx, rng, rot, trs)
        mg.save(f"fwd2_syn_64_48_p{i}.npz", R=64, L=48, cx_seed=7, **slim(forward_case(net, cx, lp, t, seed=10 + i)))
    d = load_db5_pt(os.path.join(mg.REF, "data/db5_test/7CEI.pt"))
    cx = {"rec_x": np.concatenate([d["rec_esm"].astype(np.float16).astype(np.float32), d["rec_x"][:, 1280:]], 1),
          "lig_x": np.concatenate([d["lig_esm"].astype(np.float16).astype(np.float32), d["lig_x"][:, 1280:]], 1),
          "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    for i, (t, rot, trs) in enumerate([(0.001, 0.0, 0.0), (1.0, 60.0, 8.0), (0.3, 12.0, 2.5)]):
        lp = mg.noised_pose(cx, rng, rot, trs) if i else cx["lig_pos"]
        mg.save(f"fwd2_7CEI_p{i}.npz", **slim(forward_case(net, cx, lp, t, seed=40 + i)))
    # DFMDock.modify_coords / move_to_lig_center (DFMDock.py:246-257): rotation about the ALL-ATOM centroid
    x = torch.from_numpy(rng.standard_normal((9, 3, 3)).astype(np.float32) * 10)
    rot, tr = torch.tensor([[0.2, -0.1, 0.4]]), torch.tensor([[1.0, -2.0, 0.5]])
    w = Wrapper(net)
    x2 = dd.DFMDock.modify_coords(w, x.clone(), rot, tr)
    b2 = {"rec_pos": x.clone() + 3.0, "lig_pos": x.clone()}
    w.move_to_lig_center(b2)
    mg.save("pair_kats.npz", mc_x=x.numpy(), mc_rot=rot.numpy(), mc_tr=tr.numpy(), mc_out=x2.numpy(),
            centred_rec=b2["rec_pos"].numpy(), centred_lig=b2["lig_pos"].numpy())
    # `agg: sum` variant (egnn_net.py:438-441,:459-474) on one small case
    hp_sum = HParams(family=1, mask_dist=20.0, agg_mean=False)
    net_s = build_net(0, hp_sum)
    cx = make_complex(24, 16, seed=5)
    mg.save("fwd2_sum_syn_24_16.npz", R=24, L=16, cx_seed=5, **slim(forward_case(net_s, cx, cx["lig_pos"], 0.5, seed=1)))


if __name__ == "__main__":
    if sys.argv[1:] == ["dist"]:
        gen_dist(build_net(0))
    else:
        main()

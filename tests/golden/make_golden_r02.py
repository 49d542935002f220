#!/usr/bin/env python3
"""Round-2 golden vectors, all produced by RUNNING THE REFERENCE (same rules as make_golden.py, whose stand-ins
for the absent third-party modules and recording helpers are reused by importing it):

  rollout_anneal_*.npz   inference_base.Euler_Maruyama_sampler(noise_annealing=True)  (src/inference_base.py:428-430)
  rollout_ode_*.npz      inference_mlsb.Sampler.Euler_Maruyama_sampler(ode=True)      (src/inference_mlsb.py:264-350,
                         src/utils/so3_diffuser.py:367-368); that sampler first moves both chains to their CA centroids
                         (:352-378), so its poses equal inference_base's shifted by -c1 (recorded as `c1`)
  rollout2_*.npz         second model family: src/inference.py:292-372 (all-atom centroids in randomize_pose /
                         modify_coords, :220-254) driving DFMDock.forward
  fwd2_sym_*.npz         EGNN_Net with positional_embed_dim = 67 (configs/model/DFMDock.yaml:5), position matrix =
                         [relpos66 | is_homomer] for a homomer (sym column = 1)
  fwd_c3_300_300.npz, fwd_c5_1000_1000.npz
                         one reference score evaluation at the BASELINE C3 / C5 sizes (no h taps: small files)
  fwd_db5_<id>.npz       reference score evaluations on DB5 backbones with seeded node features (the ESM blocks are too
                         large to commit; features = N(0,1) seeded by the id || one-hot(seq))
  db5_backbones.npz      backbone (N, CA, C) + sequence of the 24 DB5 test complexes present in the reference
                         (data/db5_test/*.pt): data for the C4-shaped set run on one GPU

Usage:  python tests/golden/make_golden_r02.py [names...]
"""
import glob
import os
import sys
import zlib
from types import SimpleNamespace as NS

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_pair as mgp  # noqa: E402

import inference as inf1  # noqa: E402   (src/inference.py: the all-atom-centroid sampler of the second family)
import inference_base as ib  # noqa: E402
import inference_mlsb as im  # noqa: E402
import models.egnn_net as en  # noqa: E402
import models.score_net_mlsb as snm  # noqa: E402
from utils.crop import get_position_matrix  # noqa: E402

from dfmdock_amd.db5 import load_db5_pt  # noqa: E402
from dfmdock_amd.synthetic import make_complex, seq_to_onehot  # noqa: E402
from dfmdock_amd.weights import HParams, make_random_weights  # noqa: E402

inf1.tqdm = lambda x: x
im.tqdm = lambda x: x


def seeded_features(name, seq):
    """Node features for a DB5 chain without its ESM block: N(0,1) seeded by (name) || one-hot(seq)."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    return np.concatenate([rng.standard_normal((len(seq), 1280)).astype(np.float32), seq_to_onehot(seq)], 1)


def db5_complex(cid, d=None):
    d = d or load_db5_pt(os.path.join(mg.REF, f"data/db5_test/{cid}.pt"))
    return {"rec_x": seeded_features(cid + ":rec", d["rec_seq"]), "lig_x": seeded_features(cid + ":lig", d["lig_seq"]),
            "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}


class Recorder:
    """Patches the random sources of one sampler module and records every draw + every pose."""

    def __init__(self, mod, modify_owner=None):
        self.mod, self.rec = mod, {"z": [], "poses": []}
        self.modify_owner = modify_owner

    def __enter__(self):
        r, mod = self.rec, self.mod
        self.o_rot, self.o_normal, self.o_randn = mod.Rotation, torch.normal, torch.randn
        orig_random = mod.Rotation.random

        def rot_random(*a, **k):
            x = orig_random(*a, **k)
            r["R0"] = x.as_matrix().copy()
            return x

        def normal(*a, **k):
            v = self.o_normal(*a, **k)
            r["tr_draw"] = v.numpy().copy()
            return v

        def randn(*a, **k):
            v = self.o_randn(*a, **k)
            r["z"].append(v.numpy().copy())
            return v

        mod.Rotation = NS(random=rot_random)
        torch.normal, torch.randn = normal, randn
        return self

    def __exit__(self, *a):
        self.mod.Rotation = self.o_rot
        torch.normal, torch.randn = self.o_normal, self.o_randn


class Wrapped(nn.Module):
    def __init__(self, m, keys=("tr_score", "rot_score", "energy", "f")):
        super().__init__()
        self.m, self.keys, self.outs = m, keys, []
        self.r3_diffuser, self.so3_diffuser = m.r3_diffuser, m.so3_diffuser

    def forward(self, b):
        o = self.m(b)
        self.outs.append({k: o[k].detach().numpy().copy() for k in self.keys})
        self.outs[-1]["num_clashes"] = int(o["num_clashes"].item())
        return o


def pack_rollout(cx, num_steps, rec, edges, wm, poses, extra):
    z = np.concatenate(rec["z"], 0).reshape(-1, 2, 3) if rec["z"] else np.zeros((num_steps, 2, 3), np.float32)
    return dict(R=cx["rec_pos"].shape[0], L=cx["lig_pos"].shape[0], num_steps=num_steps,
                R0=rec["R0"].astype(np.float64), tr_draw=rec["tr_draw"].astype(np.float32),
                z_rot=z[:, 0].astype(np.float32), z_tr=z[:, 1].astype(np.float32), edges=edges,
                poses=np.stack(poses).astype(np.float32),
                tr_score=np.stack([o["tr_score"][0] for o in wm.outs]), rot_score=np.stack([o["rot_score"][0] for o in wm.outs]),
                energy=np.array([o["energy"] for o in wm.outs]), num_clashes=np.array([o["num_clashes"] for o in wm.outs]),
                **extra)


def gen_rollout_anneal(model, cx, name, num_steps, seed):
    """inference_base.Euler_Maruyama_sampler(noise_annealing=True): noise scale = time_step (:428-430)."""
    batch = mg.make_batch(cx)
    recd = mg.EdgeRecorder()
    snm.get_knn_and_sample = recd
    poses = []
    orig_modify = ib.modify_coords

    def modify(x, rot, tr):
        y = orig_modify(x, rot, tr)
        poses.append(y.numpy().copy())
        return y

    wm = Wrapped(model)
    ib.modify_coords = modify
    try:
        with Recorder(ib) as rc:
            np.random.seed(seed)
            torch.manual_seed(seed)
            _, lig_pos, rot_update, tr_update, output = ib.Euler_Maruyama_sampler(
                model=wm, batch=dict(batch), num_steps=num_steps, device="cpu", noise_annealing=True)
    finally:
        ib.modify_coords = orig_modify
        snm.get_knn_and_sample = mg._orig_knn
    edges = np.stack([mg.edges_of(k, s) for k, s in recd.rec])
    mg.save(name, **pack_rollout(cx, num_steps, rc.rec, edges, wm, poses, dict(
        final_lig_pos=lig_pos.numpy(), rot_update=rot_update.numpy(), tr_update=tr_update.numpy(),
        final_energy=np.float32(output["energy"].item()), final_num_clashes=np.int64(output["num_clashes"].item()))))


def gen_rollout_ode(model, cx, name, num_steps, seed):
    """inference_mlsb.Sampler.Euler_Maruyama_sampler(ode=True) on a hand-built Sampler (its __init__ wants a checkpoint
    and a dataset; the sampler itself touches only the attributes set here)."""
    batch = mg.make_batch(cx)
    recd = mg.EdgeRecorder()
    snm.get_knn_and_sample = recd
    wm = Wrapped(model)
    s = im.Sampler.__new__(im.Sampler)
    s.device = torch.device("cpu")
    s.model = wm
    s.perturb_tr = s.perturb_rot = True
    s.data_conf = NS(num_steps=num_steps, tr_noise_scale=0.5, rot_noise_scale=0.5, use_clash_force=False)
    c1 = batch["rec_pos"][:, 1, :].mean(0).numpy().copy()
    try:
        with Recorder(im) as rc:
            np.random.seed(seed)
            torch.manual_seed(seed)
            rec_trj, lig_trj, energy, clashes = s.Euler_Maruyama_sampler(dict(batch), ode=True)
    finally:
        snm.get_knn_and_sample = mg._orig_knn
    assert not rc.rec["z"], "the ODE step draws no noise"
    edges = np.stack([mg.edges_of(k, s_) for k, s_ in recd.rec])
    poses = [p.numpy().copy() for p in lig_trj[1:]]
    mg.save(name, **pack_rollout(cx, num_steps, rc.rec, edges, wm, poses, dict(
        init_pose=lig_trj[0].numpy().copy(), rec_centered=rec_trj[0].numpy().copy(), c1=c1,
        final_energy=np.float32(energy.item()), final_num_clashes=np.int64(clashes.item()))))


class PairModel(nn.Module):
    """What src/inference.py's sampler touches of the DFMDock LightningModule: forward + the two diffusers."""

    def __init__(self, net, base):
        super().__init__()
        self.w = mgp.Wrapper(net)
        self.r3_diffuser, self.so3_diffuser = base.r3_diffuser, base.so3_diffuser

    def forward(self, batch):
        return self.w.forward(batch)


def gen_rollout_pair(base_model, cx, name, num_steps, seed):
    """Second family: src/inference.py:292-372 with DFMDock.forward; all-atom centroids (:220-254)."""
    net = mgp.build_net(0)
    batch = {k: torch.from_numpy(np.ascontiguousarray(cx[k])).float() for k in ("rec_x", "lig_x", "rec_pos", "lig_pos")}
    batch = get_position_matrix(batch)
    recd = mg.EdgeRecorder()
    mg._orig_knn, keep = mgp._orig_knn, mg._orig_knn
    en.get_knn_and_sample = recd
    poses = []
    orig_modify, orig_randomize = inf1.modify_coords, inf1.randomize_pose
    init = {}

    def modify(x, rot, tr):
        y = orig_modify(x, rot, tr)
        poses.append(y.numpy().copy())
        return y

    def randomize(x1, x2):
        out = orig_randomize(x1, x2)
        init["pose"], init["tr"], init["rot"] = out[0].numpy().copy(), out[1].numpy().copy(), out[2].numpy().copy()
        return out

    wm = Wrapped(PairModel(net, base_model), keys=("tr_score", "rot_score", "energy", "f", "confidence_logits"))
    inf1.modify_coords, inf1.randomize_pose = modify, randomize
    try:
        with Recorder(inf1) as rc:
            np.random.seed(seed)
            torch.manual_seed(seed)
            _, lig_pos, rot_update, tr_update, output = inf1.Euler_Maruyama_sampler(
                model=wm, batch=dict(batch), num_steps=num_steps, device="cpu")
    finally:
        inf1.modify_coords, inf1.randomize_pose = orig_modify, orig_randomize
        en.get_knn_and_sample = mgp._orig_knn
        mg._orig_knn = keep
    edges = np.stack([mg.edges_of(k, s) for k, s in recd.rec])
    mg.save(name, **pack_rollout(cx, num_steps, rc.rec, edges, wm, poses, dict(
        init_pose=init["pose"].astype(np.float32), init_tr=init["tr"], init_rot=init["rot"],
        final_lig_pos=lig_pos.numpy(), rot_update=rot_update.numpy(), tr_update=tr_update.numpy(),
        final_energy=np.float32(output["energy"].item()), final_num_clashes=np.int64(output["num_clashes"].item()),
        final_confidence=np.float32(output["confidence_logits"].item()))))


def gen_sym():
    """EGNN_Net with the YAML's positional_embed_dim = 67: the 67th position channel is the homomer flag."""
    hp = HParams(family=1, mask_dist=20.0, positional_embed_dim=67)
    net = mgp.build_net(0, hp)
    cx = make_complex(24, 16, seed=5)
    batch = {k: torch.from_numpy(np.ascontiguousarray(cx[k])).float() for k in ("rec_x", "lig_x", "rec_pos", "lig_pos")}
    batch = get_position_matrix(batch)
    for flag in (0, 1):
        b = dict(batch)
        pm = batch["position_matrix"]
        b["position_matrix"] = torch.cat([pm, torch.full(pm.shape[:-1] + (1,), float(flag))], -1)
        b["t"] = torch.tensor([0.5])
        recd = mg.EdgeRecorder()
        mg._orig_knn, keep = mgp._orig_knn, mg._orig_knn
        en.get_knn_and_sample = recd
        torch.manual_seed(1)
        out = mgp.Wrapper(net).forward(b)
        en.get_knn_and_sample = mgp._orig_knn
        mg._orig_knn = keep
        k, s = recd.rec[0]
        mg.save(f"fwd2_sym{flag}_syn_24_16.npz", R=24, L=16, cx_seed=5, sym=flag, lig_pos=cx["lig_pos"], t=np.float32(0.5),
                edges=mg.edges_of(k, s), tr_score=out["tr_score"].detach().numpy(), rot_score=out["rot_score"].detach().numpy(),
                energy=out["energy"].detach().numpy(), f=out["f"].detach().numpy(),
                num_clashes=np.int64(out["num_clashes"].item()), confidence_logits=out["confidence_logits"].detach().numpy())


def slim_forward(net, cx, lig_pos, t, seed):
    """mg.forward_case without the [N,256] taps and with int16 edges: fixtures for the large configurations."""
    r = mg.forward_case(net, cx, lig_pos, t, seed)
    keep = ("lig_pos", "t", "tr_score", "rot_score", "energy", "f", "num_clashes", "ires", "h_absmean", "h_absmax")
    out = {k: r[k] for k in keep}
    out["edges"] = r["edges"].astype(np.int16)
    out["codes_sum"] = np.int64(r["bins"].astype(np.int64).sum())      # checksum of the feature bins
    out["bins_sample"] = r["bins"][::37].copy()                       # every 37th node's bins, exact
    out["relpos_sample"] = r["relpos"][::37].copy()
    return out


def gen_big(net):
    rng = np.random.Generator(np.random.PCG64(61))
    cx = make_complex(300, 300, seed=1)
    mg.save("fwd_c3_300_300.npz", R=300, L=300, cx_seed=1, **slim_forward(net, cx, mg.noised_pose(cx, rng, 20.0, 4.0), 0.6, seed=3))
    cx = make_complex(1000, 1000, seed=1)
    mg.save("fwd_c5_1000_1000.npz", R=1000, L=1000, cx_seed=1, **slim_forward(net, cx, mg.noised_pose(cx, rng, 10.0, 3.0), 0.4, seed=4))


def gen_db5(net):
    ids, arrs = [], {}
    for path in sorted(glob.glob(os.path.join(mg.REF, "data/db5_test/*.pt"))):
        cid = os.path.basename(path)[:-3]
        d = load_db5_pt(path)
        ids.append(cid)
        arrs[cid + "_rec_pos"], arrs[cid + "_lig_pos"] = d["rec_pos"].astype(np.float32), d["lig_pos"].astype(np.float32)
        arrs[cid + "_rec_seq"], arrs[cid + "_lig_seq"] = d["rec_seq"], d["lig_seq"]
    mg.save("db5_backbones.npz", ids=np.array(ids), **arrs)
    rng = np.random.Generator(np.random.PCG64(67))
    for cid, t, rot, trs in (("1AVX", 0.7, 25.0, 5.0), ("4POU", 0.2, 6.0, 1.5)):
        cx = db5_complex(cid)
        mg.save(f"fwd_db5_{cid}.npz", **slim_forward(net, cx, mg.noised_pose(cx, rng, rot, trs), t, seed=5))


def main(which):
    net = mg.build_net(0)
    model = mg.Model(net).eval()
    d = load_db5_pt(os.path.join(mg.REF, "data/db5_test/7CEI.pt"))
    cx7 = {"rec_x": np.concatenate([d["rec_esm"].astype(np.float16).astype(np.float32), d["rec_x"][:, 1280:]], 1),
           "lig_x": np.concatenate([d["lig_esm"].astype(np.float16).astype(np.float32), d["lig_x"][:, 1280:]], 1),
           "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    cxs = make_complex(24, 16, seed=5)
    jobs = {
        "anneal": lambda: (gen_rollout_anneal(model, cxs, "rollout_anneal_syn_24_16.npz", 40, 211),
                           gen_rollout_anneal(model, cx7, "rollout_anneal_7CEI.npz", 6, 212)),
        "ode": lambda: (gen_rollout_ode(model, cxs, "rollout_ode_syn_24_16.npz", 40, 221),
                        gen_rollout_ode(model, cx7, "rollout_ode_7CEI.npz", 6, 222)),
        "pair": lambda: (gen_rollout_pair(model, cxs, "rollout2_syn_24_16.npz", 40, 231),
                         gen_rollout_pair(model, cx7, "rollout2_7CEI.npz", 6, 232)),
        "sym": gen_sym,
        "big": lambda: gen_big(net),
        "db5": lambda: gen_db5(net),
    }
    for k in (which or list(jobs)):
        jobs[k]()


if __name__ == "__main__":
    main(sys.argv[1:])

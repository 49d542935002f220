#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it never travels to the GPU box; only the .npz files do).
The reference is imported unmodified; the third-party modules it imports that
are absent here are replaced by the stand-ins of SURVEY.md Appendix B:
``torch_geometric.nn.norm.GraphNorm`` (restated from the torch_geometric 2.6.0
formula - the one piece of arithmetic on the path that is not under
/root/reference, so parity at that boundary is pinned only by this formula),
MagicMocks for esm/biotite/hydra/omegaconf/tree and a minimal
pytorch_lightning.

Usage:  python tests/golden/make_golden.py            (rewrites every .npz)
"""
import os
import sys
import types
from types import SimpleNamespace as NS
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "src"))


# ----------------------------------------------------------------------------
# stand-ins for absent third-party modules
class GraphNorm(nn.Module):
    """torch_geometric.nn.norm.GraphNorm (2.6.0) with batch=None (single graph)."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def forward(self, x, batch=None, batch_size=None):
        mean = x.mean(0, keepdim=True)
        out = x - mean * self.mean_scale
        var = out.pow(2).mean(0, keepdim=True)
        return self.weight * out / (var + self.eps).sqrt() + self.bias


def install_stubs():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgn = types.ModuleType("torch_geometric.nn.norm")
    tgl = types.ModuleType("torch_geometric.loader")
    tgn.GraphNorm = GraphNorm
    tgl.DataLoader = object
    tgd = types.ModuleType("torch_geometric.data")
    tgd.HeteroData = type("HeteroData", (), {})
    tg.data = tgd
    sys.modules["torch_geometric.data"] = tgd
    tgnn.norm = tgn
    tg.nn = tgnn
    tg.loader = tgl
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn,
                        "torch_geometric.nn.norm": tgn, "torch_geometric.loader": tgl})
    for name in ["esm", "biotite", "biotite.structure", "biotite.structure.io",
                 "biotite.structure.io.pdb", "tree", "hydra", "omegaconf"]:
        sys.modules[name] = MagicMock()
    sys.modules["hydra"].main = lambda **kw: (lambda f: f)
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl


install_stubs()

import models.score_net_mlsb as snm  # noqa: E402
import inference_base as ib  # noqa: E402
from utils import geometry as geo  # noqa: E402
from utils.coords6d import get_coords6d  # noqa: E402
from utils.metrics import compute_metrics  # noqa: E402
from utils.r3_diffuser import R3Diffuser  # noqa: E402
from utils.so3_diffuser import SO3Diffuser  # noqa: E402

from dfmdock_amd.weights import HParams, make_random_weights  # noqa: E402
from dfmdock_amd.synthetic import make_complex  # noqa: E402
from dfmdock_amd.db5 import load_db5_pt  # noqa: E402

ib.tqdm = lambda x: x
torch.set_num_threads(8)

HP = HParams()


def build_net(seed=0):
    conf = NS(lm_embed_dim=HP.lm_embed_dim, positional_embed_dim=HP.positional_embed_dim,
              spatial_embed_dim=HP.spatial_embed_dim, node_dim=HP.node_dim, edge_dim=HP.edge_dim,
              inner_dim=HP.inner_dim, depth=HP.depth, dropout=0.1, cut_off=HP.cut_off, normalize=True)
    net = snm.Score_Net(conf)
    w = make_random_weights(seed, HP)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    net.eval()
    return net


class Model(nn.Module):
    """What Euler_Maruyama_sampler touches of Score_Model (score_model_mlsb.py:52-63)."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.r3_diffuser = R3Diffuser(NS(min_sigma=HP.r3_min_sigma, max_sigma=HP.r3_max_sigma))
        so3 = SO3Diffuser.__new__(SO3Diffuser)  # skip the 61 s IGSO3 table build (unused at inference)
        so3.schedule = "logarithmic"
        so3.min_sigma = HP.so3_min_sigma
        so3.max_sigma = HP.so3_max_sigma
        self.so3_diffuser = so3

    def forward(self, batch):
        return self.net(batch, predict=True)


def make_batch(cx):
    batch = {k: torch.from_numpy(np.ascontiguousarray(cx[k])).float()
             for k in ("rec_x", "lig_x", "rec_pos", "lig_pos")}
    return ib.get_position_matrix(batch)


_orig_knn = snm.get_knn_and_sample


class EdgeRecorder:
    """Replace get_knn_and_sample by a recording (or replaying) wrapper."""

    def __init__(self, replay=None):
        self.rec = []
        self.replay = replay
        self.i = 0

    def __call__(self, points, knn=20, sample_size=40, epsilon=1e-10):
        if self.replay is not None:
            k, s = self.replay[self.i]
            self.i += 1
            return torch.from_numpy(k).long(), (None if s is None else torch.from_numpy(s).long())
        k, s = _orig_knn(points, knn=knn, sample_size=sample_size, epsilon=epsilon)
        self.rec.append((k.numpy().copy(), None if s is None else s.numpy().copy()))
        return k, s


def edges_of(k, s):
    e = k if s is None else np.concatenate([k, s], axis=1)
    return e.astype(np.int32)


def forward_case(net, cx, lig_pos, t, seed):
    """One reference score evaluation; returns inputs, recorded edges, outputs, intermediates."""
    batch = make_batch(cx)
    batch["lig_pos"] = torch.from_numpy(lig_pos).float()
    batch["t"] = torch.tensor([t], dtype=torch.float32)
    recd = EdgeRecorder()
    snm.get_knn_and_sample = recd
    hs = []
    hooks = []
    for l in range(HP.depth):
        hooks.append(net.network._modules[f"EGNN_{l}"].register_forward_hook(
            lambda m, i, o: hs.append((o[0].detach().numpy().copy(), o[1].detach().numpy().copy()))))
    torch.manual_seed(seed)
    out = net(batch, predict=True)
    for h in hooks:
        h.remove()
    snm.get_knn_and_sample = _orig_knn
    k, s = recd.rec[0]
    edges = edges_of(k, s)
    # per-edge feature bins as the reference bins them (score_net_mlsb.py:30-70)
    R = cx["rec_pos"].shape[0]
    center = batch["lig_pos"][:, 1, :].mean(0)
    pos = torch.cat([batch["rec_pos"] - center, batch["lig_pos"] - center], 0)
    sp = snm.get_spatial_matrix(pos).numpy()  # [N,N,100] one-hot
    N = pos.shape[0]
    rows = np.arange(N)[:, None]
    sel = sp[rows, edges]  # [N,K,100]
    bins = np.stack([sel[..., 0:40].argmax(-1), sel[..., 40:64].argmax(-1),
                     sel[..., 64:88].argmax(-1), sel[..., 88:100].argmax(-1)], -1).astype(np.int8)
    rel = batch["position_matrix"].numpy()[rows, edges].argmax(-1).astype(np.int8)
    res = {
        "lig_pos": lig_pos.astype(np.float32), "t": np.float32(t), "edges": edges,
        "bins": bins, "relpos": rel,
        "tr_score": out["tr_score"].detach().numpy(), "rot_score": out["rot_score"].detach().numpy(),
        "energy": out["energy"].detach().numpy(), "f": out["f"].detach().numpy(),
        "num_clashes": np.int64(out["num_clashes"].item()),
        "ires": out["ires"].detach().numpy(),
        "pos_out": hs[-1][1],
        "h_absmean": np.array([np.abs(h).mean() for h, _ in hs], dtype=np.float64),
        "h_absmax": np.array([np.abs(h).max() for h, _ in hs], dtype=np.float64),
        "h_last": hs[-1][0].astype(np.float32),
        "h_first": hs[0][0].astype(np.float32),
    }
    return res


def noised_pose(cx, rng, rot_deg, tr_sigma):
    """Rigidly perturb the ligand about its CA centroid (test poses near native)."""
    lig = cx["lig_pos"].astype(np.float64)
    c = lig[:, 1].mean(0)
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rot_deg)
    Rm = geo.axis_angle_to_matrix(torch.tensor(axis * ang)[None]).numpy()[0]
    out = (lig - c) @ Rm.T + c + rng.standard_normal(3) * tr_sigma
    return out.astype(np.float32)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def flat(prefix, d):
    return {f"{prefix}{k}": v for k, v in d.items()}


# ----------------------------------------------------------------------------
def gen_scalar_kats(model):
    """a-3, a-4, a-14 known answers (r3_diffuser.py:20-55, so3_diffuser.py:210-227,344-369,
    geometry.py:18-200, inference_base.py:311-352)."""
    ts = np.array([1.0, 0.974384606, 0.75, 0.487692297, 0.25, 0.001])
    g_r3 = np.array([model.r3_diffuser.diffusion_coef(t) for t in ts])
    g_so3 = np.array([model.so3_diffuser.diffusion_coef(t) for t in ts])
    s_r3 = np.array([model.r3_diffuser.sigma(t) for t in ts])
    s_so3 = np.array([model.so3_diffuser.sigma(t) for t in ts])
    time_steps = torch.linspace(1.0, 1e-3, 40)
    dt = time_steps[0] - time_steps[1]
    rng = np.random.Generator(np.random.PCG64(7))
    aa = rng.standard_normal((16, 3)).astype(np.float32)
    aa[0] = [0.3, -0.4, 1.2]
    aa[1] = [1e-8, 0, 0]
    aa[2] = [0, 0, 0]
    aa[3] *= 3.0 / np.linalg.norm(aa[3])  # angle 3.0 rad (near pi)
    aa[4] *= 3.14 / np.linalg.norm(aa[4])
    aat = torch.from_numpy(aa)
    mats = geo.axis_angle_to_matrix(aat)
    back = geo.matrix_to_axis_angle(mats)
    quat = geo.matrix_to_quaternion(mats)
    comp = ib.rot_compose(aat[:8], aat[8:])  # R(r2) @ R(r1) -> axis angle
    # torch_reverse with zero noise: perturb = g^2 * s * dt  (f32 rounding as the reference does it)
    score = torch.tensor([[0.25, -0.5, 0.125]])
    rev_r3 = np.stack([model.r3_diffuser.torch_reverse(score_t=score, t=float(t), dt=dt, noise_scale=0.0).numpy()
                       for t in ts])
    rev_so3 = np.stack([model.so3_diffuser.torch_reverse(score_t=score, t=float(t), dt=dt, noise_scale=0.0).numpy()
                        for t in ts])
    # modify_coords (inference_base.py:342-352)
    x = torch.from_numpy(rng.standard_normal((9, 3, 3)).astype(np.float32) * 10)
    rot = torch.tensor([[0.2, -0.1, 0.4]])
    tr = torch.tensor([[1.0, -2.0, 0.5]])
    x2 = ib.modify_coords(x, rot, tr)
    # clash force (inference_base.py:366-384)
    rp = torch.from_numpy(rng.standard_normal((12, 3, 3)).astype(np.float32) * 4)
    lp = torch.from_numpy(rng.standard_normal((10, 3, 3)).astype(np.float32) * 4 + 2.0)
    cf = ib.get_clash_force(rp.clone(), lp.clone())
    save("scalar_kats.npz", ts=ts, g_r3=g_r3, g_so3=g_so3, sigma_r3=s_r3, sigma_so3=s_so3,
         time_steps=time_steps.numpy(), dt=np.float32(dt.item()), sqrt_dt=np.float32(torch.sqrt(dt).item()),
         axis_angle=aa, matrices=mats.numpy(), axis_angle_back=back.numpy(), quaternions=quat.numpy(),
         compose_r1=aa[:8], compose_r2=aa[8:], compose_out=comp.numpy(),
         rev_score=score.numpy(), rev_r3=rev_r3, rev_so3=rev_so3,
         mc_x=x.numpy(), mc_rot=rot.numpy(), mc_tr=tr.numpy(), mc_out=x2.numpy(),
         cf_rec=rp.numpy(), cf_lig=lp.numpy(), cf_out=cf.numpy())


def gen_geometry(cx_small):
    """a-6, a-7, a-8, a-10 on a small synthetic complex: full N x N tables."""
    batch = make_batch(cx_small)
    center = batch["lig_pos"][:, 1, :].mean(0)
    pos = torch.cat([batch["rec_pos"] - center, batch["lig_pos"] - center], 0)
    dist, omega, theta, phi = get_coords6d(pos)
    sp = snm.get_spatial_matrix(pos).numpy()
    bins = np.stack([sp[..., 0:40].argmax(-1), sp[..., 40:64].argmax(-1),
                     sp[..., 64:88].argmax(-1), sp[..., 88:100].argmax(-1)], -1).astype(np.int8)
    assert np.all(sp.sum(-1) == 4)
    rel = batch["position_matrix"].numpy().argmax(-1).astype(np.int8)
    torch.manual_seed(3)
    k, s = _orig_knn(pos[:, 1, :])
    dm = torch.cdist(pos[:, 1, :], pos[:, 1, :])
    save("geometry_small.npz", rec_pos=cx_small["rec_pos"], lig_pos=cx_small["lig_pos"],
         pos_centered=pos.numpy(), dist=dist.numpy(), omega=omega.numpy(), theta=theta.numpy(),
         phi=phi.numpy(), bins=bins, relpos=rel, knn=k.numpy().astype(np.int32),
         sampled=s.numpy().astype(np.int32), cdist=dm.numpy())


def gen_forward_cases(net):
    rng = np.random.Generator(np.random.PCG64(11))
    # (1) N < 60 edge case: 24 + 16 residues -> 20 kNN + 20 sampled (score_net_mlsb.py:89-94)
    cx = make_complex(24, 16, seed=5)
    r = forward_case(net, cx, cx["lig_pos"], 0.5, seed=1)
    save("fwd_syn_24_16.npz", R=24, L=16, cx_seed=5, **r)
    # (1b) N < 20: 9 + 7 residues -> K = N, no sampling
    cx = make_complex(9, 7, seed=6)
    r = forward_case(net, cx, cx["lig_pos"], 0.3, seed=1)
    save("fwd_syn_9_7.npz", R=9, L=7, cx_seed=6, **r)
    # (2) 64 + 48
    cx = make_complex(64, 48, seed=7)
    for i, (t, rot, trs) in enumerate([(1.0, 40.0, 6.0), (0.49, 10.0, 2.0), (0.001, 0.0, 0.0)]):
        lp = noised_pose(cx, rng, rot, trs)
        r = forward_case(net, cx, lp, t, seed=10 + i)
        save(f"fwd_syn_64_48_p{i}.npz", R=64, L=48, cx_seed=7, **r)
    # (3) DB5 7CEI (87 + 127), ESM block stored as float16 (both sides use the rounded values)
    d = load_db5_pt(os.path.join(REF, "data/db5_test/7CEI.pt"))
    esm16_r = d["rec_esm"].astype(np.float16)
    esm16_l = d["lig_esm"].astype(np.float16)
    cx = {"rec_x": np.concatenate([esm16_r.astype(np.float32), d["rec_x"][:, 1280:]], 1),
          "lig_x": np.concatenate([esm16_l.astype(np.float32), d["lig_x"][:, 1280:]], 1),
          "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
    save("cx_7CEI.npz", rec_esm16=esm16_r, lig_esm16=esm16_l, rec_seq=d["rec_seq"], lig_seq=d["lig_seq"],
         rec_pos=d["rec_pos"], lig_pos=d["lig_pos"])
    poses = [(0.001, 0.0, 0.0), (1.0, 60.0, 8.0), (0.49, 15.0, 3.0), (0.05, 4.0, 0.7)]
    for i, (t, rot, trs) in enumerate(poses):
        lp = noised_pose(cx, rng, rot, trs) if i else cx["lig_pos"]
        r = forward_case(net, cx, lp, t, seed=20 + i)
        save(f"fwd_7CEI_p{i}.npz", **r)
    # metrics KAT (metrics.py:3-121)
    nat = (torch.from_numpy(d["rec_pos"]), torch.from_numpy(d["lig_pos"]))
    m0 = compute_metrics([nat[0].clone(), nat[1].clone()], nat)
    sh = nat[1].clone()
    sh[..., 0] += 5.0
    m1 = compute_metrics([nat[0].clone(), sh], nat)
    lp = torch.from_numpy(noised_pose(cx, rng, 25.0, 4.0))
    m2 = compute_metrics([nat[0].clone(), lp], nat)
    keys = ["c_rmsd", "i_rmsd", "l_rmsd", "fnat", "DockQ"]
    save("metrics_7CEI.npz", keys=np.array(keys), native=np.array([m0[k] for k in keys]),
         shifted=np.array([m1[k] for k in keys]), noised=np.array([m2[k] for k in keys]),
         noised_lig=lp.numpy())
    return cx


def gen_rollout(model, cx, name, num_steps, seed):
    """Full reference Euler_Maruyama_sampler run (inference_base.py:390-468) with every random
    draw recorded: R0, the N(0,30^2) draw, per-step z_rot / z_tr, per-forward edge lists; plus the
    ligand pose after every step and the per-step scores for teacher forcing."""
    batch = make_batch(cx)
    rec = {"R0": None, "tr_draw": None, "z": [], "poses": [], "scores": []}
    recd = EdgeRecorder()
    snm.get_knn_and_sample = recd

    orig_Rotation = ib.Rotation
    orig_random = ib.Rotation.random
    orig_normal = torch.normal
    orig_randn = torch.randn
    orig_modify = ib.modify_coords
    orig_randomize = ib.randomize_pose

    def randomize(x1, x2):
        out = orig_randomize(x1, x2)
        rec["init_pose"] = out[0].numpy().copy()
        rec["init_tr"] = out[1].numpy().copy()
        rec["init_rot"] = out[2].numpy().copy()
        return out

    def rot_random(*a, **k):
        r = orig_random(*a, **k)
        rec["R0"] = r.as_matrix().copy()
        return r

    def normal(*a, **k):
        v = orig_normal(*a, **k)
        rec["tr_draw"] = v.numpy().copy()
        return v

    def randn(*a, **k):
        v = orig_randn(*a, **k)
        rec["z"].append(v.numpy().copy())
        return v

    def modify(x, rot, tr):
        y = orig_modify(x, rot, tr)
        rec["poses"].append(y.numpy().copy())
        rec["scores"].append((rot.numpy().copy(), tr.numpy().copy()))
        return y

    class Wrapped(nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m
            self.r3_diffuser = m.r3_diffuser
            self.so3_diffuser = m.so3_diffuser
            self.outs = []

        def forward(self, b):
            o = self.m(b)
            self.outs.append({k: o[k].detach().numpy().copy() for k in ("tr_score", "rot_score", "energy", "f")})
            self.outs[-1]["num_clashes"] = int(o["num_clashes"].item())
            return o

    wm = Wrapped(model)
    ib.Rotation = NS(random=rot_random)
    torch.normal = normal
    torch.randn = randn
    ib.modify_coords = modify
    ib.randomize_pose = randomize
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        rec_pos, lig_pos, rot_update, tr_update, output = ib.Euler_Maruyama_sampler(
            model=wm, batch=dict(batch), num_steps=num_steps, device="cpu")
    finally:
        ib.Rotation = orig_Rotation
        torch.normal = orig_normal
        torch.randn = orig_randn
        ib.modify_coords = orig_modify
        ib.randomize_pose = orig_randomize
        snm.get_knn_and_sample = _orig_knn
    z = np.concatenate(rec["z"], 0).reshape(num_steps, 2, 3)  # [step, (so3, r3), 3]
    edges = np.stack([edges_of(k, s) for k, s in recd.rec])   # [steps+1, N, K]
    R = cx["rec_pos"].shape[0]
    arrs = dict(
        R=R, L=cx["lig_pos"].shape[0], num_steps=num_steps,
        R0=rec["R0"].astype(np.float64), tr_draw=rec["tr_draw"].astype(np.float32),
        z_rot=z[:, 0].astype(np.float32), z_tr=z[:, 1].astype(np.float32), edges=edges,
        poses=np.stack(rec["poses"]).astype(np.float32),           # pose AFTER step i
        init_pose=rec["init_pose"].astype(np.float32), init_tr=rec["init_tr"], init_rot=rec["init_rot"],
        step_rot=np.stack([a for a, _ in rec["scores"]])[:, 0], step_tr=np.stack([b for _, b in rec["scores"]])[:, 0],
        tr_score=np.stack([o["tr_score"][0] for o in wm.outs]), rot_score=np.stack([o["rot_score"][0] for o in wm.outs]),
        energy=np.array([o["energy"] for o in wm.outs]), num_clashes=np.array([o["num_clashes"] for o in wm.outs]),
        final_lig_pos=lig_pos.numpy(), rot_update=rot_update.numpy(), tr_update=tr_update.numpy(),
        final_energy=np.float32(output["energy"].item()), final_num_clashes=np.int64(output["num_clashes"].item()),
    )
    save(name, **arrs)


def gen_io_kats(cx7):
    """f-1: reference DB5-set driver pieces that are pure functions: get_full_coords (inference_mlsb.py:68-85),
    save_PDB text (utils/pdb.py:59-84), random_rotation (ppi_dataset.py:212-219) for a recorded rotation."""
    import tempfile
    import inference_mlsb as im
    from utils.pdb import save_PDB
    import datasets.ppi_dataset as pds
    coords = torch.from_numpy(cx7["rec_pos"][:12].copy())
    full = im.get_full_coords(coords)
    seq = "MELKNSISDYTG"
    with tempfile.NamedTemporaryFile("r", suffix=".pdb") as f:
        save_PDB(out_pdb=f.name, coords=full, seq=seq, delim=6)
        text = open(f.name).read()
    rec = {}
    orig = pds.Rotation

    class R:
        @staticmethod
        def random():
            r = orig.random()
            rec["q"] = r.as_quat().copy()     # scalar-last
            rec["m"] = r.as_matrix().copy()
            return r
    pds.Rotation = R
    np.random.seed(5)
    rp, lp = pds.random_rotation(torch.from_numpy(cx7["rec_pos"].copy()), torch.from_numpy(cx7["lig_pos"].copy()))
    pds.Rotation = orig
    save("io_kats.npz", coords=coords.numpy(), full=full.numpy(), seq=seq, pdb_text=text, rot_quat=rec["q"], rot_mat=rec["m"],
         rot_rec=rp.numpy(), rot_lig=lp.numpy())


def main():
    net = build_net(0)
    model = Model(net).eval()
    gen_scalar_kats(model)
    gen_geometry(make_complex(40, 30, seed=3))
    cx7 = gen_forward_cases(net)
    gen_io_kats(cx7)
    cxs = make_complex(24, 16, seed=5)
    gen_rollout(model, cxs, "rollout_syn_24_16.npz", num_steps=40, seed=123)
    gen_rollout(model, make_complex(64, 48, seed=7), "rollout_syn_64_48.npz", num_steps=40, seed=321)
    gen_rollout(model, cx7, "rollout_7CEI.npz", num_steps=6, seed=77)


if __name__ == "__main__":
    main()

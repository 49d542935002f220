"""Checkpoint-driven command line of the sampling engine.

    python -m dfmdock_amd dock REC.pdb LIG.pdb --ckpt model_0.ckpt --features F.npz [--num-samples 120] [--out output.pdb]
    python -m dfmdock_amd sweep --db5 data/db5_test --ckpt model_0.ckpt [--num-samples 40] [--out-csv results.csv]
    python -m dfmdock_amd selfcheck REC.pdb LIG.pdb --ckpt model_0.ckpt --features F.npz

  dock       <- src/inference_single.py:1-12 -> inference() (src/inference_base.py:601-670): num_samples (120) trajectories of
                num_steps (40), the minimum-energy one applied to the all-atom ligand, `output.pdb` written, {"energy": E} printed.
                The reference embeds both sequences with ESM-2 650M inside inference() (:606-609, get_esm_rep); that language
                model is outside this engine's scope, so its per-residue representations come in as
                a file: --features F.npz with `rec_esm [R,1280]`, `lig_esm [L,1280]` (the one-hot block is appended here from
                the PDB's own sequence, inference_base.py:192-215) or ready `rec_x [R,1301]`, `lig_x [L,1301]`.
  sweep      <- inference_mlsb.Sampler.run_sampling over PPIDataset('db5_test') (src/inference_mlsb.py:188-262,
                src/datasets/ppi_dataset.py:224-329): every `<id>.pt` of --db5 (ids from `test.txt` when present), the
                reference-schema CSV (src/inference_base.py:495-499) and the DockQ success-rate table.
  selfcheck  no reference counterpart: dfm_complex_selfcheck on the pair (what `dock` and `sweep` run once per complex anyway).

--ckpt takes the Lightning checkpoint the reference loads (src/inference_base.py:611-616; read without Lightning / omegaconf by
weights.load_lightning_checkpoint) or a bare state_dict; the model family (Score_Net / EGNN_Net) is read off the keys.
Everything computes on the MI355X through the C ABI; without a GPU the commands fail (no CPU fallback).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

DOCKQ_THRESHOLDS = (("acceptable", 0.23), ("medium", 0.49), ("high", 0.80))      # CAPRI classes by DockQ


def _add_common(p):
    p.add_argument("--ckpt", required=True, help="Lightning checkpoint or bare state_dict (torch.save)")
    p.add_argument("--num-steps", type=int, default=40)
    p.add_argument("--precision", default="mfma16", choices=["mfma16", "f16", "fp32", "bf16"],
                   help="mfma16 (default): 16-bit MFMA engine, self-checked against fp32 once per complex; fp32: exact; bf16: deprecated alias of mfma16")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK, else 0)")
    p.add_argument("--max-batch", type=int, default=256, help="trajectories in flight per dfm_sample call")
    p.add_argument("--no-selfcheck", action="store_true", help="skip the per-complex fp32-vs-16-bit check")
    p.add_argument("--on-selfcheck-fail", default="fp32", choices=["fp32", "raise", "warn"],
                   help="a failed check switches the complex to the fp32 engine (default), aborts, or only warns")


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m dfmdock_amd", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("dock", help="dock one receptor / ligand PDB pair (inference_single.py)")
    d.add_argument("pdb_1", help="receptor PDB")
    d.add_argument("pdb_2", help="ligand PDB")
    d.add_argument("--features", required=True, help=".npz with rec_esm / lig_esm [n,1280] (or rec_x / lig_x [n,1301])")
    d.add_argument("--num-samples", type=int, default=120)
    d.add_argument("--out", default="output.pdb")
    d.add_argument("--json", default=None, help="also write the result line to this file")
    _add_common(d)
    s = sub.add_parser("sweep", help="sample every complex of a DB5-style directory (inference_mlsb.py run_sampling)")
    s.add_argument("--db5", required=True, help="directory of <id>.pt files (+ optional test.txt with the ids to run)")
    s.add_argument("--num-samples", type=int, default=40)
    s.add_argument("--out-csv", default="results.csv")
    s.add_argument("--summary", default=None, help="write the success-rate table + self-check records as JSON")
    s.add_argument("--traj-dir", default=None, help="write one multi-MODEL PDB per trajectory (save_trj)")
    s.add_argument("--no-global-rotation", action="store_true", help="skip the loader's random rotation (ppi_dataset.py:309)")
    s.add_argument("--limit", type=int, default=None, help="first N ids only")
    _add_common(s)
    c = sub.add_parser("selfcheck", help="fp32-vs-16-bit check + fp16 range telemetry of one pair")
    c.add_argument("pdb_1")
    c.add_argument("pdb_2")
    c.add_argument("--features", required=True)
    c.add_argument("--n-eval", type=int, default=4)
    _add_common(c)
    return ap


def load_model(args):
    from . import distributed as D
    from . import engine
    from .weights import load_lightning_checkpoint, pack_blob
    dev = args.device if args.device is not None else D.dist_env()[1]
    engine.set_device(dev)
    sd, hp = load_lightning_checkpoint(args.ckpt)
    return engine.Model(pack_blob(sd, hp), hp), hp


def load_pair(pdb_1, pdb_2, features, lm_embed_dim=1301):
    """Two PDB files + the features file -> (rec, lig, rec_x, lig_x); the residue counts of the features must match the residues
    get_info_from_pdb keeps (N, CA and C present, HETATM dropped - pdbio.backbone_from_atoms)."""
    from . import pdbio
    from .synthetic import seq_to_onehot
    rec = pdbio.backbone_from_atoms(pdbio.read_pdb(pdb_1))
    lig = pdbio.backbone_from_atoms(pdbio.read_pdb(pdb_2))
    f = np.load(features, allow_pickle=False)
    xs = []
    for side, chain in (("rec", rec), ("lig", lig)):
        n = len(chain["seq"])
        if side + "_x" in f:
            x = np.asarray(f[side + "_x"], np.float32)
        elif side + "_esm" in f:
            esm = np.asarray(f[side + "_esm"], np.float32)
            if esm.ndim != 2 or esm.shape[0] != n:
                raise ValueError(f"{features}: {side}_esm is {esm.shape}, the PDB has {n} residues with N, CA and C")
            x = np.concatenate([esm, seq_to_onehot(chain["seq"])], axis=1)
        else:
            raise ValueError(f"{features}: need `{side}_esm` [n,1280] or `{side}_x` [n,{lm_embed_dim}]")
        if x.shape != (n, lm_embed_dim):
            raise ValueError(f"{features}: {side} features are {x.shape}, the PDB has {n} residues with N, CA and C "
                             f"(expected ({n}, {lm_embed_dim}))")
        if side + "_seq" in f and str(f[side + "_seq"]) != chain["seq"]:
            raise ValueError(f"{features}: {side}_seq does not match the sequence read from the PDB")
        xs.append(x)
    return rec, lig, xs[0], xs[1]


def success_table(rows):
    """Per complex: DockQ of the minimum-energy trajectory (what inference() keeps) and the best DockQ among its trajectories;
    success rates at the CAPRI thresholds."""
    by = {}
    for r in rows:
        by.setdefault(r["id"], []).append(r)
    per = {}
    for cid, rs in sorted(by.items()):
        top = min(rs, key=lambda r: (r["energy"], int(r["index"])))
        per[cid] = {"n": len(rs), "top1_DockQ": float(top["DockQ"]), "top1_energy": float(top["energy"]),
                    "best_DockQ": float(max(r["DockQ"] for r in rs)), "mean_DockQ": float(np.mean([r["DockQ"] for r in rs]))}
    n = max(len(per), 1)
    table = {name: {"threshold": thr, "top1": sum(p["top1_DockQ"] >= thr for p in per.values()) / n,
                    "best_of_n": sum(p["best_DockQ"] >= thr for p in per.values()) / n} for name, thr in DOCKQ_THRESHOLDS}
    return per, table


def format_table(per, table):
    lines = [f"{'id':8s} {'n':>4s} {'top1 DockQ':>11s} {'best DockQ':>11s} {'top1 energy':>12s}"]
    for cid, p in per.items():
        lines.append(f"{cid:8s} {p['n']:4d} {p['top1_DockQ']:11.4f} {p['best_DockQ']:11.4f} {p['top1_energy']:12.4f}")
    lines.append(f"success rate over {len(per)} complexes (DockQ of the minimum-energy trajectory | best of the trajectories):")
    for name, t in table.items():
        lines.append(f"  {name:10s} DockQ >= {t['threshold']:.2f}: {100 * t['top1']:5.1f} % | {100 * t['best_of_n']:5.1f} %")
    return "\n".join(lines)


def cmd_dock(args):
    from . import driver
    model, _ = load_model(args)
    rec, lig, rec_x, lig_x = load_pair(args.pdb_1, args.pdb_2, args.features, model.hp.lm_embed_dim)
    res = driver.dock_pair(model, rec, lig, rec_x, lig_x, num_samples=args.num_samples, num_steps=args.num_steps, seed=args.seed,
                           precision=args.precision, out_pdb=args.out, max_batch=args.max_batch, selfcheck=not args.no_selfcheck,
                           on_selfcheck_fail=args.on_selfcheck_fail)
    line = {"energy": res["energy"], "output": os.path.abspath(args.out), "num_samples": args.num_samples, "precision": res["precision"],
            "rot_update": [float(v) for v in res["rot_update"]], "tr_update": [float(v) for v in res["tr_update"]],
            "selfcheck_ok": None if res["selfcheck"] is None else bool(res["selfcheck"]["ok"])}
    print(json.dumps(line), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(line, selfcheck=res["selfcheck"]), f, default=float)
    return 0


def db5_ids(root, limit=None):
    lst = os.path.join(root, "test.txt")
    if os.path.exists(lst):
        ids = [l.strip() for l in open(lst) if l.strip()]
    else:
        ids = sorted(f[:-3] for f in os.listdir(root) if f.endswith(".pt"))
    present = [i for i in ids if os.path.exists(os.path.join(root, i + ".pt"))]
    for i in ids:
        if i not in present:
            print(f"sweep: {i}.pt is listed but missing - skipped", file=sys.stderr)
    return present[:limit] if limit else present


def cmd_sweep(args):
    from . import distributed as D
    from . import driver
    from .db5 import load_db5_pt
    model, _ = load_model(args)
    rank, local_rank, world = D.dist_env()
    grp = D.init(device_index=args.device if args.device is not None else local_rank) if world > 1 else None
    cxs = []
    for cid in db5_ids(args.db5, args.limit):
        c = load_db5_pt(os.path.join(args.db5, cid + ".pt"))
        c["id"] = c.get("id") or cid
        cxs.append(c)
    if not cxs:
        raise SystemExit(f"sweep: no <id>.pt under {args.db5}")
    checks = []
    rows, _ = driver.run_set(model, cxs, num_samples=args.num_samples, num_steps=args.num_steps, seed=args.seed,
                             precision=args.precision, global_rotation=not args.no_global_rotation, out_csv=args.out_csv,
                             traj_dir=args.traj_dir, max_batch=args.max_batch, selfcheck=not args.no_selfcheck,
                             on_selfcheck_fail=args.on_selfcheck_fail, checks_out=checks)
    all_rows = driver._gather_rows(rows, world)
    all_checks = [c for part in D.gather_objects(checks) for c in part] if world > 1 else checks
    if rank == 0:
        per, table = success_table(all_rows)
        print(format_table(per, table), flush=True)
        if args.summary:
            with open(args.summary, "w") as f:
                json.dump({"csv": os.path.abspath(args.out_csv), "complexes": per, "success": table, "selfcheck": all_checks,
                           "world": world, "backend": grp.backend if grp else "single"}, f, default=float, indent=1)
    if world > 1:
        D.shutdown()
    return 0


def cmd_selfcheck(args):
    from . import engine
    model, _ = load_model(args)
    rec, lig, rec_x, lig_x = load_pair(args.pdb_1, args.pdb_2, args.features, model.hp.lm_embed_dim)
    gx = engine.Complex(model, rec_x, lig_x, rec["bb_coords"], lig["bb_coords"])
    prec = args.precision if engine.canonical_precision(args.precision) != "fp32" else "mfma16"
    r = gx.selfcheck(n_eval=args.n_eval, seed=args.seed, precision=prec)
    print(engine.format_selfcheck(r, os.path.basename(args.pdb_1) + "+" + os.path.basename(args.pdb_2)), file=sys.stderr)
    print(json.dumps(r, default=float), flush=True)
    return 0 if r["ok"] else 1


def main(argv=None):
    args = build_parser().parse_args(argv)
    return {"dock": cmd_dock, "sweep": cmd_sweep, "selfcheck": cmd_selfcheck}[args.cmd](args)

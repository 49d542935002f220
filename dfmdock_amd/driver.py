"""Drivers around the sampling engine: the DB5-set sweep and the single-pair run.

  run_set      <- inference_mlsb.Sampler.run_sampling (src/inference_mlsb.py:188-262) over
                  PPIDataset('db5_test', training=False) (src/datasets/ppi_dataset.py:224-329): ESM || one-hot
                  features, relpos from (R, L), and the RANDOM GLOBAL ROTATION the loader applies even at test time
                  (:212-219, :309); CSV rows {id,index,c_rmsd,i_rmsd,l_rmsd,fnat,DockQ,energy,num_clashes}
                  (src/inference_base.py:495-499,:589-596)
  dock_pair    <- inference() (src/inference_base.py:601-670): num_samples trajectories, keep the minimum energy,
                  apply its (rot, tr) to the all-atom ligand, write output.pdb

Work is sharded over ranks by complexes (longest first, distributed.assign_work) - or, with fewer than two complexes per
rank, by TRAJECTORIES: every rank then samples its block of each complex's trajectories (all ranks hold all complexes: weights
7 MB + <= 3.6 MB of features per complex), so that a short list of complexes still fills every GPU with one large batch
instead of leaving ranks idle.  The energy-ranked records are gathered with the single collective of
distributed.gather_records.
"""
from __future__ import annotations

import csv
import os

import numpy as np

from . import distributed as D
from . import engine, pdbio
from .metrics import NativeContext, compute_metrics

CSV_FIELDS = ["id", "index", "c_rmsd", "i_rmsd", "l_rmsd", "fnat", "DockQ", "energy", "num_clashes"]


def rotate_complex(rec_pos, lig_pos, Rm):
    """ppi_dataset.py:212-219: rotate the whole complex about its joint CA centroid (which also centres it)."""
    pos = np.concatenate([rec_pos, lig_pos], 0).astype(np.float32)
    cen = pos[:, 1, :].mean(0)
    pos = (pos - cen) @ np.asarray(Rm, np.float32).T
    return pos[: rec_pos.shape[0]], pos[rec_pos.shape[0]:]


def random_rotation(rec_pos, lig_pos, rng):
    """The loader's test-time augmentation with a Haar-uniform rotation (scipy Rotation.random(): normalised
    Gaussian quaternion, scalar last)."""
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return rotate_complex(rec_pos, lig_pos, Rm)


def checked_precision(gx: engine.Complex, precision: str, name: str, selfcheck=True, on_fail="fp32", log=None, seed=0):
    """Once per complex, before any trajectory: dfm_complex_selfcheck of the 16-bit engine on the complex's own pose (fp32 vs
    16-bit on the same graphs + fp16 range telemetry), printed as one line.  A failed check - activations outside the fp16 range
    or deviations beyond SURVEY 8(d)'s gates, i.e. a checkpoint this engine's precision plan does not hold for - switches the
    complex to the fp32 ENGINE (`on_fail="fp32"`: slower, the reference's own arithmetic, still the HIP path), raises
    (`"raise"`) or only warns (`"warn"`).  Returns (precision to use, the check's dict or None)."""
    import sys
    if not selfcheck or engine.canonical_precision(precision) == "fp32":
        return precision, None
    r = gx.selfcheck(precision=precision, seed=seed)
    line = engine.format_selfcheck(r, name)
    (log or (lambda m: print(m, file=sys.stderr, flush=True)))(line)
    if not r["ok"]:
        if on_fail == "raise":
            raise RuntimeError(line)
        if on_fail == "fp32":
            (log or (lambda m: print(m, file=sys.stderr, flush=True)))(f"selfcheck {name}: running this complex on the fp32 engine")
            return "fp32", r
    return precision, r


def run_set(model: engine.Model, complexes, num_samples=40, num_steps=40, seed=0, precision="mfma16", global_rotation=True,
            out_csv=None, traj_dir=None, max_batch=256, selfcheck=True, on_selfcheck_fail="fp32", checks_out=None, **sampler_kw):
    """Sample `num_samples` trajectories for every complex dict (id, rec_x, lig_x, rec_pos, lig_pos[, rec_seq, lig_seq]);
    returns the metric rows of this rank's share; rank 0 writes the gathered CSV when `out_csv` is given.  Every complex is
    self-checked first (checked_precision); `checks_out` (a list) collects {id, precision used, check dict}."""
    rank, _, world = D.dist_env()
    complexes = list(complexes)
    split_trajectories = world > 1 and len(complexes) < 2 * world
    if split_trajectories:      # (complex, trajectory block) work items: rank r takes block r of every complex
        share = list(range(len(complexes)))
        t_lo, t_hi = D.shard_range(num_samples, world, rank)
    else:
        share = D.assign_work([c["rec_x"].shape[0] + c["lig_x"].shape[0] for c in complexes], world)[rank]
        t_lo, t_hi = 0, num_samples
    rng = np.random.default_rng(seed)
    rots = [rng.integers(0, 2 ** 31) for _ in complexes]      # per-complex streams, identical on every rank
    rows, records = [], []
    for ci in share:
        if t_hi <= t_lo:
            break
        c = complexes[ci]
        rec_pos, lig_pos = np.asarray(c["rec_pos"], np.float32), np.asarray(c["lig_pos"], np.float32)
        if global_rotation:
            rec_pos, lig_pos = random_rotation(rec_pos, lig_pos, np.random.default_rng(rots[ci]))
        gx = engine.Complex(model, c["rec_x"], c["lig_x"], rec_pos, lig_pos)
        native = NativeContext((rec_pos, lig_pos))
        use_prec, chk = checked_precision(gx, precision, str(c.get("id", ci)), selfcheck, on_selfcheck_fail, seed=seed)
        if checks_out is not None:
            checks_out.append({"id": c.get("id", str(ci)), "precision": use_prec, "selfcheck": chk})
        done = t_lo
        while done < t_hi:
            b = min(max_batch, t_hi - done)
            r = gx.sample(B=b, num_steps=num_steps, seed=seed * 100003 + ci * 1009 + done, trace=traj_dir is not None,
                          **engine.precision_kwargs(use_prec), **sampler_kw)
            for k in range(b):
                m = compute_metrics((rec_pos, r["lig_pos"][k]), (rec_pos, lig_pos), native)
                rows.append({"id": c.get("id", str(ci)), "index": str(done + k), **m, "energy": float(r["energy"][k]),
                             "num_clashes": int(r["num_clashes"][k])})
                if traj_dir is not None and "rec_seq" in c:
                    os.makedirs(traj_dir, exist_ok=True)
                    frames = r["trace_pose"][k]
                    pdbio.write_trajectory_pdb(os.path.join(traj_dir, f"{c.get('id', ci)}_p{done + k}.pdb"),
                                               [rec_pos] * len(frames), frames, c["rec_seq"], c["lig_seq"])
            records.append(D.make_records(ci, np.arange(done, done + b), r))
            done += b
        gx.close()
    recs = np.concatenate(records, 0) if records else np.zeros((0, D.RECORD_WIDTH), np.float32)
    ranked = D.rank_by_energy(D.gather_records(recs)) if world > 1 or len(recs) else {}
    if out_csv is not None:
        all_rows = _gather_rows(rows, world)
        if rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(out_csv)), exist_ok=True)
            with open(out_csv, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=CSV_FIELDS)
                w.writeheader()
                for row in sorted(all_rows, key=lambda x: (x["id"], int(x["index"]))):
                    w.writerow(row)
    return rows, ranked


def _gather_rows(rows, world):
    if world == 1:
        return rows
    return [r for part in D.gather_objects(rows) for r in part]


def dock_pair(model: engine.Model, rec, lig, rec_x, lig_x, num_samples=120, num_steps=40, seed=0, precision="mfma16",
              out_pdb="output.pdb", max_batch=256, selfcheck=True, on_selfcheck_fail="fp32"):
    """inference() of the reference for two parsed PDB chains (pdbio.backbone_from_atoms dicts) and their
    pre-computed node features; returns {'energy': min energy} and writes the best pose."""
    gx = engine.Complex(model, rec_x, lig_x, rec["bb_coords"], lig["bb_coords"])
    precision, chk = checked_precision(gx, precision, "pair", selfcheck, on_selfcheck_fail, seed=seed)
    best = None
    done = 0
    while done < num_samples:
        b = min(max_batch, num_samples - done)
        r = gx.sample(B=b, num_steps=num_steps, seed=seed + done, **engine.precision_kwargs(precision))
        k = int(np.argmin(r["energy"]))
        if best is None or r["energy"][k] < best[0]:     # strict <: the first minimum wins, as in the reference
            best = (float(r["energy"][k]), r["rot_update"][k].copy(), r["tr_update"][k].copy())
        done += b
    gx.close()
    lig_aa = pdbio.apply_pose_all_atom(lig["aa_coords"], lig["bb_coords"], best[1], best[2],
                                       center="all_atoms" if model.hp.family == 1 else "ca")
    if out_pdb:
        rec_atoms = [a for a in rec["atoms"]]
        pdbio.write_complex_pdb(out_pdb, rec_atoms, lig["atoms"], lig_aa)
    return {"energy": best[0], "rot_update": best[1], "tr_update": best[2], "lig_aa_coords": lig_aa, "precision": precision,
            "selfcheck": chk}

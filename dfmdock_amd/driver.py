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


class _Prepared:
    """One complex ready to sample: its handle, the pose it was created on and everything derived from that pose alone."""
    __slots__ = ("ci", "c", "gx", "N", "rec_pos", "lig_pos", "native", "precision", "check", "ms")


def _prepare(model, c, ci, rot_seed, global_rotation, precision, selfcheck, on_selfcheck_fail, seed, log=None) -> _Prepared:
    """Stage 1 of run_set (host + a little device work on the new handle's own stream): loader semantics, handle, self-check."""
    import time
    t0 = time.perf_counter()
    p = _Prepared()
    p.ci, p.c = ci, c
    rec_pos, lig_pos = np.asarray(c["rec_pos"], np.float32), np.asarray(c["lig_pos"], np.float32)
    if global_rotation:
        rec_pos, lig_pos = random_rotation(rec_pos, lig_pos, np.random.default_rng(rot_seed))
    p.rec_pos, p.lig_pos = rec_pos, lig_pos
    p.gx = engine.Complex(model, c["rec_x"], c["lig_x"], rec_pos, lig_pos)
    p.N = p.gx.N
    p.native = NativeContext((rec_pos, lig_pos))
    p.precision, p.check = checked_precision(p.gx, precision, str(c.get("id", ci)), selfcheck, on_selfcheck_fail, log=log, seed=seed)
    p.ms = {"prepare": (time.perf_counter() - t0) * 1e3}
    return p


def _sample(p: _Prepared, t_lo, t_hi, num_steps, seed, max_batch, trace, sampler_kw, keep_open=False):
    """Stage 2: the trajectories of this rank's share, in batches of at most max_batch.  The results are host arrays, so the
    handle (and its ~GB of device workspace) is released HERE, not after the host-side stage 3: however far the metrics /
    trajectory-PDB stage falls behind, the live handles are bounded by the prepare gate + the samplers."""
    import time
    t0 = time.perf_counter()
    batches, done = [], t_lo
    try:
        while done < t_hi:
            b = min(max_batch, t_hi - done)
            r = p.gx.sample(B=b, num_steps=num_steps, seed=seed * 100003 + p.ci * 1009 + done, trace=trace,
                            **engine.precision_kwargs(p.precision), **sampler_kw)
            batches.append((done, b, r))
            done += b
    finally:
        if not keep_open:
            p.gx.close()
    p.ms["sample"] = (time.perf_counter() - t0) * 1e3
    return batches


def _post(p: _Prepared, batches, traj_dir):
    """Stage 3 (host only): per-trajectory metrics against the native pose (inference_mlsb.py:232-262), trajectory PDBs, records."""
    import time
    t0 = time.perf_counter()
    c, rows, records = p.c, [], []
    for done, b, r in batches:
        for k in range(b):
            m = compute_metrics((p.rec_pos, r["lig_pos"][k]), (p.rec_pos, p.lig_pos), p.native)
            rows.append({"id": c.get("id", str(p.ci)), "index": str(done + k), **m, "energy": float(r["energy"][k]),
                         "num_clashes": int(r["num_clashes"][k])})
            if traj_dir is not None and "rec_seq" in c:
                os.makedirs(traj_dir, exist_ok=True)
                frames = r["trace_pose"][k]
                pdbio.write_trajectory_pdb(os.path.join(traj_dir, f"{c.get('id', p.ci)}_p{done + k}.pdb"),
                                           [p.rec_pos] * len(frames), frames, c["rec_seq"], c["lig_seq"])
        records.append(D.make_records(p.ci, np.arange(done, done + b), r))
    p.ms["post"] = (time.perf_counter() - t0) * 1e3
    return rows, records


def run_set(model: engine.Model, complexes, num_samples=40, num_steps=40, seed=0, precision="mfma16", global_rotation=True,
            out_csv=None, traj_dir=None, max_batch=256, selfcheck=True, on_selfcheck_fail="fp32", checks_out=None,
            overlap=True, samplers=2, timings_out=None, log=None, canary=True, canary_out=None, **sampler_kw):
    """Sample `num_samples` trajectories for every complex dict (id, rec_x, lig_x, rec_pos, lig_pos[, rec_seq, lig_seq]);
    returns the metric rows of this rank's share; rank 0 writes the gathered CSV when `out_csv` is given.  Every complex is
    self-checked first (checked_precision); `checks_out` (a list) collects {id, precision used, check dict}.

    The reference's loop (src/inference_mlsb.py:415-439 over :188-262) is serial: load, sample, score, next.  Here the three
    stages of a complex run on three host threads (`overlap=True`): while complex k samples on its handle's stream, complex k+1
    is created and self-checked on ITS handle's stream (dfmdock_amd.h: handles are independent) and complex k-1's 40 Kabsch
    fits / CSV rows are computed on the host; `samplers` complexes sample concurrently (default 2: a B = 40 batch of a 200-400-residue
    complex does not fill 256 CUs; measured +4 ... +8 % over one sampler on the C4 set, profiles/r05_c4.txt; 3 buys nothing more).  Results do not depend on any of this: a trajectory is a pure function of (seed, complex,
    trajectory index), so the rows equal the serial driver's (`overlap=False`) bit for bit.  `timings_out` (a list) collects
    per-complex {id, N, prepare, sample, post} milliseconds; `log` receives the self-check lines
    (default: stderr).

    `canary` (overlapped driver only): r05 found - and fenced, without naming the mechanism - a silent miscompute between two handles
    running at once (profiles/r05_concurrency.txt; tests/test_gpu_concurrency.py holds the shipped build to 0 deviations).  As a
    run-time tripwire on whatever box / ROCm this runs on, the cheapest complex that sampled with others in flight is sampled AGAIN
    after the pipeline has drained, alone, and compared bit for bit; on a mismatch the whole share is re-run by the serial driver and
    those rows are returned (`canary_out`, a dict, receives {checked, id, ok, reran_serial})."""
    rank, _, world = D.dist_env()
    complexes = list(complexes)
    split_trajectories = world > 1 and len(complexes) < 2 * world
    if split_trajectories:      # (complex, trajectory block) work items: rank r takes block r of every complex
        share = list(range(len(complexes)))
        t_lo, t_hi = D.shard_range(num_samples, world, rank)
    else:
        # longest-first over the estimated sampling time of each complex (affine in N: a B = 40 call has a fixed cost no batch fills)
        share = D.assign_work([D.complex_cost(c["rec_x"].shape[0] + c["lig_x"].shape[0], num_samples) for c in complexes], world)[rank]
        t_lo, t_hi = 0, num_samples
    rng = np.random.default_rng(seed)
    rots = [rng.integers(0, 2 ** 31) for _ in complexes]      # per-complex streams, identical on every rank
    if t_hi <= t_lo:
        share = []
    trace = traj_dir is not None

    def prep(ci):
        return _prepare(model, complexes[ci], ci, rots[ci], global_rotation, precision, selfcheck, on_selfcheck_fail, seed, log)

    def samp(p):
        return _sample(p, t_lo, t_hi, num_steps, seed, max_batch, trace, sampler_kw)

    done = []      # (prepared, rows, records) in share order
    if not overlap or len(share) < 2:
        for ci in share:
            p = prep(ci)
            rows_c, recs_c = _post(p, samp(p), traj_dir)
            done.append((p, rows_c, recs_c))
    else:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        ahead = threading.Semaphore(samplers + 1)      # handles prepared but not yet sampling: bounds the device memory in flight

        def prep_gated(ci):
            ahead.acquire()
            try:
                return prep(ci)
            except BaseException:
                ahead.release()
                raise

        def samp_of(fp):
            p = fp.result()
            ahead.release()
            return p, samp(p)

        # the canary complex: cheapest of those that are neither first nor last of the share (so something else was always in flight)
        inner = share[1:-1] if len(share) > 2 else share[-1:]
        canary_ci = min(inner, key=lambda ci: complexes[ci]["rec_x"].shape[0] + complexes[ci]["lig_x"].shape[0]) if canary else None
        canary_ref = {}

        def post_of(fs):
            p, batches = fs.result()
            if p.ci == canary_ci:
                canary_ref["batches"], canary_ref["precision"] = batches, p.precision
            rows_c, recs_c = _post(p, batches, traj_dir)
            return p, rows_c, recs_c

        with ThreadPoolExecutor(1, thread_name_prefix="dfm-prep") as ex_prep, \
                ThreadPoolExecutor(max(1, int(samplers)), thread_name_prefix="dfm-sample") as ex_samp, \
                ThreadPoolExecutor(1, thread_name_prefix="dfm-post") as ex_post:
            f_prep = [ex_prep.submit(prep_gated, ci) for ci in share]
            f_samp = [ex_samp.submit(samp_of, f) for f in f_prep]
            f_post = [ex_post.submit(post_of, f) for f in f_samp]
            try:
                done = [f.result() for f in f_post]
            except BaseException:
                for f in f_prep + f_samp + f_post:
                    f.cancel()
                for _ in share:      # un-block a prepare stage parked on the gate
                    ahead.release()
                raise
        if canary_ci is not None and "batches" in canary_ref:
            again = _prepare(model, complexes[canary_ci], canary_ci, rots[canary_ci], global_rotation, canary_ref["precision"], False,
                             on_selfcheck_fail, seed, log)
            solo = samp(again)
            same = len(solo) == len(canary_ref["batches"]) and all(
                all(np.array_equal(ra[k], rb[k]) for k in ("lig_pos", "energy", "num_clashes", "rot_update", "tr_update"))
                for (_, _, ra), (_, _, rb) in zip(solo, canary_ref["batches"]))
            info = {"checked": True, "id": complexes[canary_ci].get("id", str(canary_ci)), "ok": bool(same), "reran_serial": False}
            if not same:
                import sys
                (log or (lambda m: print(m, file=sys.stderr, flush=True)))(
                    f"run_set canary: complex {info['id']} sampled next to other handles differs from the same complex sampled alone - "
                    "concurrent handles are NOT independent on this system; re-running this rank's share serially")
                done = []
                for ci in share:
                    p = prep(ci)
                    rows_c, recs_c = _post(p, samp(p), traj_dir)
                    done.append((p, rows_c, recs_c))
                info["reran_serial"] = True
            if canary_out is not None:
                canary_out.update(info)
    rows, records = [], []
    for p, rows_c, recs_c in done:
        rows += rows_c
        records += recs_c
        if checks_out is not None:
            checks_out.append({"id": p.c.get("id", str(p.ci)), "precision": p.precision, "selfcheck": p.check})
        if timings_out is not None:
            timings_out.append({"id": p.c.get("id", str(p.ci)), "N": p.N, **p.ms})
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
              out_pdb="output.pdb", max_batch=256, selfcheck=True, on_selfcheck_fail="fp32", **sampler_kw):
    """inference() of the reference for two parsed PDB chains (pdbio.backbone_from_atoms dicts) and their
    pre-computed node features; returns {'energy': min energy} and writes the best pose.  `sampler_kw` are the sampler options
    the reference's pair loop passes (src/inference_base.py:483-491: use_clash_force, noise_annealing, tr_noise_scale,
    rot_noise_scale, ode)."""
    gx = engine.Complex(model, rec_x, lig_x, rec["bb_coords"], lig["bb_coords"])
    precision, chk = checked_precision(gx, precision, "pair", selfcheck, on_selfcheck_fail, seed=seed)
    best = None
    done = 0
    while done < num_samples:
        b = min(max_batch, num_samples - done)
        r = gx.sample(B=b, num_steps=num_steps, seed=seed + done, **engine.precision_kwargs(precision), **sampler_kw)
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

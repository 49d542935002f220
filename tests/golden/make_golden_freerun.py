#!/usr/bin/env python3
"""Outcome samples of the REFERENCE's free-running sampler (SURVEY.md 8(d) gate 4).

Runs only in the build container (imports /root/reference/src unmodified through make_golden.py's stand-ins).  For each case
the reference's own `Euler_Maruyama_sampler` (src/inference_base.py:390-468) is run T times with NOTHING injected: R0 from scipy
`Rotation.random`, the N(0,30^2) draw and the per-step z from `torch.normal` / `torch.randn`, the 40 sampled edges of every
evaluation from `torch.multinomial` (src/models/score_net_mlsb.py:85-131).  Per run the script keeps what a caller of the
sampler sees: tr_update[3], rot_update[3], final energy, final num_clashes.  The GPU engine's native (Philox) runs are compared
with these samples by two-sample Kolmogorov-Smirnov tests (tests/test_gpu_freerun.py); no bitwise agreement is possible between
the two random streams, only agreement in distribution.

Usage:  python tests/golden/make_golden_freerun.py [--runs 512] [--workers 8]
        -> tests/golden/freerun_<case>.npz   (16 KiB each)
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"syn_24_16": (24, 16, 5), "syn_64_48": (64, 48, 7)}      # (R, L, make_complex seed): the complexes of fwd_syn_*.npz
# r06: the same on the "sticky" weight draw (dfmdock_amd.weights.make_sticky_weights), whose free runs END IN
# CONTACT - non-degenerate final energies / clash counts, and the arg-min selection of inference() (src/inference_base.py:638-657)
# has something to select on.  These cases also keep l_rmsd of every final pose against the start pose (src/utils/metrics.py).
STICKY = {"sticky_syn_24_16": ("syn_24_16", None), "sticky_syn_64_48": ("syn_64_48", None), "sticky_7CEI": (None, "7CEI")}
NUM_STEPS = 40


def _worker(job):
    case, lo, hi = job
    import torch
    sys.path.insert(0, HERE)
    import make_golden as mg      # installs the stand-ins and imports the reference
    torch.set_num_threads(1)
    sticky = case in STICKY
    if sticky:
        from dfmdock_amd.weights import make_sticky_weights
        from utils.metrics import compute_metrics
        syn, db5 = STICKY[case]
        if syn:
            R, L, seed = CASES[syn]
            cx = mg.make_complex(R, L, seed=seed)
        else:      # 7CEI with its real ESM-2 block (cx_7CEI.npz: the fp16-rounded block both sides use)
            from dfmdock_amd.synthetic import seq_to_onehot
            d = dict(np.load(os.path.join(HERE, "cx_7CEI.npz"), allow_pickle=False))
            cx = {"rec_x": np.concatenate([d["rec_esm16"].astype(np.float32), seq_to_onehot(str(d["rec_seq"]))], 1),
                  "lig_x": np.concatenate([d["lig_esm16"].astype(np.float32), seq_to_onehot(str(d["lig_seq"]))], 1),
                  "rec_pos": d["rec_pos"], "lig_pos": d["lig_pos"]}
        net = mg.build_net(0)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_sticky_weights().items()}, strict=True)
        model = mg.Model(net.eval())
    else:
        R, L, seed = CASES[case]
        cx = mg.make_complex(R, L, seed=seed)
        model = mg.Model(mg.build_net(0))
    batch = mg.make_batch(cx)
    out = np.zeros((hi - lo, 9), np.float64)
    for i, run in enumerate(range(lo, hi)):
        np.random.seed(100000 + run)          # scipy Rotation.random draws from numpy's global state
        torch.manual_seed(100000 + run)
        with torch.no_grad():
            rec_pos, lig_pos, rot_update, tr_update, output = mg.ib.Euler_Maruyama_sampler(
                model=model, batch=dict(batch), num_steps=NUM_STEPS, device="cpu")
        if sticky:
            m = compute_metrics((rec_pos, lig_pos), (batch["rec_pos"], batch["lig_pos"]))
            out[i, 8] = float(m["l_rmsd"])
        out[i, 0:3] = tr_update.numpy().reshape(3)
        out[i, 3:6] = rot_update.numpy().reshape(3)
        out[i, 6] = float(output["energy"])
        out[i, 7] = float(output["num_clashes"])
    return case, lo, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=512)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--cases", nargs="*", default=list(CASES) + list(STICKY))
    a = ap.parse_args()
    chunk = 8
    jobs = [(c, lo, min(lo + chunk, a.runs)) for c in a.cases for lo in range(0, a.runs, chunk)]
    res = {c: np.zeros((a.runs, 9), np.float64) for c in a.cases}
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.workers) as pool:
        for n, (case, lo, out) in enumerate(pool.imap_unordered(_worker, jobs)):
            res[case][lo:lo + out.shape[0]] = out
            print(f"{n + 1}/{len(jobs)} chunks, {time.time() - t0:.0f} s", flush=True)
    for c in a.cases:
        r = res[c]
        path = os.path.join(HERE, f"freerun_{c}.npz")
        if c in STICKY:
            np.savez_compressed(path, tr_update=r[:, 0:3].astype(np.float32), rot_update=r[:, 3:6].astype(np.float32),
                                energy=r[:, 6].astype(np.float32), num_clashes=r[:, 7].astype(np.int32), l_rmsd=r[:, 8].astype(np.float32),
                                num_steps=NUM_STEPS, weights="make_sticky_weights()")
        else:
            np.savez_compressed(path, tr_update=r[:, 0:3].astype(np.float32), rot_update=r[:, 3:6].astype(np.float32),
                                energy=r[:, 6].astype(np.float32), num_clashes=r[:, 7].astype(np.int32),
                                num_steps=NUM_STEPS, weight_seed=0, R=CASES[c][0], L=CASES[c][1], cx_seed=CASES[c][2])
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; |tr| median {np.median(np.linalg.norm(r[:, 0:3], axis=1)):.2f} "
              f"energy median {np.median(r[:, 6]):.4f} nonzero-energy fraction {(r[:, 6] != 0).mean():.3f} "
              f"clashes mean {r[:, 7].mean():.2f}")


if __name__ == "__main__":
    main()

"""Backbone docking metrics (host side): c/i/l-RMSD, Fnat, DockQ - reference src/utils/metrics.py:3-121.

This is the evaluation function that defines "DockQ parity"; it is not on the GPU path.
"""
from __future__ import annotations

import numpy as np


def find_rigid_alignment(A, B):
    """Kabsch: R, t aligning A onto B (metrics.py:91-121)."""
    a_mean, b_mean = A.mean(0), B.mean(0)
    H = (A - a_mean).T @ (B - b_mean)
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        R = (Vt.T @ np.diag([1.0, 1.0, -1.0])) @ U.T
    t = b_mean - R @ a_mean
    return R, t


def _rmsd(p, q):
    return float(np.sqrt(np.mean(np.sum((p - q) ** 2, -1))))


def _min_dist(x1, x2):
    """[n1,n2] minimum backbone-atom distance between residues (metrics.py:75-83)."""
    d = x1[:, None, :, None, :] - x2[None, :, None, :, :]
    return np.sqrt((d ** 2).sum(-1)).reshape(x1.shape[0], x2.shape[0], -1).min(-1)


def _min_dist_pairs(x1, x2, i1, i2):
    """Minimum backbone-atom distance of the residue pairs (i1[k], i2[k]) only - same arithmetic as _min_dist."""
    if len(i1) == 0:
        return np.zeros((0,), np.float64)
    d = x1[i1][:, :, None, :] - x2[i2][:, None, :, :]
    return np.sqrt((d ** 2).sum(-1)).reshape(len(i1), 9).min(-1)


class NativeContext:
    """Everything compute_metrics derives from the native pose alone (interface residues < 10 A, native contacts < 5.5 A):
    computed once per complex instead of once per sampled trajectory (the R x L x 9 distance tensor dominates otherwise)."""

    def __init__(self, native):
        self.nr, self.nl = (np.asarray(x, np.float32).astype(np.float64) for x in native)
        md = _min_dist(self.nr, self.nl)
        near = np.where(md < 10.0)
        self.r1, self.r2 = np.unique(near[0]), np.unique(near[1])
        self.act = np.where(md < 5.5)


def compute_metrics(model, native, ctx: NativeContext | None = None):
    """model = (rec [R,3,3], lig [L,3,3]), native likewise -> dict like the reference's compute_metrics.
    `ctx` = NativeContext(native) to reuse the native-only parts across many models of one complex."""
    ctx = ctx or NativeContext(native)
    mr, ml = (np.asarray(x, np.float32).astype(np.float64) for x in model)
    nr, nl, r1, r2, act = ctx.nr, ctx.nl, ctx.r1, ctx.r2, ctx.act
    flat = lambda x: x.reshape(-1, 3)
    # c_rmsd: align everything
    P, Q = np.concatenate([flat(mr), flat(ml)]), np.concatenate([flat(nr), flat(nl)])
    R, t = find_rigid_alignment(P, Q)
    c_rmsd = _rmsd(P @ R.T + t, Q)
    # i_rmsd: residues within 10 A (min backbone distance) in the native
    P, Q = np.concatenate([flat(mr[r1]), flat(ml[r2])]), np.concatenate([flat(nr[r1]), flat(nl[r2])])
    R, t = find_rigid_alignment(P, Q)
    i_rmsd = _rmsd(P @ R.T + t, Q)
    # l_rmsd: align receptors, measure ligand
    R, t = find_rigid_alignment(flat(mr), flat(nr))
    l_rmsd = _rmsd(flat(ml) @ R.T + t, flat(nl))
    # fnat: native contacts (< 5.5 A) recovered
    pred = _min_dist_pairs(mr, ml, act[0], act[1])
    fnat = round(int((pred < 5.5).sum()) / (len(act[0]) + 1e-6), 6)
    dockq = (fnat + 1.0 / (1.0 + (i_rmsd / 1.5) ** 2) + 1.0 / (1.0 + (l_rmsd / 8.5) ** 2)) / 3
    return {"c_rmsd": c_rmsd, "i_rmsd": i_rmsd, "l_rmsd": l_rmsd, "fnat": fnat, "DockQ": dockq}

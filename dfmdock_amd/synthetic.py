"""Synthetic complexes for benchmarks and parity tests.

Per chain: CA = cumulative sum of N(0, 2.2^2) steps, N = CA + N(0, 0.8^2),
C = CA + N(0, 0.8^2); ligand chain offset by +5 A along x; node features
x = [N(0,1) x 1280 | one-hot('A' for receptor, 'G' for ligand) x 21].
"""
from __future__ import annotations

import numpy as np

RESTYPES = "ARNDCQEGHILKMFPSTWYV"  # residue_constants.py:855-876 order; 'X' = 20


def seq_to_onehot(seq: str) -> np.ndarray:
    """21-class one-hot with unknown -> X (residue_constants.py:885-928 semantics)."""
    idx = np.array([RESTYPES.find(c) if c in RESTYPES else 20 for c in seq], dtype=np.int64)
    out = np.zeros((len(seq), 21), dtype=np.float32)
    out[np.arange(len(seq)), idx] = 1.0
    return out


def make_chain(rng, n, offset):
    ca = np.cumsum(rng.normal(0.0, 2.2, size=(n, 3)), axis=0)
    nn = ca + rng.normal(0.0, 0.8, size=(n, 3))
    cc = ca + rng.normal(0.0, 0.8, size=(n, 3))
    pos = np.stack([nn, ca, cc], axis=1) + np.asarray(offset, dtype=np.float64)
    return pos.astype(np.float32)


def make_complex(R: int, L: int, seed: int = 1, esm_dim: int = 1280):
    """Returns dict(rec_x[R,1301], lig_x[L,1301], rec_pos[R,3,3], lig_pos[L,3,3]) float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rec_pos = make_chain(rng, R, (0.0, 0.0, 0.0))
    lig_pos = make_chain(rng, L, (5.0, 0.0, 0.0))
    rec_x = np.concatenate([rng.standard_normal((R, esm_dim)).astype(np.float32),
                            seq_to_onehot("A" * R)], axis=1)
    lig_x = np.concatenate([rng.standard_normal((L, esm_dim)).astype(np.float32),
                            seq_to_onehot("G" * L)], axis=1)
    return {"rec_x": rec_x, "lig_x": lig_x, "rec_pos": rec_pos, "lig_pos": lig_pos}

"""Multi-GPU sharding of independent trajectories.

Trajectories never interact until the final energy ranking (reference:
src/inference_base.py:644-657 keeps the arg-min energy over samples), so the
path shards with NO data-path collective: rank r samples its own block of
trajectories and ONE small all_gather of fixed-size records
(complex id, trajectory id, energy, num_clashes, rot_update[3], tr_update[3])
happens at the end - RCCL over xGMI on GPUs (backend "nccl"), gloo in CPU tests.
"""
from __future__ import annotations

import os

import numpy as np

RECORD_WIDTH = 10   # complex_id, traj_id, energy, num_clashes, rot[3], tr[3]  (float32 each)


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(total: int, world: int, rank: int):
    """Contiguous block [lo, hi) of `total` work items for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def assign_work(costs, world: int):
    """Longest-processing-time-first assignment of work items (e.g. complexes weighted by N) to ranks.

    Returns a list of index lists, one per rank.  Deterministic for equal costs (stable by index).
    """
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return out


def make_records(complex_id: int, traj_ids, result) -> np.ndarray:
    """Pack one dfm_sample result into [n, RECORD_WIDTH] float32 records."""
    n = len(traj_ids)
    rec = np.zeros((n, RECORD_WIDTH), np.float32)
    rec[:, 0] = complex_id
    rec[:, 1] = np.asarray(traj_ids, np.float32)
    rec[:, 2] = result["energy"]
    rec[:, 3] = result["num_clashes"]
    rec[:, 4:7] = result["rot_update"]
    rec[:, 7:10] = result["tr_update"]
    return rec


def gather_records(records: np.ndarray, device=None) -> np.ndarray:
    """all_gather variable-length record blocks from every rank; returns the concatenation (rank order)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(c.item() for c in counts))
    pad = torch.zeros((nmax, RECORD_WIDTH), dtype=torch.float32, device=dev)
    if records.shape[0]:
        pad[: records.shape[0]] = torch.from_numpy(records).to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return np.concatenate([b[: int(c.item())].cpu().numpy() for b, c in zip(bufs, counts)], 0)


def rank_by_energy(records: np.ndarray):
    """Per complex: records sorted by ascending energy (the reference keeps the minimum)."""
    out = {}
    for cid in np.unique(records[:, 0]).astype(int):
        r = records[records[:, 0] == cid]
        out[cid] = r[np.argsort(r[:, 2], kind="stable")]
    return out

"""Multi-GPU sharding of independent trajectories.

Trajectories never interact until the final energy ranking (reference:
src/inference_base.py:644-657 keeps the arg-min energy over samples), so the
path shards with NO data-path collective: rank r samples its own block of
trajectories and ONE small all_gather of fixed-size records
(complex id, trajectory id, energy, num_clashes, rot_update[3], tr_update[3])
happens at the end - RCCL over xGMI on GPUs (backend "nccl"), gloo in CPU tests.

`init()` brings the process group up so that a broken RCCL never kills a job whose data path needs no collective at all: a gloo
group (TCP on the loopback) is the control plane, the RCCL group is probed with one small all_reduce and used for the record
gather only if EVERY rank's probe succeeded; otherwise the gather runs over gloo, and with no torch.distributed rendezvous at
all (independent replicas started with RANK / WORLD_SIZE / DFM_GATHER_DIR) over files - SURVEY.md 8(e)'s last row.  What was
actually used is returned and ends up in bench.py's JSON line.

DFM_DIST_BACKEND=rccl selects the torch-free path instead: librccl through ctypes (dfmdock_amd/rccl.py) for the gather AND the few
control-plane exchanges, the communicator's unique id handed over through DFM_GATHER_DIR or a TCP socket at MASTER_PORT + 1
(DFM_RCCL_PORT).  It is an opt-in: it cannot probe-and-fall-back the way the default chain does, and it has only ever run with one
rank on the one-GPU boxes this was built on (tests/test_gpu_multiproc.py).
"""
from __future__ import annotations

import os

import numpy as np

RECORD_WIDTH = 10   # complex_id, traj_id, energy, num_clashes, rot[3], tr[3]  (float32 each)


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


class Group:
    """What init() settled on: `backend` in {"single", "nccl", "gloo", "file", "rccl"}, the group handle the record gather uses
    (a torch process group, or the dfmdock_amd.rccl.Rccl communicator), the reason a preferred backend was not used (or None)."""

    def __init__(self, backend, rank=0, world=1, data_group=None, fallback_reason=None, gather_dir=None):
        self.backend, self.rank, self.world = backend, rank, world
        self.data_group, self.fallback_reason, self.gather_dir = data_group, fallback_reason, gather_dir
        self._file_round = 0
        self._token = None          # file backend: the job's name space inside gather_dir (see _file_token)
        self._prev_file = None      # file backend: this rank's file of the previous round (deleted once the next round is complete)


_group = Group("single")


def current_group() -> Group:
    return _group


class _stdout_to_stderr:
    """gloo / RCCL print connection banners on file descriptor 1; a bench's stdout carries ONE JSON line, so the banners of the
    rendezvous go to stderr."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        import sys
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def init(device_index: int | None = None, prefer: str | None = None, timeout_s: float = 120.0) -> Group:
    """Bring up the collective layer for this rank (see the module docstring).  `prefer`: "nccl" (default on a GPU box; env
    DFM_DIST_BACKEND overrides), "gloo" or "file"."""
    with _stdout_to_stderr():
        return _init(device_index, prefer, timeout_s)


def _init(device_index, prefer, timeout_s) -> Group:
    global _group
    import datetime
    rank, _, world = dist_env()
    prefer = prefer or os.environ.get("DFM_DIST_BACKEND", "nccl")
    if world == 1:
        _group = Group("single")
        return _group
    gather_dir = os.environ.get("DFM_GATHER_DIR")
    if prefer == "rccl":      # librccl directly, no torch.distributed (opt-in; failures are raised, not papered over)
        from . import rccl
        if gather_dir:
            os.makedirs(gather_dir, exist_ok=True)
            token = "j" + "".join(ch if ch.isalnum() else "-" for ch in os.environ.get("DFM_JOB_ID", os.environ.get("MASTER_PORT", "0")))
            exch = lambda uid: rccl.exchange_uid_file(uid, rank, world, gather_dir, token, timeout_s)
        elif "MASTER_PORT" in os.environ:
            port = int(os.environ.get("DFM_RCCL_PORT", int(os.environ["MASTER_PORT"]) + 1))
            exch = lambda uid: rccl.exchange_uid_tcp(uid, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, timeout_s)
        else:
            raise RuntimeError("DFM_DIST_BACKEND=rccl needs DFM_GATHER_DIR or MASTER_ADDR / MASTER_PORT to hand the unique id over")
        comm = rccl.Rccl(rank, world, device_index if device_index is not None else int(os.environ.get("LOCAL_RANK", 0)), exch)
        comm.barrier()
        _group = Group("rccl", rank, world, data_group=comm)
        return _group
    if prefer == "file" or "MASTER_PORT" not in os.environ:
        if not gather_dir:
            raise RuntimeError("WORLD_SIZE > 1 without a torch.distributed rendezvous needs DFM_GATHER_DIR for the file gather")
        _group = Group("file", rank, world, gather_dir=gather_dir,
                       fallback_reason=None if prefer == "file" else "no MASTER_PORT: independent replicas")
        return _group
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    try:
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    except Exception as e:      # no control plane either: replicas + files, if a directory was provided.  (gloo connects the full
        # mesh inside init_process_group: it fails on every rank or on none; a rank that did come up alone would leave its
        # first collective with a timeout error after timeout_s, not hang)
        if not gather_dir:
            raise
        _group = Group("file", rank, world, gather_dir=gather_dir, fallback_reason=f"gloo init failed: {e}")
        return _group
    reason, data_group, ok = None, None, 0
    if prefer == "nccl":
        try:
            if not torch.cuda.is_available():
                raise RuntimeError("no GPU visible to torch")
            if device_index is not None:
                torch.cuda.set_device(device_index)
            data_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe, group=data_group)
            torch.cuda.synchronize()
            ok = int(round(float(probe.item()))) == world
            if not ok:
                reason = f"nccl probe all_reduce returned {float(probe.item())} on {world} ranks"
        except Exception as e:
            reason = f"nccl failed: {type(e).__name__}: {str(e).splitlines()[0][:200]}"
            ok = 0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # over gloo: do ALL ranks have a working RCCL group?
        if int(flag.item()) == 1:
            _group = Group("nccl", rank, world, data_group=data_group)
            return _group
        reasons = [None] * world
        dist.all_gather_object(reasons, reason)
        reason = next((r for r in reasons if r), "nccl failed on another rank")
    _group = Group("gloo", rank, world, data_group=None, fallback_reason=reason)
    return _group


def shutdown():
    global _group
    if _group.backend == "rccl":
        try:
            _group.data_group.barrier()
            _group.data_group.close()
        finally:
            _group = Group("single")
        return
    if _group.backend == "file" and _group._token:
        # a closing round (nobody may still be reading this rank's last block when it goes); the closing round's own small file
        # stays behind under the job's token and is swept by the next job's rank of the same number
        try:
            _file_gather(np.zeros((0, RECORD_WIDTH), np.float32), _group)
        except TimeoutError:
            pass
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    finally:
        _group = Group("single")


def _sweep_own(g: Group, keep_token=None):
    """Delete this rank's files of EARLIER jobs in gather_dir (any file named for this rank whose token is not the live one)."""
    suffixes = (f"_rank{g.rank}.npy", f"_rank{g.rank}.json")
    for name in os.listdir(g.gather_dir):
        mine = name.endswith(suffixes) or name.startswith(f"hello_rank{g.rank}_") or (g.rank == 0 and name.startswith("ack_"))
        if mine and (keep_token is None or not name.startswith(keep_token + "_")):
            try:
                os.remove(os.path.join(g.gather_dir, name))
            except OSError:
                pass


def _service_hellos(g: Group):
    """Rank 0, inside every wait loop: answer `hello_rank<r>_<nonce>` with `ack_<nonce>` holding the job token.  A nonce is fresh
    per process, so an ack can only come from the LIVE rank 0 - a stale directory cannot hand out a dead job's token."""
    for name in os.listdir(g.gather_dir):
        if name.startswith("hello_rank"):
            nonce = name.rsplit("_", 1)[1]
            ack = os.path.join(g.gather_dir, "ack_" + nonce)
            if not os.path.exists(ack):
                tmp = os.path.join(g.gather_dir, f".ack_{nonce}.tmp")
                with open(tmp, "w") as f:
                    f.write(g._token)
                os.replace(tmp, ack)


def _wait_for(path: str, g: Group, what: str, poll_s: float = 0.02, timeout_s: float = 600.0):
    import time
    t0 = time.time()
    while not os.path.exists(path):
        if g.rank == 0 and g._token:
            _service_hellos(g)
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"file gather: {what} never appeared ({path})")
        time.sleep(poll_s)


def _file_token(g: Group) -> str:
    """Name space of this job's files inside DFM_GATHER_DIR.  The files of a gather are `<token>_round000001_rank0.npy` ...: a
    second job (or a restarted rank) that reuses the directory must never read the previous run's records as its own, and a
    barrier must never return on them.  DFM_JOB_ID names the job when the launcher provides one; otherwise rank 0 draws a token
    and hands it to the other ranks through a nonce handshake (hello / ack, see _service_hellos).  Every rank first removes its
    own leftovers of earlier jobs."""
    if g._token:
        return g._token
    import uuid
    os.makedirs(g.gather_dir, exist_ok=True)
    env = os.environ.get("DFM_JOB_ID")
    if env:
        g._token = "j" + "".join(ch if ch.isalnum() else "-" for ch in env)
        _sweep_own(g, keep_token=g._token)
        return g._token
    _sweep_own(g)
    if g.rank == 0:
        g._token = "j" + uuid.uuid4().hex[:12]
        return g._token
    nonce = uuid.uuid4().hex[:12]
    hello = os.path.join(g.gather_dir, f"hello_rank{g.rank}_{nonce}")
    open(hello, "w").close()
    ack = os.path.join(g.gather_dir, "ack_" + nonce)
    _wait_for(ack, g, "rank 0's answer to this rank's hello")
    g._token = open(ack).read().strip()
    for pth in (hello, ack):
        try:
            os.remove(pth)
        except OSError:
            pass
    return g._token


def _file_round(g: Group, kind: str, ext: str, write, read):
    """One all-to-all round over files: every rank drops its block (atomic rename), then reads all of them.  Once round n is
    complete every rank has finished READING round n - 1 (it read before it wrote), so this rank's file of round n - 1 goes."""
    token = _file_token(g)
    g._file_round += 1
    tag = f"{token}_{kind}{g._file_round:06d}"
    mine = os.path.join(g.gather_dir, f"{tag}_rank{g.rank}{ext}")
    tmp = os.path.join(g.gather_dir, f".{tag}_rank{g.rank}.tmp{ext}")
    write(tmp)
    os.replace(tmp, mine)
    out = []
    for r in range(g.world):
        path = os.path.join(g.gather_dir, f"{tag}_rank{r}{ext}")
        _wait_for(path, g, f"rank {r}'s block of {kind} round {g._file_round}")
        out.append(read(path))
    if g._prev_file:
        try:
            os.remove(g._prev_file)
        except OSError:
            pass
    g._prev_file = mine
    return out


def _file_gather(records: np.ndarray, g: Group) -> np.ndarray:
    """SURVEY 8(e) last row: the record gather without any rendezvous, over a shared directory."""
    blocks = _file_round(g, "round", ".npy", lambda p: np.save(p, np.ascontiguousarray(records, np.float32)),
                         lambda p: np.load(p).reshape(-1, RECORD_WIDTH))
    return np.concatenate(blocks, 0)


def shard_range(total: int, world: int, rank: int):
    """Contiguous block [lo, hi) of `total` work items for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Cost of one complex of a set run on one MI355X, milliseconds (profiles/r05_c4.txt: 24 DB5-sized complexes x 40 trajectories x 40
# steps through the pipelined driver; least-squares over the per-complex `sample` times, N = 197 ... 695): the sampling call is
# affine in the residue count - a fixed part (41 evaluations x ~27 dependent launches that no batch of 40 fills) plus a per-residue
# part - and scales with the trajectories per call; creation, self-check and metrics overlap with the previous / next complex's
# sampling in the pipelined driver and only add to the first and last complex of a rank.
SET_COST_FIXED_MS = 10.0
SET_COST_PER_RESIDUE_MS = 0.16
SET_COST_EDGE_MS = 30.0      # un-overlapped prepare of a rank's first complex + post of its last


def complex_cost(n_residues: int, num_samples: int = 40) -> float:
    """Estimated milliseconds of one complex's sampling (see SET_COST_*): what driver.run_set balances over ranks."""
    return (SET_COST_FIXED_MS + SET_COST_PER_RESIDUE_MS * float(n_residues)) * max(num_samples, 1) / 40.0


def makespan(costs, assignment) -> float:
    """Largest per-rank sum of `costs` under `assignment` (index lists per rank) + the un-overlapped pipeline edges."""
    return max((sum(costs[i] for i in part) + (SET_COST_EDGE_MS if part else 0.0)) for part in assignment)


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


def gather_records(records: np.ndarray, device=None, force_collective: bool = False) -> np.ndarray:
    """all_gather variable-length record blocks from every rank; returns the concatenation (rank order).  A one-rank group skips
    the collectives unless `force_collective` (the one-rank RCCL test runs them on purpose)."""
    g = _group
    if g.backend == "file":
        return _file_gather(records, g)
    if g.backend == "rccl":
        blocks = g.data_group.all_gather_var(np.ascontiguousarray(records, np.float32).tobytes())
        return np.concatenate([np.frombuffer(b, np.float32).reshape(-1, RECORD_WIDTH) for b in blocks], 0)
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return records
    world = dist.get_world_size()
    grp = g.data_group if g.backend == "nccl" else None          # None = the default group (gloo when init() made it)
    backend = dist.get_backend(grp) if grp is not None else dist.get_backend()
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=grp)
    nmax = int(max(c.item() for c in counts))
    pad = torch.zeros((nmax, RECORD_WIDTH), dtype=torch.float32, device=dev)
    if records.shape[0]:
        pad[: records.shape[0]] = torch.from_numpy(records).to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=grp)
    return np.concatenate([b[: int(c.item())].cpu().numpy() for b, c in zip(bufs, counts)], 0)


def allgather_scalars(values) -> np.ndarray:
    """[world, len(values)] float64 table of a few host scalars per rank (timings, device ids), on the control plane."""
    v = np.atleast_1d(np.asarray(values, np.float64))
    g = _group
    if g.backend == "single":
        return v[None]
    if g.backend == "rccl":
        return np.stack([np.frombuffer(b, np.float64) for b in g.data_group.all_gather_bytes(v.tobytes())])
    if g.backend == "file":
        assert v.size <= RECORD_WIDTH // 2
        rec = np.zeros((1, RECORD_WIDTH), np.float32)
        rec[0, : 2 * v.size] = v.astype(np.float64).view(np.float32)      # bit-exact float64 through the float32 records
        out = _file_gather(rec, g)
        return np.ascontiguousarray(out[:, : 2 * v.size]).view(np.float64)
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(v.copy())
    bufs = [torch.zeros_like(t) for _ in range(g.world)]
    dist.all_gather(bufs, t)
    return np.stack([b.numpy() for b in bufs])


def allreduce_max(value: float) -> float:
    """max over ranks of a host scalar (the bench's max-over-ranks time), on the control plane."""
    return float(allgather_scalars([value])[:, 0].max())


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (small host objects: the CSV rows of driver.run_set)."""
    g = _group
    if g.backend == "rccl":
        import json
        return [json.loads(b.decode()) for b in g.data_group.all_gather_var(json.dumps(obj, default=float).encode())]
    if g.backend == "file":
        import json

        def write(pth):
            with open(pth, "w") as f:
                json.dump(obj, f, default=float)

        def read(pth):
            with open(pth) as f:
                return json.load(f)
        return _file_round(g, "obj", ".json", write, read)
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier():
    g = _group
    if g.backend == "single":
        return
    if g.backend == "file":
        _file_gather(np.zeros((0, RECORD_WIDTH), np.float32), g)
        return
    if g.backend == "rccl":
        g.data_group.barrier()
        return
    import torch.distributed as dist
    dist.barrier()


def rank_by_energy(records: np.ndarray):
    """Per complex: records sorted by ascending energy (the reference keeps the minimum)."""
    out = {}
    for cid in np.unique(records[:, 0]).astype(int):
        r = records[records[:, 0] == cid]
        out[cid] = r[np.argsort(r[:, 2], kind="stable")]
    return out

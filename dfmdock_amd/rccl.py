"""librccl through ctypes: the job's only collective without torch (north_star keeps PyTorch "only for checkpoint load").

The final energy-ranked gather (reference site: src/inference_base.py:644-657) is ONE all_gather of 40-byte records per trajectory.
This module gives it a path that needs nothing but the ROCm runtime: `librccl.so` for the communicator and the collective,
`libamdhip64.so` for the two device buffers it moves, and a 128-byte unique id handed from rank 0 to the others over a file in
DFM_GATHER_DIR or a TCP socket next to the launcher's MASTER_PORT.  Selected with DFM_DIST_BACKEND=rccl (dfmdock_amd/distributed.py);
the torch.distributed chain (nccl -> gloo -> files) stays the default because it can probe and fall back - see DESIGN.md section 7.

Everything is moved as bytes (ncclUint8): counts, records, timings, JSON rows.
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time

import numpy as np

NCCL_UNIQUE_ID_BYTES = 128
NCCL_UINT8 = 1          # ncclDataType_t
NCCL_SUM = 0            # ncclRedOp_t
HIP_H2D, HIP_D2H = 1, 2


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _load(names):
    err = None
    for n in names:
        try:
            return C.CDLL(n)
        except OSError as e:
            err = e
    raise RuntimeError(f"cannot load any of {names}: {err}")


def _write_atomic(path: str, data: bytes):
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _read(path: str):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return None


def exchange_uid_file(uid: bytes | None, rank: int, world: int, gather_dir: str, token: str, timeout_s: float = 120.0) -> bytes:
    """Rank 0 hands the 128-byte id to the other ranks through files in `gather_dir`, with a handshake that a leftover file of
    an EARLIER job with the same token (no DFM_JOB_ID: the token is just MASTER_PORT) cannot satisfy:

      rank r > 0   removes its own leftover ack, writes `<token>_rccl_hello_<r>` = a fresh 16-byte nonce, then waits for a
                   `<token>_rccl_uid` file that carries ITS nonce in slot r, answers with `<token>_rccl_ack_<r>` = nonce + the first
                   16 bytes of the id it just read, and returns the id;
      rank 0       collects the hello nonces, publishes id + nonces (atomic rename), waits for every ack to equal nonce + ITS id's
                   first 16 bytes - a stale hello gives a nonce nobody acknowledges, and the hello / ack PAIR a crashed job left
                   behind (equal nonces: rank r had acked, rank 0 died before the clean-up) acknowledges the DEAD job's id, not this
                   one - so it re-reads the hellos and publishes again; finally it removes every file of the exchange.

    A dead id read from a stale file would make ncclCommInitRank hang."""
    base = os.path.join(gather_dir, f"{token}_rccl")
    uid_path = base + "_uid"
    t0 = time.time()
    if rank == 0:
        assert uid is not None and len(uid) == NCCL_UNIQUE_ID_BYTES
        others = list(range(1, world))
        published = None
        while True:
            nonces = {r: _read(f"{base}_hello_{r}") for r in others}
            if all(n is not None and len(n) == 16 for n in nonces.values()):
                blob = uid + b"".join(nonces[r] for r in others)
                if blob != published:
                    _write_atomic(uid_path, blob)
                    published = blob
                if all(_read(f"{base}_ack_{r}") == nonces[r] + uid[:16] for r in others):
                    for r in others:
                        for kind in ("hello", "ack"):
                            try:
                                os.remove(f"{base}_{kind}_{r}")
                            except OSError:
                                pass
                    try:
                        os.remove(uid_path)
                    except OSError:
                        pass
                    return uid
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rccl unique id: ranks {[r for r in others if _read(f'{base}_ack_{r}') != (nonces.get(r) or b'') + uid[:16]]} never acknowledged {uid_path}")
            time.sleep(0.01)
    try:
        os.remove(f"{base}_ack_{rank}")      # an ack of an earlier job must not outlive this rank's new hello
    except OSError:
        pass
    nonce = os.urandom(16)
    _write_atomic(f"{base}_hello_{rank}", nonce)
    lo = NCCL_UNIQUE_ID_BYTES + 16 * (rank - 1)
    while True:
        blob = _read(uid_path)
        if blob is not None and len(blob) == NCCL_UNIQUE_ID_BYTES + 16 * (world - 1) and blob[lo:lo + 16] == nonce:
            _write_atomic(f"{base}_ack_{rank}", nonce + blob[:16])
            return blob[:NCCL_UNIQUE_ID_BYTES]
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rccl unique id for this job never appeared at {uid_path}")
        time.sleep(0.01)


def exchange_uid_tcp(uid: bytes | None, rank: int, world: int, addr: str, port: int, timeout_s: float = 120.0) -> bytes:
    """Rank 0 listens on (addr, port) and sends the 128 bytes to each of the world - 1 ranks that connect; they retry until it is up."""
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout_s)
        try:
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(uid)
        finally:
            srv.close()
        return uid
    t0 = time.time()
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < NCCL_UNIQUE_ID_BYTES:
                    chunk = s.recv(NCCL_UNIQUE_ID_BYTES - len(buf))
                    if not chunk:
                        break
                    buf += chunk
                if len(buf) == NCCL_UNIQUE_ID_BYTES:
                    return buf
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"no rccl unique id from rank 0 at {addr}:{port}")
        time.sleep(0.05)


class Rccl:
    """One communicator over `world` ranks; this rank drives HIP device `device`."""

    def __init__(self, rank: int, world: int, device: int, exchange):
        self.rank, self.world = rank, world
        self.hip = _load(["libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"])
        self.nccl = _load(["librccl.so", "/opt/rocm/lib/librccl.so", "librccl.so.1"])
        self.nccl.ncclGetErrorString.restype = C.c_char_p
        self.hip.hipGetErrorString.restype = C.c_char_p
        self.nccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        self.nccl.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        self.nccl.ncclCommDestroy.argtypes = [C.c_void_p]
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip(self.hip.hipSetDevice(int(device)), "hipSetDevice")
        uid = _UniqueId()
        if rank == 0:
            self._nccl(self.nccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = exchange(C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES) if rank == 0 else None)      # (c_char arrays stop at a NUL)
        if len(raw) != NCCL_UNIQUE_ID_BYTES:
            raise RuntimeError(f"rccl unique id has {len(raw)} bytes")
        C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        self._nccl(self.nccl.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _hip(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.hip.hipGetErrorString(rc).decode()}")

    def _nccl(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.nccl.ncclGetErrorString(rc).decode()}")

    def all_gather_bytes(self, block: bytes) -> list:
        """Equal-length byte blocks of every rank, in rank order."""
        n = len(block)
        if n == 0:
            return [b""] * self.world
        send, recv = C.c_void_p(), C.c_void_p()
        self._hip(self.hip.hipMalloc(C.byref(send), n), "hipMalloc")
        self._hip(self.hip.hipMalloc(C.byref(recv), n * self.world), "hipMalloc")
        try:
            src = C.create_string_buffer(block, n)
            self._hip(self.hip.hipMemcpy(send, C.cast(src, C.c_void_p), n, HIP_H2D), "hipMemcpy H2D")
            self._nccl(self.nccl.ncclAllGather(send, recv, n, NCCL_UINT8, self.comm, None), "ncclAllGather")      # the null stream
            self._hip(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
            dst = C.create_string_buffer(n * self.world)
            self._hip(self.hip.hipMemcpy(C.cast(dst, C.c_void_p), recv, n * self.world, HIP_D2H), "hipMemcpy D2H")
            raw = dst.raw
        finally:
            self.hip.hipFree(send)
            self.hip.hipFree(recv)
        return [raw[r * n:(r + 1) * n] for r in range(self.world)]

    def all_gather_var(self, block: bytes) -> list:
        """Variable-length blocks: lengths first (8 bytes each), then the blocks padded to the longest."""
        lens = [int(np.frombuffer(b, np.int64)[0]) for b in self.all_gather_bytes(np.int64(len(block)).tobytes())]
        nmax = max(lens)
        if nmax == 0:
            return [b""] * self.world
        out = self.all_gather_bytes(block + b"\0" * (nmax - len(block)))
        return [o[:n] for o, n in zip(out, lens)]

    def barrier(self):
        self.all_gather_bytes(b"\1")

    def close(self):
        if getattr(self, "comm", None):
            self.nccl.ncclCommDestroy(self.comm)
            self.comm = None

"""ctypes binding of the CPU oracle (oracle/libdfm_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from dfmdock_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DFM_ORACLE_LIB") or os.path.join(_HERE, "libdfm_oracle.so")      # DFM_ORACLE_LIB: a sanitizer build (oracle/Makefile: asan)

F32P = C.POINTER(C.c_float)
I32P = C.POINTER(C.c_int32)
I8P = C.POINTER(C.c_int8)
F64P = C.POINTER(C.c_double)


class OraHParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("lm_embed_dim", "positional_embed_dim", "spatial_embed_dim",
                                       "node_dim", "edge_dim", "inner_dim", "depth", "knn", "n_sample")] + \
               [(n, C.c_float) for n in ("cut_off", "mask_dist")] + \
               [(n, C.c_double) for n in ("r3_min_sigma", "r3_max_sigma", "so3_min_sigma", "so3_max_sigma")] + \
               [("family", C.c_int), ("agg_mean", C.c_int), ("homomer", C.c_int)]


class OraScoreOut(C.Structure):
    _fields_ = [("tr_score", C.c_float * 3), ("rot_score", C.c_float * 3), ("energy", C.c_float),
                ("num_clashes", C.c_int64), ("confidence", C.c_float)]


class OraDebug(C.Structure):
    _fields_ = [("f", F32P), ("pos_out", F32P), ("h_layers", F32P), ("bins", I8P), ("relpos", I8P),
                ("edges", I32P), ("ires", F32P), ("dist", F32P), ("bins_in", I8P)]


class OraInject(C.Structure):
    _fields_ = [("R0", F64P), ("tr_draw", F32P), ("z_rot", F32P), ("z_tr", F32P), ("edges", I32P)]


class OraTrajOut(C.Structure):
    _fields_ = [("lig_pos", F32P), ("rot_update", C.c_float * 3), ("tr_update", C.c_float * 3),
                ("energy", C.c_float), ("num_clashes", C.c_int64), ("trace_pose", F32P),
                ("trace_scores", F32P), ("init_pose", F32P)]


def build(force: bool = False) -> str:
    """(Re)build the oracle library; `make` decides whether anything is stale."""
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    if os.path.exists(os.path.join(_HERE, "dfm_oracle.c")) and not os.environ.get("DFM_ORACLE_LIB"):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError("oracle library missing: run `make -C oracle`")
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ora_param_count.restype = C.c_int64
        for n in ("ora_r3_sigma", "ora_r3_g", "ora_so3_sigma", "ora_so3_g"):
            getattr(L, n).restype = C.c_double
            getattr(L, n).argtypes = [C.POINTER(OraHParams), C.c_double]
        L.ora_torch_reverse.argtypes = [C.c_double, F32P, C.c_float, C.c_float, F32P, C.c_int, F32P]
        L.ora_knn_sample.argtypes = [C.POINTER(OraHParams), F32P, C.c_int, C.c_uint64, I32P, C.POINTER(C.c_int)]
        L.ora_score.argtypes = [C.POINTER(OraHParams), F32P, C.c_int, C.c_int, F32P, F32P, F32P, F32P,
                                C.c_float, I32P, C.c_uint64, C.c_int, C.POINTER(OraScoreOut), C.POINTER(OraDebug)]
        L.ora_sample.argtypes = [C.POINTER(OraHParams), F32P, C.c_int, C.c_int, F32P, F32P, F32P, F32P,
                                 C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_uint64, C.POINTER(OraInject), C.POINTER(OraTrajOut)]
        L.ora_sample_many.argtypes = [C.POINTER(OraHParams), F32P, C.c_int, C.c_int, F32P, F32P, F32P, F32P,
                                      C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                      F32P, C.POINTER(C.c_int64), I32P, F32P]
        L.ora_modify_coords_all_atom.argtypes = [F32P, C.c_int, F32P, F32P]
        L.ora_modify_coords_all_atom.restype = None
        L.ora_set_num_threads.argtypes = [C.c_int]
        L.ora_set_num_threads.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=F32P):
    return a.ctypes.data_as(t) if a is not None else None


def hparams(hp=None) -> OraHParams:
    from dfmdock_amd.weights import HParams
    hp = hp or HParams()
    return OraHParams(homomer=0, **hp.as_dict())


class Oracle:
    """Convenience wrapper: one model (blob) + one complex."""

    def __init__(self, blob, cx, hp=None, homomer=False):
        self.hp = hparams(hp)
        self.hp.homomer = int(bool(homomer))      # 67-channel position matrix only: the complex's "sym" flag
        self.blob = _f32(blob)
        assert lib().ora_param_count(C.byref(self.hp)) == self.blob.size
        self.rec_x, self.lig_x = _f32(cx["rec_x"]), _f32(cx["lig_x"])
        self.rec_pos, self.lig_pos = _f32(cx["rec_pos"]), _f32(cx["lig_pos"])
        self.R, self.L = self.rec_x.shape[0], self.lig_x.shape[0]
        self.N = self.R + self.L
        knn, ns = self.hp.knn, self.hp.n_sample
        if self.N < knn:
            knn, ns = self.N, 0
        if self.N < knn + ns:
            ns = self.N - knn
        self.K = knn + ns

    def score(self, lig_pos, t, edges=None, seed=0, want_energy=True, debug=True, dist=False, bins=None):
        """bins [N,K,4] int8: evaluate with THESE feature bins instead of the computed ones (isolates a bin-boundary flip)."""
        lig_pos = _f32(lig_pos)
        N, K, L, H = self.N, self.K, self.L, self.hp.node_dim
        out = OraScoreOut()
        d = {}
        dbg = None
        bins_in = None if bins is None else np.ascontiguousarray(bins, dtype=np.int8).reshape(N, K, 4)
        if debug:
            d = dict(f=np.zeros((L, 3), np.float32), pos_out=np.zeros((N, 3), np.float32),
                     h_layers=np.zeros((self.hp.depth, N, H), np.float32), bins=np.zeros((N, K, 4), np.int8),
                     relpos=np.zeros((N, K), np.int8), edges=np.zeros((N, K), np.int32),
                     ires=np.zeros((N,), np.float32))
            if dist and self.hp.family == 1:
                d["dist_logits"] = np.zeros((self.R, L, 64), np.float32)
            dbg = OraDebug(_p(d["f"]), _p(d["pos_out"]), _p(d["h_layers"]), _p(d["bins"], I8P),
                           _p(d["relpos"], I8P), _p(d["edges"], I32P), _p(d["ires"]), _p(d.get("dist_logits")),
                           _p(bins_in, I8P))
        elif bins_in is not None:      # the bins override applies without the debug outputs too
            dbg = OraDebug(None, None, None, None, None, None, None, None, _p(bins_in, I8P))
        e = None if edges is None else np.ascontiguousarray(edges, dtype=np.int32)
        if e is not None:
            assert e.shape == (N, K), (e.shape, N, K)
        rc = lib().ora_score(C.byref(self.hp), _p(self.blob), self.R, self.L, _p(self.rec_x), _p(self.lig_x),
                             _p(self.rec_pos), _p(lig_pos), float(t), _p(e, I32P), int(seed), int(want_energy),
                             C.byref(out), C.byref(dbg) if dbg is not None else None)
        assert rc == 0
        res = dict(tr_score=np.array(out.tr_score, np.float32)[None], rot_score=np.array(out.rot_score, np.float32)[None],
                   energy=np.float32(out.energy), num_clashes=int(out.num_clashes), confidence=np.float32(out.confidence))
        res.update(d)
        return res

    def sample(self, num_steps=40, eps=1e-3, tr_noise_scale=0.5, rot_noise_scale=0.5, noise_annealing=False,
               use_clash_force=False, ode=False, max_forwards=0, seed=0, inject=None, trace=False):
        L = self.L
        lig = np.zeros((L, 3, 3), np.float32)
        tp = np.zeros((num_steps, L, 3, 3), np.float32) if trace else None
        tsc = np.zeros((num_steps + 1, 8), np.float32) if trace else None
        ip = np.zeros((L, 3, 3), np.float32)
        out = OraTrajOut()
        out.lig_pos = _p(lig)
        out.trace_pose = _p(tp)
        out.trace_scores = _p(tsc)
        out.init_pose = _p(ip)
        inj = None
        keep = []
        if inject is not None:
            inj = OraInject()
            if inject.get("R0") is not None:
                a = np.ascontiguousarray(inject["R0"], dtype=np.float64); keep.append(a); inj.R0 = _p(a, F64P)
            for k in ("tr_draw", "z_rot", "z_tr"):
                if inject.get(k) is not None:
                    a = _f32(inject[k]); keep.append(a); setattr(inj, k, _p(a))
            if inject.get("edges") is not None:
                a = np.ascontiguousarray(inject["edges"], dtype=np.int32); keep.append(a); inj.edges = _p(a, I32P)
        nf = lib().ora_sample(C.byref(self.hp), _p(self.blob), self.R, self.L, _p(self.rec_x), _p(self.lig_x),
                              _p(self.rec_pos), _p(self.lig_pos), int(num_steps), float(eps), float(tr_noise_scale),
                              float(rot_noise_scale), int(noise_annealing), int(use_clash_force), int(ode),
                              int(max_forwards), int(seed), C.byref(inj) if inj is not None else None, C.byref(out))
        return dict(lig_pos=lig, rot_update=np.array(out.rot_update, np.float32)[None],
                    tr_update=np.array(out.tr_update, np.float32)[None], energy=np.float32(out.energy),
                    num_clashes=int(out.num_clashes), trace_pose=tp, trace_scores=tsc, init_pose=ip, forwards=nf)


    def sample_many(self, n_traj, num_steps=40, eps=1e-3, tr_noise_scale=0.5, rot_noise_scale=0.5, max_forwards=0, seed=0, n_threads=0):
        """n_traj independent trajectories (seeds seed .. seed + n_traj - 1), one single-threaded trajectory per OpenMP thread
        (ora_sample_many): the trajectory-parallel CPU form of inference_base.py:644-657."""
        en = np.zeros(n_traj, np.float32)
        cl = np.zeros(n_traj, np.int64)
        fw = np.zeros(n_traj, np.int32)
        up = np.zeros((n_traj, 6), np.float32)
        tot = lib().ora_sample_many(C.byref(self.hp), _p(self.blob), self.R, self.L, _p(self.rec_x), _p(self.lig_x),
                                    _p(self.rec_pos), _p(self.lig_pos), int(num_steps), float(eps), float(tr_noise_scale),
                                    float(rot_noise_scale), int(max_forwards), int(seed), int(n_traj), int(n_threads),
                                    _p(en), cl.ctypes.data_as(C.POINTER(C.c_int64)), _p(fw, I32P), _p(up))
        return dict(energy=en, num_clashes=cl, forwards=fw, total_forwards=int(tot), rot_update=up[:, :3], tr_update=up[:, 3:])


# ---------------------------------------------------------------------------
# thin functional wrappers used by the golden-vector tests
def _vec_fn(name, n_in, n_out):
    def fn(a):
        a = _f32(a).reshape(-1, n_in)
        o = np.zeros((a.shape[0], n_out), np.float32)
        f = getattr(lib(), name)
        for i in range(a.shape[0]):
            f(_p(a[i]), _p(o[i]))
        return o
    return fn


axis_angle_to_matrix = _vec_fn("ora_axis_angle_to_matrix", 3, 9)
matrix_to_axis_angle = _vec_fn("ora_matrix_to_axis_angle", 9, 3)
matrix_to_quaternion = _vec_fn("ora_matrix_to_quaternion", 9, 4)


def rot_compose(r1, r2):
    r1, r2 = _f32(r1).reshape(-1, 3), _f32(r2).reshape(-1, 3)
    o = np.zeros_like(r1)
    for i in range(r1.shape[0]):
        lib().ora_rot_compose(_p(r1[i]), _p(r2[i]), _p(o[i]))
    return o


def modify_coords(x, rot, tr):
    x = _f32(x).copy()
    lib().ora_modify_coords(_p(x), x.shape[0], _p(_f32(rot).reshape(3)), _p(_f32(tr).reshape(3)))
    return x


def modify_coords_all_atom(x, rot, tr):
    """Second family (src/inference.py:244-254): rotation about the all-backbone-atom centroid."""
    x = _f32(x).copy()
    lib().ora_modify_coords_all_atom(_p(x), x.shape[0], _p(_f32(rot).reshape(3)), _p(_f32(tr).reshape(3)))
    return x


def clash_force(rec, lig):
    rec, lig = _f32(rec), _f32(lig)
    o = np.zeros(3, np.float32)
    lib().ora_clash_force(_p(rec), rec.shape[0], _p(lig), lig.shape[0], _p(o))
    return o


def diffusion_coefs(ts, hp=None):
    h = hparams(hp)
    L = lib()
    return (np.array([L.ora_r3_g(C.byref(h), float(t)) for t in ts]),
            np.array([L.ora_so3_g(C.byref(h), float(t)) for t in ts]),
            np.array([L.ora_r3_sigma(C.byref(h), float(t)) for t in ts]),
            np.array([L.ora_so3_sigma(C.byref(h), float(t)) for t in ts]))


def torch_reverse(g, score, dt, noise_scale, z, ode=False):
    o = np.zeros(3, np.float32)
    lib().ora_torch_reverse(float(g), _p(_f32(score).reshape(3)), float(dt), float(noise_scale),
                            _p(_f32(z).reshape(3)), int(ode), _p(o))
    return o


def coords6d_full(pos):
    pos = _f32(pos)
    N = pos.shape[0]
    outs = [np.zeros((N, N), np.float32) for _ in range(4)]
    lib().ora_coords6d_full(_p(pos), N, *[_p(o) for o in outs])
    return outs


def bins_full(pos, hp=None):
    pos = _f32(pos)
    N = pos.shape[0]
    b = np.zeros((N, N, 4), np.int8)
    h = hparams(hp)
    lib().ora_bins_full(C.byref(h), _p(pos), N, _p(b, I8P))
    return b


def relpos_full(R, Lg):
    N = R + Lg
    r = np.zeros((N, N), np.int8)
    lib().ora_relpos_full(int(R), int(Lg), _p(r, I8P))
    return r


def knn_sample(ca, seed=0, hp=None):
    ca = _f32(ca)
    N = ca.shape[0]
    h = hparams(hp)
    e = np.zeros((N, h.knn + h.n_sample), np.int32)
    K = C.c_int(0)
    lib().ora_knn_sample(C.byref(h), _p(ca), N, int(seed), _p(e, I32P), C.byref(K))
    return e.reshape(-1)[: N * K.value].reshape(N, K.value)

"""ctypes binding of the C ABI declared in include/dfmdock_amd.h.

The shared library is the product: if it is missing this module raises - there
is no Python / CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFM_LIB") or os.path.join(_HERE, "libdfmdock_amd.so")   # DFM_LIB: A/B builds of the same engine

F32P = C.POINTER(C.c_float)
I32P = C.POINTER(C.c_int32)
U32P = C.POINTER(C.c_uint32)

DFM_F_MFMA16 = 1 << 0
DFM_F_BF16 = DFM_F_MFMA16      # name of rounds 1-2
DFM_F_ENERGY = 1 << 1
DFM_F_NOISE_ANNEALING = 1 << 2
DFM_F_CLASH_FORCE = 1 << 3
DFM_F_ODE = 1 << 4
DFM_F_PROFILE = 1 << 5
DFM_F_STEP_ENERGY = 1 << 6
DFM_F_F16 = 1 << 7
DFM_F_IRES = 1 << 8
DFM_F_BF16_OPS = 1 << 9
DFM_F_DIST = 1 << 10
DFM_F_L0_TABLE = 1 << 11       # dfm_score: layer 0 through the per-complex message table
DFM_F_NO_L0_TABLE = 1 << 12    # dfm_sample: layer 0 evaluated directly
DFM_F_GRAPH = 1 << 13          # dfm_sample: replay one captured step as a hipGraph (opt-in)

EXPORTS = [
    "dfm_last_error", "dfm_config_string", "dfm_device_count", "dfm_set_device", "dfm_default_hparams", "dfm_param_count",
    "dfm_model_create", "dfm_model_destroy", "dfm_complex_create", "dfm_complex_destroy", "dfm_complex_degree",
    "dfm_complex_set_pose", "dfm_complex_set_homomer",
    "dfm_score", "dfm_sample", "dfm_get_profile", "dfm_diffusion_coef", "dfm_complex_selfcheck", "dfm_trim_cache",
]


class HParamsC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("lm_embed_dim", "positional_embed_dim", "spatial_embed_dim", "node_dim",
                                       "edge_dim", "inner_dim", "depth", "knn", "n_sample")] + \
               [("cut_off", C.c_float), ("mask_dist", C.c_float)] + \
               [(n, C.c_double) for n in ("r3_min_sigma", "r3_max_sigma", "so3_min_sigma", "so3_max_sigma")] + \
               [("family", C.c_int), ("agg_mean", C.c_int)]


class ScoreOutC(C.Structure):
    _fields_ = [("tr_score", F32P), ("rot_score", F32P), ("energy", F32P), ("num_clashes", I32P), ("f", F32P),
                ("h_last", F32P), ("h_first", F32P), ("edges", I32P), ("edge_codes", U32P), ("confidence", F32P),
                ("ires", F32P), ("dist_logits", F32P)]


class InjectC(C.Structure):
    _fields_ = [("R0", F32P), ("tr_draw", F32P), ("z_rot", F32P), ("z_tr", F32P), ("edges", I32P)]


class TrajOutC(C.Structure):
    _fields_ = [("lig_pos", F32P), ("rot_update", F32P), ("tr_update", F32P), ("energy", F32P),
                ("num_clashes", I32P), ("final_scores", F32P), ("trace_pose", F32P), ("trace_scores", F32P),
                ("init_pose", F32P)]


class ProfileC(C.Structure):
    _fields_ = [("edge_kernel_ms", C.c_double), ("edge_kernel_launches", C.c_int64), ("edge_rows", C.c_int64),
                ("total_ms", C.c_double), ("phase_cycles", C.c_double * 4), ("slot_cycles", C.c_double * 16),
                ("l0_evals", C.c_int64), ("l0_edges", C.c_int64), ("l0_miss_rows", C.c_int64),
                ("l0_rows_ms", C.c_double), ("l0_gather_ms", C.c_double), ("l0_build_ms", C.c_double),
                ("edge_lig_launches", C.c_int64), ("edge_lig_ms", C.c_double),
                ("edge_shader_cycles", C.c_double), ("edge_ref_ticks", C.c_double)]


class SelfcheckC(C.Structure):
    _fields_ = [("n_eval", C.c_int), ("depth", C.c_int),
                ("dev_f", C.c_float), ("dev_tr_score", C.c_float), ("dev_rot_score", C.c_float), ("dev_energy", C.c_float),
                ("cancel_ratio", C.c_float * 2), ("score_bound", C.c_float * 2),
                ("gate_f", C.c_float), ("gate_score", C.c_float), ("gate_energy", C.c_float), ("limit", C.c_float),
                ("max_h", C.c_float * 9), ("max_A", C.c_float * 8), ("max_Bm", C.c_float * 8), ("max_tab", C.c_float * 8),
                ("max_sum16", C.c_float * 8), ("max_pre", C.c_float * 8), ("max_acc", C.c_float * 8), ("headroom", C.c_float),
                ("saturated", C.c_int64), ("range_ok", C.c_int), ("dev_ok", C.c_int), ("ok", C.c_int)]


_lib = None


def lib():
    """Load libdfmdock_amd.so (built by `make -C dfmdock_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (or `make -C dfmdock_amd/csrc`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.dfm_last_error.restype = C.c_char_p
    L.dfm_config_string.restype = C.c_char_p
    L.dfm_device_count.argtypes = [C.POINTER(C.c_int)]
    L.dfm_set_device.argtypes = [C.c_int]
    L.dfm_default_hparams.argtypes = [C.POINTER(HParamsC)]
    L.dfm_default_hparams.restype = None
    L.dfm_param_count.argtypes = [C.POINTER(HParamsC)]
    L.dfm_param_count.restype = C.c_int64
    L.dfm_model_create.argtypes = [F32P, C.c_size_t, C.POINTER(HParamsC)]
    L.dfm_model_create.restype = C.c_void_p
    L.dfm_model_destroy.argtypes = [C.c_void_p]
    L.dfm_model_destroy.restype = None
    L.dfm_complex_create.argtypes = [C.c_void_p, F32P, F32P, F32P, F32P, C.c_int, C.c_int]
    L.dfm_complex_create.restype = C.c_void_p
    L.dfm_complex_destroy.argtypes = [C.c_void_p]
    L.dfm_complex_destroy.restype = None
    L.dfm_complex_degree.argtypes = [C.c_void_p]
    L.dfm_complex_set_pose.argtypes = [C.c_void_p, F32P, F32P]
    L.dfm_complex_set_homomer.argtypes = [C.c_void_p, C.c_int]
    L.dfm_score.argtypes = [C.c_void_p, C.c_int, F32P, F32P, I32P, C.c_uint64, C.c_uint32, C.POINTER(ScoreOutC)]
    L.dfm_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint64,
                             C.POINTER(InjectC), C.POINTER(TrajOutC)]
    L.dfm_get_profile.argtypes = [C.c_void_p, C.POINTER(ProfileC)]
    L.dfm_complex_selfcheck.argtypes = [C.c_void_p, C.c_int, F32P, C.c_uint64, C.c_uint32, C.POINTER(SelfcheckC)]
    L.dfm_trim_cache.argtypes = [C.c_int]
    L.dfm_trim_cache.restype = C.c_longlong
    L.dfm_diffusion_coef.argtypes = [C.POINTER(HParamsC), C.c_int, C.c_double, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    _lib = L
    return L


class DfmError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dfm_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")   # the reference raises ValueError for these
        raise DfmError(f"{what}: status {rc}: {msg}")

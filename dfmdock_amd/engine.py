"""Numpy-level wrapper over the C ABI: Model / Complex handles, batched score and sample calls."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .weights import HParams


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=L.F32P):
    return a.ctypes.data_as(t) if a is not None else None


PRECISIONS = ("mfma16", "f16", "fp32")
_warned_bf16 = False


def precision_kwargs(precision: str) -> dict:
    """Engine selector -> keyword arguments of Complex.score / Complex.sample.

      "mfma16"  the 16-bit MFMA engine as shipped (DFM_F_MFMA16): fp16 MFMA operands, fp32 accumulation (config_string())
      "f16"     the same with fp32 A_i (DFM_F_F16)
      "fp32"    exact fp32 (the reference's own arithmetic)
      "bf16"    DEPRECATED alias of "mfma16" - the name of rounds 1-3, kept so that old command lines keep working; the
                engine it selects has computed on fp16 operands since r03, which is what the name now says
    """
    global _warned_bf16
    if precision == "bf16":
        if not _warned_bf16:
            import sys
            print('dfmdock_amd: precision "bf16" is a deprecated alias of "mfma16" (fp16 MFMA operands, fp32 accumulation)', file=sys.stderr)
            _warned_bf16 = True
        precision = "mfma16"
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS} (or the deprecated alias 'bf16'), got {precision!r}")
    return {"mfma16": precision == "mfma16", "f16": precision == "f16"}


def canonical_precision(precision: str) -> str:
    return "mfma16" if precision == "bf16" else precision


def hparams_c(hp: HParams | None = None) -> L.HParamsC:
    hp = hp or HParams()
    return L.HParamsC(**hp.as_dict())


def set_device(index: int = 0):
    L.check(L.lib().dfm_set_device(int(index)), "dfm_set_device")


def config_string() -> str:
    """The precision plan and every diagnostic switch in force in this process (dfm_config_string)."""
    return L.lib().dfm_config_string().decode()


def trim_cache(device: int = -1) -> int:
    """Hand the device blocks parked by destroyed handles back to the driver (dfm_trim_cache); returns the bytes freed."""
    return int(L.lib().dfm_trim_cache(int(device)))


def device_count() -> int:
    n = C.c_int(0)
    rc = L.lib().dfm_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def diffusion_coef(which: int, t: float, hp: HParams | None = None):
    """(g, sigma) of the R^3 (which=0) or SO(3) (which=1) VE-SDE, float64 like the reference."""
    h = hparams_c(hp)
    g, s = C.c_double(0), C.c_double(0)
    L.check(L.lib().dfm_diffusion_coef(C.byref(h), int(which), float(t), C.byref(g), C.byref(s)), "diffusion_coef")
    return g.value, s.value


def format_selfcheck(r: dict, name: str = "") -> str:
    """One line for logs: what a checkpoint owner reads before trusting the 16-bit engine."""
    worst = {k: max(r[k]) for k in ("max_A", "max_Bm", "max_sum16", "max_pre", "max_acc")}
    top = max(worst, key=worst.get)
    return (f"selfcheck {name}: {r['precision']} vs fp32 on {r['n_eval']} graphs: f {r['dev_f']:.2e} tr {r['dev_tr_score']:.2e} "
            f"rot {r['dev_rot_score']:.2e} E {r['dev_energy']:.2e} (gates {r['gate_f']:.0e} / {r['gate_score']:.0e} / {r['gate_energy']:.0e}; "
            f"cancellation tr {r['cancel_ratio'][0]:.3f} rot {r['cancel_ratio'][1]:.3f}, pooled-vector bounds {r['score_bound'][0]:.1e} / "
            f"{r['score_bound'][1]:.1e}) | fp16 range: max|h| {max(r['max_h']):.3g}, largest stored magnitude {worst[top]:.3g} ({top[4:]}), "
            f"headroom x{r['headroom']:.3g}, saturated {r['saturated']} | {'OK' if r['ok'] else 'FAILED: ' + ('range ' if not r['range_ok'] else '') + ('deviation' if not r['dev_ok'] else '')}")


class Model:
    """Device-resident weights (dfm_model).  `blob` is the flat float32 state_dict (weights.pack_blob)."""

    def __init__(self, blob, hp: HParams | None = None):
        self.hp = hp or HParams()
        self._hp_c = hparams_c(self.hp)
        blob = _f32(blob).reshape(-1)
        self._h = L.lib().dfm_model_create(_p(blob), blob.size, C.byref(self._hp_c))
        if not self._h:
            L.check(-1 if b"blob" in L.lib().dfm_last_error() or b"unsupported" in L.lib().dfm_last_error() else -2,
                    "dfm_model_create")

    def close(self):
        if getattr(self, "_h", None):
            L.lib().dfm_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Complex:
    """One receptor/ligand pair resident on the GPU (dfm_complex)."""

    def __init__(self, model: Model, rec_x, lig_x, rec_pos, lig_pos):
        self.model = model
        rec_x, lig_x = _f32(rec_x), _f32(lig_x)
        rec_pos, lig_pos = _f32(rec_pos).reshape(-1, 9), _f32(lig_pos).reshape(-1, 9)
        self.R, self.L = rec_x.shape[0], lig_x.shape[0]
        if rec_x.shape[1] != model.hp.lm_embed_dim or lig_x.shape[1] != model.hp.lm_embed_dim:
            raise ValueError("node features must be [n, lm_embed_dim]")
        if rec_pos.shape[0] != self.R or lig_pos.shape[0] != self.L:
            raise ValueError("positions must be [n, 3, 3] matching the features")
        self.N = self.R + self.L
        self._h = L.lib().dfm_complex_create(model._h, _p(rec_x), _p(lig_x), _p(rec_pos), _p(lig_pos), self.R, self.L)
        if not self._h:
            L.check(-2, "dfm_complex_create")
        self.K = L.lib().dfm_complex_degree(self._h)
        self.lig_pos0 = lig_pos.reshape(self.L, 3, 3).copy()

    def close(self):
        if getattr(self, "_h", None):
            L.lib().dfm_complex_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def set_pose(self, rec_pos=None, lig_pos=None):
        """Replace the resident receptor pose and / or the ligand start pose of `sample` (features stay resident)."""
        rp = None if rec_pos is None else _f32(rec_pos).reshape(-1, 9)
        lp = None if lig_pos is None else _f32(lig_pos).reshape(-1, 9)
        if (rp is not None and rp.shape[0] != self.R) or (lp is not None and lp.shape[0] != self.L):
            raise ValueError("set_pose: positions must keep the residue counts of the complex")
        L.check(L.lib().dfm_complex_set_pose(self._h, _p(rp), _p(lp)), "dfm_complex_set_pose")
        if lp is not None:
            self.lig_pos0 = lp.reshape(self.L, 3, 3).copy()

    def set_homomer(self, flag: bool):
        """Value of the 67th ("sym") position channel (positional_embed_dim = 67 models only)."""
        L.check(L.lib().dfm_complex_set_homomer(self._h, int(bool(flag))), "dfm_complex_set_homomer")

    def score(self, lig_pos, t, edges=None, seed=0, mfma16=False, energy=True, debug=False, profile=False, f16=False,
              ires=False, return_edges=False, bf16_ops=False, dist=False, bf16=False, l0_table=False):
        """B score evaluations.  lig_pos [B,L,3,3] (or [L,3,3]), t [B] (or scalar).  mfma16: the 16-bit MFMA engine
        (`bf16=` is its deprecated keyword of rounds 1-3).  l0_table: layer 0 through the per-complex message table
        (DFM_F_L0_TABLE; the mfma16 and the fp32 engine, a table each - `sample` uses it by default, `score` only on request)."""
        mfma16 = mfma16 or bf16
        lig_pos = _f32(lig_pos)
        if lig_pos.ndim == 3:
            lig_pos = lig_pos[None]
        B = lig_pos.shape[0]
        t = np.broadcast_to(_f32(t).reshape(-1), (B,)).copy()
        N, Lg, K, H = self.N, self.L, self.K, self.model.hp.node_dim
        o = dict(tr_score=np.zeros((B, 3), np.float32), rot_score=np.zeros((B, 3), np.float32),
                 energy=np.zeros((B,), np.float32), num_clashes=np.zeros((B,), np.int32),
                 f=np.zeros((B, Lg, 3), np.float32), confidence=np.zeros((B,), np.float32))
        out = L.ScoreOutC()
        out.tr_score, out.rot_score = _p(o["tr_score"]), _p(o["rot_score"])
        out.energy, out.num_clashes, out.f = _p(o["energy"]), _p(o["num_clashes"], L.I32P), _p(o["f"])
        out.confidence = _p(o["confidence"])
        if ires:
            o["ires"] = np.zeros((B, N), np.float32)
            out.ires = _p(o["ires"])
        if dist:      # family 1 only: dist_logits [B,R,L,64] (egnn_net.py:447)
            o["dist_logits"] = np.zeros((B, self.R, Lg, 64), np.float32)
            out.dist_logits = _p(o["dist_logits"])
        if return_edges and not debug:      # the graph each evaluation used, without the [B,N,H] debug taps
            o["edges"] = np.zeros((B, N, K), np.int32)
            out.edges = _p(o["edges"], L.I32P)
        if debug:
            o.update(h_last=np.zeros((B, N, H), np.float32), h_first=np.zeros((B, N, H), np.float32),
                     edges=np.zeros((B, N, K), np.int32), edge_codes=np.zeros((B, N, K), np.uint32))
            out.h_last, out.h_first = _p(o["h_last"]), _p(o["h_first"])
            out.edges, out.edge_codes = _p(o["edges"], L.I32P), _p(o["edge_codes"], L.U32P)
        e = None
        if edges is not None:
            e = np.ascontiguousarray(edges, dtype=np.int32)
            if e.ndim == 2:
                e = e[None]
            if e.shape != (B, N, K):
                raise ValueError(f"edges must be [B,N,K] = {(B, N, K)}, got {e.shape}")
        flags = (L.DFM_F_MFMA16 if mfma16 else 0) | (L.DFM_F_ENERGY if energy else 0) | (L.DFM_F_PROFILE if profile else 0) | \
                (L.DFM_F_F16 if f16 else 0) | (L.DFM_F_IRES if ires else 0) | (L.DFM_F_BF16_OPS if bf16_ops else 0) | \
                (L.DFM_F_DIST if dist else 0) | (L.DFM_F_L0_TABLE if l0_table else 0)
        rc = L.lib().dfm_score(self._h, B, _p(lig_pos), _p(t), _p(e, L.I32P), int(seed), flags, C.byref(out))
        L.check(rc, "dfm_score")
        if debug:
            c = o["edge_codes"]
            o["bins"] = np.stack([c & 63, (c >> 6) & 31, (c >> 11) & 31, (c >> 16) & 15], -1).astype(np.int8)
            o["relpos"] = ((c >> 20) & 127).astype(np.int8)
        return o

    def sample(self, B=1, num_steps=40, eps=1e-3, tr_noise_scale=0.5, rot_noise_scale=0.5, noise_annealing=False,
               use_clash_force=False, ode=False, seed=0, mfma16=False, inject=None, trace=False, profile=False, f16=False, bf16_ops=False,
               bf16=False, l0_table=True, graph=False, step_energy=None):
        """B independent Euler-Maruyama trajectories (inference_base.py:390-468 batched).  l0_table=False: DFM_F_NO_L0_TABLE
        (layer 0 evaluated edge by edge even where the per-complex message table applies).  graph=True: DFM_F_GRAPH (one captured
        step replayed as a hipGraph instead of every launch enqueued by the host; bitwise the same results, no faster on MI355X).
        trace=True returns the pose after every step and the scores of every evaluation; by default it also asks for the energy
        head on every step (DFM_F_STEP_ENERGY - the step evaluations then run their last layer in full); step_energy=False keeps
        the step evaluations exactly as an untraced call runs them (ligand-only last layer, no energy in trace_scores[:, :-1])."""
        mfma16 = mfma16 or bf16
        if step_energy is None:
            step_energy = bool(trace)
        Lg, N, K, S = self.L, self.N, self.K, int(num_steps)
        o = dict(lig_pos=np.zeros((B, Lg, 3, 3), np.float32), rot_update=np.zeros((B, 3), np.float32),
                 tr_update=np.zeros((B, 3), np.float32), energy=np.zeros((B,), np.float32),
                 num_clashes=np.zeros((B,), np.int32), final_scores=np.zeros((B, 6), np.float32))
        out = L.TrajOutC()
        out.lig_pos, out.rot_update, out.tr_update = _p(o["lig_pos"]), _p(o["rot_update"]), _p(o["tr_update"])
        out.energy, out.num_clashes = _p(o["energy"]), _p(o["num_clashes"], L.I32P)
        out.final_scores = _p(o["final_scores"])
        if trace:
            o.update(trace_pose=np.zeros((B, S, Lg, 3, 3), np.float32), trace_scores=np.zeros((B, S + 1, 8), np.float32),
                     init_pose=np.zeros((B, Lg, 3, 3), np.float32))
            out.trace_pose, out.trace_scores, out.init_pose = _p(o["trace_pose"]), _p(o["trace_scores"]), _p(o["init_pose"])
        inj, keep = None, []
        if inject:
            inj = L.InjectC()
            shapes = {"R0": (B, 9), "tr_draw": (B, 3), "z_rot": (B, S, 3), "z_tr": (B, S, 3)}
            for k, shp in shapes.items():
                if inject.get(k) is not None:
                    a = _f32(inject[k]).reshape(shp)
                    keep.append(a)
                    setattr(inj, k, _p(a))
            if inject.get("edges") is not None:
                a = np.ascontiguousarray(inject["edges"], dtype=np.int32).reshape(B, S + 1, N, K)
                keep.append(a)
                inj.edges = _p(a, L.I32P)
        flags = (L.DFM_F_MFMA16 if mfma16 else 0) | (L.DFM_F_NOISE_ANNEALING if noise_annealing else 0) | \
                (L.DFM_F_CLASH_FORCE if use_clash_force else 0) | (L.DFM_F_ODE if ode else 0) | \
                (L.DFM_F_PROFILE if profile else 0) | (L.DFM_F_STEP_ENERGY if step_energy else 0) | (L.DFM_F_F16 if f16 else 0) | \
                (L.DFM_F_BF16_OPS if bf16_ops else 0) | (0 if l0_table else L.DFM_F_NO_L0_TABLE) | \
                (L.DFM_F_GRAPH if graph else 0)
        rc = L.lib().dfm_sample(self._h, int(B), S, float(eps), float(tr_noise_scale), float(rot_noise_scale), flags,
                                int(seed), C.byref(inj) if inj is not None else None, C.byref(out))
        L.check(rc, "dfm_sample")
        return o

    def selfcheck(self, n_eval=4, t=None, seed=0, precision="mfma16", bf16_ops=False):
        """dfm_complex_selfcheck: the stored pose through the fp32 engine and the 16-bit engine `precision` names on the same
        n_eval engine-drawn graphs; deviations, cancellation ratios and the fp16 range telemetry as a dict (include/dfmdock_amd.h:
        dfm_selfcheck_out).  `ok` False = do not trust the 16-bit engine on this model / complex: run precision="fp32"."""
        kw = precision_kwargs(precision)
        if not (kw["mfma16"] or kw["f16"]):
            raise ValueError("selfcheck compares a 16-bit engine with the fp32 engine: precision must be 'mfma16' or 'f16'")
        flags = (L.DFM_F_F16 if kw["f16"] else L.DFM_F_MFMA16) | (L.DFM_F_BF16_OPS if bf16_ops else 0)
        tt = None
        if t is not None:
            tt = _f32(t).reshape(-1)
            n_eval = tt.size
        out = L.SelfcheckC()
        L.check(L.lib().dfm_complex_selfcheck(self._h, int(n_eval), _p(tt), int(seed), flags, C.byref(out)), "dfm_complex_selfcheck")
        d = out.depth
        r = {k: getattr(out, k) for k in ("n_eval", "depth", "dev_f", "dev_tr_score", "dev_rot_score", "dev_energy", "gate_f", "gate_score",
                                          "gate_energy", "limit", "headroom", "saturated")}
        r.update(cancel_ratio=list(out.cancel_ratio), score_bound=list(out.score_bound), max_h=list(out.max_h)[: d + 1],
                 range_ok=bool(out.range_ok), dev_ok=bool(out.dev_ok), ok=bool(out.ok), precision=canonical_precision(precision))
        for k in ("max_A", "max_Bm", "max_tab", "max_sum16", "max_pre", "max_acc"):
            r[k] = list(getattr(out, k))[:d]
        return r

    def profile(self):
        p = L.ProfileC()
        L.check(L.lib().dfm_get_profile(self._h, C.byref(p)), "dfm_get_profile")
        return dict(edge_kernel_ms=p.edge_kernel_ms, edge_kernel_launches=p.edge_kernel_launches,
                    edge_rows=p.edge_rows, total_ms=p.total_ms, phase_cycles=list(p.phase_cycles), slot_cycles=list(p.slot_cycles),
                    l0_evals=p.l0_evals, l0_edges=p.l0_edges, l0_miss_rows=p.l0_miss_rows, l0_rows_ms=p.l0_rows_ms,
                    l0_gather_ms=p.l0_gather_ms, l0_build_ms=p.l0_build_ms, edge_lig_launches=p.edge_lig_launches, edge_lig_ms=p.edge_lig_ms,
                    edge_shader_cycles=p.edge_shader_cycles, edge_ref_ticks=p.edge_ref_ticks,
                    edge_sclk_mhz=(100.0 * p.edge_shader_cycles / p.edge_ref_ticks) if p.edge_ref_ticks > 0 else None)

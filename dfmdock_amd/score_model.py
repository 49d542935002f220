"""Host-side mirror of the reference's Python interface for the sampling path.

Same names, argument meaning, dict keys, shapes and error behaviour as the reference, implemented
over the C ABI (include/dfmdock_amd.h) - nothing here computes on the CPU except scalar schedule
math.  PyTorch is only the tensor container the reference's callers expect.

  Score_Model                 <- src/models/score_model_mlsb.py:22-63 (forward, r3_diffuser, so3_diffuser)
  R3Diffuser / SO3Diffuser    <- src/utils/r3_diffuser.py:15-55, src/utils/so3_diffuser.py:140-369
                                 (sigma, diffusion_coef, torch_reverse only: the IGSO(3) tables are training-only)
  Euler_Maruyama_sampler      <- src/inference_base.py:390-468
  sample_trajectories         <- the `for i in range(num_samples)` loops of src/inference_base.py:483,:644,
                                 batched on the GPU
  DFMDock                     <- src/models/DFMDock.py:21-75 (second model family: EGNN_Net behind
                                 move_to_lig_center), forward(batch) only - the training half is out of scope
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import engine
from .weights import HParams, pack_blob


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def _digest(a: np.ndarray) -> bytes:
    return hashlib.blake2b(memoryview(np.ascontiguousarray(a)).cast("B"), digest_size=16).digest()


def _version(x):
    """Mutation counter of a torch tensor (None for anything else: its content is then hashed on every call)."""
    return getattr(x, "_version", None) if hasattr(x, "detach") else None


class R3Diffuser:
    """VE-SDE on R^3 (src/utils/r3_diffuser.py)."""

    def __init__(self, hp: HParams):
        self._hp = hp
        self.min_sigma, self.max_sigma = hp.r3_min_sigma, hp.r3_max_sigma

    def sigma(self, t):
        return engine.diffusion_coef(0, float(t), self._hp)[1]

    def diffusion_coef(self, t):
        return engine.diffusion_coef(0, float(t), self._hp)[0]

    def torch_reverse(self, score_t, dt, t, noise_scale=1.0, ode=False):
        return _torch_reverse(self.diffusion_coef, score_t, dt, t, noise_scale, ode)


class SO3Diffuser:
    """VE-SDE on SO(3), logarithmic schedule (src/utils/so3_diffuser.py:210-227,:344-369)."""

    def __init__(self, hp: HParams):
        self._hp = hp
        self.min_sigma, self.max_sigma, self.schedule = hp.so3_min_sigma, hp.so3_max_sigma, "logarithmic"

    def sigma(self, t):
        return engine.diffusion_coef(1, float(t), self._hp)[1]     # ValueError outside [0, 1]

    def diffusion_coef(self, t):
        return engine.diffusion_coef(1, float(t), self._hp)[0]

    def torch_reverse(self, score_t, dt, t, noise_scale=1.0, ode=False):
        return _torch_reverse(self.diffusion_coef, score_t, dt, t, noise_scale, ode)


def _torch_reverse(coef, score_t, dt, t, noise_scale, ode):
    """One reverse-SDE increment (r3_diffuser.py:40-55 == so3_diffuser.py:344-369), float32 tensor maths."""
    import torch
    if not np.isscalar(t):
        raise ValueError(f"{t} must be a scalar.")
    g_t = coef(t)
    dt = torch.as_tensor(dt, dtype=torch.float32)
    if not ode:
        z = noise_scale * torch.randn(1, 3, device=score_t.device)
        perturb = (g_t ** 2) * score_t * dt + g_t * torch.sqrt(dt) * z
    else:
        perturb = 0.5 * (g_t ** 2) * score_t * dt
    return perturb.float()


class Score_Model:
    """Drop-in for the reference's Score_Model at inference: `model(batch) -> dict`.

    batch: {rec_x [R,1301], lig_x [L,1301], rec_pos [R,3,3], lig_pos [L,3,3], t [1]} (position_matrix is
    accepted and ignored: relpos is derived from (R, L) on the GPU).  Output keys / shapes follow
    score_net_mlsb.py:413-425: tr_score [1,3], rot_score [1,3], energy [], f [L,3], num_clashes [] (int64),
    ires [N,1].  `with_ires=False` skips the interface-residue head (three fp32 GEMMs on [N,512] per call that no sampler
    reads) and leaves the key out.
    """

    def __init__(self, weights, hp: HParams | None = None, precision: str = "mfma16", device_index: int = 0, seed: int = 0,
                 with_ires: bool = True):
        self.hp = hp or HParams()
        self.with_ires = bool(with_ires)
        blob = weights if isinstance(weights, np.ndarray) and weights.ndim == 1 else pack_blob(weights, self.hp)
        engine.set_device(device_index)
        self.model = engine.Model(blob, self.hp)
        engine.precision_kwargs(precision)                      # validates; "bf16" = deprecated alias of "mfma16" (prints a note once)
        self.precision = engine.canonical_precision(precision)
        self.r3_diffuser = R3Diffuser(self.hp)
        self.so3_diffuser = SO3Diffuser(self.hp)
        self._cx = None
        self._cx_key = None
        self._pose_key = None
        self._feat_refs = None
        self._homomer = False
        self._calls = 0
        self.seed = seed

    # reference spelling ---------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, path, map_location=None, **kw):
        from .weights import load_lightning_checkpoint
        sd, hp = load_lightning_checkpoint(path)
        return cls(sd, hp=hp, **kw)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ----------------------------------------------------------------------------------------------
    def complex_for(self, batch) -> engine.Complex:
        """The device-resident complex of `batch`.  The expensive part (feature upload + node embedding) is keyed on the
        CONTENT of rec_x / lig_x; the poses are compared on every call and, when they changed (a second ligand
        conformation, a re-centred receptor - DFMDock.move_to_lig_center), pushed with dfm_complex_set_pose, so that
        sampler calls always start from batch['lig_pos'] (inference_base.py:408-412)."""
        rec_x, lig_x = batch["rec_x"], batch["lig_x"]
        same_obj = (self._feat_refs is not None and rec_x is self._feat_refs[0] and lig_x is self._feat_refs[1]
                    and _version(rec_x) is not None and (_version(rec_x), _version(lig_x)) == self._feat_refs[2])
        rec_pos, lig_pos = _np(batch["rec_pos"]), _np(batch["lig_pos"])
        if not same_obj:
            rx, lx = _np(rec_x), _np(lig_x)
            key = (rx.shape, lx.shape, _digest(rx), _digest(lx))
            if key != self._cx_key:
                if self._cx is not None:
                    self._cx.close()
                self._cx = engine.Complex(self.model, rx, lx, rec_pos, lig_pos)
                self._cx_key = key
                self._pose_key = (_digest(rec_pos), _digest(lig_pos))
                self._homomer = False
            # holding the objects keeps their ids from being reused; _version catches in-place edits of torch tensors
            self._feat_refs = (rec_x, lig_x, (_version(rec_x), _version(lig_x)))
        pose_key = (_digest(rec_pos), _digest(lig_pos))
        if pose_key != self._pose_key:
            self._cx.set_pose(rec_pos if pose_key[0] != self._pose_key[0] else None,
                              lig_pos if pose_key[1] != self._pose_key[1] else None)
            self._pose_key = pose_key
        if self.hp.positional_embed_dim == 67:
            hom = bool(batch.get("is_homomer", False))
            if hom != self._homomer:
                self._cx.set_homomer(hom)
                self._homomer = hom
        return self._cx

    def selfcheck(self, batch, n_eval=4, seed=0):
        """Runtime parity evidence on THIS checkpoint and THIS complex (no reference counterpart - the reference is fp32 throughout):
        batch's pose through the fp32 engine and through the 16-bit engine this model was built for, on the same engine-drawn graphs.
        Returns engine.Complex.selfcheck's dict (deviations of f / tr_score / rot_score / energy against SURVEY 8(d)'s gates, the
        cancellation ratios that condition the scores, per-layer fp16 range telemetry, `ok`).  A model built with precision="fp32" is
        checked against the default 16-bit engine - what switching it to "mfma16" would cost."""
        cx = self.complex_for(batch)
        return cx.selfcheck(n_eval=n_eval, seed=seed, precision=self.precision if self.precision != "fp32" else "mfma16")

    _ires_key = "ires"              # score_net_mlsb.py:413-425

    def _score_dict(self, batch):
        import torch
        cx = self.complex_for(batch)
        t = _np(batch["t"]).reshape(-1)
        if t.size != 1:
            raise ValueError("batch['t'] must hold one time value (the reference runs batch_size=1)")
        self._calls += 1
        r = cx.score(_np(batch["lig_pos"]), t, seed=self.seed + self._calls, energy=True, **engine.precision_kwargs(self.precision),
                     ires=self.with_ires, dist=getattr(self, "with_dist", False))
        out = {"tr_score": torch.from_numpy(r["tr_score"]), "rot_score": torch.from_numpy(r["rot_score"]),
               "energy": torch.tensor(float(r["energy"][0]), dtype=torch.float32), "f": torch.from_numpy(r["f"][0]),
               "num_clashes": torch.tensor(int(r["num_clashes"][0]), dtype=torch.int64)}
        if self.with_ires:
            out[self._ires_key] = torch.from_numpy(r["ires"][0].reshape(-1, 1).copy())
        return out, r

    def forward(self, batch):
        return self._score_dict(batch)[0]

    __call__ = forward


class DFMDock(Score_Model):
    """Drop-in for the reference's second model family at inference: ``DFMDock.forward(batch)`` =
    ``move_to_lig_center`` + ``EGNN_Net(batch, predict=True)`` (DFMDock.py:68-75, egnn_net.py:408-505).

    Output keys follow egnn_net.py:486-495: tr_score [1,3], rot_score [1,3], energy [], f [L,3], num_clashes [],
    confidence_logits [], ires_logits [N,1]; ``dist_logits`` [R,L,64] - a training-loss input (DFMDock.py:196-215) - with
    ``with_dist=True`` (off by default: 64 floats per residue pair).  With a 67-channel checkpoint (configs/model/DFMDock.yaml:5) ``batch['is_homomer']`` selects the
    value of the sym channel (default False).  The sampler of this family rotates about the all-backbone-atom centroids
    (src/inference.py:220-254); the engine does the same for ``family=1`` models.  The diffusers and the Euler-Maruyama sampler are shared with
    Score_Model, so ``Euler_Maruyama_sampler(model, batch)`` / ``sample_trajectories`` accept this class too.
    """

    def __init__(self, weights, hp: HParams | None = None, precision: str = "mfma16", device_index: int = 0, seed: int = 0,
                 with_ires: bool = True, with_dist: bool = False):
        hp = hp or HParams(family=1, mask_dist=20.0)
        if hp.family != 1:
            raise ValueError("DFMDock needs HParams(family=1)")
        super().__init__(weights, hp=hp, precision=precision, device_index=device_index, seed=seed, with_ires=with_ires)
        self.with_dist = bool(with_dist)      # dist_logits [R,L,64] (egnn_net.py:447,:500): 64 floats per residue pair, off by default

    _ires_key = "ires_logits"       # egnn_net.py:486-495

    def forward(self, batch):
        import torch
        out, r = self._score_dict(batch)
        out["confidence_logits"] = torch.tensor(float(r["confidence"][0]), dtype=torch.float32)
        if self.with_dist:
            out["dist_logits"] = torch.from_numpy(r["dist_logits"][0].copy())
        return out

    __call__ = forward

    # host-side helpers of the reference wrapper, kept for callers that build their own perturbed poses ---------------
    @staticmethod
    def move_to_lig_center(batch):
        """DFMDock.py:254-257: subtract the mean over all L x 3 ligand backbone atoms from both chains (in place)."""
        center = batch["lig_pos"].mean(dim=(0, 1)) if hasattr(batch["lig_pos"], "dim") else batch["lig_pos"].mean(axis=(0, 1))
        batch["rec_pos"] = batch["rec_pos"] - center
        batch["lig_pos"] = batch["lig_pos"] - center

    @staticmethod
    def modify_coords(lig_pos, rot_update, tr_update):
        """DFMDock.py:246-252: rotate about the all-atom centroid (not the CA centroid of inference_base.modify_coords), then translate."""
        from .pdbio import axis_angle_to_matrix
        x = np.asarray(lig_pos, np.float32)
        cen = x.mean(axis=(0, 1))
        rot = axis_angle_to_matrix(np.asarray(rot_update, np.float64).reshape(3)).astype(np.float32)
        return ((x - cen) @ rot.T + cen + np.asarray(tr_update, np.float32).reshape(3)).astype(np.float32)


def Euler_Maruyama_sampler(model: Score_Model, batch, num_steps=40, device="cpu", batch_size=1, eps=1e-3,
                           use_clash_force=False, noise_annealing=False, tr_noise_scale=0.5, rot_noise_scale=0.5,
                           seed=None):
    """One trajectory, reference signature and return tuple (inference_base.py:390-468):
    (rec_pos, lig_pos [L,3,3], rot_update [1,3], tr_update [1,3], output dict).  The whole 40-step loop runs
    inside one dfm_sample call on the GPU."""
    import torch
    if batch_size != 1:
        raise ValueError("batch_size must be 1 (as in the reference); use sample_trajectories for batches")
    cx = model.complex_for(batch)
    model._calls += 1
    r = cx.sample(B=1, num_steps=num_steps, eps=eps, tr_noise_scale=tr_noise_scale, rot_noise_scale=rot_noise_scale,
                  noise_annealing=noise_annealing, use_clash_force=use_clash_force,
                  seed=(model.seed + model._calls) if seed is None else seed, **engine.precision_kwargs(model.precision))
    output = {"energy": torch.tensor(float(r["energy"][0])), "num_clashes": torch.tensor(int(r["num_clashes"][0])),
              "tr_score": torch.from_numpy(r["final_scores"][:, 0:3].copy()),
              "rot_score": torch.from_numpy(r["final_scores"][:, 3:6].copy())}
    rec_pos = batch["rec_pos"].clone() if hasattr(batch["rec_pos"], "clone") else torch.from_numpy(_np(batch["rec_pos"]))
    return (rec_pos, torch.from_numpy(r["lig_pos"][0]), torch.from_numpy(r["rot_update"]),
            torch.from_numpy(r["tr_update"]), output)


def sample_trajectories(model: Score_Model, batch, num_samples=120, num_steps=40, eps=1e-3, use_clash_force=False,
                        noise_annealing=False, tr_noise_scale=0.5, rot_noise_scale=0.5, seed=0, max_batch=256):
    """`num_samples` independent trajectories (the reference loops Euler_Maruyama_sampler sequentially,
    inference_base.py:644-657); returns numpy arrays sorted as drawn, plus the arg-min-energy index."""
    cx = model.complex_for(batch)
    outs = []
    done = 0
    while done < num_samples:
        b = min(max_batch, num_samples - done)
        outs.append(cx.sample(B=b, num_steps=num_steps, eps=eps, tr_noise_scale=tr_noise_scale,
                              rot_noise_scale=rot_noise_scale, noise_annealing=noise_annealing,
                              use_clash_force=use_clash_force, seed=seed + done, **engine.precision_kwargs(model.precision)))
        done += b
    res = {k: np.concatenate([o[k] for o in outs], 0) for k in ("lig_pos", "rot_update", "tr_update", "energy", "num_clashes")}
    res["best"] = int(np.argmin(res["energy"]))     # `if outputs["energy"] < min_energy` keeps the first minimum
    return res

"""Randomised size stress of the 16-bit engine against the fp32 engine on the same graphs: odd receptor / ligand sizes and batch
sizes on both sides of every launch-shape threshold (tile tasks, 64 x 128 GEMM tiles, wave-major tails, ligand-only last layer),
score evaluations with and without the energy head, a short sampler run; prints the worst relative deviations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}
def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    R, L = int(rng.integers(3, 400)), int(rng.integers(2, 300))
    B = int(rng.choice([1, 2, 3, 5, 8, 13, 33, 70]))
    if (R + L) * B > 40000:
        B = max(1, 40000 // (R + L))
    cx = make_complex(R, L, seed=100 + it)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    poses = np.stack([cx["lig_pos"] + rng.normal(0, 1.0, 3).astype(np.float32) for _ in range(B)])
    ref = gx.score(poses, 0.3, seed=it, energy=True, debug=True)
    for tag, kw in (("mfma16", dict(mfma16=True)), ("f16", dict(f16=True))):
        full = gx.score(poses, 0.3, edges=ref["edges"], energy=True, **kw)
        lean = gx.score(poses, 0.3, edges=ref["edges"], **kw)
        assert all((full[k] == lean[k]).all() for k in ("f", "tr_score", "rot_score")), (R, L, B, tag)
        for k in ("f", "tr_score", "rot_score", "energy"):
            d = rel(full[k], ref[k]) if k != "energy" else float(np.abs(full[k] - ref[k]).max())
            assert np.isfinite(d), (R, L, B, tag, k)
            if d > worst.get((tag, k), (0,))[0]:
                worst[(tag, k)] = (d, R, L, B)
    s1 = gx.sample(B=B, num_steps=3, seed=it, mfma16=True)
    s2 = gx.sample(B=1, num_steps=3, seed=it, mfma16=True)
    assert np.isfinite(s1["lig_pos"]).all() and (s1["lig_pos"][0] == s2["lig_pos"][0]).all(), (R, L, B, "batch-size invariance")
    gx.close()
for k, v in sorted(worst.items()):
    print(f"{k[0]:5s} {k[1]:9s} worst {v[0]:.2e} at {v[1]}+{v[2]} B={v[3]}")
print("stress ok")

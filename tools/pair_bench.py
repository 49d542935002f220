"""Deviation table and throughput of the second model family (EGNN_Net / DFMDock.forward) on the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import complex_for, load_golden, pair_hparams
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

hp = pair_hparams()
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0, hp), hp), hp)
cases = ["fwd2_syn_24_16", "fwd2_syn_64_48_p0", "fwd2_syn_64_48_p1", "fwd2_syn_64_48_p2", "fwd2_7CEI_p0", "fwd2_7CEI_p1", "fwd2_7CEI_p2"]
cache = {}
STEPS = int(os.environ.get("PAIR_STEPS", "40"))      # PAIR_ONLY_BENCH=1 PAIR_STEPS=3: the C3-shaped launches alone (counter passes)
for prec in (() if os.environ.get("PAIR_ONLY_BENCH") else ("fp32", "mfma16", "f16")):
    worst = np.zeros(6)
    for c in cases:
        g = load_golden(c + ".npz")
        key = c.split("_p")[0]
        if key not in cache:
            cx = complex_for(c)
            cache[key] = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        r = cache[key].score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, mfma16=prec == "mfma16", f16=prec == "f16", debug=True)
        v = np.array([rel(r["h_last"][0], g["h_last"]), rel(r["f"][0], g["f"]), rel(r["tr_score"][0], g["tr_score"][0]),
                      rel(r["rot_score"][0], g["rot_score"][0]), abs(float(r["energy"][0]) - float(g["energy"])) / max(abs(float(g["energy"])), 0.1),
                      abs(float(r["confidence"][0]) - float(g["confidence_logits"]))])
        worst = np.maximum(worst, v)
    print(f"{prec:5s} worst (h_last f tr rot energy conf):", " ".join(f"{x:.2e}" for x in worst))
if len(sys.argv) > 1:
    B = int(sys.argv[1])
    cx = make_complex(300, 300, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    for prec in (("fp32",) if os.environ.get("PAIR_FP32") else ("mfma16",)):      # PAIR_FP32=1: the fp32 engine of this family
        m16 = prec == "mfma16"
        gx.sample(B=B, num_steps=4 if m16 else 2, seed=1, mfma16=m16)
        t0 = time.perf_counter()
        gx.sample(B=B, num_steps=STEPS, seed=2, mfma16=m16)
        dt = time.perf_counter() - t0
        print(f"C3-shaped 300+300, B={B}, {STEPS} steps, {prec}: {dt*1e3:.0f} ms -> {B/dt:.1f} trajectories/s")

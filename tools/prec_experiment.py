"""Print bf16-path deviations from the reference's golden vectors (set DFM_GATHER_PREC=0..3 to compare variants)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import complex_for, load_golden
from dfmdock_amd import engine
from dfmdock_amd.weights import make_random_weights, pack_blob

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cases = ["fwd_syn_24_16", "fwd_syn_64_48_p0", "fwd_syn_64_48_p1", "fwd_syn_64_48_p2", "fwd_7CEI_p0", "fwd_7CEI_p1", "fwd_7CEI_p2", "fwd_7CEI_p3"]
cache = {}
F16 = len(sys.argv) > 1 and sys.argv[1] == "f16"
print("MFMA operands:", "fp16" if F16 else "mfma16")
worst = np.zeros(5)
for c in cases:
    g = load_golden(c + ".npz")
    key = c.split("_p")[0]
    if key not in cache:
        cx = complex_for(c)
        cache[key] = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = cache[key].score(g["lig_pos"], float(g["t"]), edges=g["edges"], energy=True, mfma16=not F16, f16=F16, debug=True)
    v = np.array([rel(r["h_last"][0], g["h_last"]), rel(r["f"][0], g["f"]), rel(r["tr_score"][0], g["tr_score"][0]),
                  rel(r["rot_score"][0], g["rot_score"][0]), abs(float(r["energy"][0]) - float(g["energy"])) / max(abs(float(g["energy"])), 0.1)])
    worst = np.maximum(worst, v)
    print(f"{c:20s} h_last {v[0]:.2e}  f {v[1]:.2e}  tr {v[2]:.2e}  rot {v[3]:.2e}  energy {v[4]:.2e}")
print("worst".ljust(20), " ".join(f"{x:.2e}" for x in worst))

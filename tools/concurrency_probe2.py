"""Where do two concurrently driven handles first deviate from their solo runs?  (diagnostic)"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
ca, cb = make_complex(223, 172, seed=5), make_complex(120, 90, seed=6)
A = engine.Complex(model, ca["rec_x"], ca["lig_x"], ca["rec_pos"], ca["lig_pos"])
Bc = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])

def conc(fa, fb):
    out = {}
    ta = threading.Thread(target=lambda: out.update(a=fa())); tb = threading.Thread(target=lambda: out.update(b=fb()))
    ta.start(); tb.start(); ta.join(); tb.join()
    return out["a"], out["b"]

kw = dict(mfma16=True, l0_table=False, trace=True, step_energy=False)
fa = lambda: A.sample(B=40, num_steps=6, seed=1, **kw)
fb = lambda: Bc.sample(B=40, num_steps=6, seed=2, **kw)
sa, sb = fa(), fb()
for rep in range(3):
    a, b = conc(fa, fb)
    for name, x, s in (("A", a, sa), ("B", b, sb)):
        d = np.abs(x["trace_scores"][:, :, :6] - s["trace_scores"][:, :, :6]).max(-1)      # [B, S+1]
        first = [int(np.argmax(d[t] > 0)) if (d[t] > 0).any() else -1 for t in range(d.shape[0])]
        print(rep, name, "trajectories differing:", int((d.max(1) > 0).sum()), "first differing evaluation per trajectory:", sorted(set(first)),
              "max |d score| at that evaluation:", float(d.max()), "rel:", float(d.max() / np.abs(s["trace_scores"][:, :, :6]).max()))
# single evaluations with debug taps, concurrently, many reps
def sc(g, c, seed, **k2):
    return g.score(np.repeat(c["lig_pos"][None], 40, 0), 0.5, seed=seed, energy=False, debug=True, **k2)
s1, s2 = sc(A, ca, 3, mfma16=True), sc(Bc, cb, 4, mfma16=True)
nd = {"edges": 0, "edge_codes": 0, "h_first": 0, "h_last": 0, "f": 0}
for rep in range(12):
    a, b = conc(lambda: sc(A, ca, 3, mfma16=True), lambda: sc(Bc, cb, 4, mfma16=True))
    for k in nd:
        nd[k] += int((a[k] != s1[k]).any()) + int((b[k] != s2[k]).any())
print("score mfma16 direct, 12 concurrent reps x 2 handles: runs with a differing tap:", nd)
a, b = conc(lambda: sc(A, ca, 3, mfma16=True), lambda: sc(Bc, cb, 4, mfma16=True))
for k in ("h_first", "h_last"):
    d = np.abs(a[k] - s1[k]); print(k, "A: differing elements", int((d > 0).sum()), "of", d.size, "max", float(d.max()), "trajectories", sorted(set(np.nonzero(d.reshape(40, -1).max(1))[0].tolist()))[:10])

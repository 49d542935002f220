"""Two complex handles driven from two host threads at once: do their results equal the solo results bit for bit?
(diagnostic for cross-handle interference; dfmdock_amd.h promises independence of handles)"""
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
KEYS = ("lig_pos", "energy", "tr_update", "rot_update")

def both(fa, fb, reps=4):
    sa, sb = fa(), fb()
    bad = [0, 0]
    for _ in range(reps):
        out = {}
        ta = threading.Thread(target=lambda: out.update(a=fa())); tb = threading.Thread(target=lambda: out.update(b=fb()))
        ta.start(); tb.start(); ta.join(); tb.join()
        bad[0] += any((out["a"][k] != sa[k]).any() for k in KEYS if k in sa)
        bad[1] += any((out["b"][k] != sb[k]).any() for k in KEYS if k in sb)
    return bad

for name, kw in (("mfma16 table", dict(mfma16=True)), ("mfma16 no table", dict(mfma16=True, l0_table=False)), ("fp32 table", dict()),
                 ("fp32 no table", dict(l0_table=False)), ("f16", dict(f16=True))):
    for Bn in (40, 8):
        r = both(lambda: A.sample(B=Bn, num_steps=10, seed=1, **kw), lambda: Bc.sample(B=Bn, num_steps=10, seed=2, **kw))
        print(f"sample {name:16s} B={Bn:3d}: runs differing from solo (of 4): A {r[0]} B {r[1]}", flush=True)
SK = ("f", "tr_score", "rot_score", "energy")
def sc(g, c, **kw):
    r = g.score(np.repeat(c["lig_pos"][None], 16, 0), 0.5, seed=3, energy=True, **kw)
    return {k: r[k] for k in SK}
KEYS = SK
for name, kw in (("mfma16", dict(mfma16=True)), ("mfma16 table", dict(mfma16=True, l0_table=True)), ("fp32", dict())):
    r = both(lambda: sc(A, ca, **kw), lambda: sc(Bc, cb, **kw))
    print(f"score  {name:16s}      : runs differing from solo (of 4): A {r[0]} B {r[1]}", flush=True)
# same handle twice in sequence, other handle idle: sanity
x, y = A.sample(B=40, num_steps=10, seed=1, mfma16=True), A.sample(B=40, num_steps=10, seed=1, mfma16=True)
print("solo repeat identical:", all((x[k] == y[k]).all() for k in ("lig_pos", "energy")))

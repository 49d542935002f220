"""Victim x aggressor matrix: handle A repeats one call while handle B loops over one kind of work in another thread."""
import os, sys, threading, time
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
pb = np.repeat(cb["lig_pos"][None], 40, 0)
victims = {
    "A.sample mfma16 table": lambda: A.sample(B=40, num_steps=4, seed=1, mfma16=True),
    "A.sample mfma16 direct": lambda: A.sample(B=40, num_steps=4, seed=1, mfma16=True, l0_table=False),
    "A.sample fp32 table": lambda: A.sample(B=8, num_steps=3, seed=1),
}
aggr = {
    "B.sample mfma16 table": lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True),
    "B.sample mfma16 direct": lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False),
    "B.score fp32 direct": lambda: Bc.score(pb[:8], 0.5, seed=4, energy=True),
    "B.score mfma16 direct": lambda: Bc.score(pb, 0.5, seed=4, energy=True, mfma16=True),
    "B.selfcheck": lambda: Bc.selfcheck(seed=1),
    "B.create+close": lambda: engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"]).close(),
}
n = int(os.environ.get("REPS", "16"))
for vn, v in victims.items():
    solo = v()
    for an, a in aggr.items():
        stop = [False]
        def loop():
            while not stop[0]:
                a()
        t = threading.Thread(target=loop); t.start()
        bad = sum(int(any((v()[k] != solo[k]).any() for k in ("lig_pos", "energy"))) for _ in range(n))
        stop[0] = True; t.join()
        print(f"{vn:24s} | {an:24s}: {bad:2d} of {n} calls differ from solo", flush=True)

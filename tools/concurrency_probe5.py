"""Detail of one perturbed evaluation: which edges get a different theta bin when another handle runs the message kernel."""
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
rng = np.random.default_rng(0)
poses = (ca["lig_pos"][None] + rng.standard_normal((40, 1, 1, 3)).astype(np.float32) * 3).astype(np.float32)
pb = np.repeat(cb["lig_pos"][None], 40, 0)
def scoreA(**kw):
    return A.score(poses, 0.5, seed=3, energy=False, debug=True, mfma16=True, **kw)
solo = scoreA()
stop = [False]
def loop():
    while not stop[0]:
        Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False)
t = threading.Thread(target=loop); t.start()
found = 0
for rep in range(300):
    r = scoreA(edges=solo["edges"])
    w = np.argwhere(r["edge_codes"] != solo["edge_codes"])
    if len(w):
        found += 1
        flat = (w[:, 0] * 395 + w[:, 1]) * 60 + w[:, 2]
        x = r["edge_codes"][tuple(w.T)] ^ solo["edge_codes"][tuple(w.T)]
        fields = {"dist": int(((x & 63) != 0).sum()), "omega": int((((x >> 6) & 31) != 0).sum()), "theta": int((((x >> 11) & 31) != 0).sum()),
                  "phi": int((((x >> 16) & 15) != 0).sum()), "relpos": int((((x >> 20) & 127) != 0).sum())}
        nodes = sorted(set((w[:, 0] * 395 + w[:, 1]).tolist()))
        runs = np.split(np.array(nodes), np.where(np.diff(nodes) != 1)[0] + 1)
        print(f"rep {rep}: {len(w)} codes differ, fields {fields}; {len(nodes)} nodes in {len(runs)} contiguous runs: "
              f"{[(int(q[0]), len(q)) for q in runs][:12]}; new theta bins {sorted(set(((r['edge_codes'][tuple(w.T)] >> 11) & 31).tolist()))[:12]} "
              f"old {sorted(set(((solo['edge_codes'][tuple(w.T)] >> 11) & 31).tolist()))[:12]}; h_first differs: {bool((r['h_first'] != solo['h_first']).any())}", flush=True)
        if found >= 4:
            break
stop[0] = True; t.join()
print("perturbed evaluations found:", found)

"""Handle A scores in a loop while handle B does one kind of work in another thread: which work perturbs A, and which tap first?"""
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
poses = np.repeat(ca["lig_pos"][None], 40, 0)
def scoreA(**kw):
    return A.score(poses, 0.5, seed=3, energy=False, debug=True, mfma16=True, **kw)
solo = scoreA()
solo_e = A.score(poses, 0.5, edges=solo["edges"], energy=False, debug=True, mfma16=True)
assert all((solo[k] == solo_e[k]).all() for k in ("edge_codes", "h_first", "h_last", "f"))
works = {
    "idle": lambda: time.sleep(0.002),
    "B.score mfma16": lambda: Bc.score(np.repeat(cb["lig_pos"][None], 40, 0), 0.5, seed=4, energy=False, mfma16=True),
    "B.score fp32": lambda: Bc.score(np.repeat(cb["lig_pos"][None], 8, 0), 0.5, seed=4, energy=False),
    "B.sample mfma16": lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True),
    "B.create": lambda: engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"]).close(),
    "B.selfcheck": lambda: Bc.selfcheck(seed=1),
}
for name, work in works.items():
    stop = [False]
    def loop():
        while not stop[0]:
            work()
    t = threading.Thread(target=loop); t.start()
    cnt = {"edges": 0, "edge_codes": 0, "h_first": 0, "h_last": 0, "f": 0}
    detail = None
    n = 30
    for rep in range(n):
        r = scoreA(edges=solo["edges"])      # injected edges: only geometry + network
        for k in cnt:
            cnt[k] += int((r[k] != solo[k]).any())
        if detail is None and (r["edge_codes"] != solo["edge_codes"]).any():
            w = np.argwhere(r["edge_codes"] != solo["edge_codes"])
            detail = (len(w), sorted(set(w[:, 0].tolist()))[:8], [hex(int(r["edge_codes"][tuple(x)]) ^ int(solo["edge_codes"][tuple(x)])) for x in w[:4]])
    stop[0] = True; t.join()
    print(f"{name:18s}: of {n} A.score calls, differing taps {cnt}; first code diff (count, trajectories, xor): {detail}", flush=True)

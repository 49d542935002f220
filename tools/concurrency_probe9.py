"""Build variant DFM_FEAT_PROBE: k_edge_feat loads N_i twice and bins theta twice; which of the two disagrees when another handle perturbs it?"""
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
rng = np.random.default_rng(0)
def poses(k):
    return (ca["lig_pos"][None] + np.random.default_rng(k).standard_normal((40, 1, 1, 3)).astype(np.float32) * 3).astype(np.float32)
solo = {k: A.score(poses(k), 0.5, seed=3, energy=False, debug=True, mfma16=True) for k in range(4)}
for k in solo:
    assert (solo[k]["edge_codes"] >> 30).max() == 0, "probe bits set in a solo run"
stop = [False]
def loop():
    while not stop[0]:
        Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False)
t = threading.Thread(target=loop); t.start()
found = 0
for rep in range(400):
    k = rep % 4      # the pose set CHANGES from call to call, like the steps of a sampler
    r = A.score(poses(k), 0.5, edges=solo[k]["edges"], energy=False, debug=True, mfma16=True)
    c, s = r["edge_codes"], solo[k]["edge_codes"]
    diff = (c & 0x3FFFFFFF) != (s & 0x3FFFFFFF)
    if diff.any() or (c >> 30).any():
        found += 1
        print(f"rep {rep}: {int(diff.sum())} codes differ from solo; among ALL codes: bit30 (two loads of N_i differ) {int(((c >> 30) & 1).sum())}, "
              f"bit31 (same N_i, different theta bin) {int((c >> 31).sum())}; among the differing codes: bit30 {int((((c >> 30) & 1) != 0)[diff].sum())} "
              f"bit31 {int(((c >> 31) != 0)[diff].sum())} neither {int((((c >> 30) == 0))[diff].sum())}", flush=True)
        if found >= 6:
            break
stop[0] = True; t.join()
print("perturbed evaluations:", found)

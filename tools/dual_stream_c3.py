"""C3 (300+300, 256 trajectories in flight, 40 steps): one handle with B = 256 against n handles of the same complex sampling
B = 256 / n each from n host threads (one non-blocking stream per handle): does overlapping the VALU-bound message launches of one
sub-batch with the HBM-bound node GEMMs / latency-bound small launches of another pay?"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
R = int(os.environ.get("R", "300")); L = int(os.environ.get("L", "300")); BT = int(os.environ.get("B", "256"))
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(R, L, seed=1)
def run(n, reps=3):
    hs = [engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"]) for _ in range(n)]
    b = BT // n
    for h in hs:
        h.sample(B=b, num_steps=2, seed=1, mfma16=True)
    out = [None] * n
    def work(k):
        for r in range(reps):
            out[k] = hs[k].sample(B=b, num_steps=40, seed=10 + r, mfma16=True)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    for h in hs: h.close()
    return b * n * reps / dt, out
base, o1 = run(1)
print(f"{R}+{L}, {BT} trajectories in flight: 1 handle x B={BT}: {base:.1f} traj/s")
for n in (2, 3, 4):
    v, o = run(n)
    same = bool((o[0]["lig_pos"] == o1[0]["lig_pos"][: BT // n]).all())
    print(f"  {n} handles x B={BT // n}: {v:.1f} traj/s ({100 * (v / base - 1):+.1f} %); first handle's trajectories bitwise equal to the single-handle run's: {same}")

"""C4-shaped end-to-end run (24 complexes x 40 trajectories, N from ~200 to ~700) through driver.run_set on one GPU:
loader semantics (test-time global rotation), sampling, per-sample metrics, CSV.  Synthetic complexes stand in for DB5
(its files do not travel to the GPU box)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import driver, engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
rng = np.random.default_rng(0)
sizes = [(int(a), int(b)) for a, b in zip(rng.integers(90, 420, 24), rng.integers(60, 300, 24))]
cxs = []
for k, (R, L) in enumerate(sizes):
    c = make_complex(R, L, seed=100 + k)
    c["id"] = f"SYN{k:02d}"
    cxs.append(c)
out = os.path.join(tempfile.mkdtemp(), "c4.csv")
t0 = time.perf_counter()
rows, ranked = driver.run_set(model, cxs, num_samples=40, num_steps=40, seed=0, precision="mfma16", out_csv=out)
dt = time.perf_counter() - t0
n = sum(1 for _ in open(out)) - 1
print(f"C4-shaped: 24 complexes (N = {min(a+b for a,b in sizes)}..{max(a+b for a,b in sizes)}), 40 trajectories each, 40 steps: "
      f"{dt:.2f} s wall incl. complex creation and metrics -> {len(rows)/dt:.1f} trajectories/s; CSV rows {n}")
print(open(out).read().splitlines()[0]); print(open(out).read().splitlines()[1])

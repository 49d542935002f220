import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
for B in (1, 8, 64):
    poses = np.repeat(cx["lig_pos"][None], B, 0)
    for kw, name in ((dict(mfma16=True, energy=False), "mfma16"), (dict(mfma16=True, energy=True), "mfma16+energy"), (dict(mfma16=True, energy=True, l0_table=True), "mfma16+energy+table"), (dict(energy=True), "fp32+energy")):
        gx.score(poses, 0.5, seed=1, **kw)
        n = 20 if B < 64 else 5
        t0 = time.perf_counter()
        for i in range(n): gx.score(poses, 0.5, seed=i, **kw)
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:3d} {name:22s} {dt*1e3:8.2f} ms per dfm_score call")

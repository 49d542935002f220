import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dfmdock_amd import driver, engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for (R, L) in ((120, 90), (223, 172), (400, 280)):
    c = make_complex(R, L, seed=5)
    t0 = time.perf_counter(); gx = engine.Complex(model, c["rec_x"], c["lig_x"], c["rec_pos"], c["lig_pos"]); t1 = time.perf_counter()
    r = gx.selfcheck(precision="mfma16", seed=0); t2 = time.perf_counter()
    gx.sample(B=40, num_steps=40, seed=1, mfma16=True); t3 = time.perf_counter()
    gx.sample(B=40, num_steps=40, seed=2, mfma16=True); t4 = time.perf_counter()
    r2 = gx.selfcheck(precision="mfma16", seed=0); t5 = time.perf_counter()
    print(f"{R}+{L}: create {1e3*(t1-t0):.1f} ms  selfcheck(first) {1e3*(t2-t1):.1f} ms  sample40 first {1e3*(t3-t2):.1f} ms  second {1e3*(t4-t3):.1f} ms  selfcheck(again) {1e3*(t5-t4):.1f} ms")

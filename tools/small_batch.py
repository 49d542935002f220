"""Small-batch throughput (C1-like regimes): 300+300 at B = 1, 2, 4, 8, 16, 32 (bf16 engine, 40 steps), best of 3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
for B in tuple(int(x) for x in os.environ.get("BS", "1,2,4,8,16,32,64").split(",")):
    gx.sample(B=B, num_steps=2, seed=1, mfma16=True)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        o = gx.sample(B=B, num_steps=40, seed=2 + rep, mfma16=True)
        best = min(best, time.perf_counter() - t0)
    print(f"300+300 B={B:3d}: {best*1e3:8.1f} ms  {B/best:7.1f} traj/s  lib={os.path.basename(os.environ.get('DFM_LIB', 'product'))}")
gx.close()

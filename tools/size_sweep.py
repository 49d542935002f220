"""Throughput / sanity sweep over batch sizes and complex sizes (bf16 engine, 40 steps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for (R, L, Bs) in [(300, 300, (1, 8, 64, 256, 512)), (100, 60, (64, 1024)), (1000, 1000, (32,)), (2000, 1500, (8,))]:
    cx = make_complex(R, L, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    for B in Bs:
        gx.sample(B=B, num_steps=2, seed=1, mfma16=True)
        t0 = time.perf_counter()
        o = gx.sample(B=B, num_steps=40, seed=2, mfma16=True)
        dt = time.perf_counter() - t0
        ok = np.isfinite(o["lig_pos"]).all() and np.isfinite(o["energy"]).all()
        print(f"{R}+{L} B={B:5d}: {dt*1e3:9.1f} ms  {B/dt:8.1f} traj/s  finite={ok}  E[min,mean]={o['energy'].min():.3f},{o['energy'].mean():.3f}")
    gx.close()

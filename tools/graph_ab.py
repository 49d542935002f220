"""Replayed step graph vs plain launches at small batches (16-bit engine, 300+300, 40 steps, best of 5).   python tools/graph_ab.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
shapes = {"300+300": (300, 300), "1AVX 223+172": (223, 172), "7CEI 87+127": (87, 127)}
Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 40, 120]
for name, (R, L) in shapes.items():
    cx = make_complex(R, L, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    for B in Bs if name == "300+300" else [b for b in Bs if b in (1, 8, 40, 120)]:
        row = []
        for graph in (False, True):
            gx.sample(B=B, num_steps=3, seed=1, mfma16=True, graph=graph)
            best = 1e9
            for rep in range(5):
                t0 = time.perf_counter()
                o = gx.sample(B=B, num_steps=40, seed=2 + rep, mfma16=True, graph=graph)
                best = min(best, time.perf_counter() - t0)
            row.append(best)
        print(f"{name:14s} B={B:4d}: plain {row[0] * 1e3:8.2f} ms {B / row[0]:8.1f} traj/s | graph {row[1] * 1e3:8.2f} ms {B / row[1]:8.1f} traj/s  ({100 * (row[0] / row[1] - 1):+.1f} %)", flush=True)
    gx.close()

"""Layer-0 table: of the edges the edge model still evaluates, how many are inter-chain and how many are intra-chain pairs whose feature
bins in the pose at hand differ from the table's (rounding at a bin boundary)?  Poses: what dfm_sample draws (randomize_pose) and what
it reaches after 10 / 39 steps, 300+300, 16-bit engine."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
R = L = 300
cx = make_complex(R, L, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
B = 64
tr = gx.sample(B=B, num_steps=40, seed=3, mfma16=True, trace=True)
for name, poses in (("randomize_pose", tr["init_pose"]), ("after step 10", tr["trace_pose"][:, 10]), ("after step 39", tr["trace_pose"][:, 39]),
                    ("stored pose x B", np.repeat(cx["lig_pos"][None], B, 0))):
    r = gx.score(poses, 0.5, seed=5, mfma16=True, energy=False, l0_table=True, profile=True, return_edges=True)
    p = gx.profile()
    e = r["edges"]
    same = (np.arange(R + L)[None, :, None] < R) == (e < R)
    inter = int((~same).sum())
    print(f"{name:18s}: edges {e.size}, edge model on {p['l0_miss_rows']} = {100.0 * p['l0_miss_rows'] / e.size:.2f} %: inter-chain {inter} ({100.0 * inter / e.size:.2f} %), "
          f"bin mismatches {p['l0_miss_rows'] - inter} ({100.0 * (p['l0_miss_rows'] - inter) / max(int(same.sum()), 1):.3f} % of the intra-chain edges)")

#!/usr/bin/env python3
"""The r05 cross-handle miscompute, reproduced on purpose (diagnostics only; the shipped build never deviates, tests/test_gpu_concurrency.py):
victim = A.sample on the 16-bit engine with layer 0 evaluated directly (k_edge_feat<0>), aggressor = another handle looping the same on
its own stream in a second host thread.  Needs DFM_TOKEN_LDS=0 and a library whose kernels_geom.hip was built WITH SLP vectorisation
(tools/asm_variant.py).  Prints how many of the victim's calls deviate from its solo result and which output deviates first.

    DFM_TOKEN_LDS=0 DFM_LIB=$PWD/tools/variants/NAME.so python tools/concurrency_repro.py [calls]
"""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
ca, cb = make_complex(223, 172, seed=5), make_complex(120, 90, seed=6)
A = engine.Complex(model, ca["rec_x"], ca["lig_x"], ca["rec_pos"], ca["lig_pos"])
Bc = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])
rng = np.random.default_rng(0)
poses = (ca["lig_pos"][None] + rng.standard_normal((40, 1, 1, 3)).astype(np.float32) * 3).astype(np.float32)
# r05's most sensitive cell: the victim SAMPLES (every evaluation of every step re-runs k_edge_feat<0>); a deviating bin moves the pose
victim = lambda: A.sample(B=40, num_steps=6, seed=3, mfma16=True, l0_table=False)
KEYS = ("lig_pos", "energy", "rot_update", "tr_update")
solo = victim()
assert all(np.array_equal(solo[k], victim()[k]) for k in KEYS)
stop = [False]


def loop():
    while not stop[0]:
        Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False)


t = threading.Thread(target=loop); t.start()
bad, codes = 0, 0
for _ in range(calls):
    r = victim()
    d = [k for k in KEYS if not np.array_equal(solo[k], r[k])]
    if d:
        bad += 1
        codes += int((np.abs(solo["lig_pos"] - r["lig_pos"]).reshape(40, -1).max(1) > 0).sum())
stop[0] = True; t.join()
print(f"{os.path.basename(os.environ.get('DFM_LIB', 'product'))}: token LDS {os.environ.get('DFM_TOKEN_LDS', '1')}: {bad} of {calls} victim calls deviate from solo ({codes} trajectories moved in total)")

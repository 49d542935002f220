"""dfm_complex_selfcheck on the DB5 complexes whose REAL ESM-2 feature blocks are committed (tests/golden/esm_<id>.npz, cx_7CEI.npz):
fp16 headroom and 16-bit-vs-fp32 deviations on the features the reference's loader produces (src/datasets/ppi_dataset.py:249-265),
next to the seeded N(0,1) stand-in of the same backbone.  -> profiles/r05_selfcheck_db5.txt
r06: plus the other 20 complexes on their int8-quantised ESM blocks (tests/golden/make_golden_r06.py), seed-0 draw, 8 graphs each."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Q8_ESM_IDS, REAL_ESM_IDS, db5_complex, real_db5_complex
from dfmdock_amd import engine
from dfmdock_amd.weights import WEIGHT_DRAWS, make_weight_draw, pack_blob
engine.set_device(0)
for draw in ("s0", "x3"):
    model = engine.Model(pack_blob(make_weight_draw(draw)))
    print(f"weight draw {draw} (seed {WEIGHT_DRAWS[draw][0]}, MLP scale x{WEIGHT_DRAWS[draw][1]}):")
    for cid in REAL_ESM_IDS:
        for kind, cx in (("real ESM-2", real_db5_complex(cid)), ("seeded N(0,1)", db5_complex(cid))):
            gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
            r = gx.selfcheck(n_eval=4, seed=3, precision="mfma16")
            x = np.concatenate([cx["rec_x"][:, :1280], cx["lig_x"][:, :1280]])
            print(f"  {engine.format_selfcheck(r, cid + ' ' + kind)}  [features: max |x| {np.abs(x).max():.2f}, rms {np.sqrt((x ** 2).mean()):.3f}]")
            gx.close()
    model.close()

model = engine.Model(pack_blob(make_weight_draw("s0")))
print("all 24 DB5 test complexes on ESM-2 features (4 fp16 blocks, 20 int8-quantised), weight draw s0, 8 engine-drawn graphs each, native pose:")
fails = 0
for cid in REAL_ESM_IDS + Q8_ESM_IDS:
    cx = real_db5_complex(cid)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.selfcheck(n_eval=8, seed=3, precision="mfma16")
    fails += 0 if r["ok"] else 1
    print("  " + engine.format_selfcheck(r, cid))
    gx.close()
print(f"self-check failed on {fails} of 24")

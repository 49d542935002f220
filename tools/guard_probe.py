import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for (R, L, B) in ((120, 90, 40), (223, 172, 40), (300, 300, 16), (24, 16, 3)):
    c = make_complex(R, L, seed=6)
    g = engine.Complex(model, c["rec_x"], c["lig_x"], c["rec_pos"], c["lig_pos"])
    g.sample(B=B, num_steps=3, seed=2, mfma16=True)
    g.sample(B=B, num_steps=3, seed=2, mfma16=True, l0_table=False)
    g.sample(B=B, num_steps=3, seed=2)
    g.score(np.repeat(c["lig_pos"][None], B, 0), 0.5, seed=1, energy=True, mfma16=True, debug=True)
    g.selfcheck(seed=1)
    g.close()
    print("done", R, L, B, flush=True)
model.close()

"""CA-RMSD between the engine's poses and the reference's over injected full rollouts (every random draw and every edge list
of the reference run replayed): per precision mode, after 5 steps and the maximum over the trajectory."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import complex_for, load_golden
from dfmdock_amd import engine
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for case, steps in (("rollout_syn_24_16", 40), ("rollout_syn_64_48", 40), ("rollout_7CEI", 6)):
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    inj = dict(R0=g["R0"].astype(np.float32), tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    for prec in ("fp32", "f16", "mfma16"):
        r = gx.sample(B=1, num_steps=steps, inject=inj, trace=True, mfma16=prec == "mfma16", f16=prec == "f16")
        ca, ref = r["trace_pose"][0][:, :, 1, :], g["poses"][:, :, 1, :]
        rmsd = np.sqrt(((ca - ref) ** 2).sum(-1).mean(-1))
        print(f"{case:20s} {prec:5s} steps {steps:2d}: CA-RMSD vs reference after 5 steps {rmsd[:5].max():.2e} A, max {rmsd.max():.2e} A, final {rmsd[-1]:.2e} A")

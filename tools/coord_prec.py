"""Experiment: coordinate-MLP operand type in the bf16 engine (DFM_COORD_F16=0|1), deviation of f / its mean from the reference goldens."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import complex_for, load_golden
from dfmdock_amd import engine
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
m = engine.Model(pack_blob(make_random_weights(0)))
for case in ["fwd_syn_24_16", "fwd_7CEI_p1", "fwd_db5_1AVX", "fwd_db5_4POU", "fwd_c3_300_300"]:
    g = load_golden(case + ".npz")
    cx = complex_for(case)
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    for prec in ("fp32", "mfma16", "f16"):
        r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, mfma16=prec == "mfma16", f16=prec == "f16")
        df = r["f"][0].astype(np.float64) - g["f"]
        print(f"COORD_F16={os.environ.get('DFM_COORD_F16', '0')} {case:16s} {prec:5s} max|f| {np.abs(g['f']).max():.3e} |mean f| {np.abs(g['f'].mean(0)).max():.3e} "
              f"max|df| {np.abs(df).max():.2e} |mean df| {np.abs(df.mean(0)).max():.2e} rms df {np.sqrt((df**2).mean()):.2e} "
              f"tr_err {np.abs(r['tr_score'][0] - g['tr_score'].reshape(3)).max() / np.abs(g['tr_score']).max():.2e}")
    gx.close()

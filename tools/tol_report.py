"""Measured deviations behind the parity gates (run on the GPU box): 16-bit engines vs the reference goldens (both families),
fp32 engine vs the oracle on odd sizes.  Prints one line per case; the gates in tests/ are SURVEY 8(d)'s."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from conftest import complex_for, load_golden, pair_hparams
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import HParams, make_random_weights, pack_blob
from oracle import oracle as ora


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


engine.set_device(0)
blob = pack_blob(make_random_weights(0))
hp1 = pair_hparams()
blob1 = pack_blob(make_random_weights(0, hp1), hp1)
for fam, bl, hp, cases in ((0, blob, HParams(), ["fwd_syn_9_7", "fwd_syn_24_16", "fwd_syn_64_48_p0", "fwd_syn_64_48_p1", "fwd_syn_64_48_p2",
                                                 "fwd_7CEI_p0", "fwd_7CEI_p1", "fwd_7CEI_p2", "fwd_7CEI_p3", "fwd_c3_300_300", "fwd_db5_1AVX",
                                                 "fwd_db5_4POU", "fwd_c5_1000_1000"]),
                           (1, blob1, hp1, ["fwd2_syn_9_7", "fwd2_syn_24_16", "fwd2_syn_64_48_p0", "fwd2_syn_64_48_p1", "fwd2_syn_64_48_p2",
                                            "fwd2_7CEI_p0", "fwd2_7CEI_p1", "fwd2_7CEI_p2"])):
    m = engine.Model(bl, hp)
    for case in cases:
        g = load_golden(case + ".npz")
        cx = complex_for(case)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        for prec in ("fp32", "mfma16", "f16"):
            r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, mfma16=prec == "mfma16", f16=prec == "f16")
            print(f"fam{fam} {case:20s} {prec:5s} f {rel(r['f'][0], g['f']):.2e} tr {rel(r['tr_score'][0], g['tr_score'].reshape(3)):.2e} "
                  f"rot {rel(r['rot_score'][0], g['rot_score'].reshape(3)):.2e} E {abs(float(r['energy'][0]) - float(g['energy'])) / max(abs(float(g['energy'])), 0.1):.2e}"
                  f"  |rot| {np.abs(g['rot_score']).max():.3e} |tr| {np.abs(g['tr_score']).max():.3e}")
        gx.close()
    m.close()
import draw_report
draw_report.main()
print()
for family in (0, 1):
    hp = hp1 if family else HParams()
    bl = blob1 if family else blob
    m = engine.Model(bl, hp)
    for (R, L) in [(1, 1), (70, 1), (2, 65), (19, 1), (33, 27), (64, 31), (65, 130), (129, 67)]:
        cx = make_complex(R, L, seed=11 + R + L)
        gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
        poses = np.stack([cx["lig_pos"], cx["lig_pos"] + np.float32(1.5)])
        t = np.array([0.7, 0.05], np.float32)
        r = gx.score(poses, t, seed=3, energy=True, debug=True)
        o = ora.Oracle(bl, cx, hp)
        for b in range(2):
            ref = o.score(poses[b], float(t[b]), edges=r["edges"][b])
            print(f"odd fam{family} {R}+{L} b{b}: f {rel(r['f'][b], ref['f']):.2e} tr {rel(r['tr_score'][b], ref['tr_score'].reshape(3)):.2e} "
                  f"rot {rel(r['rot_score'][b], ref['rot_score'].reshape(3)):.2e} (|rot| {np.abs(ref['rot_score']).max():.2e}) "
                  f"E {abs(float(r['energy'][b]) - float(ref['energy'])):.2e} (|E| {abs(float(ref['energy'])):.2e})")
        gx.close()
    m.close()

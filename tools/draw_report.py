"""Per weight draw (dfmdock_amd/weights.py: WEIGHT_DRAWS): worst deviation of each engine from the REFERENCE's outputs over the
draw's forward cases and the 40-step replayed rollout (run on the GPU box).

    python tools/draw_report.py [--fam 0,1] [--draws s0,s1,s2,x3] [--prec fp32,bf16,f16,bf16ops] [--cases]

`--cases` prints every case, not only the worst.  Precision-changing environment switches that are set are echoed.
"""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from conftest import DRAWS, DRAW_CASES, complex_for, draw_blob, draw_golden, draw_hparams, load_golden, pair_hparams


# engine selections: "mfma16" = the 16-bit MFMA engine as shipped (DFM_F_MFMA16: fp16 operands), "bf16ops" = + DFM_F_BF16_OPS
# "mfma16+tab" = ... with layer 0 through the per-complex message table (DFM_F_L0_TABLE; what dfm_sample runs by default)
KW = {"fp32": {}, "mfma16": dict(mfma16=True), "f16": dict(f16=True), "bf16ops": dict(mfma16=True, bf16_ops=True),
      "mfma16+tab": dict(mfma16=True, l0_table=True)}


def sample_kw(prec):      # Complex.sample takes the table by default: the plain "mfma16" row switches it off
    kw = dict(KW[prec])
    if prec == "mfma16":
        kw["l0_table"] = False
    return kw


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main(fams=(0, 1), draws=("s0",) + tuple(DRAWS), precs=("fp32", "mfma16", "mfma16+tab", "f16"), per_case=False):
    from dfmdock_amd import engine
    from dfmdock_amd.weights import make_random_weights, pack_blob
    engine.set_device(0)
    sw = {k: v for k, v in os.environ.items() if k.startswith("DFM_")}
    print("\n# per weight draw: worst relative deviation over the draw's forward cases (f / tr_score / rot_score / energy) and the worst")
    print("# ligand CA-RMSD of the 40-step replayed rollout; gates fp32 1e-4, bf16 1e-2 (3e-2 energy), f16 3e-3 (5e-3 energy), 0.5 A")
    print("# yardstick (SURVEY 7): the oracle's own bf16-autocast deviations on 7CEI are 5e-4 (tr) / 3e-3 (rot) / 1e-2 (f) / 1.8e-2 (E)")
    print(f"# environment switches: {sw if sw else 'none'}")
    print(f"# engine: {engine.config_string()}")
    for fam in fams:
        for draw in draws:
            hp = draw_hparams(fam)
            bl = draw_blob(fam, draw) if draw != "s0" else pack_blob(make_random_weights(0, hp), hp)
            m = engine.Model(bl, hp)
            worst = {p: np.zeros(4) for p in precs}
            wcase = {p: [""] * 4 for p in worst}
            for case in DRAW_CASES[fam]:
                if draw == "s0":
                    if fam == 1 and case.startswith("fwd_"):
                        continue
                    g = {k: np.asarray(v) for k, v in load_golden(case + ".npz").items()}
                else:
                    g = draw_golden(fam, draw, case)
                cx = complex_for(case)
                gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
                for prec in worst:
                    r = gx.score(g["lig_pos"], float(g["t"]), edges=g["edges"].astype(np.int32), energy=True, **KW[prec])
                    d = [rel(r["f"][0], g["f"]), rel(r["tr_score"][0], np.asarray(g["tr_score"]).reshape(3)),
                         rel(r["rot_score"][0], np.asarray(g["rot_score"]).reshape(3)),
                         abs(float(r["energy"][0]) - float(g["energy"])) / max(abs(float(g["energy"])), 0.1)]
                    if per_case:
                        print(f"  fam{fam} {draw} {case:18s} {prec:5s} f {d[0]:.2e} tr {d[1]:.2e} rot {d[2]:.2e} E {d[3]:.2e}")
                    for i in range(4):
                        if d[i] > worst[prec][i]:
                            worst[prec][i], wcase[prec][i] = d[i], case
                gx.close()
            roll = {}
            g = draw_golden(fam, draw, "rollout") if draw != "s0" else \
                load_golden(("rollout2_syn_24_16" if fam else "rollout_syn_24_16") + ".npz")
            cx = complex_for("syn_24_16")
            gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
            inj = dict(R0=g["R0"].astype(np.float32), tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
            for prec in worst:
                r = gx.sample(B=1, num_steps=40, inject=inj, trace=True, **sample_kw(prec))
                roll[prec] = float(np.sqrt(((r["trace_pose"][0][:, :, 1, :] - g["poses"][:, :, 1, :]) ** 2).sum(-1).mean(-1)).max())
            gx.close()
            for prec in worst:
                w = worst[prec]
                print(f"draw fam{fam} {draw} {prec:10s} f {w[0]:.2e} tr {w[1]:.2e} rot {w[2]:.2e} E {w[3]:.2e} rollout40 {roll[prec]:.2e} A"
                      f"   worst cases: {wcase[prec][0]} / {wcase[prec][1]} / {wcase[prec][2]} / {wcase[prec][3]}")
            m.close()
    print()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fam", default="0,1")
    ap.add_argument("--draws", default="s0," + ",".join(DRAWS))
    ap.add_argument("--prec", default="fp32,mfma16,mfma16+tab,f16")
    ap.add_argument("--cases", action="store_true")
    a = ap.parse_args()
    main([int(x) for x in a.fam.split(",")], a.draws.split(","), a.prec.split(","), a.cases)

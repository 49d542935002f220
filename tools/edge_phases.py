"""Per-phase cycle counts of the per-edge message kernel (diagnostic build: bash tools/build_variant.sh libdfm_stamp -DDFM_EDGE_STAMP).

    DFM_LIB=$PWD/dfmdock_amd/libdfm_stamp.so python tools/edge_phases.py [B]
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
poses = np.repeat(cx["lig_pos"][None], B, 0)
for it in range(3):
    gx.score(poses, 0.5, seed=it, mfma16=True, energy=False, profile=True)
    p = gx.profile()
    ph = p["phase_cycles"]
    print(f"edge launch avg {p['edge_kernel_ms'] / p['edge_kernel_launches']:.3f} ms | cycles per tile and wave: prologue {ph[0]:.0f}  "
          f"chunks0-6 {ph[1]:.0f}  chunk7+bias {ph[2]:.0f}  epilogue {ph[3]:.0f}  sum {sum(ph):.0f}")
    sl = np.array(p["slot_cycles"])
    if sl.sum() > 0:
        tiles = B * 600 * 2 / 2048.0
        print("   chunk-3 slot cycles (wave 0, per tile): " + " ".join(f"{x / tiles:.0f}" for x in sl[1:]))

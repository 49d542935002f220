#!/usr/bin/env python3
"""Phase relationship of the waves of one workgroup of the message kernel (diagnostic build -DDFM_EDGE_TRACE):

    bash tools/build_edge_variant.sh trace -DDFM_EDGE_TRACE
    DFM_LIB=$PWD/tools/variants/trace.so DFM_EDGE_TRACE_FILE=/tmp/tr.bin python tools/edge_trace.py

Per wave of workgroup 0: its SIMD (HW_ID bits 5:4), the cycle at which each of its first 64 tiles started and entered its epilogue.
For the two waves of a SIMD the script prints, tile by tile, where the partner was when this wave's epilogue began: the fraction of
the partner's current tile already done (0 = partner starts a tile, 0.73 = partner enters its own epilogue on the shipped kernel).
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
path = os.environ["DFM_EDGE_TRACE_FILE"]
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
poses = np.repeat(cx["lig_pos"][None], B, 0)
for it in range(int(os.environ.get("REPS", "2"))):
    gx.score(poses, 0.5, seed=it, mfma16=True, energy=False, profile=True)
    gx.profile()
raw = np.fromfile(path, dtype=np.uint64)
rt = raw[:16].reshape(8, 2).astype(np.int64)      # s_memrealtime (100 MHz) of every wave at the start of its tiles 0 and 63
tr = raw[48:].reshape(8, 130)
simd = [(int(t[0]) >> 4) & 3 for t in tr]
t0 = tr[:, 1::2][:, :64].astype(np.int64); te = tr[:, 2::2][:, :64].astype(np.int64)
base = t0[t0 > 0].min()
print("wave simd  first tile start   tile period (median)   epilogue share of a tile")
for w in range(8):
    per = np.diff(t0[w]); per = per[per > 0]
    ep = (t0[w][1:] - te[w][:-1]); ep = ep[(ep > 0) & (ep < 10 ** 6)]
    print(f"  {w}    {simd[w]}   {int(t0[w][0] - base):8d}          {np.median(per):8.0f}             {np.median(ep) / np.median(per):.3f}")
for s in range(4):
    ws = [w for w in range(8) if simd[w] == s]
    if len(ws) != 2:
        print(f"SIMD {s}: waves {ws}"); continue
    a, b = ws
    ph = []
    for k in range(4, 60):
        t = te[a][k]                               # wave a enters its epilogue
        j = np.searchsorted(t0[b], t, side="right") - 1
        if 0 <= j < 63 and t0[b][j + 1] > t0[b][j]:
            ph.append((t - t0[b][j]) / (t0[b][j + 1] - t0[b][j]))
    ph = np.array(ph)
    print(f"SIMD {s}: waves {a},{b}: partner's tile fraction when wave {a} enters its epilogue: " + " ".join(f"{x:.2f}" for x in ph[:24]) + f"  | mean {ph.mean():.2f} std {ph.std():.2f}")

# the shader clock this launch ran at: s_memtime (shader cycles) against s_memrealtime (100 MHz) between tile 0 and tile 63 of every wave
for w in range(8):
    dc, dr = int(t0[w][63] - t0[w][0]), int(rt[w][1] - rt[w][0])
    if dr > 0:
        print(f"wave {w}: tiles 0..63 took {dc} shader cycles = {dr / 100:.1f} us -> {100.0 * dc / dr:7.1f} MHz")

per_old = np.median([np.median(np.diff(t0[w])[np.diff(t0[w]) > 0]) for w in range(4)])
per_young = np.median([np.median(np.diff(t0[w])[np.diff(t0[w]) > 0]) for w in range(4, 8)])
mhz = np.median([100.0 * (t0[w][63] - t0[w][0]) / max(int(rt[w][1] - rt[w][0]), 1) for w in range(8)])
pair = 1.0 / (1.0 / per_old + 1.0 / per_young)
print(f"SUMMARY {os.path.basename(os.environ.get('DFM_LIB', 'product'))}: tile period older {per_old:.0f} younger {per_young:.0f} cycles -> {pair:.0f} cycles per tile and SIMD; "
      f"shader clock {mhz:.0f} MHz -> {pair / mhz:.3f} us per tile and SIMD")

#!/usr/bin/env python3
"""fp16 against bf16 MFMA operands in the message kernel: time AND the clock the kernel holds (the kernel's own stamps, dfm_profile).
bf16 operands (DFM_F_BF16_OPS) are outside the parity gates (1.5e-2 on the 3x-scaled draw, r03) - this is an energy measurement, not a
shipping option.   python tools/bf16_clock.py   -> profiles/r06_clock.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
for rnd in range(2):
    for name, kw in (("fp16 operands (shipped)", {}), ("bf16 operands (DFM_F_BF16_OPS)", {"bf16_ops": True})):
        gx.sample(B=256, num_steps=6, seed=1, mfma16=True, **kw)
        t0 = time.perf_counter()
        gx.sample(B=256, num_steps=40, seed=2, mfma16=True, profile=True, **kw)
        dt = time.perf_counter() - t0
        p = gx.profile()
        n_full = p["edge_kernel_launches"] - p["edge_lig_launches"]
        print(f"{name:32s} {256 / dt:6.1f} traj/s   message launch full {(p['edge_kernel_ms'] - p['edge_lig_ms']) / n_full:.4f} ms  "
              f"lig-only {p['edge_lig_ms'] / max(p['edge_lig_launches'], 1):.4f} ms   sclk {p['edge_sclk_mhz']:.0f} MHz")
gx.close()

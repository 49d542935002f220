"""Layer 0 behind the per-complex message table vs the direct evaluation: same-process A/B of dfm_sample (16-bit engine, 40 steps).

    python tools/l0_ab.py [R L B ...]      default: 300 300 256, 300 300 8, 300 300 1, 1000 1000 32
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

a = [int(x) for x in sys.argv[1:]]
cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(300, 300, 256), (300, 300, 64), (300, 300, 8), (300, 300, 1), (1000, 1000, 32)]
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for R, L, B in cases:
    cx = make_complex(R, L, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    res = {}
    for on in (False, True, False, True):
        gx.sample(B=B, num_steps=2, seed=1, mfma16=True, l0_table=on)
        t0 = time.perf_counter()
        o = gx.sample(B=B, num_steps=40, seed=2, mfma16=True, l0_table=on, profile=True)
        dt = time.perf_counter() - t0
        p = gx.profile()
        res[on] = o
        msg = f"{R}+{L} B={B:4d} table={'on ' if on else 'off'}: {dt * 1e3:8.1f} ms {B / dt:8.1f} traj/s | message launches {p['edge_kernel_launches']} avg {p['edge_kernel_ms'] / max(p['edge_kernel_launches'], 1):.3f} ms"
        if on:
            n = max(p["l0_evals"], 1)
            msg += (f" | layer 0: rows {p['l0_rows_ms'] / n:.3f} ms + gather {p['l0_gather_ms'] / n:.3f} ms per evaluation, "
                    f"edge model on {100.0 * p['l0_miss_rows'] / max(p['l0_edges'], 1):.2f} % of the edges, build {p['l0_build_ms']:.2f} ms")
        print(msg, flush=True)
    d = np.sqrt(((res[True]["lig_pos"][:, :, 1] - res[False]["lig_pos"][:, :, 1]) ** 2).sum(-1).mean(-1))
    print(f"   final poses, table vs direct (same seed, native graphs): CA-RMSD median {np.median(d):.3f} max {d.max():.3f} A; energies {np.abs(res[True]['energy'] - res[False]['energy']).max():.2e}")
    gx.close()

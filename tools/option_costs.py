"""What the sampler's options cost at C3 (300+300, B = 256, 40 steps, 16-bit engine): clash force, noise annealing, ODE, trace - wall time of
one batched call each, against the plain call (the options of inference_base.py:390-468)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gx.sample(B=B, num_steps=3, seed=1, mfma16=True)
for name, kw in (("plain", {}), ("use_clash_force", dict(use_clash_force=True)), ("noise_annealing", dict(noise_annealing=True)),
                 ("ode", dict(ode=True)), ("trace", dict(trace=True)), ("plain", {})):
    t0 = time.perf_counter()
    gx.sample(B=B, num_steps=40, seed=2, mfma16=True, **kw)
    dt = time.perf_counter() - t0
    print(f"{name:16s} {dt * 1e3:8.1f} ms  {B / dt:7.1f} traj/s")

#!/usr/bin/env python3
"""Shader clock and socket power while the engine runs (hwmon of the amdgpu card: freq1_input = sclk in Hz, power1_input in uW), sampled
every few ms from a side thread: what clock does the chip hold in the message kernel, and which of its pipes pays for it?

  python tools/clock_power.py                      C3-shaped dfm_sample in a loop for ~6 s with the product library
  DFM_LIB=tools/variants/ko1.so python tools/clock_power.py ko1     a knock-out build (tools/build_edge_variant.sh)

Prints one line: label, samples, sclk mean / p10 / p90 (MHz), power mean (W), trajectories/s of the loop.  -> profiles/r06_clock_power.txt"""
import glob
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmdock_amd import engine  # noqa: E402
from dfmdock_amd.synthetic import make_complex  # noqa: E402
from dfmdock_amd.weights import make_random_weights, pack_blob  # noqa: E402


def hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.exists(d + "/freq1_input") and os.path.exists(d + "/power1_input"):
            return d
    raise SystemExit("no amdgpu hwmon with freq1_input / power1_input")


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
    secs = float(os.environ.get("SECS", "6"))
    mode = os.environ.get("MODE", "sample")
    h = hwmon()
    engine.set_device(0)
    model = engine.Model(pack_blob(make_random_weights(0)))
    cx = make_complex(300, 300, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    run = (lambda: gx.sample(B=256, num_steps=10, seed=1, mfma16=True)) if mode == "sample" else \
          (lambda: gx.sample(B=256, num_steps=10, seed=1))
    run()
    clk, pw, stop = [], [], [False]

    def sampler():
        while not stop[0]:
            try:
                clk.append(int(open(h + "/freq1_input").read()) / 1e6)
                pw.append(int(open(h + "/power1_input").read()) / 1e6)
            except (OSError, ValueError):
                pass
            time.sleep(0.004)

    t = threading.Thread(target=sampler)
    t.start()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        run(); n += 1
    dt = time.time() - t0
    stop[0] = True
    t.join()
    c, p = np.array(clk[len(clk) // 10:]), np.array(pw[len(pw) // 10:])
    print(f"{label:18s} {mode:7s} samples {c.size:5d}  sclk mean {c.mean():6.0f} MHz (p10 {np.percentile(c, 10):5.0f}, p90 {np.percentile(c, 90):5.0f})  "
          f"power mean {p.mean():6.0f} W (max {p.max():5.0f})  loop {256 * n * 10 / 40 / dt:6.1f} traj/s-equivalent")
    gx.close()


if __name__ == "__main__":
    main()

"""C4-shaped end-to-end run (24 complexes x 40 trajectories; reference loop src/inference_mlsb.py:415-439 over
:188-262) through driver.run_set on one GPU: loader semantics (test-time global rotation), handle creation + self-check,
sampling, per-sample metrics, CSV.  Runs the SERIAL driver (overlap=False: the r04 driver), the pipelined one (create / check of
the next complex and the metrics of the previous one overlapped with sampling) and the pipelined one with two complexes
sampling concurrently; checks that the three CSV files are byte-identical and prints the per-phase breakdown.

    python tools/c4_run.py [--db5]      --db5: the 24 DB5 test backbones (tests/golden/db5_backbones.npz, seeded features + the
                                        committed real ESM blocks) instead of synthetic chains of the same size range
"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dfmdock_amd import driver, engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob

engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
if "--db5" in sys.argv:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from conftest import REAL_ESM_IDS, db5_ids, real_db5_complex
    cxs = [real_db5_complex(c) for c in db5_ids()]
    label = "the 24 DB5 test complexes on ESM-2 features (fp16 blocks: %s; int8-quantised blocks elsewhere, tests/golden/make_golden_r06.py)" % ", ".join(REAL_ESM_IDS)
else:
    rng = np.random.default_rng(0)
    sizes = [(int(a), int(b)) for a, b in zip(rng.integers(90, 420, 24), rng.integers(60, 300, 24))]
    cxs = []
    for k, (R, L) in enumerate(sizes):
        c = make_complex(R, L, seed=100 + k)
        c["id"] = f"SYN{k:02d}"
        cxs.append(c)
    label = "24 synthetic complexes"
Ns = [c["rec_x"].shape[0] + c["lig_x"].shape[0] for c in cxs]
tmp = tempfile.mkdtemp()
print(f"C4-shaped run: {label}, N = {min(Ns)}..{max(Ns)}, 40 trajectories each, 40 steps; wall clock includes handle creation, "
      f"self-check, metrics and the CSV")
driver.run_set(model, cxs[:2], num_samples=40, num_steps=40, seed=0, precision="mfma16", out_csv=os.path.join(tmp, "warm.csv"),
               selfcheck=True, on_selfcheck_fail="warn")      # warm-up: code objects, the block cache
csvs = {}
for name, kw in (("serial (r04 driver)", dict(overlap=False)), ("pipelined", dict(overlap=True, samplers=1)),
                 ("pipelined, 2 samplers", dict(overlap=True, samplers=2)), ("pipelined, 3 samplers", dict(overlap=True, samplers=3))):
    out = os.path.join(tmp, name.split()[0] + str(kw.get("samplers", 1)) + ".csv")
    tim = []
    t0 = time.perf_counter()
    rows, ranked = driver.run_set(model, cxs, num_samples=40, num_steps=40, seed=0, precision="mfma16", out_csv=out, timings_out=tim, **kw)
    dt = time.perf_counter() - t0
    csvs[name] = open(out, "rb").read()
    if name == "pipelined":
        per_complex = [(t["N"], t["prepare"], t["sample"], t["post"]) for t in tim]
    ph = {k: sum(t[k] for t in tim) / 1e3 for k in ("prepare", "sample", "post")}
    print(f"{name:24s}: {dt:6.3f} s wall -> {len(rows) / dt:6.1f} trajectories/s | summed over complexes: prepare (rotation, handle, "
          f"self-check) {ph['prepare']:.3f} s, sample {ph['sample']:.3f} s, post (metrics, records, close) {ph['post']:.3f} s "
          f"| not sampling: {100 * (1 - ph['sample'] / dt) if 'serial' in name else float('nan'):.0f} % of the wall")
base = csvs["serial (r04 driver)"]
print("CSV files byte-identical to the serial driver's:", all(v == base for v in csvs.values()), f"({len(base.splitlines()) - 1} rows)")
print(base.decode().splitlines()[0]); print(base.decode().splitlines()[1])
a = np.array(sorted(per_complex))
fit = np.polyfit(a[:, 0], a[:, 2], 1)
print(f"per complex (pipelined driver), ms: sample = {fit[1]:.1f} + {fit[0]:.4f} * N (least squares over {len(a)} complexes); "
      f"prepare mean {a[:, 1].mean():.1f}, post mean {a[:, 3].mean():.1f}")
print("  N: " + " ".join(f"{int(x):5d}" for x in a[:, 0]))
print("  sample ms: " + " ".join(f"{x:5.0f}" for x in a[:, 2]))

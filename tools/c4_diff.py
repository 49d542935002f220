"""Which numbers of a set run change between drivers / allocator settings (diagnostic for cross-handle interference and reads of
uninitialised memory): DFM_ALLOC_POISON / DFM_ALLOC_CACHE in the environment, MODE=serial|pipelined, writes rows to OUT."""
import os, sys, csv, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dfmdock_amd import driver, engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
rng = np.random.default_rng(0)
sizes = [(int(a), int(b)) for a, b in zip(rng.integers(90, 420, 12), rng.integers(60, 300, 12))]
cxs = []
for k, (R, L) in enumerate(sizes):
    c = make_complex(R, L, seed=100 + k); c["id"] = f"SYN{k:02d}"; cxs.append(c)
mode, out = os.environ.get("MODE", "serial"), os.environ["OUT"]
chk = []
kw = dict(overlap=False) if mode == "serial" else dict(overlap=True, samplers=int(os.environ.get("SAMPLERS", "1")))
rows, _ = driver.run_set(model, cxs, num_samples=40, num_steps=40, seed=0, log=lambda m: None, checks_out=chk, **kw)
json.dump({"rows": [[r["id"], r["index"], r["energy"], r["c_rmsd"]] for r in sorted(rows, key=lambda r: (r["id"], int(r["index"])))],
           "dev_f": {c["id"]: c["selfcheck"]["dev_f"] for c in chk}}, open(out, "w"))
print(mode, os.environ.get("DFM_ALLOC_POISON"), os.environ.get("DFM_ALLOC_CACHE"), "finite:", all(np.isfinite(r[2]) and np.isfinite(r[3]) for r in json.load(open(out))["rows"]))

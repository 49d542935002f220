"""Synthetic victims (tools/ubench/libvictim.so) next to the REAL engine as aggressor: which property of the victim matters?"""
import ctypes as C, os, sys, threading
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
V = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libvictim.so"))
V.victim_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
assert V.victim_init() == 0
model = engine.Model(pack_blob(make_random_weights(0)))
cb = make_complex(120, 90, seed=6)
Bc = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])
names = {0: "computed boundaries", 1: "__constant__ boundaries (s_load)", 2: "__constant__ boundaries, no dihedral"}
for aggr_name, aggr in (("idle", None), ("engine: B.sample mfma16 direct", lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False)),
                        ("engine: B.sample fp32", lambda: Bc.sample(B=8, num_steps=3, seed=2))):
    stop = [False]
    def loop():
        while not stop[0]:
            aggr()
    t = threading.Thread(target=loop) if aggr else None
    if t: t.start()
    for v in (0, 1, 2):
        for lds in (0, 64):
            w = C.c_longlong(0)
            bad = V.victim_run(v, 24, lds, C.byref(w))
            print(f"{aggr_name:32s} | victim {names[v]:40s} LDS {lds:2d} B: {bad:2d} of 24 launches differ from solo (worst {w.value} of 948000 codes)", flush=True)
    stop[0] = True
    if t: t.join()

#!/usr/bin/env python3
"""Does the ONE instruction form the bisection of k_edge_feat<0> ends at - an in-place packed fp32 multiply with crossed halves - go
wrong on its own next to the engine's message kernel?  tools/ubench/libpkmul.so (pkmul_victim.hip) on the main thread, the real engine
sampling on another handle in a second thread (or nothing: "idle").  Prints launches / elements that differ from two plain multiplies."""
import ctypes as C, os, sys, threading
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
V = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libpkmul.so"))
V.pk_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
assert V.pk_init() == 0
model = engine.Model(pack_blob(make_random_weights(0)))
cb = make_complex(120, 90, seed=6)
Bc = engine.Complex(model, cb["rec_x"], cb["lig_x"], cb["rec_pos"], cb["lig_pos"])
names = {0: "in place, crossed halves (the bisected instruction)", 1: "crossed halves, separate destination", 2: "in place, straight halves",
         3: "in place, crossed, operand roles swapped", 4: "op_sel:[0,1]  (lo from S1.hi)", 5: "op_sel_hi:[1,0]  (hi from S1.lo: the library's form)",
         6: "op_sel:[1,0]  (lo from S0.hi)", 7: "v_pk_add_f32 crossed S1", 8: "v_pk_fma_f32 crossed S1", 9: "both sources crossed"}
pb = np.repeat(cb["lig_pos"][None], 40, 0)
for aggr_name, aggr in (("idle", None), ("engine: B.sample mfma16 direct", lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=False)),
                        ("engine: B.sample mfma16 table", lambda: Bc.sample(B=40, num_steps=4, seed=2, mfma16=True, l0_table=True)),
                        ("engine: B.sample fp32", lambda: Bc.sample(B=8, num_steps=3, seed=2))):
    stop = [False]

    def loop():
        while not stop[0]:
            aggr()

    t = threading.Thread(target=loop) if aggr else None
    if t:
        t.start()
    for mode in range(10):
        for lds in ((0, 64) if mode < 2 else (0,)):
            w = C.c_longlong(0)
            bad = V.pk_run(mode, 48, lds, 64, C.byref(w))
            print(f"{aggr_name:32s} | {names[mode]:52s} LDS {lds:2d} B: {bad:2d} of 48 launches wrong ({w.value} of {48 * 948000 * 64} element-rounds)", flush=True)
    stop[0] = True
    if t:
        t.join()

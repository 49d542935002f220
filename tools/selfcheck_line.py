import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import complex_for
from dfmdock_amd import engine
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
m = engine.Model(pack_blob(make_random_weights(0)))
cx = complex_for("fwd_7CEI_p0")
gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
print(engine.format_selfcheck(gx.selfcheck(n_eval=4, seed=0), "7CEI"))

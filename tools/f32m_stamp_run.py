import numpy as np, sys
sys.path.insert(0, "/root/repo")
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
m = engine.Model(pack_blob(make_random_weights(0)))
cx = make_complex(300, 300, seed=1)
gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
poses = np.repeat(cx["lig_pos"][None], 256, 0)
gx.score(poses, 0.5, seed=1, energy=False)

import sys, numpy as np
sys.path.insert(0, ".")
from dfmdock_amd import engine
from dfmdock_amd.synthetic import make_complex
from dfmdock_amd.weights import make_random_weights, pack_blob
engine.set_device(0)
model = engine.Model(pack_blob(make_random_weights(0)))
for (R, L) in [(300, 300), (1000, 1000)]:
    cx = make_complex(R, L, seed=1)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    N = R + L
    def frac(pose, tag):
        r = gx.score(pose[None], 0.5, seed=3, mfma16=True, debug=True)
        e = r["edges"][0]                      # [N, K]
        src = np.arange(N)[:, None] < R
        inter = (e < R) != src
        per_node = inter.sum(1)
        print(f"{R}+{L} {tag}: inter-chain edges {inter.mean()*100:.1f} %  nodes with >=1: {(per_node>0).mean()*100:.0f} %  mean per node {per_node.mean():.1f} max {per_node.max()}  tiles32 per node if compacted per node: {np.ceil(per_node/32).mean():.2f}")
    frac(cx["lig_pos"], "native pose")
    s = gx.sample(B=4, num_steps=40, seed=1, mfma16=True, trace=True)
    for k in (0, 10, 20, 39):
        frac(s["trace_pose"][0][k], f"trajectory 0 after step {k}")
    frac(s["init_pose"][0], "random start")
    gx.close()

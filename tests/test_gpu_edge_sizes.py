"""Odd complex sizes through both model families (fp32 engine vs the CPU oracle on the engine's own graph): single-residue
chains, N < knn, N < knn + n_sample, receptor tile boundaries of the pair kernel (R = 64, 65), non-multiples of every tile."""
import numpy as np
import pytest

from conftest import pair_hparams

pytestmark = pytest.mark.gpu

SIZES = [(1, 1), (70, 1), (2, 65), (19, 1), (33, 27), (64, 31), (65, 130), (129, 67)]


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("family", [0, 1])
@pytest.mark.parametrize("R,L", SIZES)
def test_odd_sizes_fp32_vs_oracle(R, L, family, blob, blob_pair):
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import HParams
    from oracle import oracle as ora
    hp = pair_hparams() if family else HParams()
    bl = blob_pair if family else blob
    cx = make_complex(R, L, seed=11 + R + L)
    engine.set_device(0)
    m = engine.Model(bl, hp)
    gx = engine.Complex(m, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    poses = np.stack([cx["lig_pos"], cx["lig_pos"] + np.float32(1.5)])
    t = np.array([0.7, 0.05], np.float32)
    r = gx.score(poses, t, seed=3, energy=True, debug=True)
    o = ora.Oracle(bl, cx, hp)
    for b in range(2):
        ref = o.score(poses[b], float(t[b]), edges=r["edges"][b])
        assert int(r["num_clashes"][b]) == ref["num_clashes"]
        assert rel_inf(r["f"][b], ref["f"]) < 2e-4
        assert rel_inf(r["tr_score"][b], ref["tr_score"].reshape(3)) < 2e-4
        assert rel_inf(r["rot_score"][b], ref["rot_score"].reshape(3)) < 2e-4 or np.abs(ref["rot_score"]).max() < 1e-6
        assert abs(float(r["energy"][b]) - float(ref["energy"])) < 2e-4 * max(1.0, abs(float(ref["energy"])))
    # the 16-bit engine and the sampler run on these shapes too
    s = gx.sample(B=3, num_steps=3, seed=5, bf16=True)
    assert np.isfinite(s["lig_pos"]).all() and np.isfinite(s["energy"]).all()
    gx.close(); m.close()

"""Host-side logic on CPU: metrics (a-16) vs the reference's golden values, checkpoint reader,
work sharding and the world_size-2 record gather over gloo."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_compute_metrics_matches_reference():
    from dfmdock_amd.metrics import compute_metrics
    g, cx = load_golden("metrics_7CEI.npz"), load_golden("cx_7CEI.npz")
    keys = [str(k) for k in g["keys"]]
    nat = (cx["rec_pos"], cx["lig_pos"])
    m0 = compute_metrics(nat, nat)
    assert m0["fnat"] == 1.0 and abs(m0["DockQ"] - 1.0) < 1e-6 and m0["l_rmsd"] < 1e-4
    sh = cx["lig_pos"].copy()
    sh[..., 0] += 5.0
    m1 = compute_metrics((cx["rec_pos"], sh), nat)
    m2 = compute_metrics((cx["rec_pos"], g["noised_lig"]), nat)
    for got, ref in ((m1, g["shifted"]), (m2, g["noised"])):
        for k, v in zip(keys, ref):
            assert got[k] == pytest.approx(float(v), rel=2e-5, abs=2e-5), k
    assert m1["DockQ"] == pytest.approx(0.4223329224, abs=1e-5)    # SURVEY 8(c) known answer


def test_lightning_checkpoint_reader(tmp_path):
    import torch
    from dfmdock_amd.weights import load_lightning_checkpoint, make_random_weights, pack_blob
    w = make_random_weights(5)
    sd = {"net." + k: torch.from_numpy(v.copy()) for k, v in w.items()}
    path = tmp_path / "model_0.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"model": {"node_dim": 256, "depth": 6, "cut_off": 20.0}},
                "epoch": 3}, path)
    out, hp = load_lightning_checkpoint(str(path))
    np.testing.assert_array_equal(pack_blob(out), pack_blob(w))
    assert hp.depth == 6 and hp.cut_off == 20.0


def test_shard_and_assign():
    from dfmdock_amd.distributed import assign_work, shard_range
    for total, world in ((960, 8), (7, 3), (2, 4), (0, 2)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    costs = [395, 695, 343, 456, 575, 626, 561, 329, 197, 352, 320, 430, 377, 430, 382, 339, 628, 404, 240, 535, 373, 588,
             492, 214]     # N of the 24 DB5 test complexes
    parts = assign_work(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(24))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) / (sum(costs) / 8) < 1.15


GLOO_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from dfmdock_amd import distributed as D
    rank, _, world = D.dist_env()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lo, hi = D.shard_range(11, world, rank)            # 11 trajectories over 2 ranks: 6 + 5
    n = hi - lo
    rng = np.random.default_rng(100 + rank)
    res = dict(energy=rng.standard_normal(n).astype(np.float32), num_clashes=np.arange(n, dtype=np.int32),
               rot_update=rng.standard_normal((n, 3)).astype(np.float32), tr_update=rng.standard_normal((n, 3)).astype(np.float32))
    rec = D.make_records(7, np.arange(lo, hi), res)
    allrec = D.gather_records(rec)
    assert allrec.shape == (11, D.RECORD_WIDTH), allrec.shape
    assert (allrec[:, 1] == np.arange(11)).all()                      # rank order, every trajectory exactly once
    assert (allrec[lo:hi] == rec).all()
    ranked = D.rank_by_energy(allrec)[7]
    assert (np.diff(ranked[:, 2]) >= 0).all()
    np.save(os.path.join({out!r}, f"ranked_{{rank}}.npy"), ranked)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_record_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = 29500 + (os.getpid() % 400)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    a, b = np.load(tmp_path / "ranked_0.npy"), np.load(tmp_path / "ranked_1.npy")
    np.testing.assert_array_equal(a, b)         # every rank ends with the same energy-ranked table


def test_dfmdock_wrapper_helpers_match_reference():
    """DFMDock.modify_coords / move_to_lig_center (DFMDock.py:246-257) against values produced by the reference."""
    import numpy as np
    from conftest import load_golden
    from dfmdock_amd.score_model import DFMDock
    g = load_golden("pair_kats.npz")
    out = DFMDock.modify_coords(g["mc_x"], g["mc_rot"], g["mc_tr"])
    assert np.abs(out - g["mc_out"]).max() < 2e-5
    b = {"rec_pos": g["mc_x"] + 3.0, "lig_pos": g["mc_x"].copy()}
    DFMDock.move_to_lig_center(b)
    assert np.abs(b["rec_pos"] - g["centred_rec"]).max() < 1e-5 and np.abs(b["lig_pos"] - g["centred_lig"]).max() < 1e-5


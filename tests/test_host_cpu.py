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


def _omegaconf_shaped(tree):
    """Wrap a plain dict tree into instances of classes NAMED like omegaconf's (omegaconf itself is not installed): the
    pickle stream then references omegaconf.dictconfig.DictConfig / omegaconf.nodes.AnyNode exactly as a Lightning
    checkpoint written by the reference does (save_hyperparameters on Hydra configs, score_model_mlsb.py:30)."""
    import types
    mods = {}
    for name in ("omegaconf", "omegaconf.dictconfig", "omegaconf.nodes", "omegaconf.base"):
        mods[name] = types.ModuleType(name)

    class DictConfig:
        def __init__(self, content):
            self.__dict__["_content"] = content
            self.__dict__["_metadata"] = Metadata()

        def __getstate__(self):
            return dict(self.__dict__)

        def __setstate__(self, d):
            self.__dict__.update(d)

    class AnyNode:
        def __init__(self, val):
            self._val = val
            self._metadata = Metadata()

    class Metadata:
        def __init__(self):
            self.optional, self.key, self.flags = True, None, {}

    DictConfig.__module__, DictConfig.__qualname__ = "omegaconf.dictconfig", "DictConfig"
    AnyNode.__module__, AnyNode.__qualname__ = "omegaconf.nodes", "AnyNode"
    Metadata.__module__, Metadata.__qualname__ = "omegaconf.base", "Metadata"
    mods["omegaconf.dictconfig"].DictConfig = DictConfig
    mods["omegaconf.nodes"].AnyNode = AnyNode
    mods["omegaconf.base"].Metadata = Metadata

    def wrap(x):
        if isinstance(x, dict):
            return DictConfig({k: wrap(v) for k, v in x.items()})
        return AnyNode(x)
    return mods, wrap(tree)


@pytest.mark.parametrize("family", [0, 1])
def test_lightning_checkpoint_with_omegaconf_hyperparameters(family, tmp_path):
    """src/inference_base.py:611-614: the real checkpoints carry omegaconf DictConfig hyper-parameters.  The reader must get
    the hyper-parameters (model dims, agg, diffuser sigmas) and the weights out of such a pickle WITHOUT omegaconf or
    pytorch_lightning installed, for both model families, and must not execute anything the pickle names."""
    import torch
    from dfmdock_amd.weights import HParams, load_lightning_checkpoint, make_random_weights, pack_blob
    hp0 = HParams(family=1, mask_dist=20.0, positional_embed_dim=67, agg_mean=False) if family else HParams()
    w = make_random_weights(6, hp0)
    sd = {"net." + k: torch.from_numpy(v.copy()) for k, v in w.items()}
    model_cfg = {"lm_embed_dim": 1301, "positional_embed_dim": hp0.positional_embed_dim, "spatial_embed_dim": 100, "node_dim": 256,
                 "edge_dim": 128, "inner_dim": 128, "depth": 6, "dropout": 0.1, "cut_off": 20.0, "normalize": True}
    if family:
        model_cfg["agg"] = "sum"
    tree = {"model": model_cfg,
            "diffuser": {"r3": {"min_sigma": 0.2, "max_sigma": 25.0, "schedule": "VE"},
                         "so3": {"num_omega": 1000, "min_sigma": 0.05, "max_sigma": 1.25, "schedule": "logarithmic"}},
            "experiment": {"lr": 1e-4, "perturb_tr": True}}
    mods, hyper = _omegaconf_shaped(tree)
    path = tmp_path / f"family{family}.ckpt"
    sys.modules.update(mods)
    try:
        torch.save({"state_dict": sd, "hyper_parameters": hyper, "epoch": 7, "pytorch-lightning_version": "2.4.0"}, path)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert "omegaconf" not in sys.modules
    out, hp = load_lightning_checkpoint(str(path))
    np.testing.assert_array_equal(pack_blob(out, hp), pack_blob(w, hp0))
    assert (hp.family, hp.depth, hp.node_dim, hp.cut_off, hp.positional_embed_dim) == (family, 6, 256, 20.0, hp0.positional_embed_dim)
    assert (hp.r3_min_sigma, hp.r3_max_sigma, hp.so3_min_sigma, hp.so3_max_sigma) == (0.2, 25.0, 0.05, 1.25)
    if family:
        assert hp.agg_mean is False and hp.mask_dist == 20.0
    # a bare state_dict (no hyper_parameters) still resolves the family from its keys
    torch.save(sd, tmp_path / "bare.pt")
    out2, hp2 = load_lightning_checkpoint(str(tmp_path / "bare.pt"))
    assert hp2.family == family and hp2.positional_embed_dim == hp0.positional_embed_dim
    np.testing.assert_array_equal(pack_blob(out2, hp2), pack_blob(w, hp0))


def test_checkpoint_reader_runs_nothing_from_the_pickle(tmp_path):
    """Globals outside the allow-list (here: os.system through a __reduce__) are never resolved: they come back as inert
    attribute bags instead of being called."""
    import pickle
    import torch
    from dfmdock_amd.weights import load_lightning_checkpoint
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    path = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"net.single_embed.weight": torch.zeros(2, 2)}, "hyper_parameters": {"model": Evil()}}, path,
               pickle_module=pickle)
    out, hp = load_lightning_checkpoint(str(path))
    assert not marker.exists()
    assert "single_embed.weight" in out


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


INIT_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    from dfmdock_amd import distributed as D
    rank, _, world = D.dist_env()
    g = D.init()                                  # prefers nccl; this box has no GPU -> must settle on gloo (or files) and say why
    lo, hi = D.shard_range(9, world, rank)
    n = hi - lo
    res = dict(energy=np.arange(lo, hi, dtype=np.float32)[::-1].copy(), num_clashes=np.zeros(n, np.int32),
               rot_update=np.zeros((n, 3), np.float32), tr_update=np.full((n, 3), rank, np.float32))
    allrec = D.gather_records(D.make_records(rank, np.arange(lo, hi), res))
    again = D.gather_records(D.make_records(rank, np.arange(lo, hi), res))      # a second round on the same group / directory
    tmax = D.allreduce_max(1.5 + rank)
    tab = D.allgather_scalars([0.1 * (rank + 1), 7.0 + rank])
    D.barrier()
    json.dump(dict(backend=g.backend, reason=g.fallback_reason, n=int(allrec.shape[0]), ids=allrec[:, 1].tolist(),
                   ranks=sorted(set(allrec[:, 0].astype(int).tolist())), same=bool((allrec == again).all()), tmax=tmax,
                   tab=tab.tolist()), open(os.path.join({out!r}, f"init_{{rank}}.json"), "w"))
    D.shutdown()
""")


@pytest.mark.parametrize("mode", ["rendezvous", "replicas"])
def test_init_falls_back_from_nccl_and_gathers(tmp_path, mode):
    """distributed.init(): RCCL is probed and, when it does not work on every rank (no GPU here), the record gather runs over
    gloo - or, with no rendezvous at all (independent replicas: RANK / WORLD_SIZE / DFM_GATHER_DIR only), over files
    (SURVEY 8(e) last row).  Same records either way, and the group says what it is and why."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(INIT_WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", DFM_GATHER_DIR=str(tmp_path / "gather"))
        env.pop("DFM_DIST_BACKEND", None)
        if mode == "rendezvous":
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29100 + (os.getpid() % 400)))
        else:
            env.pop("MASTER_PORT", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    a, b = (json.load(open(tmp_path / f"init_{r}.json")) for r in range(2))
    assert a == b
    assert a["backend"] == ("gloo" if mode == "rendezvous" else "file") and a["reason"]
    if mode == "rendezvous":
        assert "nccl" in a["reason"] or "GPU" in a["reason"]
    assert a["n"] == 9 and a["ids"] == list(range(9)) and a["ranks"] == [0, 1] and a["same"]
    assert a["tmax"] == 2.5 and a["tab"] == [[0.1, 7.0], [0.2, 8.0]]


def test_file_gather_ignores_a_previous_jobs_files(tmp_path):
    """a second job (or a restarted rank) that reuses DFM_GATHER_DIR must not read the earlier run's records, and
    barrier() must stay a barrier.  Two jobs run back to back in ONE directory, the second with different records, and the
    directory is salted with files in the old naming scheme (round000001_rank1.npy) holding poison; rank 1 of the second job starts
    late, so rank 0 would return at once if it accepted anything already lying there."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys, time
        import numpy as np
        sys.path.insert(0, {root!r})
        from dfmdock_amd import distributed as D
        rank, _, world = D.dist_env()
        time.sleep(float(os.environ.get("START_DELAY", "0")) * rank)
        g = D.init(prefer="file")
        base = float(os.environ["JOB_BASE"])
        rec = np.full((2, D.RECORD_WIDTH), base + rank, np.float32)
        a = D.gather_records(rec)
        D.barrier()
        objs = D.gather_objects({{"rank": rank, "base": base}})
        json.dump(dict(vals=sorted(set(a[:, 0].tolist())), n=int(a.shape[0]), objs=objs), open(os.path.join({out!r}, f"job{{int(base)}}_{{rank}}.json"), "w"))
        D.shutdown()
    """).format(root=ROOT, out=str(tmp_path)))
    gdir = tmp_path / "gather"
    gdir.mkdir()
    poison = np.full((2, 10), -777.0, np.float32)
    for r in range(2):
        np.save(gdir / f"round000001_rank{r}.npy", poison)
        np.save(gdir / f"round000002_rank{r}.npy", poison[:0])
    for base, delay in ((100, "0"), (200, "1.0")):
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", DFM_GATHER_DIR=str(gdir), JOB_BASE=str(base), START_DELAY=delay)
            for k in ("MASTER_PORT", "DFM_DIST_BACKEND", "DFM_JOB_ID"):
                env.pop(k, None)
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        for r in range(2):
            o = json.load(open(tmp_path / f"job{base}_{r}.json"))
            assert o["vals"] == [float(base), float(base + 1)] and o["n"] == 4, o       # this job's records only
            assert [x["base"] for x in o["objs"]] == [float(base)] * 2 and [x["rank"] for x in o["objs"]] == [0, 1]
    # a finished job leaves only its closing round behind (two small files), not one file per round and rank
    left = [n for n in os.listdir(gdir) if not n.startswith("round0")]
    assert len(left) <= 4, left


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



def test_rccl_unique_id_handover_tcp_and_file(tmp_path):
    """dfmdock_amd/rccl.py: the 128-byte communicator id travels from rank 0 to the other ranks over a TCP socket (torchrun-style
    launches: MASTER_ADDR / MASTER_PORT + 1) or a file in DFM_GATHER_DIR; late joiners retry until rank 0 is up."""
    import socket, threading, time
    from dfmdock_amd import rccl
    uid = bytes(range(128))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    got = {}

    def client(r, delay):
        time.sleep(delay)
        got[r] = rccl.exchange_uid_tcp(None, r, 3, "127.0.0.1", port, timeout_s=30)

    th = [threading.Thread(target=client, args=(1, 0.0)), threading.Thread(target=client, args=(2, 0.3))]
    for t in th:
        t.start()
    time.sleep(0.2)                                    # rank 1 is already knocking when rank 0 starts to listen
    assert rccl.exchange_uid_tcp(uid, 0, 3, "127.0.0.1", port, timeout_s=30) == uid
    for t in th:
        t.join(30)
    assert got == {1: uid, 2: uid}
    d = str(tmp_path / "g"); os.makedirs(d)
    res = {}
    t = threading.Thread(target=lambda: res.update(r1=rccl.exchange_uid_file(None, 1, 2, d, "jx", timeout_s=30)))
    t.start()
    time.sleep(0.1)
    assert rccl.exchange_uid_file(uid, 0, 2, d, "jx") == uid
    t.join(30)
    assert res["r1"] == uid
    with pytest.raises(TimeoutError):
        rccl.exchange_uid_file(None, 1, 2, d, "other-job", timeout_s=0.2)


def test_rccl_uid_file_ignores_leftovers_of_an_earlier_job(tmp_path):
    """without DFM_JOB_ID the token of the id file is just MASTER_PORT, so a second job in the same DFM_GATHER_DIR found
    the first job's file at once and handed a dead id to ncclCommInitRank.  The exchange is now a nonce handshake: leftovers of an
    earlier (crashed) job - an id file in the old or the new format, hello / ack files - are never accepted, and a completed
    exchange leaves nothing behind."""
    import threading, time
    from dfmdock_amd import rccl
    d = str(tmp_path / "g"); os.makedirs(d)
    old, new = bytes([7] * 128), bytes(range(128, 256))
    open(os.path.join(d, "29500_rccl_uid"), "wb").write(old)                                  # r04 format
    for world in (2, 3):
        open(os.path.join(d, "29500_rccl_uid"), "wb").write(old + os.urandom(16 * (world - 1)))   # this round's format, dead nonces
        for r in range(1, world):
            open(os.path.join(d, f"29500_rccl_hello_{r}"), "wb").write(os.urandom(16))            # a crashed job's hello / ack files
            open(os.path.join(d, f"29500_rccl_ack_{r}"), "wb").write(os.urandom(16))
        res = {}
        th = [threading.Thread(target=lambda r=r: res.update({r: rccl.exchange_uid_file(None, r, world, d, "29500", timeout_s=30)}))
              for r in range(1, world)]
        th[0].start()
        time.sleep(0.15)      # rank 1 is early: it must NOT take the stale file
        assert not res
        for t in th[1:]:
            t.start()
        assert rccl.exchange_uid_file(new, 0, world, d, "29500", timeout_s=30) == new
        for t in th:
            t.join(30)
        assert res == {r: new for r in range(1, world)}
        assert [f for f in os.listdir(d) if f.startswith("29500_rccl")] == []
    # the NORMAL leftover of a crashed job - rank 1 had acknowledged (hello and ack carry the SAME nonce) and rank 0 died
    # before its clean-up.  A new rank 0 that starts first must not take that pair for this job's acknowledgement: the ack is bound
    # to the id it acknowledges, and a late rank 1 still completes the exchange.
    n_old = os.urandom(16)
    open(os.path.join(d, "29500_rccl_hello_1"), "wb").write(n_old)
    open(os.path.join(d, "29500_rccl_ack_1"), "wb").write(n_old + old[:16])       # this round's ack format, of the dead job's id
    res = {}
    t0 = threading.Thread(target=lambda: res.update({0: rccl.exchange_uid_file(new, 0, 2, d, "29500", timeout_s=30)}))
    t0.start()
    time.sleep(0.3)
    assert not res                                                                    # rank 0 is still waiting for a REAL rank 1
    assert rccl.exchange_uid_file(None, 1, 2, d, "29500", timeout_s=30) == new
    t0.join(30)
    assert res == {0: new} and [f for f in os.listdir(d) if f.startswith("29500_rccl")] == []
    open(os.path.join(d, "29500_rccl_hello_1"), "wb").write(n_old)                    # ... and the r05 ack format (bare nonce) is no ack at all
    open(os.path.join(d, "29500_rccl_ack_1"), "wb").write(n_old)
    with pytest.raises(TimeoutError):
        rccl.exchange_uid_file(new, 0, 2, d, "29500", timeout_s=0.4)


def test_set_run_assignment_over_eight_ranks_with_the_measured_cost_model():
    """C4's sharding: the 24 DB5 test complexes over 8 ranks, longest first, under the cost model measured on one MI355X
    (distributed.complex_cost: a B = 40 sampling call is AFFINE in the residue count - profiles/r05_c4.txt - so balancing on N
    alone over-weights the large complexes).  Every complex on exactly one rank, three per rank here, estimated makespan within
    10 % of the perfect split and no worse than balancing on N alone; run_set itself takes the same assignment (2-rank gloo test
    above runs it end to end)."""
    from dfmdock_amd import distributed as D
    sizes = [395, 695, 343, 456, 575, 626, 561, 329, 197, 352, 320, 430, 377, 430, 382, 339, 628, 404, 240, 535, 373, 588, 492, 214]
    assert len(sizes) == 24
    costs = [D.complex_cost(n, 40) for n in sizes]
    assert D.complex_cost(600, 80) == pytest.approx(2 * D.complex_cost(600, 40)) and costs[1] / costs[8] < sizes[1] / sizes[8]
    a = D.assign_work(costs, 8)
    assert sorted(i for part in a for i in part) == list(range(24)) and all(len(part) == 3 for part in a)
    ideal = sum(costs) / 8 + D.SET_COST_EDGE_MS
    assert D.makespan(costs, a) <= 1.10 * ideal, (D.makespan(costs, a), ideal)
    by_n = D.assign_work([float(n) for n in sizes], 8)
    assert D.makespan(costs, a) <= D.makespan(costs, by_n) + 1e-9
    # fewer complexes than 2 x ranks: run_set splits TRAJECTORIES instead (every rank holds every complex)
    assert [D.shard_range(40, 8, r) for r in range(8)] == [(5 * r, 5 * r + 5) for r in range(8)]


def test_run_set_pipeline_stages_overlap_and_fail_cleanly(monkeypatch):
    """driver.run_set's three-stage pipeline with the engine stubbed out (CPU): rows come back in share order whatever the stage
    timings, the stages of neighbouring complexes really overlap (wall clock well below the serial sum), at most `samplers + 1`
    handles are ever prepared ahead of sampling, and an exception in any stage surfaces in the caller instead of hanging the pools."""
    import threading, time
    from dfmdock_amd import driver
    log, lock = [], threading.Lock()
    live = {"prepared": 0, "max_prepared": 0}

    class P:
        pass

    def prep(model, c, ci, rot_seed, global_rotation, precision, selfcheck, on_fail, seed, log_=None):
        with lock:
            live["prepared"] += 1
            live["max_prepared"] = max(live["max_prepared"], live["prepared"])
        time.sleep(0.03)
        p = P(); p.ci, p.c, p.precision, p.check, p.ms = ci, c, precision, None, {"prepare": 30.0}
        p.N = c["rec_x"].shape[0] + c["lig_x"].shape[0]
        return p

    calls = {}

    def samp(p, t_lo, t_hi, num_steps, seed, max_batch, trace, kw):
        with lock:
            live["prepared"] -= 1
            log.append(("sample", p.ci))
            calls[p.ci] = calls.get(p.ci, 0) + 1
            nth = calls[p.ci]
        if p.c.get("boom") == "sample":
            raise RuntimeError("sampling failed")
        time.sleep(0.06)
        p.ms["sample"] = 60.0
        # results as the engine returns them; a complex marked "flaky" gives a different answer the first time it is sampled (= next to others)
        val = float(p.ci) + (0.5 if (p.c.get("flaky") and nth == 1) else 0.0)
        r = {k: np.full((t_hi - t_lo, 1), val, np.float32) for k in ("lig_pos", "energy", "num_clashes", "rot_update", "tr_update")}
        return [(t_lo, t_hi - t_lo, r)]

    def post(p, batches, traj_dir):
        if p.c.get("boom") == "post":
            raise RuntimeError("post failed")
        time.sleep(0.03)
        p.ms["post"] = 30.0
        return [{"id": p.c["id"], "index": str(k)} for k in range(batches[0][1])], []

    monkeypatch.setattr(driver, "_prepare", prep)
    monkeypatch.setattr(driver, "_sample", samp)
    monkeypatch.setattr(driver, "_post", post)
    cxs = [{"id": f"C{k}", "rec_x": np.zeros((10 + k, 4)), "lig_x": np.zeros((5, 4))} for k in range(8)]
    for samplers in (1, 2):
        log.clear(); live.update(prepared=0, max_prepared=0)
        tim = []
        t0 = time.perf_counter()
        rows, ranked = driver.run_set(None, cxs, num_samples=3, overlap=True, samplers=samplers, timings_out=tim, canary=False)
        dt = time.perf_counter() - t0
        order = [c["id"] for c in sorted(cxs, key=lambda c: -c["rec_x"].shape[0])]      # one rank: longest first
        assert [r["id"] for r in rows] == [i for i in order for _ in range(3)] and ranked == {}
        assert [t["id"] for t in tim] == order
        assert dt < 0.8 * 8 * 0.12, dt                                   # serial would be 8 x (30 + 60 + 30) ms
        assert live["max_prepared"] <= samplers + 1, live
    t0 = time.perf_counter()
    rows_serial, _ = driver.run_set(None, cxs, num_samples=3, overlap=False)
    assert rows_serial == rows and time.perf_counter() - t0 > 0.9 * 8 * 0.12
    for where in ("sample", "post"):
        bad = [dict(c) for c in cxs]
        bad[3]["boom"] = where
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match=where[:4]):
            driver.run_set(None, bad, num_samples=3, overlap=True)
        assert time.perf_counter() - t0 < 5.0
    # the canary (default on): the cheapest complex that sampled with others in flight is sampled again, alone, and compared bit for bit
    calls.clear(); can = {}
    rows_c, _ = driver.run_set(None, cxs, num_samples=3, overlap=True, canary_out=can)
    assert rows_c == rows and can == {"checked": True, "id": "C1", "ok": True, "reran_serial": False} and calls[1] == 2 and calls[0] == 1
    flaky = [dict(c) for c in cxs]
    flaky[1]["flaky"] = True                                              # C1 is the canary: its pipelined result will not reproduce
    calls.clear(); can = {}; msgs = []
    rows_f, _ = driver.run_set(None, flaky, num_samples=3, overlap=True, canary_out=can, log=msgs.append)
    assert can["ok"] is False and can["reran_serial"] is True and rows_f == rows
    assert any("canary" in m for m in msgs) and all(calls[k] >= 2 for k in range(8))      # every complex was sampled again, serially


def test_bench_c4_set_is_the_db5_fixtures(monkeypatch, tmp_path):
    """bench.py's C4 records run on the 24 DB5 test complexes with the reference loader's features (src/datasets/ppi_dataset.py:249-265) rebuilt from the
    committed fixtures - the same arrays the parity tests use - and fall back to synthetic complexes of the same sizes only when the fixtures are absent."""
    import bench
    from conftest import db5_ids, real_db5_complex
    cxs, what = bench.c4_complexes()
    assert [c["id"] for c in cxs] == db5_ids() and "ESM-2" in what
    for c in cxs[::5]:
        r = real_db5_complex(c["id"])
        for k in ("rec_x", "lig_x", "rec_pos", "lig_pos"):
            np.testing.assert_array_equal(c[k], r[k])
    assert sorted((c["rec_x"].shape[0], c["lig_x"].shape[0]) for c in cxs) == sorted(bench.DB5_SIZES)      # the sizes SURVEY Appendix A lists
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(tmp_path / "bench.py"))      # no tests/golden next to it
    syn, what2 = bench.c4_complexes()
    assert len(syn) == 24 and "synthetic" in what2 and sorted((c["rec_x"].shape[0], c["lig_x"].shape[0]) for c in syn) == sorted(bench.DB5_SIZES)

"""The multi-rank path with REAL engine processes: two ranks, one per process, both on GPU 0 of the test box,
rendezvous over gloo (the 8-GPU node runs the same code over RCCL: backend "nccl" stays the default everywhere).

  * bench.py --gpus 2 under torch.distributed.run: the weak-scaling step (per-rank dfm_sample + one record all_gather +
    max-over-ranks timing) end to end;
  * plain `python bench.py --gpus 2`: no launcher, bench.py starts its two ranks itself; RCCL cannot serve two ranks on ONE GPU,
    so the probe fails and the gather falls back to gloo - the JSON line says so;
  * driver.run_set over 4 complexes on 2 ranks: complexes sharded longest-first, disjoint (complex, trajectory) ids, every rank
    ends with the identical energy-ranked table, rank 0 writes the complete CSV; over 1 complex on 2 ranks: its trajectories
    are split between the ranks instead.

RCCL between GPUs is not executed by any test here (one GPU; the 8-GPU node is the driver's) - but a ONE-rank RCCL communicator is:
`test_rccl_one_rank_record_gather` builds the "nccl" process group on the box's GPU and runs the record gather of
dfmdock_amd.distributed through it (library load, communicator init, all_gather / all_reduce kernels on the MI355X).
"""
import csv
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _port():
    return 29900 + (os.getpid() % 300)


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, DFM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--num-steps", "6", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]            # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] > 0 and np.isfinite(out["best_energy"]) and "cpu_baseline" not in out
    assert out["config"]["trajectories_per_gpu"] == 8
    assert abs(out["value"] - 2 * 8 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]   # whole-job aggregate over both ranks
    r = out["roofline"]      # the contract's roofline object, plus the clock the dominant kernel measured for itself
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["unit"] == "TFLOP/s"
    assert 500.0 < r["sclk_mhz"] <= 2600.0 and r["frac_at_sclk"] == pytest.approx(r["frac"] * 2400.0 / r["sclk_mhz"])


def test_bench_self_spawns_its_ranks_and_survives_a_broken_rccl():
    """`python bench.py --gpus 2 --batch 8`: the driver's command shape with no torchrun around it.  Both ranks land on the one
    GPU of the test box, where RCCL refuses to build a communicator: the line must still come out, with n_gpus = 2, the backend
    that carried the gather and the reason RCCL was dropped."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DFM_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--num-steps", "6", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_in_gather"] == 2 and out["config"]["trajectories_per_gpu"] == 8
    assert out["backend"] in ("nccl", "gloo", "file")
    if out["distinct_devices"] == 1:      # two ranks on one GPU: RCCL cannot have worked
        assert out["backend"] != "nccl" and out["backend_fallback"] and "nccl" in out["backend_fallback"].lower()
    assert abs(out["value"] - 2 * 8 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]


RUN_SET_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    from dfmdock_amd import distributed as D, driver, engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob
    rank, local, world = D.dist_env()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    engine.set_device(0)                      # both ranks share the one GPU of the test box
    model = engine.Model(pack_blob(make_random_weights(0)))
    cxs = []
    for k, (R, L) in enumerate({shapes!r}):
        c = make_complex(R, L, seed=20 + k)
        c.update(id=f"SYN{{k}}", rec_seq="A" * R, lig_seq="G" * L)
        cxs.append(c)
    rows, ranked = driver.run_set(model, cxs, num_samples=5, num_steps=4, seed=3, out_csv=os.path.join({out!r}, "set.csv"), max_batch=3)
    json.dump({{"rows": [[r["id"], r["index"], r["energy"]] for r in rows],
               "ranked": {{str(k): v.tolist() for k, v in ranked.items()}}}}, open(os.path.join({out!r}, f"rank{{rank}}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
""")


def _run_set_two_ranks(tmp_path, shapes, port):
    script = tmp_path / "worker.py"
    script.write_text(RUN_SET_WORKER.format(root=ROOT, out=str(tmp_path), shapes=shapes))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return json.load(open(tmp_path / "rank0.json")), json.load(open(tmp_path / "rank1.json"))


def test_run_set_two_ranks_one_gpu(tmp_path):
    a, b = _run_set_two_ranks(tmp_path, [(30, 22), (70, 41), (41, 17), (25, 25)], _port() + 1)
    ids_a, ids_b = {(r[0], r[1]) for r in a["rows"]}, {(r[0], r[1]) for r in b["rows"]}
    assert ids_a and ids_b and not (ids_a & ids_b)                                  # every rank sampled something, nothing twice
    assert ids_a | ids_b == {(f"SYN{k}", str(i)) for k in range(4) for i in range(5)}
    assert {r[0] for r in a["rows"]}.isdisjoint({r[0] for r in b["rows"]})            # sharded by complex
    assert a["ranked"] == b["ranked"] and sorted(a["ranked"]) == ["0", "1", "2", "3"]    # identical ranked table everywhere
    for k, tab in a["ranked"].items():
        e = np.array(tab)[:, 2]
        assert len(e) == 5 and (np.diff(e) >= 0).all()
    got = list(csv.DictReader(open(tmp_path / "set.csv")))
    assert len(got) == 20 and {(r["id"], r["index"]) for r in got} == ids_a | ids_b
    by_key = {(r[0], r[1]): r[2] for r in a["rows"] + b["rows"]}
    for r in got:
        assert abs(float(r["energy"]) - by_key[(r["id"], r["index"])]) < 1e-6


def test_run_set_splits_trajectories_when_complexes_are_few(tmp_path):
    """One complex on two ranks: fewer than two complexes per rank, so each rank samples its block of the complex's trajectories
    (3 + 2 of 5) and the gathered table holds all five exactly once."""
    a, b = _run_set_two_ranks(tmp_path, [(70, 41)], _port() + 2)
    ids_a, ids_b = {(r[0], r[1]) for r in a["rows"]}, {(r[0], r[1]) for r in b["rows"]}
    assert ids_a == {("SYN0", str(i)) for i in range(3)} and ids_b == {("SYN0", str(i)) for i in (3, 4)}
    assert a["ranked"] == b["ranked"] and list(a["ranked"]) == ["0"] and len(a["ranked"]["0"]) == 5
    assert sorted(int(r[1]) for r in a["ranked"]["0"]) == [0, 1, 2, 3, 4]
    got = list(csv.DictReader(open(tmp_path / "set.csv")))
    assert [(r["id"], r["index"]) for r in got] == [("SYN0", str(i)) for i in range(5)]


def test_rccl_one_rank_record_gather():
    """RCCL itself on the test box: a one-rank "nccl" group (RCCL refuses two ranks on one device, a single rank it serves), the
    probe all_reduce of distributed.init() and the padded two-phase all_gather of trajectory records that closes a multi-GPU job
    (inference_base.py:644-657), on device tensors.  A failure to load or initialise RCCL on this image shows up here, not on the
    8-GPU node."""
    code = textwrap.dedent("""
        import datetime, os, sys
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from dfmdock_amd import distributed as D
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert float(probe.item()) == 1.0
        D._group = D.Group("nccl", 0, 1, data_group=dist.group.WORLD)
        rec = np.arange(5 * D.RECORD_WIDTH, dtype=np.float32).reshape(5, D.RECORD_WIDTH)
        out = D.gather_records(rec, force_collective=True)
        assert dist.get_backend() == "nccl" and out.shape == rec.shape and (out == rec).all(), out.shape
        dist.destroy_process_group()
        print("RCCL_OK", torch.cuda.get_device_name(0))
    """ % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port() + 301), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout.decode(), p.stderr.decode()[-3000:]


def test_rccl_direct_one_rank_without_torch():
    """The torch-free collective path (dfmdock_amd/rccl.py, DFM_DIST_BACKEND=rccl): librccl through ctypes builds a communicator on
    the MI355X and runs the record gather, the timing exchange, the object gather and a barrier as byte all_gathers - in a process
    that never imports torch."""
    code = textwrap.dedent("""
        import sys
        import numpy as np
        sys.path.insert(0, %r)
        from dfmdock_amd import distributed as D, rccl
        comm = rccl.Rccl(0, 1, 0, lambda uid: uid)
        D._group = D.Group("rccl", 0, 1, data_group=comm)
        rec = np.arange(7 * D.RECORD_WIDTH, dtype=np.float32).reshape(7, D.RECORD_WIDTH)
        out = D.gather_records(rec)
        assert out.shape == rec.shape and (out == rec).all()
        assert D.gather_records(rec[:0]).shape == (0, D.RECORD_WIDTH)
        t = D.allgather_scalars([1.25, 3.0])
        assert t.shape == (1, 2) and t[0, 0] == 1.25 and D.allreduce_max(2.5) == 2.5
        assert D.gather_objects({"rows": [["1AVX", "0", -0.5]]}) == [{"rows": [["1AVX", "0", -0.5]]}]
        D.barrier()
        D.shutdown()
        assert "torch" not in sys.modules, "the rccl path must not need torch"
        print("RCCL_DIRECT_OK")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and "RCCL_DIRECT_OK" in p.stdout.decode(), p.stderr.decode()[-3000:]


def test_bench_eight_ranks_oversubscribed_on_one_gpu():
    """The driver's 8-GPU command shape on the 1-GPU test box: `python bench.py --gpus 8` starts eight ranks itself, all of them land
    on GPU 0 (local rank modulo the device count), RCCL cannot serve them there and the gather falls back (gloo / files).  What
    must hold: ONE JSON line, n_gpus = 8, every rank's block in the gather - 8 ranks x 8 trajectories = 64 distinct record ids -
    and value = the whole job's trajectories over the slowest rank's time.  No scaling claim: eight processes share one GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DFM_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "8",
           "--num-steps", "4", "--no-cpu-baseline", "--no-fp32-line"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_in_gather"] == 8 and out["config"]["trajectories_per_gpu"] == 8
    assert out["records_in_gather"] == 64 and out["distinct_record_ids"] == 64
    assert out["backend"] in ("nccl", "gloo", "file") and out["scaling"] == "weak"
    if out["distinct_devices"] == 1:
        assert out["backend"] != "nccl" and out["backend_fallback"]
    assert abs(out["value"] - 8 * 8 * 1 / (out["ms_per_step"] / 1e3)) < 1e-6 * out["value"]
    assert "c4" not in out and "c5" not in out and "cpu_baseline" not in out      # secondary records: rank 0 at N = 1 only


def test_bench_c4_sharded_record_on_two_ranks(tmp_path):
    """BASELINE config 4 is DEFINED as a sharded run (24 DB5-sized complexes x 40 trajectories over the ranks of one node, one gather
    of the ranked energy records; the loop it replaces: src/inference_mlsb.py:415-439).  `bench.py --gpus N` puts it on the line as
    `c4_sharded`.  Two ranks on the one GPU of the test box: 960 records with 960 distinct (complex,
    trajectory) ids reach every rank, both ranks took complexes, and the sharded CSV is byte-identical to the one-rank CSV of the
    same call (a trajectory is a pure function of seed, complex and index)."""
    env = dict(os.environ, DFM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    c4 = out["c4_sharded"]
    assert out["n_gpus"] == 2 and out["config"]["workload"].startswith("C3") and "c4" not in out
    assert c4["records_in_gather"] == 960 and c4["distinct_record_ids"] == 960 and c4["complexes_in_gather"] == 24
    assert sum(c4["complexes_per_rank"]) == 24 and min(c4["complexes_per_rank"]) >= 8 and sum(c4["rows_per_rank"]) == 960
    assert len(c4["per_rank_makespan_s"]) == 2 and c4["wall_s"] >= max(c4["per_rank_makespan_s"]) > 0
    assert c4["canary"]["ok"] is True and c4["backend"] in ("gloo", "nccl", "file")

    # the same set through run_set on 2 ranks and on 1 rank: byte-identical CSV
    worker = tmp_path / "c4_worker.py"
    worker.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        from dfmdock_amd import distributed as D, driver, engine
        from dfmdock_amd.synthetic import make_complex
        from dfmdock_amd.weights import make_random_weights, pack_blob
        rank, local, world = D.dist_env()
        grp = D.init(device_index=0)
        engine.set_device(0)
        model = engine.Model(pack_blob(make_random_weights(0)))
        cxs = []
        for k, (R, L) in enumerate(bench.DB5_SIZES):
            c = make_complex(R, L, seed=300 + k); c["id"] = f"S{{k:02d}}_{{R}}_{{L}}"; cxs.append(c)
        rows, ranked = driver.run_set(model, cxs, num_samples=40, num_steps=10, seed=0, out_csv=os.path.join({str(tmp_path)!r}, f"c4_w{{world}}.csv"),
                                      log=lambda m: None)
        assert sum(len(v) for v in ranked.values()) == 960
        D.shutdown()
    """))
    port = _port() + 7
    for world in (1, 2):
        procs = []
        for r in range(world):
            e2 = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + world),
                      DFM_GATHER_DIR=str(tmp_path))
            procs.append(subprocess.Popen([sys.executable, str(worker)], env=e2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [q.communicate(timeout=900)[0].decode() for q in procs]
        assert all(q.returncode == 0 for q in procs), "\n".join(o[-3000:] for o in outs)
    a, b = open(tmp_path / "c4_w1.csv", "rb").read(), open(tmp_path / "c4_w2.csv", "rb").read()
    assert a == b and a.count(b"\n") == 961


def test_bench_eight_ranks_headline_shape_carries_c4_sharded():
    """The driver's 8-GPU command at the HEADLINE shape (no size overrides) on the 1-GPU box: eight self-spawned ranks share GPU 0.  After the weak-scaling C3
    step (8 x 256 trajectories) the line must carry BASELINE config 4 as `c4_sharded`: 24 DB5-sized complexes x 40 trajectories sharded three per rank by the
    cost model, 960 distinct (complex, trajectory) ids in ONE gather (the loop it replaces: src/inference_mlsb.py:415-439).  No scaling claim."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DFM_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_in_gather"] == 8 and out["records_in_gather"] == 8 * 256 == out["distinct_record_ids"]
    assert "300+300" in out["config"]["workload"] and out["config"]["trajectories_per_gpu"] == 256
    c4 = out["c4_sharded"]
    assert c4["records_in_gather"] == 960 == c4["distinct_record_ids"] and c4["complexes_in_gather"] == 24
    assert sorted(c4["complexes_per_rank"]) == [3] * 8 and sum(c4["rows_per_rank"]) == 960
    assert len(c4["per_rank_makespan_s"]) == 8 and c4["wall_s"] >= max(c4["per_rank_makespan_s"]) - 1e-3
    assert c4["value"] == pytest.approx(960 / c4["wall_s"]) and c4["canary"]["ok"]
    assert "c4" not in out and "cpu_baseline" not in out      # single-GPU secondary records stay at N = 1

#!/usr/bin/env python3
"""bench.py - docking trajectories/second of the sampling hot path on MI355X.

One "step" = one batched dfm_sample call: B independent trajectories x 40 Euler-Maruyama steps
(41 score evaluations each + the final energy head) on the synthetic 300+300-residue complex of
BASELINE.json configs[2] (C3).  Inputs (weights, node features, backbone) are resident in HBM before
the timed region; one call returns only ~40 B per trajectory.  With N > 1 GPUs every rank samples its own
B trajectories (weak scaling, no data-path collective) and one tiny all_gather of energy-ranked
records closes the step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision mfma16|f16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with no RANK in the environment starts the N ranks itself (one process per GPU, rendezvous on
127.0.0.1) and still prints ONE JSON line.  The record gather runs over RCCL ("nccl") when every rank's RCCL probe succeeds,
else over gloo, else over files (dfmdock_amd/distributed.py); the line says which ("backend", "backend_fallback").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of the dominant kernel (per-edge message kernel), SURVEY.md 8(d):
# per node and layer 2*K*H*H (edge_mlp.2) + 2*K*H (attention gate)
H, K_DEG = 256, 60
FLOP_PER_NODE_LAYER = 2 * K_DEG * H * H + 2 * K_DEG * H
PEAK_MFMA16_TFLOPS = 2500.0   # dense MFMA bf16 / fp16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3       # fp32 vector / f32-input MFMA
PEAK_SCLK_MHZ = 2400          # the engine clock the peak figures are quoted at


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(blob, cx, num_steps, repeats=2):
    """Oracle (plain-C port of the reference as written, OpenMP) timed on the host cores of this box on a bounded sample.

    value = TRAJECTORY-PARALLEL throughput on ALL host cores: the metric is trajectories/s and trajectories are
    independent until the final arg-min (inference_base.py:644-657), so the honest all-cores CPU figure is one single-threaded
    trajectory per core (ora_sample_many), not one evaluation spread over the cores.  Sample: host_cores trajectories x 2 score
    evaluations each after one untimed evaluation per trajectory, extrapolated to the 41 evaluations of a trajectory.  The
    intra-evaluation OpenMP numbers of r01-r05 (which peak near 32 threads and fall at 128) stay under by_mode for comparison."""
    import statistics
    from oracle import oracle as ora
    L = ora.lib()
    o = ora.Oracle(blob, cx)
    all_cores = int(L.ora_num_threads())
    n_eval = 2
    o.sample_many(all_cores, num_steps=num_steps, max_forwards=1, seed=1, n_threads=all_cores)      # warm-up: pages, thread pool
    t0 = time.perf_counter()
    res = o.sample_many(all_cores, num_steps=num_steps, max_forwards=n_eval, seed=2, n_threads=all_cores)
    dt_tp = time.perf_counter() - t0
    tp = res["total_forwards"] / dt_tp / (num_steps + 1)

    def sample(n_threads, n_forwards):
        L.ora_set_num_threads(n_threads)
        o.sample(num_steps=num_steps, max_forwards=1, seed=1)                 # warm-up
        vals = []
        for r in range(repeats):
            t0 = time.perf_counter()
            res = o.sample(num_steps=num_steps, max_forwards=n_forwards, seed=2 + r)
            dt = time.perf_counter() - t0
            vals.append(1.0 / (dt / max(res["forwards"], 1) * (num_steps + 1)))
        return vals

    counts = sorted({min(8, all_cores), min(32, all_cores), all_cores})
    runs = {n: sample(n, 4 if n >= 32 else 2) for n in counts}
    L.ora_set_num_threads(all_cores)
    med = {n: statistics.median(v) for n, v in runs.items()}
    best = max(med, key=med.get)
    return {"value": tp, "unit": "trajectories/s", "cores": all_cores, "kind": "port, trajectory-parallel", "host_cores": all_cores,
            "by_mode": {"trajectory_parallel": {"value": tp, "threads": all_cores, "trajectories": all_cores, "evaluations": int(res["total_forwards"]),
                                                "wall_s": dt_tp},
                        "intra_evaluation": {"value": med[best], "threads": best,
                                             "by_threads": {str(n): {"median": med[n], "repeats": [round(v, 5) for v in runs[n]]} for n in counts}}},
            "cpu_model": cpu_model(), "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
            "note": "kind = port: the plain-C restatement of the reference AS WRITTEN (dense [E,641] edge MLP, N x N-free but otherwise "
                    "unfactorised; 4 x 2 register tiles, no cache blocking), one single-threaded trajectory per host core (OpenMP over "
                    "independent trajectories, nested regions off).  A stated baseline, not a tuned CPU implementation and never the "
                    "target; the reference itself (PyTorch on CPU) cannot run on this box.  by_mode.intra_evaluation = r01-r05's form "
                    "(one evaluation spread over the threads: ~200 fork / join regions per evaluation, does not scale past ~32 threads).",
            "sample": f"{all_cores} independent trajectories x {n_eval} score evaluations each on {all_cores} threads after one untimed evaluation "
                      f"per trajectory, extrapolated to {num_steps + 1} evaluations per trajectory ({dt_tp:.1f} s of wall)"}


def replayed_counters(args):
    """Counters of the committed rocprofv3 --pmc run of this exact configuration (profiles/r06_traffic.json, tools/make_traffic_json.py;
    collected as MI355X_MICROARCH.md prescribes: separate passes, FETCH_SIZE x 2 + WRITE_SIZE).  PMC counters cannot be read from inside
    this process: they are REPLAYED, per launch TYPE of the message kernel (full / ligand-only), and weighted here by the launch mix this
    run measures itself; absent for any other configuration."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json"):      # the newest committed counter run of this configuration
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        c = t["config"]
        if (c["R"], c["L"], c["batch"], c["precision"], bool(c.get("layer0_table"))) == (args.R, args.L, args.batch, args.precision, not args.no_l0_table) \
                and "full" in t.get("edge", {}):
            return t, "replayed profiles/" + name
    return None, None


def valu_issue(ctr, n_full, n_lig, rows_full, rows_lig, edge_ms):
    """VALU-issue floor of the message kernel over the launches of this run: the kernel is bound by what the
    vector ALU can ISSUE, not by the matrix pipe it is priced against in `roofline.frac`.  Per launch type: T = 4 transcendentals
    per edge and channel (two SiLUs: v_exp + v_rcp each) / 64 lanes - from the algebra; the other VALU instructions = SQ_INSTS_VALU
    of the committed PMC pass - T - MFMAs, split packed / plain in the static proportion of the kernel's ISA
    (profiles/r06_valu_mix.json, tools/valu_mix.py); each class priced at its measured issue cost per wave64 instruction and SIMD
    at two waves per SIMD (tools/ubench).  frac = floor time / LIVE launch time (HIP events of this run)."""
    mix = None
    for name in ("r06_valu_mix.json", "r05_valu_mix.json"):      # static ISA mix of the shipped kernel (tools/valu_mix.py)
        try:
            mix = json.load(open(os.path.join(ROOT, "profiles", name)))
            break
        except OSError:
            continue
    if mix is None:
        return None
    k, c = mix["k_edge_msg<1,1,0>"], mix["issue_cycles_per_wave64_instruction_at_2_waves_per_simd"]
    simds, clock = 256 * 4, c["clock_GHz"] * 1e9
    pk_share = k["packed_share_of_non_transcendental"]
    pk_cost = (k["packed_f32"] * c["packed_f32"] + k["packed_16"] * c["packed_16"]) / max(k["packed_f32"] + k["packed_16"], 1)
    floor_s, detail = 0.0, {}
    for name, n, rows, key in (("full", n_full, rows_full, "full"), ("ligand_only", n_lig, rows_lig, "lig_only")):
        if not n or key not in ctr["edge"]:
            continue
        insts = ctr["edge"][key]["insts_valu"]                      # wave64 VALU instructions per launch (incl. MFMA issues)
        T = rows * H * 4 / 64.0
        mfma = rows / 32.0 * 136                                    # 8 chunks x 16 + 8 bias / epilogue MFMAs per 32-row tile
        rest = max(insts - T - mfma, 0.0)
        cyc = T * c["transcendental"] + rest * (pk_share * pk_cost + (1 - pk_share) * c["plain"])
        detail[name] = {"launches": int(n), "valu_instructions": insts, "transcendental": T, "mfma": mfma, "other": rest,
                        "floor_ms_per_launch": cyc / simds / clock * 1e3}
        floor_s += n * cyc / simds / clock
    return {"frac": floor_s / (edge_ms * 1e-3) if edge_ms > 0 else None, "floor_ms_total": floor_s * 1e3, "measured_ms_total": edge_ms,
            "by_launch_type": detail, "instruction_mix": {kk: k[kk] for kk in ("transcendental", "packed_f32", "packed_16", "plain", "mfma")},
            "issue_cycles": {kk: c[kk] for kk in ("transcendental", "packed_f32", "packed_16", "plain")},
            "source": "instruction counts: replayed SQ_INSTS_VALU (the committed counter run named in traffic_source) + algebra; class split: static ISA mix "
                      "(profiles/r06_valu_mix.json); issue costs: tools/ubench on MI355X; time: live HIP events"}


def c5_line(engine, model, pk, num_steps):
    """Secondary record: BASELINE config 5 (1000+1000 residues, batch 32): one timed dfm_sample call after one warm-up call."""
    from dfmdock_amd.synthetic import make_complex
    import torch
    cx5 = make_complex(1000, 1000, seed=1)
    g5 = engine.Complex(model, cx5["rec_x"], cx5["lig_x"], cx5["rec_pos"], cx5["lig_pos"])
    g5.sample(B=32, num_steps=2, seed=3, **pk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g5.sample(B=32, num_steps=num_steps, seed=4, profile=True, **pk)
    dt = time.perf_counter() - t0
    p = g5.profile()
    g5.close()
    fl = p["edge_rows"] / K_DEG * FLOP_PER_NODE_LAYER
    ach = fl / (p["edge_kernel_ms"] * 1e-3) / 1e12 if p["edge_kernel_ms"] > 0 else None
    return {"workload": f"C5: synthetic 1000+1000-residue complex, batch=32 trajectories, {num_steps} steps", "value": 32 / dt,
            "unit": "trajectories/s", "ms_per_step": dt * 1e3, "steps": 1,
            "roofline": {"bound": "valu", "kernel": "k_edge_msg<1,1,0>", "achieved": ach, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_MFMA16_TFLOPS if ach else None,
                         "avg_launch_ms": p["edge_kernel_ms"] / max(p["edge_kernel_launches"], 1), "launches": int(p["edge_kernel_launches"]),
                         "share_of_call": p["edge_kernel_ms"] / (dt * 1e3)},
            "layer0_table": {"evaluations": int(p["l0_evals"]), "edge_model_fraction": p["l0_miss_rows"] / max(p["l0_edges"], 1),
                             "rows_launch_ms": p["l0_rows_ms"] / max(p["l0_evals"], 1), "gather_launch_ms": p["l0_gather_ms"] / max(p["l0_evals"], 1)},
            "whole_path_reference_equivalent_tflops": FLOP_PER_NODE_EVAL * 2000 * (num_steps + 1) * 32 / dt / 1e12}


DB5_SIZES = [(223, 172), (368, 327), (242, 101), (311, 145), (470, 105), (426, 200), (432, 129), (238, 91), (102, 95), (223, 129),
             (195, 125), (269, 161), (170, 207), (355, 75), (275, 107), (275, 64), (574, 54), (263, 141), (120, 120), (420, 115),
             (127, 246), (117, 471), (427, 65), (87, 127)]      # the 24 DB5 test complexes present in the reference


def c4_complexes():
    """The C4 set: the 24 DB5 test complexes with the reference loader's node features (src/datasets/ppi_dataset.py:249-265: ESM-2 block || one-hot) when
    the committed fixtures are there - backbones + sequences tests/golden/db5_backbones.npz, ESM-2 blocks as fp16 (esm_<id>.npz, cx_7CEI.npz) or int8 +
    per-residue scale (esm_db5_q8.npz; tests/golden/make_golden_r05.py / _r06.py wrote them from the reference's data/db5_test) -, else 24 synthetic
    complexes of the same sizes.  Returns (complexes, label)."""
    from dfmdock_amd.synthetic import make_complex, seq_to_onehot
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    try:
        bb = np.load(os.path.join(gdir, "db5_backbones.npz"))
        q8 = np.load(os.path.join(gdir, "esm_db5_q8.npz"))
        cxs = []
        for cid in [str(x) for x in bb["ids"]]:
            if cid + "_q" in q8.files:
                x = q8[cid + "_q"].astype(np.float32) * q8[cid + "_s"].astype(np.float32)[:, None]
                R = len(str(bb[f"{cid}_rec_seq"]))
                esm = {"rec": x[:R], "lig": x[R:]}
            else:
                e = np.load(os.path.join(gdir, "cx_7CEI.npz" if cid == "7CEI" else f"esm_{cid}.npz"))
                esm = {"rec": e["rec_esm16"].astype(np.float32), "lig": e["lig_esm16"].astype(np.float32)}
            c = {"id": cid}
            for side in ("rec", "lig"):
                seq = str(bb[f"{cid}_{side}_seq"])
                c[side + "_x"] = np.concatenate([esm[side], seq_to_onehot(seq)], 1)
                c[side + "_pos"] = bb[f"{cid}_{side}_pos"].astype(np.float32)
                c[side + "_seq"] = seq
            cxs.append(c)
        assert len(cxs) == 24
        return cxs, "the 24 DB5 test complexes (reference backbones, ESM-2 node features: 4 fp16 blocks, 20 int8-quantised; N = 197..695)"
    except (OSError, KeyError, AssertionError):
        cxs = []
        for k, (R, L) in enumerate(DB5_SIZES):
            c = make_complex(R, L, seed=300 + k)
            c["id"] = f"S{k:02d}_{R}_{L}"
            cxs.append(c)
        return cxs, "24 complexes with the DB5 test set's sizes (N = 197..695, synthetic chains and features: tests/golden fixtures not found)"


def c4_line(engine, model, precision, num_steps):
    """Secondary record: BASELINE config 4 on ONE GPU - the 24 DB5 test complexes (c4_complexes: committed backbones + ESM-2 feature
    blocks) x 40 trajectories through driver.run_set: handle creation, self-check, sampling, 40 x compute_metrics
    and the CSV inside the clock.  Serial driver (the reference's loop shape, src/inference_mlsb.py:415-439) and the pipelined one."""
    import tempfile
    from dfmdock_amd import driver
    from dfmdock_amd.synthetic import make_complex
    cxs, what = c4_complexes()
    tmp = tempfile.mkdtemp(prefix="dfm_c4_")
    quiet = lambda m: None
    out = {"workload": f"C4 on one GPU: {what} x 40 "
                       f"trajectories x {num_steps} steps through driver.run_set; clock includes handle creation, self-check, metrics, CSV",
           "unit": "trajectories/s"}
    driver.run_set(model, cxs[:3], num_samples=40, num_steps=num_steps, seed=0, precision=precision,
                   out_csv=os.path.join(tmp, "warm.csv"), log=quiet)
    res = {}
    for name, kw in (("serial", dict(overlap=False)), ("pipelined", dict(overlap=True, samplers=1)), ("pipelined_2_samplers", dict(overlap=True, samplers=2))):
        tim = []
        t0 = time.perf_counter()
        rows, _ = driver.run_set(model, cxs, num_samples=40, num_steps=num_steps, seed=0, precision=precision,
                                 out_csv=os.path.join(tmp, name + ".csv"), timings_out=tim, log=quiet, **kw)
        dt = time.perf_counter() - t0
        res[name] = {"value": len(rows) / dt, "wall_s": dt, "trajectories": len(rows),
                     "stage_sums_s": {k: sum(t[k] for t in tim) / 1e3 for k in ("prepare", "sample", "post")}}
    base = open(os.path.join(tmp, "serial.csv"), "rb").read()
    out["csv_identical_to_serial"] = all(open(os.path.join(tmp, n + ".csv"), "rb").read() == base for n in res)
    best = max(res, key=lambda n: res[n]["value"])
    out.update(value=res[best]["value"], wall_s=res[best]["wall_s"], driver=best, by_driver=res)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def c4_sharded_line(model, precision, num_steps, rank, world, backend):
    """BASELINE config 4 as it is DEFINED: the 24 DB5-sized complexes x 40 trajectories sharded over the ranks of this job (one
    process per GPU) with one gather of the ranked energy records (the loop it replaces: src/inference_mlsb.py:415-439 over
    src/inference_base.py:561-580).  driver.run_set assigns whole complexes longest-first on the measured cost model
    (distributed.complex_cost; 3 per rank at 8 ranks).  wall_s = max over ranks of the time between two barriers around run_set
    (handle creation, self-check, sampling, metrics, record + row gathers, CSV on rank 0); per_rank_makespan_s = each rank's own
    time to the end of its share.  A record, not a scaling claim."""
    import tempfile
    from dfmdock_amd import distributed as D
    from dfmdock_amd import driver
    from dfmdock_amd.synthetic import make_complex
    import torch
    cxs, what = c4_complexes()
    tmp = tempfile.mkdtemp(prefix=f"dfm_c4s_{rank}_")
    quiet = lambda m: None
    driver.run_set(model, cxs[:2], num_samples=max(8, world), num_steps=2, seed=0, precision=precision, log=quiet, canary=False)      # warm-up (collectives included)
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tim, can = [], {}
    rows, ranked = driver.run_set(model, cxs, num_samples=40, num_steps=num_steps, seed=0, precision=precision,
                                  out_csv=os.path.join(tmp, "c4.csv"), timings_out=tim, log=quiet, canary_out=can)
    mine = time.perf_counter() - t0
    D.barrier(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_rank = D.allgather_scalars([mine, float(len(tim)), float(len(rows))])
    n_rec = int(sum(len(v) for v in ranked.values()))
    ids = {(int(cid), int(r[1])) for cid, v in ranked.items() for r in v}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    costs = [D.complex_cost(R + L, 40) for R, L in DB5_SIZES]
    assign = D.assign_work(costs, world)
    return {"workload": f"C4: {what} x 40 trajectories x {num_steps} steps, "
                        f"sharded over {world} ranks by driver.run_set, one gather of the ranked energy records",
            "wall_s": wall, "value": n_rec / wall, "unit": "trajectories/s", "per_rank_makespan_s": [float(x) for x in per_rank[:, 0]],
            "complexes_per_rank": [int(x) for x in per_rank[:, 1]], "rows_per_rank": [int(x) for x in per_rank[:, 2]],
            "predicted_makespan_ms": D.makespan(costs, assign), "backend": backend, "records_in_gather": n_rec,
            "distinct_record_ids": len(ids), "complexes_in_gather": len(ranked), "canary": can or None}


# reference-equivalent work of one score evaluation per residue (the reference's dense formulation, all six layers in full)
FLOP_PER_NODE_EVAL = 59.2e6


def workload_label(a):
    shape = f"synthetic {a.R}+{a.L}-residue complex, batch={a.batch} trajectories/GPU, {a.num_steps} steps ({a.num_steps + 1} score evaluations + energy head)"
    if (a.R, a.L, a.batch, a.num_steps) == (300, 300, 256, 40):
        return "C3: " + shape
    if (a.R, a.L, a.batch, a.num_steps) == (1000, 1000, 32, 40):
        return "C5: " + shape
    return "custom: " + shape


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (this same command line, one process per GPU) with a
    loopback rendezvous, wait for them, pass rank 0's JSON line through.  A rank that dies takes the others down."""
    port = free_port()
    gather_dir = tempfile.mkdtemp(prefix="dfm_gather_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DFM_GATHER_DIR=gather_dir, DFM_BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                code = p.poll()
                if code is not None:
                    procs.remove(p)
                    rc = rc or code
            time.sleep(0.05)
    finally:
        for p in procs:      # exact PIDs of the children started above
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        import shutil
        shutil.rmtree(gather_dir, ignore_errors=True)      # the directory made above for the file-gather fallback
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="trajectories in flight per GPU")
    ap.add_argument("--R", type=int, default=300)
    ap.add_argument("--L", type=int, default=300)
    ap.add_argument("--num-steps", type=int, default=40, help="diffusion steps per trajectory")
    ap.add_argument("--precision", choices=["mfma16", "bf16", "f16", "fp32"], default="mfma16",
                    help="mfma16: the 16-bit MFMA engine as shipped (DFM_F_MFMA16: fp16 operands, fp32 accumulation; dfm_config_string() "
                         "in the JSON line says what that is; reported as dtype f16); f16: the same with fp32 A_i; fp32: exact; "
                         "bf16: deprecated ALIAS of mfma16 kept for BASELINE's wording - the operands are fp16, not bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-l0-table", action="store_true", help="A/B: layer 0 evaluated edge by edge (DFM_F_NO_L0_TABLE)")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the secondary measurement of the fp32 engine (one more batched call)")
    ap.add_argument("--no-c5-line", action="store_true", help="skip the secondary C5 record (1000+1000 residues, batch 32: one timed call)")
    ap.add_argument("--no-c4-line", action="store_true", help="skip the secondary C4 record (24 DB5-sized complexes x 40 trajectories through driver.run_set, three drivers)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from dfmdock_amd import distributed as D
    rank, local_rank, world = D.dist_env()
    if world == 1:
        # pin the CPU baseline's OpenMP threads (must be in the environment before the oracle library starts its runtime);
        # only where the baseline is measured: N ranks pinned to the same places would share cores
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
    import torch
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob

    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev            # one process per GPU; the modulo only matters for dry runs with more ranks than GPUs
    torch.cuda.set_device(dev)
    grp = D.init(device_index=dev)     # nccl (= RCCL) -> gloo -> files, whatever works on every rank
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    engine.set_device(dev)

    blob = pack_blob(make_random_weights(0))
    cx = make_complex(args.R, args.L, seed=1)
    model = engine.Model(blob)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    pk = engine.precision_kwargs(args.precision)      # "bf16" -> "mfma16" with a note on stderr
    args.precision = engine.canonical_precision(args.precision)
    f16 = pk["f16"]
    mfma16 = pk["mfma16"] or f16
    B = args.batch

    def one_step(it, profile=False):
        r = gx.sample(B=B, num_steps=args.num_steps, seed=1000 * (rank + 1) + it, profile=profile, l0_table=not args.no_l0_table, **pk)
        rec = D.make_records(rank, np.arange(rank * B, (rank + 1) * B), r)      # record id = the rank that sampled it
        allrec = D.gather_records(rec)            # the only collective: ranked energies (RCCL all_gather)
        return r, allrec

    def barrier():
        D.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        one_step(it)
    barrier()
    t0 = time.perf_counter()
    edge_ms, edge_launches, edge_rows, lig_launches, lig_ms = 0.0, 0, 0, 0, 0.0
    clk_cycles = clk_ticks = 0.0      # the message kernel's own clock stamps (dfm_profile::edge_shader_cycles / edge_ref_ticks)
    l0 = dict(l0_evals=0, l0_edges=0, l0_miss_rows=0, l0_rows_ms=0.0, l0_gather_ms=0.0)
    allrec = None
    for it in range(args.steps):
        _, allrec = one_step(args.warmup + it, profile=True)
        p = gx.profile()
        edge_ms += p["edge_kernel_ms"]
        edge_launches += p["edge_kernel_launches"]
        edge_rows += p["edge_rows"]
        lig_launches += p["edge_lig_launches"]
        lig_ms += p["edge_lig_ms"]
        clk_cycles += p.get("edge_shader_cycles") or 0.0
        clk_ticks += p.get("edge_ref_ticks") or 0.0
        for k in l0:
            l0[k] += p[k]
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = D.allgather_scalars([elapsed, float(dev)])      # max over ranks of the time; which device every rank ran on
    elapsed = float(per_rank[:, 0].max())
    headline = (args.R, args.L, args.batch, args.num_steps) == (300, 300, 256, 40)
    c4s = None
    if world > 1 and headline and not args.no_c4_line:      # every rank takes part (outside the timed C3 region)
        c4s = c4_sharded_line(model, args.precision, args.num_steps, rank, world, grp.backend)
    fp32_line = None
    if rank == 0 and world == 1 and mfma16 and not args.no_fp32_line:
        # secondary record: the fp32 engine - the reference's own arithmetic - on the same workload, one batched call
        gx.sample(B=B, num_steps=2, seed=7, l0_table=not args.no_l0_table)      # (builds the fp32 engine's own layer-0 table)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gx.sample(B=B, num_steps=args.num_steps, seed=8, profile=True, l0_table=not args.no_l0_table)
        dt32 = time.perf_counter() - t1
        p32 = gx.profile()
        fl32 = p32["edge_rows"] / K_DEG * FLOP_PER_NODE_LAYER
        fp32_line = {"value": B / dt32, "unit": "trajectories/s", "ms_per_step": dt32 * 1e3, "dtype": "f32", "steps": 1,
                     "kernel": "k_edge_f32m (exact fp32 on v_mfma_f32_32x32x2_f32, edge model + coordinate MLP fused)",
                     "avg_launch_ms": p32["edge_kernel_ms"] / max(p32["edge_kernel_launches"], 1),
                     "achieved_tflops": fl32 / (p32["edge_kernel_ms"] * 1e-3) / 1e12 if p32["edge_kernel_ms"] > 0 else None,
                     "peak_tflops": PEAK_F32_TFLOPS,
                     "frac": (fl32 / (p32["edge_kernel_ms"] * 1e-3) / 1e12 / PEAK_F32_TFLOPS) if p32["edge_kernel_ms"] > 0 else None,
                     "layer0_table": not args.no_l0_table,
                     "note": "message FLOPs only in `achieved` (the last layer's launches also run the coordinate MLP of the ligand nodes; layer 0 runs behind the fp32 engine's own message table and is not in `achieved`); 1e-4 parity gates"}

    if rank == 0:
        total_traj = world * B * args.steps
        ranks_seen = sorted(set(allrec[:, 0].astype(int).tolist())) if allrec is not None else []
        assert allrec is None or allrec.shape[0] == world * B, f"record gather returned {allrec.shape[0]} records, expected {world * B}"
        # every launch of the message kernel inside the timed region is bracketed by HIP events on the engine's stream:
        # achieved = (edge rows processed / K) * FLOP per node and layer / total kernel time
        avg_launch_s = edge_ms / max(edge_launches, 1) * 1e-3
        flop_total = edge_rows / K_DEG * FLOP_PER_NODE_LAYER
        flop_per_launch = flop_total / max(edge_launches, 1)
        achieved = flop_total / (edge_ms * 1e-3) / 1e12 if edge_ms > 0 else 0.0
        peak = PEAK_MFMA16_TFLOPS if mfma16 else PEAK_F32_TFLOPS
        ctr, ctr_src = replayed_counters(args)
        sclk_mhz = 100.0 * clk_cycles / clk_ticks if clk_ticks > 0 else None
        n_full, n_lig = edge_launches - lig_launches, lig_launches
        traffic = mfma_busy = valu_busy = None
        if ctr:      # weight the two launch types by THIS run's mix (bytes: per launch; busy fractions: by active cycles)
            ef, el = ctr["edge"]["full"], ctr["edge"].get("lig_only", ctr["edge"]["full"])
            traffic = (n_full * ef["hbm_bytes"] + n_lig * el["hbm_bytes"]) / max(edge_launches, 1)
            wf, wl = n_full * ef["active_cycles"], n_lig * el["active_cycles"]
            mfma_busy = (wf * ef["mfma_busy"] + wl * el["mfma_busy"]) / max(wf + wl, 1)
            valu_busy = (wf * ef["valu_busy"] + wl * el["valu_busy"]) / max(wf + wl, 1)
        out = {
            "metric": "docking trajectories/sec (N_res~300+300, 40 steps)",
            "value": total_traj / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if mfma16 else "f32",
            "data": "synthetic",
            "backend": grp.backend,
            "backend_fallback": grp.fallback_reason,
            "ranks_in_gather": len(ranks_seen) if world > 1 else 1,
            "records_in_gather": int(allrec.shape[0]) if allrec is not None else 0,
            "distinct_record_ids": len({(int(a), int(b)) for a, b in allrec[:, 0:2]}) if allrec is not None else 0,
            "distinct_devices": len(set(per_rank[:, 1].astype(int).tolist())),
            "config": {"workload": workload_label(args),
                       "trajectories_per_gpu": B, "num_steps": args.num_steps, "parallelism": f"traj-shard x{world}",
                       "weights": "random-init (seeded generator; trained checkpoint not in the reference)",
                       "work": "41 score evaluations per trajectory through dfm_sample: the 40 step evaluations return f and the two scores "
                               "(all the sampler reads, inference_base.py:425-448), so their last layer runs over the ligand nodes only and "
                               "without its node model - bitwise the same f / scores as the full evaluation (tests/test_gpu_variants.py); "
                               "the final evaluation runs in full with the energy head"
                               + ("; layer 0 through the per-complex message table: the gated message of an intra-chain edge is a "
                                  "function of the residue pair alone there (pose-independent embedding, rigid chains), so it is gathered "
                                  "from a table built once per complex; inter-chain edges and bin mismatches go through the edge model "
                                  "(tests/test_gpu_l0_table.py)" if l0["l0_evals"] else ""),
                       "precision": (("16-bit MFMA engine" + (" with fp32 A_i (DFM_F_F16)" if f16 else "") + ": ") if mfma16 else "fp32 engine; library plan: ")
                                    + engine.config_string(),
                       "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DFM_") and k not in
                                        ("DFM_GATHER_DIR", "DFM_BENCH_SELF_SPAWNED")}},
            "roofline": {"bound": "valu" if mfma16 else "mfma", "kernel": "k_edge_msg<1,%d> (fp16 operands, all six layers)" % (0 if f16 else 1) if mfma16 else "k_edge_f32",
                         "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "sclk_mhz": sclk_mhz, "peak_at_sclk": peak * sclk_mhz / PEAK_SCLK_MHZ if sclk_mhz else None,
                         "frac_at_sclk": achieved / (peak * sclk_mhz / PEAK_SCLK_MHZ) if sclk_mhz else None,
                         "sclk_note": "sclk_mhz = the shader clock the message kernel ran at in THIS run: its first wave reads s_memtime (shader cycles) and "
                                      "s_memrealtime (100 MHz) at start and exit, summed over the timed launches. `peak` assumes %d MHz; the chip's power "
                                      "management holds this kernel lower (the MFMA's energy, not its issue rate, is what the clock pays for: "
                                      "profiles/r06_clock.txt), so frac_at_sclk is the fraction of what the matrix pipe can deliver at the clock it was given" % PEAK_SCLK_MHZ,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": int(edge_launches),
                         "flop_per_launch": flop_per_launch, "traffic": traffic, "traffic_source": ctr_src,
                         "traffic_gbps": (traffic / avg_launch_s / 1e9) if traffic else None,
                         "mfma_busy": mfma_busy, "valu_busy": valu_busy,
                         "valu_issue": valu_issue(ctr, n_full, n_lig, (edge_rows - lig_launches * B * args.L * K_DEG) / max(n_full, 1),
                                                  B * args.L * K_DEG, edge_ms) if (ctr and mfma16) else None,
                         "launch_mix": {"full": int(n_full), "ligand_only": int(n_lig),
                                        "avg_full_ms": (edge_ms - lig_ms) / max(n_full, 1), "avg_ligand_only_ms": lig_ms / max(n_lig, 1),
                                        "traffic_full": ctr["edge"]["full"]["hbm_bytes"] if ctr else None,
                                        "traffic_ligand_only": ctr["edge"].get("lig_only", {}).get("hbm_bytes") if ctr else None},
                         "rows_per_launch": edge_rows / max(edge_launches, 1),
                         "algorithmic_bytes_per_launch": 8 * H * edge_rows / K_DEG / max(edge_launches, 1),
                         "note": "bound = valu: the kernel is limited by VALU ISSUE (4 transcendentals + ~11 plain operations per edge and "
                                 "channel), see valu_issue.frac; achieved / peak / frac price the same launches against the dense 16-bit "
                                 "MFMA peak as BASELINE's metric asks. "
                                 "achieved = B*N*(2*K*H*H + 2*K*H) FLOP / launch time from HIP events on the engine's stream, live; "
                                 "traffic = HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE), mfma_busy / valu_busy = pipe-busy fractions "
                                 "(SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU x 4 per SIMD over GRBM_GUI_ACTIVE): not measured in this "
                                 "run, see traffic_source; algorithmic bytes per launch = 8*N*H per trajectory (SURVEY 8d)"},
            "best_energy": float(allrec[:, 2].min()),
        }
        # whole path against the matrix peak in the reference's own currency: the dense formulation's FLOPs for the evaluations done
        ref_flops = FLOP_PER_NODE_EVAL * (args.R + args.L) * (args.num_steps + 1) * total_traj
        out["whole_path"] = {"reference_equivalent_tflops": ref_flops / elapsed / 1e12, "peak_tflops": peak,
                             "frac": ref_flops / elapsed / 1e12 / peak,
                             "note": "59.2 MFLOP per residue and score evaluation (SURVEY 8d: the reference's dense formulation, six full layers) x "
                                     "residues x evaluations x trajectories / wall time; the engine executes less than that (layer-0 message "
                                     "table, ligand-only last layer, one-hot -> gather) and three split-bf16 terms in the node GEMMs"}
        if ctr:      # replayed per-kernel table (same PMC run as `roofline.traffic`): what bounds each kernel and how close it gets
            out["kernels"] = {"source": ctr_src, "table": [
                {"kernel": k, "avg_us": d.get("avg_us"), "percent_of_gpu_time": d.get("percent"), "bound": d.get("bound"), "frac": d.get("frac"),
                 "hbm_tbps": d.get("hbm_tbps"), "hbm_MB": None if d.get("hbm_bytes") is None else d["hbm_bytes"] / 1e6,
                 "mfma_busy": d.get("mfma_busy"), "valu_busy": d.get("valu_busy"), "issue_stalled": d.get("wait_frac")}
                for k, d in sorted(ctr["kernels"].items(), key=lambda kv: -(kv[1].get("percent") or 0))],
                "note": "bound hbm: frac = HBM TB/s / 8; valu: VALU-pipe busy (issue-bound kernel); l2 / mfma rows: see DESIGN section 5"}
        if fp32_line:
            out["fp32_engine"] = fp32_line
        if l0["l0_evals"]:      # layer 0 behind the message table: its launches are not part of `roofline` (HIP events, live)
            n = l0["l0_evals"]
            out["layer0_table"] = {"evaluations": int(n), "edge_model_fraction": l0["l0_miss_rows"] / max(l0["l0_edges"], 1),
                                   "rows_launch_ms": l0["l0_rows_ms"] / n, "gather_launch_ms": l0["l0_gather_ms"] / n,
                                   "gather_bytes_per_launch": 512 * l0["l0_edges"] / n,
                                   "gather_tbps": 512 * l0["l0_edges"] / max(l0["l0_gather_ms"], 1e-9) / 1e9,
                                   "note": "per evaluation: k_edge_msg<1,1,1> over the row list (inter-chain edges + bin mismatches) and "
                                           "k_l0_gather (K rows of 512 B per node, out of L2 / the Infinity Cache) replace one full message launch"}
        if world == 1 and mfma16 and headline and not args.no_c5_line:
            out["c5"] = c5_line(engine, model, pk, args.num_steps)
        if world == 1 and headline and not args.no_c4_line:
            out["c4"] = c4_line(engine, model, args.precision, args.num_steps)
        if world > 1 and headline and not args.no_c4_line:
            out["c4_sharded"] = c4s      # measured by every rank below the timed C3 region (collectives inside), reported by rank 0
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(blob, cx, args.num_steps)
        print(json.dumps(out), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()

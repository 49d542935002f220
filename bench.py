#!/usr/bin/env python3
"""bench.py - docking trajectories/second of the sampling hot path on MI355X.

One "step" = one batched dfm_sample call: B independent trajectories x 40 Euler-Maruyama steps
(41 score evaluations each + the final energy head) on the synthetic 300+300-residue complex of
BASELINE.json configs[2] (C3).  Inputs (weights, node features, backbone) are resident in HBM before
the timed region; one call returns only ~40 B per trajectory.  With N > 1 GPUs every rank samples its own
B trajectories (weak scaling, no data-path collective) and one tiny all_gather of energy-ranked
records closes the step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of the dominant kernel (per-edge message kernel), SURVEY.md 8(d):
# per node and layer 2*K*H*H (edge_mlp.2) + 2*K*H (attention gate)
H, K_DEG = 256, 60
FLOP_PER_NODE_LAYER = 2 * K_DEG * H * H + 2 * K_DEG * H
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3       # fp32 vector / f32-input MFMA


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(blob, cx, num_steps, repeats=3):
    """Oracle (plain-C port of the reference as written, OpenMP) timed on the host cores of this box on a bounded sample:
    a few score evaluations of ONE trajectory of the same complex, extrapolated to the 41 evaluations of a trajectory.
    Threads are pinned (OMP_PROC_BIND / OMP_PLACES, set in main() before the OpenMP runtime starts), one untimed
    evaluation warms the pages, and the reported value is the MEDIAN of `repeats` samples; the same is measured with 8
    threads."""
    import statistics
    from oracle import oracle as ora
    L = ora.lib()
    o = ora.Oracle(blob, cx)
    all_cores = int(L.ora_num_threads())

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

    full = sample(all_cores, 6)
    eight = sample(min(8, all_cores), 2)
    L.ora_set_num_threads(all_cores)
    return {"value": statistics.median(full), "unit": "trajectories/s", "cores": all_cores, "kind": "port",
            "repeats": [round(v, 5) for v in full], "value_8_threads": statistics.median(eight),
            "repeats_8_threads": [round(v, 5) for v in eight], "cpu_model": cpu_model(),
            "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
            "sample": f"median of {repeats} x 6 (all {all_cores} threads) / {repeats} x 2 (8 threads) score evaluations of 1 trajectory "
                      f"of the same complex, after one warm-up evaluation, extrapolated to {num_steps + 1} evaluations per trajectory"}


def replayed_traffic(args):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the number is
    REPLAYED from the committed rocprofv3 --pmc run of this exact configuration (profiles/*_traffic.json, collected as
    MI355X_MICROARCH.md prescribes: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2) and is absent otherwise."""
    for name in ("r02_traffic.json", "r01_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        c = t["config"]
        if (c["R"], c["L"], c["batch"], c["precision"]) == (args.R, args.L, args.batch, args.precision):
            return t["traffic_bytes_per_launch"], "replayed profiles/" + name
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="trajectories in flight per GPU")
    ap.add_argument("--R", type=int, default=300)
    ap.add_argument("--L", type=int, default=300)
    ap.add_argument("--num-steps", type=int, default=40, help="diffusion steps per trajectory")
    ap.add_argument("--precision", choices=["bf16", "f16", "fp32"], default="bf16",
                    help="bf16 / f16: 16-bit MFMA operands for the per-edge contractions (fp32 accumulate); fp32: exact")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # pin the CPU baseline's OpenMP threads (must be in the environment before the oracle library starts its runtime)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import torch
    import torch.distributed as dist
    from dfmdock_amd import distributed as D
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    from dfmdock_amd.weights import make_random_weights, pack_blob

    rank, local_rank, world = D.dist_env()
    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev            # one process per GPU; the modulo only matters for single-GPU dry runs
    backend = os.environ.get("DFM_DIST_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" for dry runs on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    engine.set_device(dev)
    torch.cuda.set_device(dev)

    blob = pack_blob(make_random_weights(0))
    cx = make_complex(args.R, args.L, seed=1)
    model = engine.Model(blob)
    gx = engine.Complex(model, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    bf16 = args.precision == "bf16"
    f16 = args.precision == "f16"
    mfma16 = bf16 or f16
    B = args.batch

    def one_step(it, profile=False):
        r = gx.sample(B=B, num_steps=args.num_steps, seed=1000 * (rank + 1) + it, bf16=bf16, f16=f16, profile=profile)
        rec = D.make_records(0, np.arange(rank * B, (rank + 1) * B), r)
        allrec = D.gather_records(rec)            # the only collective: ranked energies (RCCL all_gather)
        return r, allrec

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        one_step(it)
    barrier()
    t0 = time.perf_counter()
    edge_ms, edge_launches, edge_rows = 0.0, 0, 0
    for it in range(args.steps):
        _, allrec = one_step(args.warmup + it, profile=True)
        p = gx.profile()
        edge_ms += p["edge_kernel_ms"]
        edge_launches += p["edge_kernel_launches"]
        edge_rows += p["edge_rows"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_traj = world * B * args.steps
        N = args.R + args.L
        # every launch of the message kernel inside the timed region is bracketed by HIP events on the engine's stream; the last
        # layer runs as several smaller launches (trajectory chunks), so work and time are summed over launches:
        # achieved = (edge rows processed / K) * FLOP per node and layer / total kernel time
        avg_launch_s = edge_ms / max(edge_launches, 1) * 1e-3
        flop_total = edge_rows / K_DEG * FLOP_PER_NODE_LAYER
        flop_per_launch = flop_total / max(edge_launches, 1)
        achieved = flop_total / (edge_ms * 1e-3) / 1e12 if edge_ms > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if mfma16 else PEAK_F32_TFLOPS
        traffic, traffic_src = replayed_traffic(args)
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
            "dtype": args.precision if mfma16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"C3: synthetic {args.R}+{args.L}-residue complex, batch={B} trajectories/GPU, "
                                   f"{args.num_steps} steps ({args.num_steps + 1} score evaluations + energy head)",
                       "trajectories_per_gpu": B, "num_steps": args.num_steps, "parallelism": f"traj-shard x{world}",
                       "weights": "random-init (seeded generator; trained checkpoint not in the reference)",
                       "precision": ("bf16 MFMA operands for the per-edge contractions of layers 0-4, fp16 operands (same rate) for the last "
                                     "layer and the coordinate head, fp32 accumulate; split-bf16 node GEMMs; fp32 geometry / heads / SDE step")
                                    if bf16 else args.precision},
            "roofline": {"bound": "mfma", "kernel": ("k_edge_msg<%s>" % ("1" if f16 else "0|1: bf16 operands, last layer fp16")) if mfma16 else "k_edge_f32", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": int(edge_launches),
                         "flop_per_launch": flop_per_launch, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_gbps": (traffic / avg_launch_s / 1e9) if traffic else None,
                         "rows_per_launch": edge_rows / max(edge_launches, 1),
                         "algorithmic_bytes_per_launch": 8 * H * edge_rows / K_DEG / max(edge_launches, 1),
                         "note": "achieved = B*N*(2*K*H*H + 2*K*H) FLOP / launch time from HIP events on the engine's stream, live; "
                                 "traffic = HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run: see "
                                 "traffic_source; algorithmic bytes per launch = 8*N*H per trajectory (SURVEY 8d)"},
            "best_energy": float(D.rank_by_energy(allrec)[0][0, 2]),
        }
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(blob, cx, args.num_steps)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

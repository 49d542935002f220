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


def cpu_baseline(blob, cx, num_steps, n_forwards=8):
    """Oracle (plain-C port of the reference, OpenMP) timed on a bounded sample: n_forwards score
    evaluations of one trajectory; extrapolated to 41 evaluations per trajectory."""
    from oracle import oracle as ora
    o = ora.Oracle(blob, cx)
    cores = ora.lib().ora_num_threads()
    t0 = time.perf_counter()
    r = o.sample(num_steps=num_steps, max_forwards=n_forwards, seed=1)
    dt = time.perf_counter() - t0
    per_fwd = dt / max(r["forwards"], 1)
    return {"value": 1.0 / (per_fwd * (num_steps + 1)), "unit": "trajectories/s", "cores": int(cores), "kind": "port",
            "sample": f"{r['forwards']} of {num_steps + 1} score evaluations of 1 trajectory on the same 300+300 complex "
                      f"({dt:.1f} s), extrapolated"}


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC run (profiles/r01_traffic.json);
    PMC counters cannot be read from inside this process, so the number is attached only for the exact configuration
    it was measured on."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        t = json.load(open(path))
    except OSError:
        return None
    c = t["config"]
    if (c["R"], c["L"], c["batch"], c["precision"]) == (args.R, args.L, args.batch, args.precision):
        return t["traffic_bytes_per_launch"]
    return None


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
    edge_ms, edge_launches = 0.0, 0
    for it in range(args.steps):
        _, allrec = one_step(args.warmup + it, profile=True)
        p = gx.profile()
        edge_ms += p["edge_kernel_ms"]
        edge_launches += p["edge_kernel_launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_traj = world * B * args.steps
        N = args.R + args.L
        avg_launch_s = edge_ms / max(edge_launches, 1) * 1e-3
        flop_per_launch = B * N * FLOP_PER_NODE_LAYER
        achieved = flop_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if mfma16 else PEAK_F32_TFLOPS
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
                       "weights": "random-init (seeded generator; trained checkpoint not in the reference)"},
            "roofline": {"bound": "mfma", "kernel": ("k_edge_bf16<0,%d>" % int(f16)) if mfma16 else "k_edge_f32", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": int(edge_launches),
                         "flop_per_launch": flop_per_launch, "traffic": measured_traffic(args),
                         "traffic_gbps": (measured_traffic(args) / avg_launch_s / 1e9) if measured_traffic(args) else None,
                         "valu_floor_ms": 1.8 if (args.R, args.L, args.batch) == (300, 300, 256) else None,
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json); traffic_gbps = HBM GB/s "
                                         "of that kernel (peak ~8000); valu_floor_ms = the VALU-issue floor of a launch (DESIGN.md 5, "
                                         "profiles/r01_ubench_valu_rate.txt) - the bound that applies to this kernel"},
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

"""From the counter passes of tools/final_profile.sh: the JSON bench.py replays into `roofline` and `kernels` (profiles/r06_traffic.json).

  edge.full / edge.lig_only   HBM bytes and pipe-busy fractions of ONE launch of the message kernel k_edge_msg<1,1,0>, separately for a
                              full launch (all nodes) and a last-layer launch over the ligand nodes only - bench.py weights them by the
                              launch mix it measures itself (one replayed number divided by two different mixes)
  kernels.<name>              per kernel: launches seen, active cycles, HBM bytes, busy fractions, its bound and the fraction reached

Counter conventions (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950 for wide coalesced reads; SQ_* summed
over the chip (1024 SIMDs, 256 CUs); GRBM_GUI_ACTIVE counts per shader engine: / 8 = active cycles of the launch.  Dispatches of the
message kernel are classified by position: the passes run bench.py --num-steps 3 with the layer-0 table, i.e. three step evaluations of
[layers 1-4 full, layer 5 ligand-only] and a final evaluation of five full launches.

    python tools/make_traffic_json.py <dir with pmc_all/ and bench_prof_kernel_stats.csv>
"""
import collections, csv, glob, json, os, sys

out = sys.argv[1]
STEPS = int(os.environ.get("PMC_NUM_STEPS", "3"))
HBM_PEAK, HBM_ACH, MFMA_PEAK = 8.0e12, 6.3e12, 2.5e15
N, H, B, K, L = 600, 256, 256, 60, 300


def per_pass_means(d, keep):
    """{kernel: {counter: mean over the kept dispatches}}; dispatch ids differ between passes, so dispatches are ranked per pass"""
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(out, d, "*counter_collection.csv"))):
        rows = collections.defaultdict(lambda: collections.defaultdict(dict))
        for row in csv.DictReader(open(f)):
            rows[row.get("Kernel_Name", "")][int(row.get("Dispatch_Id", 0))][row["Counter_Name"]] = float(row["Counter_Value"])
        for k, disp in rows.items():
            ids = sorted(disp)
            for rank, i in enumerate(ids):
                if keep(k, rank, len(ids)):
                    for c, v in disp[i].items():
                        res[k][c].append(v)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in res.items()}


def derived(c):
    g = lambda n: c.get(n, (float("nan"), 0))[0]
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    d = {"launches": c.get("GRBM_GUI_ACTIVE", (0, 0))[1], "active_cycles": cyc,
         "hbm_bytes": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0, "fetch_KiB": g("FETCH_SIZE"), "write_KiB": g("WRITE_SIZE"),
         "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / cyc, "valu_busy": g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cyc,
         "lds_busy": g("SQ_LDS_IDX_ACTIVE") / 256.0 / cyc, "wait_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
         "parked_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
         "insts_valu": g("SQ_INSTS_VALU"), "l2_hit": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))}
    return {k: (None if isinstance(v, float) and v != v else v) for k, v in d.items()}


def is_msg(k):
    return "k_edge_msg<1, 1, 0>" in k


def lig_rank(rank, n):
    per = 5      # message launches of k_edge_msg<1,1,0> per evaluation behind the layer-0 table
    return rank < per * STEPS and rank % per == per - 1


full = per_pass_means("pmc_all", lambda k, r, n: is_msg(k) and not lig_rank(r, n))
lig = per_pass_means("pmc_all", lambda k, r, n: is_msg(k) and lig_rank(r, n))
others = per_pass_means("pmc_all", lambda k, r, n: not is_msg(k))

stats = {}
sf = os.path.join(out, "bench_prof_kernel_stats.csv")
if os.path.exists(sf):
    for r in csv.DictReader(open(sf)):
        stats[r["Name"]] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]), "percent": float(r["Percentage"])}


def stat_of(k):
    for name, v in stats.items():
        if name[:60] == k[:60] or name.startswith(k[:50]):
            return v
    return None


js = {"config": {"R": 300, "L": 300, "batch": B, "precision": "mfma16", "layer0_table": True, "num_steps_of_the_passes": STEPS},
      "source": "tools/final_profile.sh on MI355X: rocprofv3 --pmc, one counter set per run, kernel-filtered, no trace domains; FETCH_SIZE x 2 "
                "(gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE; busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs and SQ_ACTIVE_INST_VALU x 4 / 1024 "
                "over GRBM_GUI_ACTIVE / 8; durations / shares from rocprofv3 --kernel-trace --stats of the default bench command",
      "edge": {}, "kernels": {}}
for tag, m, nodes in (("full", full, N), ("lig_only", lig, L)):
    for k, c in m.items():
        d = derived(c)
        d["algorithmic_bytes"] = 8 * nodes * H * B      # 8 N H per trajectory and launch
        d["flop"] = B * nodes * (2 * K * H * H + 2 * K * H)
        js["edge"][tag] = d
        print(f"k_edge_msg<1,1,0> {tag:8s}: {d['launches']:3d} launches, HBM {d['hbm_bytes'] / 1e9:.3f} GB (FETCH {d['fetch_KiB']:.0f} KiB x 2 + WRITE {d['write_KiB']:.0f} KiB) = "
              f"{d['hbm_bytes'] / d['algorithmic_bytes']:.2f} x algorithmic {d['algorithmic_bytes'] / 1e6:.0f} MB; {d['active_cycles'] / 1e6:.2f} M cycles; MFMA busy {d['mfma_busy']:.3f} "
              f"VALU busy {d['valu_busy']:.3f} LDS {d['lds_busy']:.3f} issue-stalled {d['wait_frac']:.3f} parked {d['parked_frac']:.3f} L2 hit {d['l2_hit']:.3f}")

BOUNDS = {      # kernel -> (bound, how the fraction is formed)
    "k_gemm_split": "hbm", "k_edge_coord": "hbm", "k_l0_gather": "l2", "k_knn_sample": "valu", "k_edge_msg<1, 1, 1>": "mfma", "k_edge_feat": "valu", "k_heads": "latency"}
for k, c in sorted(others.items()):
    d = derived(c)
    st = stat_of(k)
    if st:
        d.update(st)
    bound = next((b for key, b in BOUNDS.items() if key in k), None)
    d["bound"] = bound
    if st and d["hbm_bytes"] is not None:
        d["hbm_tbps"] = d["hbm_bytes"] / (st["avg_us"] * 1e-6) / 1e12
        if bound == "hbm":
            d["frac"] = d["hbm_tbps"] * 1e12 / HBM_PEAK
    if bound == "valu":
        d["frac"] = d["valu_busy"]
    js["kernels"][k[:70]] = d
    print(f"{k[:58]:58s} {d['launches']:4d} launches" + (f" avg {st['avg_us']:8.1f} us {st['percent']:5.2f} %" if st else "") +
          f": HBM {0 if d['hbm_bytes'] is None else d['hbm_bytes'] / 1e6:8.1f} MB" + (f" = {d['hbm_tbps']:.2f} TB/s" if "hbm_tbps" in d else "") +
          f", MFMA {d['mfma_busy'] or 0:.3f} VALU {d['valu_busy'] or 0:.3f} LDS {d['lds_busy'] or 0:.3f} issue-stalled {d['wait_frac'] or 0:.3f}" +
          (f" -> {bound} fraction {d['frac']:.3f}" if "frac" in d else ""))
for k, st in stats.items():
    if is_msg(k):
        js["edge"]["stats"] = st
json.dump(js, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("-> traffic.json")

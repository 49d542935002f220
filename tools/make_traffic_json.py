"""From the counter passes of tools/final_profile.sh: the JSON bench.py replays into `roofline` (HBM bytes per launch of the dominant
kernel, pipe-busy fractions) and the roofline lines of k_knn_sample at C3 / C5.  Counter conventions (MI355X_MICROARCH.md): FETCH_SIZE /
WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950 for wide coalesced reads; SQ_* are summed over the chip (1024 SIMDs, 256 CUs);
GRBM_GUI_ACTIVE counts per shader engine: / 8 = active cycles of the launch."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]


def means(d):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out, d, "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            acc[(row.get("Kernel_Name", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    by_kernel = collections.defaultdict(dict)
    for (k, c), v in acc.items():
        by_kernel[k][c] = (sum(v) / len(v), len(v))
    return by_kernel


def derived(c):
    g = lambda n: c.get(n, (float("nan"), 0))[0]
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    return {"launches": c.get("GRBM_GUI_ACTIVE", (0, 0))[1], "active_cycles": cyc,
            "hbm_bytes": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0, "fetch_KiB": g("FETCH_SIZE"), "write_KiB": g("WRITE_SIZE"),
            "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / cyc, "valu_busy": g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cyc,
            "lds_busy": g("SQ_LDS_IDX_ACTIVE") / 256.0 / cyc, "wait_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
            "insts_valu": g("SQ_INSTS_VALU"), "insts_vmem_rd": g("SQ_INSTS_VMEM_RD"),
            "l2_hit": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))}


edge = means("pmc_edge")
res = {}
for k, c in edge.items():
    res[k] = derived(c)
    d = res[k]
    print(f"{k[:60]:60s} launches {d['launches']:3d}: HBM {d['hbm_bytes'] / 1e9:.3f} GB/launch (FETCH {d['fetch_KiB']:.0f} KiB x2 + WRITE {d['write_KiB']:.0f} KiB), "
          f"{d['active_cycles'] / 1e6:.2f} M cycles, MFMA busy {d['mfma_busy']:.3f}, VALU busy {d['valu_busy']:.3f}, LDS {d['lds_busy']:.3f}, "
          f"waiting {d['wait_frac']:.3f}, VALU insts {d['insts_valu'] / 1e6:.0f} M, L2 hit {d['l2_hit']:.3f}")
if res:
    N, H, B = 600, 256, 256
    # the counter passes run bench.py --num-steps 3: E = 4 evaluations of six launches; in the E - 1 step evaluations the last layer's
    # launch covers the ligand nodes only (half the rows at 300+300), so the launch-weighted algorithmic bytes are below 8 N H B
    E = 4
    alg = 8 * N * H * B * (6 * E - 0.5 * (E - 1)) / (6 * E)
    tot = sum(d["hbm_bytes"] * d["launches"] for d in res.values()) / sum(d["launches"] for d in res.values())
    w = lambda key: sum(d[key] * d["launches"] for d in res.values()) / sum(d["launches"] for d in res.values())
    js = {"kernel": " / ".join(sorted(res)), "config": {"R": 300, "L": 300, "batch": 256, "precision": "mfma16"},
          "source": "tools/final_profile.sh on MI355X: rocprofv3 --pmc, one counter set per run, kernel-filtered, no trace domains; "
                    "FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE; busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs and "
                    "SQ_ACTIVE_INST_VALU x 4 / 1024 over GRBM_GUI_ACTIVE / 8",
          "per_kernel": res, "traffic_bytes_per_launch": tot, "algorithmic_bytes_per_launch": alg,
          "mfma_busy": w("mfma_busy"), "valu_busy": w("valu_busy"), "wait_frac": w("wait_frac")}
    json.dump(js, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(f"-> traffic.json: {tot / 1e9:.3f} GB per launch (launch-weighted over the six layers) = {tot / alg:.2f} x algorithmic ({alg / 1e6:.1f} MB: 8 N H B per full launch, half of it for a ligand-only last layer); "
          f"MFMA busy {js['mfma_busy']:.3f}, VALU busy {js['valu_busy']:.3f}")
# k_knn_sample: a VALU-issue-bound kernel.  Roofline line = VALU-pipe busy fraction (instructions x measured issue cost over the
# launch's SIMD cycles) next to its HBM figure, which is tiny (16 N + 4 N K bytes per trajectory)
for tag, (N, B) in (("pmc_knn_c3", (600, 256)), ("pmc_knn_c5", (2000, 32))):
    for k, c in means(tag).items():
        d = derived(c)
        alg = (16 * N + 4 * N * 60) * B
        print(f"{tag} {k[:40]:40s} launches {d['launches']}: {d['active_cycles'] / 1e3:.0f} k cycles per launch, VALU insts {d['insts_valu'] / 1e6:.1f} M "
              f"({d['insts_valu'] / (B * N):.0f} per node), VALU busy {d['valu_busy']:.3f} (= roofline fraction of the VALU issue bound), waiting {d['wait_frac']:.3f}, "
              f"HBM {d['hbm_bytes'] / 1e6:.1f} MB per launch vs {alg / 1e6:.1f} MB algorithmic, pairwise distances {B * N * N / 1e6:.0f} M per launch")

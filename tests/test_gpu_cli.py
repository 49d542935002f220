"""`python -m dfmdock_amd` end to end on the GPU: a seeded checkpoint written by the test, 7CEI as two PDB files +
a features file (the reference's own backbone, sequence and ESM block: tests/golden/cx_7CEI.npz), and a DB5-style directory.

  dock   <- src/inference_single.py:1-12 / inference() (src/inference_base.py:601-670): output.pdb + {"energy": ...}
  sweep  <- src/inference_mlsb.py:188-262: reference-schema CSV + success-rate table + the self-check lines
"""
import csv
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cli_fixtures import golden_7cei, write_ckpt, write_db5_pt, write_pair
from conftest import ROOT, complex_for

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    return subprocess.run([sys.executable, "-m", "dfmdock_amd"] + args, cwd=cwd, capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))


def test_dock_7cei_end_to_end(tmp_path):
    from dfmdock_amd import cli, driver, engine, pdbio
    from dfmdock_amd.weights import load_lightning_checkpoint, pack_blob
    cx, rs, ls = golden_7cei()
    rec_pdb, lig_pdb, feat = write_pair(str(tmp_path), cx, rs, ls)
    ck = str(tmp_path / "model_0.ckpt")
    write_ckpt(ck, seed=0)
    p = _run(["dock", rec_pdb, lig_pdb, "--ckpt", ck, "--features", feat, "--num-samples", "9", "--max-batch", "4", "--seed", "5",
              "--json", str(tmp_path / "res.json")], cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "selfcheck pair" in p.stderr and " OK" in p.stderr and line["selfcheck_ok"] is True and line["precision"] == "mfma16"
    out = tmp_path / "output.pdb"                      # inference() writes output.pdb into the working directory
    assert out.exists() and os.path.samefile(line["output"], out)
    # the same call in process: same energy, same pose
    engine.set_device(0)
    sd, hp = load_lightning_checkpoint(ck)
    model = engine.Model(pack_blob(sd, hp), hp)
    rec, lig, rec_x, lig_x = cli.load_pair(rec_pdb, lig_pdb, feat)
    ref = driver.dock_pair(model, rec, lig, rec_x, lig_x, num_samples=9, num_steps=40, seed=5, max_batch=4, out_pdb=None)
    assert line["energy"] == ref["energy"]
    np.testing.assert_array_equal(np.float32(line["rot_update"]), ref["rot_update"])
    atoms = pdbio.read_pdb(str(out))
    n_rec = len(rec["atoms"])
    assert len(atoms) == n_rec + len(lig["atoms"])
    np.testing.assert_allclose(np.array([a["coord"] for a in atoms[:n_rec]]), rec["aa_coords"], atol=6e-4)      # the receptor as read
    np.testing.assert_allclose(np.array([a["coord"] for a in atoms[n_rec:]]), ref["lig_aa_coords"], atol=6e-4)  # the ligand at the best pose
    moved = np.array([a["coord"] for a in atoms[n_rec:]])
    d0 = np.linalg.norm(lig["aa_coords"][0] - lig["aa_coords"][-1])
    assert abs(np.linalg.norm(moved[0] - moved[-1]) - d0) < 2e-3                                                 # rigid
    res = json.load(open(tmp_path / "res.json"))
    assert res["selfcheck"]["ok"] and res["selfcheck"]["dev_f"] < 1e-2
    # the selfcheck command on the same inputs
    q = _run(["selfcheck", rec_pdb, lig_pdb, "--ckpt", ck, "--features", feat], cwd=str(tmp_path))
    assert q.returncode == 0 and json.loads(q.stdout.strip().splitlines()[-1])["ok"], q.stdout + q.stderr
    model.close()


def test_sweep_db5_directory(tmp_path):
    cx, rs, ls = golden_7cei()
    d = tmp_path / "db5"
    d.mkdir()
    write_db5_pt(str(d / "7CEI.pt"), "7CEI", cx, rs, ls)
    small = complex_for("fwd_syn_24_16")
    write_db5_pt(str(d / "SYN1.pt"), "SYN1", small, "A" * 24, "G" * 16)
    (d / "test.txt").write_text("7CEI\nSYN1\n1N2C\n")
    ck = str(tmp_path / "model_0.ckpt")
    write_ckpt(ck, seed=0)
    p = _run(["sweep", "--db5", str(d), "--ckpt", ck, "--num-samples", "6", "--num-steps", "8", "--out-csv", str(tmp_path / "r.csv"),
              "--summary", str(tmp_path / "s.json")], cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "1N2C.pt is listed but missing" in p.stderr and p.stderr.count("selfcheck ") >= 2
    assert "success rate over 2 complexes" in p.stdout and "DockQ >= 0.23" in p.stdout
    rows = list(csv.DictReader(open(tmp_path / "r.csv")))
    assert len(rows) == 12 and list(rows[0]) == ["id", "index", "c_rmsd", "i_rmsd", "l_rmsd", "fnat", "DockQ", "energy", "num_clashes"]
    s = json.load(open(tmp_path / "s.json"))
    assert set(s["complexes"]) == {"7CEI", "SYN1"} and len(s["selfcheck"]) == 2 and all(c["selfcheck"]["ok"] for c in s["selfcheck"])
    assert 0 <= s["success"]["acceptable"]["top1"] <= s["success"]["acceptable"]["best_of_n"] <= 1

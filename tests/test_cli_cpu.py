"""Argument / IO plumbing of `python -m dfmdock_amd` (dfmdock_amd/cli.py <- src/inference_single.py:1-12, src/inference_base.py:601-670,
src/inference_mlsb.py:188-262) on the CPU: checkpoint written by the test and read back without Lightning, PDB pair + features file
-> the arrays dfm_complex_create takes, DB5-style pickles, the success-rate table - and the commands FAIL without a GPU (the
product has no CPU path)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cli_fixtures import golden_7cei, write_ckpt, write_db5_pt, write_pair
from conftest import ROOT


def test_parser_and_defaults():
    from dfmdock_amd import cli
    a = cli.build_parser().parse_args(["dock", "r.pdb", "l.pdb", "--ckpt", "c.ckpt", "--features", "f.npz"])
    assert (a.num_samples, a.num_steps, a.out, a.precision) == (120, 40, "output.pdb", "mfma16")      # inference(): 120 x 40, output.pdb
    s = cli.build_parser().parse_args(["sweep", "--db5", "d", "--ckpt", "c"])
    assert (s.num_samples, s.out_csv, s.on_selfcheck_fail) == (40, "results.csv", "fp32")
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["dock", "r.pdb", "l.pdb", "--features", "f.npz"])      # --ckpt is required


def test_checkpoint_written_by_the_test_reads_back(tmp_path):
    from dfmdock_amd.weights import HParams, load_lightning_checkpoint, pack_blob
    for lightning in (True, False):
        p = str(tmp_path / f"m{int(lightning)}.ckpt")
        w = write_ckpt(p, seed=4, lightning=lightning)
        sd, hp = load_lightning_checkpoint(p)
        assert hp.family == 0 and hp.depth == 6 and hp.positional_embed_dim == 66
        np.testing.assert_array_equal(pack_blob(sd, hp), pack_blob(w))
    hp1 = HParams(family=1, mask_dist=20.0)
    p = str(tmp_path / "pair.ckpt")
    write_ckpt(p, hp=hp1, seed=1)
    sd, hp = load_lightning_checkpoint(p)
    assert hp.family == 1 and hp.mask_dist == 20.0      # the family is read off the state_dict keys


def test_pdb_pair_and_features_to_engine_inputs(tmp_path):
    from dfmdock_amd import cli
    cx, rs, ls = golden_7cei()
    rec_pdb, lig_pdb, feat = write_pair(str(tmp_path), cx, rs, ls)
    rec, lig, rec_x, lig_x = cli.load_pair(rec_pdb, lig_pdb, feat)
    assert rec["seq"] == rs and lig["seq"] == ls
    np.testing.assert_allclose(rec["bb_coords"], cx["rec_pos"], atol=6e-4)      # 8.3f columns
    np.testing.assert_array_equal(rec_x, cx["rec_x"])                         # ESM block + one-hot of the PDB's own sequence
    np.testing.assert_array_equal(lig_x, cx["lig_x"])
    assert lig["aa_coords"].shape[0] == sum(5 if c != "G" else 4 for c in ls)   # N CA C O CB, no CB on GLY
    _, _, feat2 = write_pair(str(tmp_path), cx, rs, ls, with_onehot=True)
    np.testing.assert_array_equal(cli.load_pair(rec_pdb, lig_pdb, feat2)[2], cx["rec_x"])
    bad = str(tmp_path / "bad.npz")
    np.savez(bad, rec_esm=cx["rec_x"][:-1, :1280], lig_esm=cx["lig_x"][:, :1280])
    with pytest.raises(ValueError, match="residues"):
        cli.load_pair(rec_pdb, lig_pdb, bad)
    np.savez(bad, lig_esm=cx["lig_x"][:, :1280])
    with pytest.raises(ValueError, match="rec_esm"):
        cli.load_pair(rec_pdb, lig_pdb, bad)
    np.savez(bad, rec_esm=cx["rec_x"][:, :1280], lig_esm=cx["lig_x"][:, :1280], rec_seq="A" * len(rs))
    with pytest.raises(ValueError, match="rec_seq"):
        cli.load_pair(rec_pdb, lig_pdb, bad)


def test_db5_directory_and_pickles(tmp_path):
    from dfmdock_amd import cli
    from dfmdock_amd.db5 import load_db5_pt
    cx, rs, ls = golden_7cei()
    d = tmp_path / "db5"
    d.mkdir()
    write_db5_pt(str(d / "7CEI.pt"), "7CEI", cx, rs, ls)
    write_db5_pt(str(d / "XXXX.pt"), "XXXX", cx, rs, ls)
    c = load_db5_pt(str(d / "7CEI.pt"))
    assert c["id"] == "7CEI" and c["rec_seq"] == rs
    np.testing.assert_array_equal(c["rec_x"], cx["rec_x"])
    np.testing.assert_array_equal(c["lig_pos"], cx["lig_pos"])
    assert cli.db5_ids(str(d)) == ["7CEI", "XXXX"]
    (d / "test.txt").write_text("XXXX\n1N2C\n7CEI\n")      # the reference's list names a file its tree does not hold (1N2C)
    assert cli.db5_ids(str(d)) == ["XXXX", "7CEI"] and cli.db5_ids(str(d), limit=1) == ["XXXX"]


def test_success_table():
    from dfmdock_amd import cli
    rows = [{"id": "A", "index": "0", "DockQ": 0.9, "energy": -1.0}, {"id": "A", "index": "1", "DockQ": 0.1, "energy": -2.0},
            {"id": "B", "index": "0", "DockQ": 0.3, "energy": -5.0}, {"id": "B", "index": "1", "DockQ": 0.5, "energy": -5.0},
            {"id": "C", "index": "0", "DockQ": 0.05, "energy": 0.0}]
    per, table = cli.success_table(rows)
    assert per["A"]["top1_DockQ"] == 0.1 and per["A"]["best_DockQ"] == 0.9      # minimum energy wins, not the best DockQ
    assert per["B"]["top1_DockQ"] == 0.3                                        # energy tie: the first minimum (inference_base.py:652)
    assert table["acceptable"]["top1"] == pytest.approx(1 / 3) and table["acceptable"]["best_of_n"] == pytest.approx(2 / 3)
    assert table["medium"]["best_of_n"] == pytest.approx(2 / 3) and table["high"]["best_of_n"] == pytest.approx(1 / 3) and table["high"]["top1"] == 0
    txt = cli.format_table(per, table)
    assert "DockQ >= 0.23" in txt and "A " in txt


def test_commands_fail_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cx, rs, ls = golden_7cei()
    rec_pdb, lig_pdb, feat = write_pair(str(tmp_path), cx, rs, ls)
    ck = str(tmp_path / "m.ckpt")
    write_ckpt(ck)
    p = subprocess.run([sys.executable, "-m", "dfmdock_amd", "dock", rec_pdb, lig_pdb, "--ckpt", ck, "--features", feat, "--num-samples", "2",
                        "--out", str(tmp_path / "o.pdb")], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "DfmError" in p.stderr and not os.path.exists(tmp_path / "o.pdb")

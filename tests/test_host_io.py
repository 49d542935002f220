"""Host-side I/O of the drivers against values produced by the reference's own functions."""
import os

import numpy as np
import pytest

from conftest import load_golden


def test_full_backbone_and_pdb_text_match_reference(tmp_path):
    from dfmdock_amd import pdbio
    g = load_golden("io_kats.npz")
    full = pdbio.full_backbone(g["coords"])
    np.testing.assert_allclose(full, g["full"], atol=2e-4)
    out = tmp_path / "bb.pdb"
    pdbio.write_backbone_pdb(str(out), g["full"], str(g["seq"]), delim=6)
    assert out.read_text() == str(g["pdb_text"])          # byte-identical save_PDB formatting (GLY has no CB)


def test_loader_global_rotation_matches_reference():
    from dfmdock_amd.driver import random_rotation, rotate_complex
    g, cx = load_golden("io_kats.npz"), load_golden("cx_7CEI.npz")
    r, l = rotate_complex(cx["rec_pos"], cx["lig_pos"], g["rot_mat"])
    np.testing.assert_allclose(r, g["rot_rec"], atol=2e-5)
    np.testing.assert_allclose(l, g["rot_lig"], atol=2e-5)
    r2, l2 = random_rotation(cx["rec_pos"], cx["lig_pos"], np.random.default_rng(0))
    both = np.concatenate([r2, l2])[:, 1]
    assert np.abs(both.mean(0)).max() < 1e-4              # centred on the joint CA centroid
    d0 = np.linalg.norm(cx["rec_pos"][0, 1] - cx["lig_pos"][0, 1])
    assert abs(np.linalg.norm(r2[0, 1] - l2[0, 1]) - d0) < 1e-3   # rigid


PDB = """\
ATOM      1  N   MET A   1      11.104  13.207   2.100  1.00  0.00           N
ATOM      2  CA  MET A   1      12.560  13.300   2.300  1.00  0.00           C
ATOM      3  C   MET A   1      13.100  14.700   2.000  1.00  0.00           C
ATOM      4  O   MET A   1      12.400  15.700   2.100  1.00  0.00           O
ATOM      5  N  AGLY A   2      14.400  14.800   1.700  1.00  0.00           N
ATOM      6  N  BGLY A   2      14.500  14.900   1.800  1.00  0.00           N
ATOM      7  CA  GLY A   2      15.000  16.100   1.400  1.00  0.00           C
ATOM      8  C   GLY A   2      16.500  16.000   1.200  1.00  0.00           C
ATOM      9  N   ALA A   3      17.100  17.100   0.800  1.00  0.00           N
ATOM     10  CA  ALA A   3      18.500  17.200   0.500  1.00  0.00           C
HETATM   11  O   HOH A 101      20.000  20.000  20.000  1.00  0.00           O
ATOM     12  N   MSE A   4      19.100  18.100   0.100  1.00  0.00           N
ATOM     13  CA  MSE A   4      20.500  18.200   0.200  1.00  0.00           C
ATOM     14  C   MSE A   4      21.000  19.600   0.300  1.00  0.00           C
END
"""


def test_pdb_parse_filter_and_roundtrip(tmp_path):
    from dfmdock_amd import pdbio
    p = tmp_path / "in.pdb"
    p.write_text(PDB)
    atoms = pdbio.read_pdb(str(p))
    assert len(atoms) == 13                                # alternate location B dropped
    info = pdbio.backbone_from_atoms(atoms)
    assert info["seq"] == "MGX"                            # ALA 3 lacks C -> dropped; MSE unknown -> X; HOH is HETATM
    assert info["bb_coords"].shape == (3, 3, 3) and info["aa_coords"].shape == (12, 3)
    np.testing.assert_allclose(info["bb_coords"][1, 0], [14.4, 14.8, 1.7])
    rot, tr = np.array([0.2, -0.1, 0.4]), np.array([1.0, -2.0, 0.5])
    moved = pdbio.apply_pose_all_atom(info["aa_coords"], info["bb_coords"], rot, tr)
    d0 = np.linalg.norm(info["aa_coords"][0] - info["aa_coords"][5])
    assert abs(np.linalg.norm(moved[0] - moved[5]) - d0) < 1e-9            # rigid
    c = info["bb_coords"][:, 1].mean(0)
    np.testing.assert_allclose(moved.mean(0) - info["aa_coords"].mean(0),
                               (pdbio.axis_angle_to_matrix(rot) - np.eye(3)) @ (info["aa_coords"].mean(0) - c) + tr, atol=1e-9)
    out = tmp_path / "out.pdb"
    pdbio.write_complex_pdb(str(out), info["atoms"], info["atoms"], moved)
    back = pdbio.read_pdb(str(out))
    assert len(back) == 24
    np.testing.assert_allclose(np.array([a["coord"] for a in back[12:]]), moved, atol=6e-4)   # 8.3f columns


def test_apply_pose_matches_sampler_convention():
    """modify_aa_coords on the backbone atoms == the sampler's own rigid transform (same centroid, same R)."""
    from dfmdock_amd import pdbio
    from oracle import oracle as ora
    cx = load_golden("cx_7CEI.npz")
    rot, tr = np.array([0.73, 1.45, -1.28], np.float32), np.array([10.0, -3.6, -30.9], np.float32)
    a = pdbio.apply_pose_all_atom(cx["lig_pos"].reshape(-1, 3), cx["lig_pos"], rot, tr).reshape(-1, 3, 3)
    b = ora.modify_coords(cx["lig_pos"], rot, tr)
    np.testing.assert_allclose(a, b, atol=2e-4)


@pytest.mark.skipif(not os.path.exists("/root/reference/data/db5_test/4POU.pt"), reason="reference data not mounted")
def test_db5_pt_reader_shapes():
    from dfmdock_amd.db5 import load_db5_pt
    d = load_db5_pt("/root/reference/data/db5_test/4POU.pt")
    assert d["rec_x"].shape == (120, 1301) and d["lig_x"].shape == (120, 1301) and d["rec_pos"].shape == (120, 3, 3)
    assert (d["rec_x"][:, 1280:].sum(1) == 1).all() and len(d["rec_seq"]) == 120

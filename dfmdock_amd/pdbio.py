"""PDB front / back end of the docking drivers without biotite.  Host side only.

  read_pdb / backbone_from_atoms   <- get_info_from_pdb            (src/inference_base.py:72-126)
  apply_pose_all_atom              <- modify_aa_coords             (src/inference_base.py:354-364)
  write_complex_pdb                <- combine_atom_arrays + PDBFile.write (src/inference_base.py:37-66,:659-668)
  place_fourth_atom, full_backbone <- utils/pdb.py:31-56, inference_mlsb.py:68-85 (O / virtual-CB placement)
  write_backbone_pdb, write_trajectory_pdb <- utils/pdb.py:59-84 (save_PDB), inference_mlsb.py:130-159 (save_trj)
"""
from __future__ import annotations

import numpy as np

THREE_TO_ONE = {
    "ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H",
    "ILE": "I", "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W",
    "TYR": "Y", "VAL": "V",
}   # residue_constants.py:959 restype_3to1
ONE_TO_THREE = {v: k for k, v in THREE_TO_ONE.items()}
ONE_TO_THREE.update({"-": "GAP", "X": "URI"})   # utils/pdb.py:4-27


def read_pdb(path):
    """ATOM records of the first model (first alternate location only) as a list of dicts."""
    atoms = []
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            alt = line[16]
            if alt not in (" ", "A"):
                continue
            atoms.append({
                "hetero": rec == "HETATM", "name": line[12:16].strip(), "res_name": line[17:20].strip(),
                "chain": line[21], "res_id": int(line[22:26]), "ins": line[26],
                "coord": (float(line[30:38]), float(line[38:46]), float(line[46:54])),
                "element": line[76:78].strip() if len(line) >= 78 else "",
            })
    return atoms


def backbone_from_atoms(atoms):
    """get_info_from_pdb semantics: drop HETATM; a residue (keyed by res_id, as the reference does) is kept when
    it has N, CA and C; seq via 3->1 (unknown -> X); bb_coords [n,3,3] float64 in file order."""
    atoms = [a for a in atoms if not a["hetero"]]
    names_by_res = {}
    for a in atoms:
        names_by_res.setdefault(a["res_id"], set()).add(a["name"])
    valid = {r for r, names in names_by_res.items() if {"N", "CA", "C"} <= names}
    kept = [a for a in atoms if a["res_id"] in valid]
    seq, order, last = [], [], None
    for a in kept:      # residue starts: change of (chain, res_id, ins, res_name)
        key = (a["chain"], a["res_id"], a["ins"], a["res_name"])
        if key != last:
            order.append(key)
            seq.append(THREE_TO_ONE.get(a["res_name"], "X"))
            last = key
    pick = lambda nm: np.array([a["coord"] for a in kept if a["name"] == nm], dtype=np.float64)
    n, ca, c = pick("N"), pick("CA"), pick("C")
    if not (len(n) == len(ca) == len(c) == len(seq)):
        raise ValueError("backbone atom counts do not match the residue count (duplicate residue ids?)")
    bb = np.stack([n, ca, c], axis=1)
    aa = np.array([a["coord"] for a in atoms], dtype=np.float64)
    return {"atoms": atoms, "seq": "".join(seq), "aa_coords": aa, "bb_coords": bb}


def axis_angle_to_matrix(aa):
    """geometry.py:154-198 in float64 (host-side pose application)."""
    aa = np.asarray(aa, np.float64).reshape(3)
    ang = np.linalg.norm(aa)
    s = 0.5 - ang * ang / 48.0 if abs(ang) < 1e-6 else np.sin(0.5 * ang) / ang
    r, (i, j, k) = np.cos(0.5 * ang), aa * s
    two_s = 2.0 / (r * r + i * i + j * j + k * k)
    return np.array([[1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r)],
                     [two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r)],
                     [two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)]])


def apply_pose_all_atom(aa_coords, bb_coords, rot, tr, center="ca"):
    """Rigid (rot, tr) of the sampler applied to all atoms.  center="ca": about the ORIGINAL backbone CA centroid
    (modify_aa_coords of src/inference_base.py:354-364, first model family); center="all_atoms": about the mean of the
    all-atom array itself, which is what the second family's script does (src/inference.py:256-266)."""
    if center == "all_atoms":
        center = np.asarray(aa_coords, np.float64).reshape(-1, 3).mean(axis=0)
    else:
        center = np.asarray(bb_coords, np.float64)[:, 1].mean(axis=0)
    R = axis_angle_to_matrix(np.asarray(rot).reshape(3))
    return (np.asarray(aa_coords, np.float64) - center) @ R.T + center + np.asarray(tr, np.float64).reshape(3)


def write_complex_pdb(path, rec_atoms, lig_atoms, lig_coords):
    """Receptor atoms as read + ligand atoms at `lig_coords`, one PDB model."""
    with open(path, "w") as f:
        k = 0
        for atoms, coords in ((rec_atoms, None), (lig_atoms, lig_coords)):
            for idx, a in enumerate(atoms):
                x, y, z = a["coord"] if coords is None else coords[idx]
                k += 1
                name = a["name"] if len(a["name"]) == 4 else " " + a["name"]
                f.write("%-6s%5d %-4s %3s %s%4d%s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % (
                    "HETATM" if a["hetero"] else "ATOM", k % 100000, name, a["res_name"], a["chain"], a["res_id"], a["ins"],
                    x, y, z, 1.0, 0.0, a["element"]))
        f.write("END\n")


def place_fourth_atom(a, b, c, length, planar, dihedral):
    """utils/pdb.py:31-56."""
    bc = b - c
    bc = bc / np.linalg.norm(bc, axis=-1, keepdims=True)
    n = np.cross(np.broadcast_to(b - a, bc.shape), bc)
    n = n / np.linalg.norm(n, axis=-1, keepdims=True)
    m = [bc, np.cross(n, bc), n]
    d = [length * np.cos(planar), length * np.sin(planar) * np.cos(dihedral), -length * np.sin(planar) * np.sin(dihedral)]
    return c + sum(mi * di for mi, di in zip(m, d))


def full_backbone(coords):
    """[n,3,3] (N,CA,C) -> [n,5,3] (N,CA,C,O,CB); O from the NEXT residue's N, wrapping at the end like the reference."""
    coords = np.asarray(coords, np.float32)
    N, CA, C = coords[:, 0], coords[:, 1], coords[:, 2]
    b, c = CA - N, C - CA
    a = np.cross(b, c)
    CB = np.float32(-0.58273431) * a + np.float32(0.56802827) * b - np.float32(0.54067466) * c + CA
    O = place_fourth_atom(np.roll(N, -1, axis=0), CA, C, np.float32(1.231), np.float32(2.108), np.float32(-3.142))
    return np.stack([N, CA, C, O.astype(np.float32), CB], axis=1)


def write_backbone_pdb(path, coords5, seq, delim=-1, mode="a"):
    """utils/pdb.py:59-84 save_PDB: chain A for residues <= delim, B after; CB skipped for GLY."""
    names = ["N", "CA", "C", "O", "CB"]
    with open(path, mode) as f:
        k = 0
        for r, residue in enumerate(coords5):
            aa3 = ONE_TO_THREE[seq[r]]
            for a, atom in enumerate(residue):
                if aa3 == "GLY" and names[a] == "CB":
                    continue
                f.write("ATOM  %5d  %-2s  %3s %s%4d    %8.3f%8.3f%8.3f  %4.2f %4.2f\n" % (
                    k + 1, names[a], aa3, "A" if r <= delim else "B", r + 1, atom[0], atom[1], atom[2], 1, 0.0))
                k += 1


def write_trajectory_pdb(path, rec_frames, lig_frames, rec_seq, lig_seq):
    """inference_mlsb.py:130-159 save_trj: one MODEL per frame (N,CA,C,O,CB), appended."""
    open(path, "w").close()
    for i, (x1, x2) in enumerate(zip(rec_frames, lig_frames)):
        with open(path, "a") as f:
            f.write("MODEL        " + str(i) + "\n")
        write_backbone_pdb(path, full_backbone(np.concatenate([x1, x2], 0)), rec_seq + lig_seq, delim=len(rec_seq) - 1)

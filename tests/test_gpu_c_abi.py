"""The C ABI from a plain C99 program (tests/c_abi/abi_client.c): built with gcc against include/dfmdock_amd.h and
dfmdock_amd/libdfmdock_amd.so, run on the GPU box, outputs bit-identical to the ctypes binding's for the same inputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_client_matches_ctypes_binding(blob, tmp_path):
    from dfmdock_amd import engine
    from dfmdock_amd.synthetic import make_complex
    libdir = os.path.join(ROOT, "dfmdock_amd")
    exe = str(tmp_path / "abi_client")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "abi_client.c"), "-o", exe, "-L", libdir, "-ldfmdock_amd",
                           "-Wl,-rpath," + libdir])
    cx = make_complex(24, 16, seed=5)
    B, R, L = 3, 24, 16
    rng = np.random.default_rng(0)
    poses = (cx["lig_pos"][None] + rng.standard_normal((B, 1, 1, 3)).astype(np.float32)).astype(np.float32)
    t = np.array([1.0, 0.5, 0.001], np.float32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<4i", R, L, B, blob.size))
        for a in (blob, cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"], poses, t):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    p = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    raw = np.fromfile(fout, np.float32)
    o = 0
    tr = raw[o:o + B * 3].reshape(B, 3); o += B * 3
    rot = raw[o:o + B * 3].reshape(B, 3); o += B * 3
    en = raw[o:o + B]; o += B
    cl = raw[o:o + B].view(np.int32); o += B
    fv = raw[o:o + B * L * 3].reshape(B, L, 3)
    engine.set_device(0)
    gx = engine.Complex(engine.Model(blob), cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.score(poses, t, seed=7, energy=True)
    np.testing.assert_array_equal(tr, r["tr_score"])
    np.testing.assert_array_equal(rot, r["rot_score"])
    np.testing.assert_array_equal(en, r["energy"])
    np.testing.assert_array_equal(cl, r["num_clashes"])
    np.testing.assert_array_equal(fv, r["f"])

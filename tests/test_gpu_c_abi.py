"""The C ABI from a plain C99 program (tests/c_abi/abi_client.c): built with gcc against include/dfmdock_amd.h and
dfmdock_amd/libdfmdock_amd.so, run on the GPU box, outputs bit-identical to the ctypes binding's for the same inputs - both
halves of the boundary: dfm_score, and dfm_sample with a dfm_inject holding every draw of a reference sampler run
(rollout_syn_24_16.npz: the C program's trajectory must also follow the reference's at the fp32 gate) + dfm_diffusion_coef."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

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
        g = load_golden("rollout_syn_24_16.npz")      # same complex (24 + 16, seed 5): the reference's 40-step run
        S = int(g["num_steps"])
        f.write(struct.pack("<i", S))
        for a in (g["R0"], g["tr_draw"], g["z_rot"], g["z_tr"]):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
        f.write(np.ascontiguousarray(g["edges"], np.int32).tobytes())
    p = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    raw = np.fromfile(fout, np.float32)
    o = 0
    tr = raw[o:o + B * 3].reshape(B, 3); o += B * 3
    rot = raw[o:o + B * 3].reshape(B, 3); o += B * 3
    en = raw[o:o + B]; o += B
    cl = raw[o:o + B].view(np.int32); o += B
    fv = raw[o:o + B * L * 3].reshape(B, L, 3); o += B * L * 3
    c_pose = raw[o:o + L * 9].reshape(L, 3, 3); o += L * 9
    c_rot, c_tr, c_en = raw[o:o + 3], raw[o + 3:o + 6], raw[o + 6]; o += 7
    c_cl = raw[o:o + 1].view(np.int32)[0]; o += 1
    c_trace = raw[o:o + S * L * 9].reshape(S, L, 3, 3); o += S * L * 9
    c_g = raw[o:o + 8].view(np.float64)
    engine.set_device(0)
    gx = engine.Complex(engine.Model(blob), cx["rec_x"], cx["lig_x"], cx["rec_pos"], cx["lig_pos"])
    r = gx.score(poses, t, seed=7, energy=True)
    np.testing.assert_array_equal(tr, r["tr_score"])
    np.testing.assert_array_equal(rot, r["rot_score"])
    np.testing.assert_array_equal(en, r["energy"])
    np.testing.assert_array_equal(cl, r["num_clashes"])
    np.testing.assert_array_equal(fv, r["f"])
    # the sampler half: bit-identical to ctypes, and on the reference's trajectory (SURVEY 8d gate 3: 0.05 A over 5 steps)
    inj = dict(R0=g["R0"].astype(np.float32), tr_draw=g["tr_draw"], z_rot=g["z_rot"], z_tr=g["z_tr"], edges=g["edges"])
    rs = gx.sample(B=1, num_steps=S, inject=inj, trace=True)
    np.testing.assert_array_equal(c_pose, rs["lig_pos"][0])
    np.testing.assert_array_equal(c_trace, rs["trace_pose"][0])
    np.testing.assert_array_equal(c_rot, rs["rot_update"][0])
    np.testing.assert_array_equal(c_tr, rs["tr_update"][0])
    assert c_en == rs["energy"][0] and c_cl == rs["num_clashes"][0]
    rmsd = np.sqrt(((c_trace[:, :, 1, :] - g["poses"][:, :, 1, :]) ** 2).sum(-1).mean(-1))
    assert rmsd[:5].max() < 0.05 and rmsd.max() < 0.5, rmsd
    assert abs(float(c_en) - float(g["final_energy"])) < 1e-3 and int(c_cl) == int(g["final_num_clashes"])
    k = load_golden("scalar_kats.npz")
    i = int(np.argmin(np.abs(k["ts"] - 0.487692297)))
    np.testing.assert_allclose(c_g, [k["g_r3"][i], k["sigma_r3"][i], k["g_so3"][i], k["sigma_so3"][i]], rtol=1e-12)
    assert c_g.tolist() == list(engine.diffusion_coef(0, 0.487692297)) + list(engine.diffusion_coef(1, 0.487692297))

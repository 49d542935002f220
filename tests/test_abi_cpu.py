"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/*.h declares,
host-only entry points agree with the reference's known answers, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dfmdock_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dfm_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dfmdock_amd import _lib
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dfmdock_amd.h but not exported"
    assert sorted(_lib.EXPORTS) == syms


def test_ctypes_structs_have_the_c_layout(tmp_path):
    """Every public struct of include/dfmdock_amd.h as gcc lays it out against the ctypes mirror in dfmdock_amd/_lib.py: a field added
    on one side only would shift everything behind it silently."""
    import subprocess
    from dfmdock_amd import _lib
    names = {"dfm_hparams": _lib.HParamsC, "dfm_score_out": _lib.ScoreOutC, "dfm_inject": _lib.InjectC, "dfm_traj_out": _lib.TrajOutC,
             "dfm_profile": _lib.ProfileC, "dfm_selfcheck_out": _lib.SelfcheckC}
    src = tmp_path / "sz.c"
    probes = {"dfm_selfcheck_out": ["dev_f", "cancel_ratio", "max_h", "max_acc", "saturated", "ok"], "dfm_hparams": ["cut_off", "r3_min_sigma", "agg_mean"],
              "dfm_traj_out": ["init_pose"], "dfm_score_out": ["dist_logits"], "dfm_profile": ["slot_cycles"], "dfm_inject": ["edges"]}
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names)
    body += "".join(f'printf("{n}.{f} %zu\\n", offsetof({n}, {f}));' for n, fs in probes.items() for f in fs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dfmdock_amd.h"\nint main(void){' + body + "return 0;}\n")
    exe = str(tmp_path / "sz")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(line.split() for line in subprocess.check_output([exe], text=True).splitlines())
    for n, cls in names.items():
        assert int(got[n]) == C.sizeof(cls), n
        for f in probes[n]:
            assert int(got[f"{n}.{f}"]) == getattr(cls, f).offset, (n, f)


def test_param_count_and_default_hparams():
    from dfmdock_amd import _lib, engine
    from dfmdock_amd.weights import HParams, n_params
    hp = _lib.HParamsC()
    _lib.lib().dfm_default_hparams(C.byref(hp))
    ref = HParams()
    from dfmdock_amd.weights import HParams as HP
    hp1 = engine.hparams_c(HP(family=1, mask_dist=20.0))
    assert _lib.lib().dfm_param_count(C.byref(hp1)) == n_params(HP(family=1)) == 3913543      # EGNN_Net (f-2)
    for k, v in ref.as_dict().items():
        assert getattr(hp, k) == pytest.approx(float(v)), k
    assert _lib.lib().dfm_param_count(C.byref(engine.hparams_c())) == n_params() == 3566919


def test_diffusion_coefficients_host_entry_point():
    from dfmdock_amd import engine
    k = load_golden("scalar_kats.npz")
    for i, t in enumerate(k["ts"]):
        g3, s3 = engine.diffusion_coef(0, t)
        gso, sso = engine.diffusion_coef(1, t)
        assert g3 == pytest.approx(k["g_r3"][i], rel=1e-13) and s3 == pytest.approx(k["sigma_r3"][i], rel=1e-13)
        assert gso == pytest.approx(k["g_so3"][i], rel=1e-13) and sso == pytest.approx(k["sigma_so3"][i], rel=1e-13)
    with pytest.raises(ValueError):       # so3_diffuser.py:212-213 raises ValueError
        engine.diffusion_coef(1, 1.5)
    with pytest.raises(ValueError):
        engine.diffusion_coef(1, -0.1)


def test_weights_blob_roundtrip():
    from dfmdock_amd.weights import make_random_weights, pack_blob, param_specs, unpack_blob
    w = make_random_weights(3)
    blob = pack_blob(w)
    back = unpack_blob(blob)
    assert list(back) == [n for n, _ in param_specs()]
    for k in w:
        np.testing.assert_array_equal(w[k], back[k])
    pref = {"net." + k: v for k, v in w.items()}       # Lightning checkpoints prefix keys with net.
    np.testing.assert_array_equal(pack_blob(pref), blob)
    with pytest.raises(KeyError):
        pack_blob({k: v for k, v in list(w.items())[:-1]})


def test_no_gpu_fails_loudly():
    """Without a device the product raises; it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dfmdock_amd import _lib, engine
    with pytest.raises(_lib.DfmError):
        engine.set_device(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dfmdock_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} references the oracle"


def test_no_kernel_uses_scratch():
    """Every kernel of the shipped code objects must have private_segment_fixed_size == 0 (no register spills to scratch, no
    private arrays).  r05 found two concurrently running instances of the message kernel - one per complex handle, each on its own
    stream - storing their segment sums through EACH OTHER's spilled output pointer: the one 64-bit value that kernel kept in
    scratch.  Scratch-free kernels cannot interfere that way; this test keeps it so (llvm-readelf on the gfx950 bundles)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from dfmdock_amd import _lib
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "llvm-readelf")):
        pytest.skip("llvm-readelf not available")
    td = tempfile.mkdtemp()
    try:
        lib = os.path.join(td, "lib.so")
        shutil.copy(_lib.LIB_PATH, lib)
        subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", lib], cwd=td, check=True, capture_output=True)
        seen, bad = 0, []
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", notes, re.S):
                seen += 1
                if int(m.group(2)) != 0:
                    bad.append((m.group(1), int(m.group(2))))
        assert seen >= 40, seen
        assert not bad, bad
    finally:
        shutil.rmtree(td, ignore_errors=True)


def test_no_compiler_made_packed_fp32_in_geometry_and_head_kernels():
    """r05: hipcc's SLP vectoriser turned the dihedral code of k_edge_feat<0> into packed-fp32 instructions (v_pk_mul_f32 /
    v_pk_add_f32 with SGPR-pair operands and op_sel), and waves of that kernel then computed WRONG theta bins from correct inputs
    whenever another complex handle's message kernel was resident - never with the -O1 / -fno-slp-vectorize builds of the same source
    (profiles/r05_concurrency.txt; reference arithmetic: src/utils/coords6d.py:25-43).  The mechanism is not identified, so the whole
    library is built with -fno-slp-vectorize (csrc/Makefile) and this test keeps compiler-made packed fp32 out of every kernel of
    kernels_geom.hip and kernels_heads.hip - the parity-sensitive fp32 code that never writes float2 arithmetic itself.  (The packed
    fp32 of the message / GEMM / pair kernels is hand-written vector code and has run next to itself since r01.)"""
    import re
    import shutil
    import subprocess
    import tempfile
    from dfmdock_amd import _lib
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    names = set()
    for src in ("kernels_geom.hip", "kernels_heads.hip"):
        txt = open(os.path.join(ROOT, "dfmdock_amd", "csrc", src)).read()
        names |= set(re.findall(r"__global__[^;{]*?\bvoid\s+(k_\w+)\s*\(", txt))
    assert {"k_edge_feat", "k_knn_sample", "k_prep_pose", "k_heads", "k_init_pose"} <= names, names
    mk = open(os.path.join(ROOT, "dfmdock_amd", "csrc", "Makefile")).read()
    assert re.search(r"^COMMON\s*:=.*-fno-slp-vectorize", mk, re.M), "csrc/Makefile: every translation unit is built without SLP vectorisation"
    td = tempfile.mkdtemp()
    try:
        lib = os.path.join(td, "lib.so")
        shutil.copy(_lib.LIB_PATH, lib)
        subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", lib], cwd=td, check=True, capture_output=True)
        seen, bad = set(), {}
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            asm = subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", os.path.join(td, f)], capture_output=True, text=True).stdout
            cur = None
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    hit = [n for n in names if re.search(r"\d+" + n + r"(I|E|P|N|\b)", m.group(1))]
                    cur = m.group(1) if hit else None
                    if cur:
                        seen.add(hit[0])
                    continue
                if cur and re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
                    bad[cur] = bad.get(cur, 0) + 1
        assert {"k_edge_feat", "k_knn_sample", "k_prep_pose", "k_heads"} <= seen, seen
        assert not bad, bad
    finally:
        shutil.rmtree(td, ignore_errors=True)


def test_no_packed_instruction_reads_source1_high_half_for_its_low_result():
    """r06 took the cross-handle miscompute of r05 down to one instruction form (profiles/r06_concurrency.txt): a packed fp32 VALU
    instruction with op_sel = [0,1] - the LOW result reads the HIGH half of source 1 - returns wrong values in a wave that holds no LDS
    while workgroups of the 16-bit message kernel are resident (stand-alone reproducer: tools/pkmul_probe.py; in the reference's
    arithmetic it was the cross product of a dihedral, src/utils/coords6d.py:25-43, as hipcc's SLP vectoriser packed it).  The shipped
    library contains NO packed instruction with an op_sel modifier at all (hand-written packed code uses op_sel_hi only, every
    translation unit is built with -fno-slp-vectorize): this test disassembles every kernel of the code objects and keeps it so."""
    import re
    import shutil
    import subprocess
    import tempfile
    from dfmdock_amd import _lib
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    td = tempfile.mkdtemp()
    try:
        lib = os.path.join(td, "lib.so")
        shutil.copy(_lib.LIB_PATH, lib)
        subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", lib], cwd=td, check=True, capture_output=True)
        packed, bad, cur = 0, [], None
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            asm = subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", os.path.join(td, f)], capture_output=True, text=True).stdout
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    continue
                m = re.search(r"\b(v_pk_\w+)\s+([^/]*)", line)
                if not m:
                    continue
                packed += 1
                sel = re.search(r"\bop_sel:\[([0-9,]+)\]", m.group(2))
                if sel:      # any op_sel on a packed instruction; the known-bad pattern is source 0 low, source 1 high
                    bad.append((cur, m.group(1), sel.group(0)))
        assert packed > 2000, packed      # the audit saw the library's packed code (message, GEMM and pair kernels)
        assert not bad, bad[:10]
    finally:
        shutil.rmtree(td, ignore_errors=True)


def _split_top_level(argstr):
    out, depth, cur = [], 0, ""
    for ch in argstr:
        if ch in "([{":      # (template arguments of a kernel are inside the parentheses the launch macro needs anyway)
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def test_every_kernel_launch_holds_some_lds():
    """a kernel launched WITHOUT any LDS can be placed on a CU whose LDS another handle's message kernel holds entirely - the
    one placement in which the packed-fp32 erratum of profiles/r06_concurrency.txt was ever observed.  Every hipLaunchKernelGGL of the
    library must therefore either pass a dynamic LDS size that is not the literal 0 (token_lds() or a real size) or launch a kernel whose
    code object has a static LDS segment; a new LDS-free launch fails here."""
    import re
    import shutil
    import subprocess
    import tempfile
    from dfmdock_amd import _lib
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "llvm-readelf")):
        pytest.skip("llvm-readelf not available")
    static_lds = {}
    td = tempfile.mkdtemp()
    try:
        lib = os.path.join(td, "lib.so")
        shutil.copy(_lib.LIB_PATH, lib)
        subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", lib], cwd=td, check=True, capture_output=True)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                static_lds[name] = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
    finally:
        shutil.rmtree(td, ignore_errors=True)
    launches, bad = 0, []
    csrc = os.path.join(ROOT, "dfmdock_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith(".hip"):
            continue
        txt = open(os.path.join(csrc, fn)).read()
        for m in re.finditer(r"hipLaunchKernelGGL\(", txt):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(txt[i], 0)
                i += 1
            args = _split_top_level(txt[m.end():i - 1])
            launches += 1
            if args[3] != "0":
                continue
            full = re.sub(r"[()\s]", "", args[0])
            kern = full.split("<")[0].split("::")[-1]
            targs = re.match(r"[^<]*<([0-9,]+)>$", full)      # literal template arguments select ONE instantiation (k_edge_feat<1> has static LDS, <0> is launched with the token)
            mangled = kern + ("I" + "".join(f"Li{a}E" for a in targs.group(1).split(",")) if targs else "")      # (prefix: defaulted template parameters follow)
            insts = {n: v for n, v in static_lds.items() if re.search(r"\d+" + re.escape(mangled) + (r"" if targs else r"(I|E|P|N|\b)"), n)}
            if not insts or min(insts.values()) == 0:
                bad.append((fn, kern, insts))
    assert launches >= 30, launches
    assert not bad, bad

#!/usr/bin/env python3
"""ISA-level variants of ONE kernel of kernels_geom.hip for the cross-handle miscompute hunt (profiles/r06_concurrency.txt): compile the
translation unit the way r05 shipped it (-O3 WITH the SLP vectoriser), edit the gfx950 assembly of the named kernel, re-assemble and
link a variant library next to the product's other objects.

    python tools/asm_variant.py NAME TRANSFORM [kernel-substring]        -> tools/variants/NAME.so
      TRANSFORM: none | nop_after_pk | nop_before_pk | nop_after_div | nop_after_trans | nop_everywhere |
                 nop_after_pk:<first>-<last>   (only the packed instructions number first..last of the kernel, 0-based)
                 sepdst:<i>-<i>     packed instruction i writes a fresh register pair, two moves copy it to its destination
                 longnop:<i>-<i>    64 wait states behind packed instruction i
                 unpack=<what>[:<first>-<last>]  replace packed instructions by two plain ones (same arithmetic, same rounding):
                     what = all | neg (those with neg_lo / neg_hi) | noneg | mov (v_pk_mov_b32) | add | mul | opsel | sgpr
"""
import os, re, shlex, subprocess, sys, tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "dfmdock_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=off"]      # r05's kernels_geom flags: SLP on


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)


def _halves(tok):
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok)
    if m:
        return m.group(1) + m.group(2), m.group(1) + m.group(3)
    return tok, tok      # inline constant / literal: the same value in both halves


def unpack(line, tmp):
    """`v_pk_{add,mul}_f32` / `v_pk_mov_b32` -> two plain VOP3 / VOP1 instructions with the same per-half operands and negations."""
    body = line.split(";")[0].strip()
    op, rest = body.split(None, 1)
    mods = dict((k, [int(x) for x in v.split(",")]) for k, v in re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]", rest))
    rest = re.sub(r"\s*(op_sel|op_sel_hi|neg_lo|neg_hi):\[[0-9,]+\]", "", rest)
    toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)]
    dst, srcs = _halves(toks[0]), [_halves(t) for t in toks[1:]]
    n = len(srcs)
    sel_lo, sel_hi = mods.get("op_sel", [0] * n), mods.get("op_sel_hi", [1] * n)
    neg_lo, neg_hi = mods.get("neg_lo", [0] * n), mods.get("neg_hi", [0] * n)
    plain = {"v_pk_add_f32": "v_add_f32_e64", "v_pk_mul_f32": "v_mul_f32_e64", "v_pk_mov_b32": "v_mov_b32_e32"}[op]

    def half(dreg, sel, neg):
        ops = [("-" if neg[i] else "") + srcs[i][sel[i]] for i in range(n)]
        return f"\t{plain} {dreg}, " + ", ".join(ops), [srcs[i][sel[i]] for i in range(n)]

    if op == "v_pk_mov_b32":      # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
        a, b2 = srcs[0][sel_lo[0]], srcs[1][sel_lo[1]]
        lo, lo_src, hi, hi_src = f"\t{plain} {dst[0]}, {a}", [a], f"\t{plain} {dst[1]}, {b2}", [b2]
        if dst[0] not in hi_src:
            return [lo, hi]
        if dst[1] not in lo_src:
            return [hi, lo]
        return [f"\t{plain} {tmp}, {a}", hi, f"\tv_mov_b32_e32 {dst[0]}, {tmp}"]
    lo, lo_src = half(dst[0], sel_lo, neg_lo)
    hi, hi_src = half(dst[1], sel_hi, neg_hi)
    if dst[0] not in hi_src:
        return [lo, hi]
    if dst[1] not in lo_src:
        return [hi, lo]
    lo_t, _ = half(tmp, sel_lo, neg_lo)
    return [lo_t, hi, f"\tv_mov_b32_e32 {dst[0]}, {tmp}"]


def main():
    name, transform = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "k_edge_featILi0E"
    td = tempfile.mkdtemp(prefix="asmv_")
    src = os.path.join(SRC, "kernels_geom.hip")
    run(["hipcc"] + FLAGS + ["-c", src, "-o", os.path.join(td, "ref.o"), "-save-temps"], cwd=td)
    dev_s = os.path.join(td, "kernels_geom-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(dev_s).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3dfm\d+" + kern[:-0 or None].replace("ILi0E", r"ILi0E") + r".*:", l) or (kern in l and l.rstrip().endswith(":") and l.startswith("_Z")))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    rng = None
    if ":" in transform:
        transform, r = transform.split(":")
        rng = tuple(int(x) for x in r.split("-"))
    what = None
    if transform.startswith("unpack="):
        transform, what = "unpack", transform.split("=")[1]
    TMP = "v63"
    out, n_pk, edits = [], 0, 0
    for i, l in enumerate(lines):
        op = l.split()[0] if l.strip() and not l.strip().startswith((";", ".", "#")) and not l.rstrip().endswith(":") else ""
        inside = start < i < end
        before = after = False
        if inside and op:
            is_pk = op.startswith("v_pk_")
            sel = True
            if is_pk:
                if rng is not None:
                    sel = rng[0] <= n_pk <= rng[1]
                n_pk += 1
            if transform == "nop_after_pk" and is_pk and sel: after = True
            if transform == "nop_before_pk" and is_pk and sel: before = True
            if transform == "nop_after_div" and op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")): after = True
            if transform == "nop_after_trans" and re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)", op): after = True
            if transform == "nop_everywhere" and op.startswith("v_"): after = True
        if inside and op and transform == "unpack" and op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_mov_b32"):
            body = l.split(";")[0]
            hit = {"all": True, "neg": "neg_" in body, "noneg": "neg_" not in body, "mov": op == "v_pk_mov_b32", "add": op == "v_pk_add_f32",
                   "mul": op == "v_pk_mul_f32", "opsel": "op_sel" in body, "sgpr": re.search(r"\bs\[", body) is not None}[what]
            if hit and (rng is None or rng[0] <= n_pk - 1 <= rng[1]):
                out.extend(unpack(l, TMP)); edits += 1
                continue
        if inside and op and op.startswith("v_pk_") and transform in ("sepdst", "longnop") and rng is not None and rng[0] <= n_pk - 1 <= rng[1]:
            if transform == "longnop":      # the instruction as it is, 64 wait states before anything else issues
                out.append(l); out.extend(["\ts_nop 15"] * 4); edits += 1
                continue
            # the same packed instruction into a fresh register pair (no destination / source overlap), then two moves
            body = l.split(";")[0]
            m = re.match(r"(\s*\S+\s+)(v\[(\d+):(\d+)\])(,.*)", body)
            out.append(m.group(1) + "v[62:63]" + m.group(5))
            out.append(f"\tv_mov_b32_e32 v{m.group(3)}, v62"); out.append(f"\tv_mov_b32_e32 v{m.group(4)}, v63"); edits += 1
            continue
        if before:
            out.append("\ts_nop 7"); edits += 1
        out.append(l)
        if after:
            out.append("\ts_nop 7"); edits += 1
    if transform in ("unpack", "sepdst"):      # the temporaries v62 / v63: make sure the kernel's descriptor covers them
        mangled = lines[start].split(":")[0]
        k0 = next(i for i, l in enumerate(out) if l.strip().startswith(".amdhsa_kernel") and mangled in l)
        for i in range(k0, k0 + 80):
            m = re.match(r"(\s*\.amdhsa_next_free_vgpr\s+)(\d+)", out[i])
            if m:
                out[i] = m.group(1) + str(max(int(m.group(2)), 64))
                break
    mod_s = os.path.join(td, "mod.s")
    open(mod_s, "w").write("\n".join(out))
    clang = os.path.join(LLVM, "clang")
    run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", mod_s, "-o", os.path.join(td, "mod.o")])
    run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", os.path.join(td, "mod.o"), "-o", os.path.join(td, "mod.out")])
    run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", "-input=" + os.path.join(td, "mod.out"), "-output=" + os.path.join(td, "mod.hipfb")])
    # the host half: hipcc's own cc1 command with the edited device binary
    dash = subprocess.run(["hipcc"] + FLAGS + ["-c", src, "-o", os.path.join(td, "host.o"), "-###"], capture_output=True, text=True).stderr
    host = [l for l in dash.split("\n") if '"-triple" "x86_64-unknown-linux-gnu"' in l and "-fcuda-include-gpubinary" in l][0]
    argv = shlex.split(host)
    argv[argv.index("-fcuda-include-gpubinary") + 1] = os.path.join(td, "mod.hipfb")
    argv[argv.index("-o") + 1] = os.path.join(td, "kernels_geom.o")
    run(argv)
    os.makedirs(os.path.join(ROOT, "tools", "variants"), exist_ok=True)
    so = os.path.join(ROOT, "tools", "variants", name + ".so")
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(td, "kernels_geom.o")] +
        [os.path.join(SRC, f) for f in ("api.o", "kernels_heads.o", "kernels_dense.o", "kernels_edge.o", "kernels_pair.o")])
    print(f"{name}: transform {transform}{'' if what is None else '=' + what}{'' if rng is None else rng} on {lines[start][:40]}: {n_pk} packed instructions in the kernel, {edits} edits -> {so}")


if __name__ == "__main__":
    main()

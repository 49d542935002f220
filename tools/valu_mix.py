#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel of libdfmdock_amd.so, by class, from its gfx950 disassembly (llvm-objdump) ->
profiles/r06_valu_mix.json.  bench.py prices the message kernel's VALU-issue floor with it:

    floor = (T * c_T + P * c_P + Q * c_Q) / (SIMDs * clock)

T = transcendental wave instructions of a launch (known from the algebra: 4 per edge and channel / 64 lanes), the remaining VALU
instructions of the launch (SQ_INSTS_VALU of the committed PMC pass minus T minus the MFMAs) split into packed (P) and plain (Q)
in the STATIC proportion found here, c_* = issue cost per wave64 instruction and SIMD at two waves per SIMD from
tools/ubench/valu_rate*.hip (profiles/r01_ubench_valu_rate.txt, r02_ubench_valu_rate16.txt).

    python tools/valu_mix.py            (needs /opt/rocm/lib/llvm/bin; extracts into a temporary directory)
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LIB = os.path.join(ROOT, "dfmdock_amd", "libdfmdock_amd.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNELS = {"k_edge_msg<1,1,0>": "_ZN3dfm10k_edge_msgILi1ELi1ELi0EEEvNS_9EdgeKArgsE",
           "k_edge_coord<1>": "_ZN3dfm12k_edge_coordILi1EEEvNS_9EdgeKArgsE"}
TRANS = re.compile(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)_")


def main():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        lib = os.path.join(td, "lib.so")
        os.symlink(LIB, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=td, check=True, capture_output=True)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            asm = subprocess.run([OBJDUMP, "-d", os.path.join(td, f)], capture_output=True, text=True).stdout
            for name, sym in KERNELS.items():
                m = re.search(r"^[0-9a-f]+ <" + re.escape(sym) + r">:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.S | re.M)
                if not m:
                    continue
                c = collections.Counter()
                for line in m.group(1).splitlines():
                    mm = re.match(r"\s+(v_\S+)", line)
                    if mm:
                        c[mm.group(1)] += 1
                mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
                tr = sum(v for k, v in c.items() if TRANS.match(k))
                pk32 = sum(v for k, v in c.items() if k.startswith("v_pk_") and k.endswith("f32"))
                pk16 = sum(v for k, v in c.items() if k.startswith("v_pk_") and not k.endswith("f32"))
                tot = sum(c.values())
                out[name] = {"static_valu_instructions": tot, "mfma": mfma, "transcendental": tr, "packed_f32": pk32, "packed_16": pk16,
                             "plain": tot - mfma - tr - pk32 - pk16,
                             "packed_share_of_non_transcendental": (pk32 + pk16) / max(tot - mfma - tr, 1),
                             "top": dict(c.most_common(16))}
    out["issue_cycles_per_wave64_instruction_at_2_waves_per_simd"] = {
        "transcendental": 12.18, "packed_f32": 6.26, "packed_16": 5.70, "plain": 3.16, "clock_GHz": 2.4,
        "source": "profiles/r01_ubench_valu_rate.txt, profiles/r02_ubench_valu_rate16.txt (tools/ubench/valu_rate*.hip on MI355X)"}
    out["method"] = __doc__.split("\n\n")[1].strip()
    path = os.path.join(ROOT, "profiles", "r06_valu_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "top"} for k, v in out.items() if k.startswith("k_")}, indent=1))


if __name__ == "__main__":
    main()

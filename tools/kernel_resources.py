#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of the kernels of a built library whose mangled name contains PATTERN (llvm-readelf notes of the gfx950 bundle).

    python tools/kernel_resources.py dfmdock_amd/libdfmdock_amd.so k_edge_msg
"""
import os, re, subprocess, sys, tempfile
T = "/opt/rocm/lib/llvm/bin"
lib, pat = os.path.abspath(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as td:
    l2 = os.path.join(td, "lib.so"); os.symlink(lib, l2)
    subprocess.run([os.path.join(T, "llvm-objdump"), "--offloading", l2], cwd=td, check=True, capture_output=True)
    for f in sorted(os.listdir(td)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(T, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if pat not in name:
                continue
            g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
            print(f"{name[:60]:60s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} spill_v {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size')}")

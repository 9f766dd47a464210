#!/usr/bin/env python3
"""Disassembly of ONE kernel of libdcs_hip.so (gfx950), for reading what the compiler made of a hot loop:
    python3 tools/disasm_kernel.py k_pose_opt2 [lib.so] > /tmp/k.s
Finds the code object that holds the kernel (the library's .hip_fatbin is a run of clang offload bundles, tools/check_codeobj.py reads the
same), runs llvm-objdump -d on it and prints the kernel's symbol only."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_codeobj as cc

name = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb-slam2-dualcam_amd", "lib", "libdcs_hip.so")
fat = cc.section_bytes(lib, ".hip_fatbin")
for elf in cc.code_objects(fat):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf); path = f.name
    try:
        syms = subprocess.run([f"{cc.LLVM}/llvm-readelf", "-s", "-W", path], check=True, capture_output=True, text=True).stdout
        hit = [l.split()[7] for l in syms.splitlines() if len(l.split()) >= 8 and l.split()[3] == "FUNC" and name in l.split()[7]]
        if not hit:
            continue
        out = subprocess.run([f"{cc.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={hit[0]}", path], check=True, capture_output=True, text=True).stdout
        sys.stdout.write(out)
        break
    finally:
        os.unlink(path)

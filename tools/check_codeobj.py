#!/usr/bin/env python3
"""Build gate over the gfx950 code objects inside libdcs_hip.so.

For every kernel: code bytes (its symbol's size), VGPRs / SGPRs, spills, scratch, LDS -- read from the code object's own
metadata (llvm-readelf --notes) and symbol table. The build FAILS (exit 1) when a kernel spills a register, uses
scratch, or is larger than the limit (default 48 KB: the instruction cache of a CU pair is 64 KB). Round 5 shipped a
kernel of 940 KB with 13 375 spilled SGPRs without anybody noticing; this is the tripwire.

    python3 tools/check_codeobj.py [lib.so] [--max-code BYTES] [--out table.txt] [--allow NAME=REASON ...] [--sgpr-only-allow NAME=REASON ...]

Kernels on an allow list are reported, marked, and do not fail the build (each needs a reason: the table prints it). --sgpr-only-allow
tolerates only spilled SCALAR registers (they go to lanes of a vector register, not to memory); build.sh carries the list.
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("DCS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def section_bytes(path, name):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-S", "-W", path], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.match(r"\s*\[\s*\d+\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m and m.group(1) == name:
            off, size = int(m.group(3), 16), int(m.group(4), 16)
            with open(path, "rb") as f:
                f.seek(off)
                return f.read(size)
    return None


def code_objects(fat):
    """every gfx950 ELF of the concatenated clang offload bundles"""
    objs, at = [], 0
    while True:
        at = fat.find(MAGIC, at)
        if at < 0:
            break
        n = struct.unpack_from("<Q", fat, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                objs.append(fat[at + off:at + off + size])
        at += 24
    return objs


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf_bytes)
        path = f.name
    try:
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", path], check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(path)
    size_of = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size_of[f[7]] = int(f[2], 0)
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key, d=0: (lambda m: m.group(1) if m else d)(re.search(r"\." + key + r":\s*'?([^'\n]+)'?", blk))
        sym = get("symbol", "")
        if not sym.endswith(".kd"):
            continue
        name = sym[:-3]
        out.append(dict(mangled=name, code=size_of.get(name, 0), vgpr=int(get("vgpr_count")), agpr=int(get("agpr_count")), sgpr=int(get("sgpr_count")),
                        vspill=int(get("vgpr_spill_count")), sspill=int(get("sgpr_spill_count")), scratch=int(get("private_segment_fixed_size")),
                        lds=int(get("group_segment_fixed_size")), wg=int(get("max_flat_workgroup_size"))))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return [re.sub(r"^void ", "", re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", s)).replace("dcs::", "").replace("(anonymous namespace)::", "") for s in r]
    except Exception:
        return names


def names(entry, kernel):
    """True when an allow-list entry names this kernel: the whole demangled name, or the name before its template arguments
    (k_octree does not name k_octree_hist<1>, k_pose_opt does not name k_pose_opt2)."""
    return kernel == entry or ("<" not in entry and kernel.startswith(entry + "<"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(os.path.dirname(__file__), "..", "orb-slam2-dualcam_amd", "lib", "libdcs_hip.so"))
    ap.add_argument("--max-code", type=int, default=48 * 1024)
    ap.add_argument("--out")
    ap.add_argument("--sgpr-only-allow-file", help="file of NAME=REASON lines (tools/codeobj_allow.txt)")
    ap.add_argument("--verbose", action="store_true", help="print the whole table even when every kernel passes")
    ap.add_argument("--allow", action="append", default=[], help="NAME=REASON (the demangled name, with or without its template arguments): anything goes for this kernel")
    ap.add_argument("--sgpr-only-allow", nargs="*", default=[], help="NAME=REASON ...: spilled SCALAR registers (to VGPR lanes) are tolerated for these kernels; vector spills, scratch and size still fail")
    a = ap.parse_args()
    fat = section_bytes(a.lib, ".hip_fatbin")
    if fat is None:
        print(f"check_codeobj: no .hip_fatbin in {a.lib}", file=sys.stderr)
        return 2
    ks = []
    for co in code_objects(fat):
        ks += kernels_of(co)
    if not ks:
        print("check_codeobj: no gfx950 kernels found", file=sys.stderr)
        return 2
    for k, d in zip(ks, demangle([k["mangled"] for k in ks])):
        k["name"] = d
    allow = dict(x.split("=", 1) for x in a.allow)
    allow_s = dict(x.split("=", 1) for x in a.sgpr_only_allow)
    if a.sgpr_only_allow_file:
        with open(a.sgpr_only_allow_file) as f:
            allow_s.update(dict(ln.strip().split("=", 1) for ln in f if "=" in ln and not ln.lstrip().startswith("#")))
    ks.sort(key=lambda k: -k["code"])
    lines = [f"# {os.path.basename(a.lib)}: {len(ks)} gfx950 kernels; gate: spills == 0, scratch == 0, code <= {a.max_code} B",
             f"{'code B':>8} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'v-spill':>7} {'s-spill':>7} {'scratch':>7} {'lds B':>7} {'wg':>5}  kernel"]
    bad = 0
    for k in ks:
        why = []
        if k["vspill"] or k["sspill"]:
            why.append("spills")
        if k["scratch"]:
            why.append("scratch")
        if k["code"] > a.max_code:
            why.append("code size")
        mark = ""
        if why:
            reason = next((r for n, r in allow.items() if names(n, k["name"])), None)
            if reason is None and why == ["spills"] and not k["vspill"]:
                reason = next((r for n, r in allow_s.items() if names(n, k["name"])), None)
            if reason is None:
                bad += 1
                mark = "   <-- FAIL: " + ", ".join(why)
            else:
                mark = f"   (allowed: {', '.join(why)}; {reason})"
        lines.append(f"{k['code']:>8} {k['vgpr']:>5} {k['agpr']:>5} {k['sgpr']:>5} {k['vspill']:>7} {k['sspill']:>7} {k['scratch']:>7} {k['lds']:>7} {k['wg']:>5}  {k['name']}{mark}")
    lines.append(f"# total code {sum(k['code'] for k in ks)} B; largest {ks[0]['code']} B; failing kernels: {bad}")
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    sys.stdout.write(text if bad or a.verbose else "\n".join(lines[:1] + lines[-1:]) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

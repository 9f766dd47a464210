#!/usr/bin/env python3
"""Static instruction mix of every gfx950 kernel of the library, by ISSUE CLASS (profiles/r04_valu_rate_probe.txt): what bench.py prices
`valu_issue_frac` with. Compiles each csrc/*.hip to assembly (hipcc -S --cuda-device-only; no GPU needed), walks every kernel and counts,
separately for ALL basic blocks and for the blocks LLVM marks as part of a loop (the hot code):
  valu_fast   plain 32-bit add / sub / logic / right shifts, 16-bit min / max / add / sub / mul, f32 add / mul / fma, v_bitop3_b32, v_mov
              (0.75-0.85 ticks of 720 MHz per wave instruction and SIMD = 2.5 shader cycles) -- unless an operand is an SGPR or the
              instruction is an SDWA / DPP form: those issue at the slow rate
  valu_slow   everything else on the vector ALU (1.25-1.4 ticks = 4.3 cycles); v_max3/min3/med3_*16 and v_rcp/rsq/sqrt count double (2.5 ticks)
  mfma, lds_read / lds_write (ds_*: plain reads 2.5 CU-cycles; writes, atomics, permutes and 12-/16-byte reads 4.2), vmem (buffer_/global_/flat_/scratch_), sop2 (two-operand scalar: 2.4 ticks), sop1_other (1.3), s_wait_nop (0.3-0.4)
usage: tools/isa_class_mix.py <out.json>"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "orb-slam2-dualcam_amd", "csrc")
FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16", "v_add_u16", "v_sub_u16", "v_mul_lo_u16", "v_lshrrev_b16",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_bitop3_b32", "v_add_i32", "v_sub_i32"}
DOUBLE = re.compile(r"^v_(max3|min3|med3)_[iu]16|^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
SOP1 = re.compile(r"^s_(mov|not|bcnt|ff|flbit|brev|sext|and_saveexec|or_saveexec|andn2_saveexec|xor_saveexec|cmov|abs|bitset|getpc|wqm|quadmask|movk|cmpk)")
def classify(line):
    m = line.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m)
    if m.startswith("v_mfma") or m.startswith("v_smfma"): return "mfma"
    if m.startswith("v_"):
        if DOUBLE.match(base): return "valu_slow2"
        ops = line.split(None, 1)[1] if " " in line else ""
        srcs = ",".join(ops.split(",")[1:])
        sgpr_src = re.search(r"\b(s\d+|s\[\d+:\d+\]|vcc|exec)", srcs) is not None
        if base in FAST and not m.endswith(("_sdwa", "_dpp")) and "dpp" not in ops and "sel:" not in ops and not sgpr_src: return "valu_fast"
        return "valu_slow"
    if m.startswith("ds_"): return "lds_write" if m.startswith(("ds_write", "ds_add", "ds_min", "ds_max", "ds_or", "ds_and", "ds_cmpst", "ds_bpermute", "ds_permute", "ds_swizzle")) or "b128" in m or "b96" in m else "lds_read"
    if m.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if m.startswith("s_"):
        if m in ("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep") or m.startswith(("s_branch", "s_cbranch", "s_cmp", "s_load", "s_buffer_load", "s_setprio", "s_sendmsg", "s_setreg", "s_getreg")): return "s_other"
        return "sop1" if SOP1.match(m) else "sop2"
    return None
def mix_of(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "a.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-x", "hip", "--cuda-device-only", "-S", path, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur, in_loop = {}, None, False
    names = {l.split()[1] for l in text if l.strip().startswith(".amdhsa_kernel")}
    for l in text:
        s = l.strip()
        if s.endswith(":") or (":" in s and s.split(":")[0] in names) or re.match(r"^[\w.$]+:", s):
            label = s.split(":")[0]
            if label in names: cur, in_loop = label, False; kernels[cur] = {"all": {}, "loops": {}}
            elif label.startswith(".LBB") and cur: in_loop = "Loop" in s
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section"): cur = None if s.startswith(".Lfunc_end") else cur
        if not cur or not s or s.startswith((";", ".")): continue
        c = classify(s)
        if c:
            kernels[cur]["all"][c] = kernels[cur]["all"].get(c, 0) + 1
            if in_loop: kernels[cur]["loops"][c] = kernels[cur]["loops"].get(c, 0) + 1
    return kernels
def demangle(n):
    try: return subprocess.check_output(["c++filt", n], text=True).strip().split("(")[0].replace("void ", "").replace("dcs::", "").replace("(anonymous namespace)::", "")
    except Exception: return n
res = {}
for f in sorted(os.listdir(SRC)):
    if not f.endswith(".hip"): continue
    for k, v in mix_of(os.path.join(SRC, f)).items():
        name = demangle(k)
        for scope in ("all", "loops"):
            m = v[scope]
            fast, slow = m.get("valu_fast", 0), m.get("valu_slow", 0) + 2 * m.get("valu_slow2", 0)
            m["valu_fast_share"] = round(fast / max(fast + slow, 1), 3)
            m["ticks_per_valu_instruction"] = round((0.8 * fast + 1.33 * slow) / max(fast + m.get("valu_slow", 0) + m.get("valu_slow2", 0), 1), 3)
        res[name] = dict(file=f, **v)
json.dump({"tool": "tools/isa_class_mix.py (static counts from hipcc -S; 'loops' = basic blocks inside a loop)", "kernels": res}, open(sys.argv[1], "w"), indent=1)
for n in ("k_fast_cells<48, true, true>", "k_fast_cells<48, true, false>", "k_describe<true>", "k_pose_opt2", "k_resize<true>", "k_knn2_pairs_fp4", "k_octree_hist<2>"):
    for k, v in res.items():
        if k.startswith(n.split("<")[0]) and (n in k or "<" not in n): print(k, v["loops"]); break

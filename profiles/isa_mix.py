#!/usr/bin/env python3
"""Static vector-instruction mix of one kernel of libgsplat_mi355x, from the ISA hipcc emits for the Makefile's flags
(`hipcc -S --cuda-device-only`), and the issue cost that mix has on this part.

    python profiles/isa_mix.py rasterize 'rasterize_bwd_kernel<4, true>'

Classes and their measured issue cost in ns per wave-instruction per SIMD at 4 resident waves per SIMD (the occupancy the
rasterizers run at), from profiles/r02_valu_ubench.txt (scratch/ubench/valu.hip on gfx950):
    fma   v_fma / v_fmac / v_mul / v_add / v_sub / v_mad (f32)      1.45   (v_fma 1.57, v_mul 1.35, v_add 1.27)
    mov   v_mov_b32 / b64                                            1.34
    cmp   v_cmp_*                                                    2.04
    sel   v_cndmask (SGPR mask), v_min / v_max / v_med3              2.05   (cndmask 2.19, max 1.92)
    trans v_exp / v_rcp / v_rsq / v_sqrt / v_log                     3.60
    dpp   any VALU op with a DPP modifier                            1.91
    swap  v_permlane32_swap / v_permlane16_swap                      3.59
    pk    v_pk_*                                                     2.25
    other integer / bit / convert / readlane                         1.45
A STATIC count over the kernel's INNERMOST loop blocks (the compiler's loop-depth annotations), every instruction once:
for the rasterizers that is the per-splat loop with its four unrolled quadrant evaluations, the reduce-scatter and the
atomic -- the instructions that dominate the executed stream; branches taken or not are not weighed."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST_NS = {"fma": 1.45, "mov": 1.34, "cmp": 2.04, "sel": 2.05, "trans": 3.60, "dpp": 1.91, "swap": 3.59, "pk": 2.25,
           "other": 1.45}
EXTRA = {"rasterize": ["-fno-slp-vectorize"]}


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return None
    if "permlane" in op:
        return "swap"
    if "_dpp" in op or " row_" in line or "quad_perm" in line:
        return "dpp"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith(("v_cndmask", "v_min", "v_max", "v_med3")):
        return "sel"
    if op.startswith("v_mov"):
        return "mov"
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|mad|mac)_f32", op) or re.match(r"v_(fma|fmac)_", op):
        return "fma"
    return "other"


def kernel_mix(src, demangled_substring):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "3d-reconstruction_amd", "csrc", src + ".hip")] + EXTRA.get(src, [])
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    names = re.findall(r"^(_Z\w+):", text, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    want = [n for n, d in zip(names, dem) if demangled_substring in d]
    if not want:
        raise SystemExit(f"no kernel matching {demangled_substring!r} in {src}.hip")
    name = want[0]
    body = text[text.index(f"\n{name}:"):]
    body = body[:body.index("s_endpgm")]
    # instructions of the innermost loops only (the compiler annotates every block of a loop with its depth)
    depth_of, cur, pending = [], 0, False
    lines = body.split("\n")
    for line in lines:
        t = line.strip()
        if re.match(r"\.LBB\d+_\d+:", t):
            m = re.search(r"Depth=(\d+)", t)
            cur, pending = (int(m.group(1)) if m else 0), True
        elif pending and t.startswith(";"):
            m = re.search(r"Depth=(\d+)", t)
            if m:
                cur = max(cur, int(m.group(1)))
        elif t and not t.startswith(";"):
            pending = False
        depth_of.append(cur)
    deepest = max(depth_of) if depth_of else 0
    counts = {}
    for line, d in zip(lines, depth_of):
        line = line.strip()
        if d != deepest or not line or line.startswith((";", ".")):
            continue
        c = classify(line)
        if c:
            counts[c] = counts.get(c, 0) + 1
    m = re.search(r"\.name:\s+%s\b.*?\.vgpr_count:\s+(\d+)" % re.escape(name), text, re.S)
    m2 = re.search(r"\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", text[text.index(f".name:           {name}"):]
                   if f".name:           {name}" in text else text)
    total = sum(counts.values())
    mean_ns = sum(COST_NS[c] * n for c, n in counts.items()) / max(total, 1)
    return {"kernel": demangled_substring, "loop_depth": deepest, "static_valu_instructions_in_innermost_loops": total,
            "share": {c: round(n / total, 4) for c, n in sorted(counts.items())},
            "mean_issue_ns_per_wave_instruction": round(mean_ns, 4),
            "vgprs": int(m2.group(1)) if m2 else (int(m.group(1)) if m else None),
            "vgpr_spills": int(m2.group(2)) if m2 else None}


if __name__ == "__main__":
    print(json.dumps(kernel_mix(sys.argv[1], sys.argv[2]), indent=1))

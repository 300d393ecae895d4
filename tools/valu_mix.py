#!/usr/bin/env python3
"""tools/valu_mix.py [out.json] -- static VALU instruction mix of the extractor / matcher kernels by ISSUE-COST CLASS.

Disassembles dvm_slam_amd/csrc/{orb_kernels,octree_kernel,match_kernels}.hip for gfx950 (hipcc --cuda-device-only -S, the same flags
as the product build), splits the listing per kernel and puts every VALU mnemonic into the cost class measured on the MI355X by
tools/valu_issue.hip / valu_issue2.hip (profiles/r02_valu_issue*.jsonl: cycles a wave64 instruction occupies its SIMD at 8 waves
per SIMD, independent chains):
   full   2.25  v_add / sub / and / or / xor / shifts, 16-bit min / max / sub, v_mul_f32, v_add_f32, v_mov, v_cndmask and v_cmp in
                their 32-bit (e32) encodings
   half   4.15  everything else that is 32-bit: 32-bit min / max, v_pk_*, v_perm, v_alignbyte, v_dot*, v_mad*, v_add3, v_lshl_or /
                v_lshl_add, v_and_or, v_bfe, v_mul_lo / hi / u24, conversions, v_fma_f32 (3.66 measured, counted as half), DPP and
                SDWA forms, v_readlane / v_readfirstlane, every e64 (VOP3) encoding, 64-bit integer ops
   f64    4.6   v_add / mul / fma_f64
   slow   8.2   v_min3_u16 / v_max3_u16, v_sqrt_f32, v_rcp / rsq / log / exp_f32 (quarter-rate transcendentals)
   vslow 16.3   v_rcp / rsq / sqrt_f64, v_div_*_f64
The mean cost per static VALU instruction of a kernel x its DYNAMIC wave-instruction count (SQ_INSTS_VALU, a rocprofv3 --pmc pass)
is the kernel's VALU issue time; bench.py turns the sum over a launch group into ONE number, roofline.valu_issue.frac.  Static
mix, dynamic count: loops are not weighted (stated in the output)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dvm_slam_amd", "csrc")
COST = {"full": 2.25, "half": 4.15, "f64": 4.6, "slow": 8.2, "vslow": 16.3}
FULL = re.compile(r"^v_(add|sub|subrev)_(u32|i32|u16|i16|co_u32|f32)|^v_(and|or|xor|not)_b32|^v_(lshrrev|lshlrev|ashrrev)_(b32|i32|b16|i16)|"
                  r"^v_(min|max)_(u16|i16)|^v_mul_f32|^v_mov_b32|^v_cndmask_b32|^v_cmp_|^v_cmpx_|^v_bfrev|^v_subb|^v_addc")


def classify(m):
    if re.match(r"^v_(rcp|rsq|sqrt)_f64|^v_div_", m):
        return "vslow"
    if re.search(r"_f64", m):
        return "f64"
    if re.match(r"^v_(min3|max3)_(u16|i16)|^v_sqrt_f32|^v_(rcp|rsq|log|exp|sin|cos)_f32", m):
        return "slow"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if m.endswith(("_e64", "_dpp", "_sdwa")) or re.search(r"_(b64|u64|i64)$", base):
        return "half"
    if m.endswith("_e32") and FULL.match(base):
        return "full"
    return "half"


def kernels_of(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                               os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    res = {}
    for m in re.finditer(r"; -- Begin function (\S+)\n(.*?); -- End function", txt, re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name)
        short = re.sub(r"\(.*$", "", name)
        body = m.group(2)
        hist, ops = collections.Counter(), collections.Counter()
        other = collections.Counter()
        ldw, stw = collections.Counter(), collections.Counter()      # static global load / store instructions by bytes per lane
        for line in body.splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr")):
                hist[classify(op)] += 1; ops[op] += 1
            wm = re.match(r"^(global|buffer|flat)_(load|store)_(ubyte|sbyte|ushort|sshort|short|byte|dword|dwordx2|dwordx3|dwordx4)(_d16\w*)?$", op)
            if wm:
                w = {"ubyte": 1, "sbyte": 1, "byte": 1, "ushort": 2, "sshort": 2, "short": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}[wm.group(3)]
                (ldw if wm.group(2) == "load" else stw)[w] += 1
            if op.startswith("v_") and False:
                pass
            elif op.startswith(("ds_", "global_", "buffer_", "flat_", "s_load", "s_buffer")):
                other["ds" if op.startswith("ds_") else ("smem" if op.startswith("s_") else "vmem")] += 1
            elif op.startswith("s_"):
                other["salu"] += 1
        n = sum(hist.values())
        if n == 0:
            continue
        res[short if short not in res else name] = {"static_valu": n, "classes": dict(hist), "mean_issue_cycles": sum(COST[c] * k for c, k in hist.items()) / n,
                                                    "other_static": dict(other), "global_load_bytes_per_lane": dict(ldw), "global_store_bytes_per_lane": dict(stw), "top_ops": ops.most_common(12), "signature": name}
    return res


def main():
    out = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "cost_cycles": COST, "kernels": {}}
    for src in ("orb_kernels.hip", "octree_kernel.hip", "match_kernels.hip", "ba_kernels.hip", "ba_window.hip"):
        out["kernels"].update(kernels_of(src))
    js = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(js)
    for k, v in out["kernels"].items():
        print(f"{k:60s} {v['static_valu']:6d} static VALU  mean {v['mean_issue_cycles']:.2f} cycles  {v['classes']}")


if __name__ == "__main__":
    main()

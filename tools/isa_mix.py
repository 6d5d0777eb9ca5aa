#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels, per kernel and per basic block.

  python tools/isa_mix.py cracks_amd/csrc/pfm_cart_phi4.hip --kernel 'k_cart_phi4ILi3ELi0ELb0ELb1E' [--blocks] [--json out.json]

Compiles the source to ISA (hipcc -S --cuda-device-only, the flags of cracks_amd/build.py) and counts, per kernel:
VALU instructions by class (FP64 arithmetic, moves, selects/compares, integer, lane permutes, read/writelane = SGPR
spill traffic), LDS, VMEM, SALU, waits.  With --blocks every basic block of more than --min instructions is listed so
that the loop bodies (the blocks that execute thousands of times) can be told from the straight-line set-up code; the
dynamic mix of profiles/r03 weights these blocks with the trip counts of the kernel's loops."""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(op: str) -> str:
    if op.startswith("v_"):
        if op in ("v_readlane_b32", "v_writelane_b32"):
            return "valu_rdwrlane"
        if op.startswith(("v_permlane", "v_mov_b32_dpp", "v_readfirstlane")) or "_dpp" in op:
            return "valu_lane_permute"
        if op.endswith("_f64") or op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_fmac_f64")):
            if op.startswith(("v_cmp", "v_cmpx")):
                return "valu_cmp_select"
            if op.startswith("v_cvt"):
                return "valu_other"
            return "valu_fp64"
        if op.startswith(("v_mov_b64", "v_mov_b32", "v_accvgpr")):
            return "valu_mov"
        if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
            return "valu_cmp_select"
        return "valu_int_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_barrier", "s_nop", "s_sleep")):
        return "s_sync_nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "s_branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def instructions(body: str):
    for line in body.split("\n"):
        if not line.startswith("\t"):
            continue
        s = line.strip()
        if not s or s[0] in ".;":
            continue
        yield s.split()[0]


def compile_to_asm(src: str) -> str:
    from cracks_amd import build as B

    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    subprocess.check_call([B.hipcc()] + flags + ["-I/opt/rocm/include", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    return out


def analyse(asm_path: str, pattern: str, blocks: bool, min_ins: int):
    txt = open(asm_path).read()
    res = {}
    for m in re.finditer(r"\n(_Z\S+):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S):
        name, body = m.group(1), m.group(2)
        if pattern and not re.search(pattern, name):
            continue
        tot = collections.Counter(classify(op) for op in instructions(body))
        ops = collections.Counter(instructions(body))
        valu = sum(v for k, v in tot.items() if k.startswith("valu"))
        rec = {"instructions": sum(tot.values()), "valu": valu, "classes": dict(sorted(tot.items())),
               "fp64_share_of_valu": round(tot["valu_fp64"] / max(valu, 1), 4),
               "top_valu_ops": dict(collections.Counter({k: v for k, v in ops.items() if k.startswith("v_")}).most_common(14))}
        if blocks:
            parts = re.split(r"\n(\.LBB\d+_\d+):[^\n]*", body)
            labels = ["entry"] + parts[1::2]
            bodies = [parts[0]] + parts[2::2]
            bl = []
            for lab, b in zip(labels, bodies):
                c = collections.Counter(classify(op) for op in instructions(b))
                n = sum(c.values())
                if n >= min_ins:
                    v = sum(x for k, x in c.items() if k.startswith("valu"))
                    bl.append({"block": lab, "n": n, "valu": v, "fp64": c["valu_fp64"], "rdwrlane": c["valu_rdwrlane"],
                               "mov": c["valu_mov"], "cmp_sel": c["valu_cmp_select"], "int": c["valu_int_other"],
                               "permute": c["valu_lane_permute"], "lds": c["lds"], "vmem": c["vmem"], "wait": c["s_waitcnt"],
                               "salu": c["salu"] + c["smem"]})
            rec["blocks"] = bl
        res[name] = rec
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default="", help="regex on the mangled kernel name")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--min", type=int, default=40)
    ap.add_argument("--json", default="")
    ap.add_argument("--asm", default="", help="use this .s instead of compiling")
    a = ap.parse_args()
    asm = a.asm or compile_to_asm(a.source)
    res = analyse(asm, a.kernel, a.blocks, a.min)
    for name, r in res.items():
        print(name)
        print("  instructions %d, VALU %d, FP64 share of VALU %.1f %%" % (r["instructions"], r["valu"], 100 * r["fp64_share_of_valu"]))
        print("  classes:", r["classes"])
        print("  top VALU ops:", r["top_valu_ops"])
        for b in r.get("blocks", []):
            print("   %-12s n=%5d valu=%5d fp64=%5d rdwr=%4d mov=%4d cmpsel=%4d int=%4d perm=%4d lds=%4d vmem=%3d wait=%3d salu=%4d" % (
                b["block"], b["n"], b["valu"], b["fp64"], b["rdwrlane"], b["mov"], b["cmp_sel"], b["int"], b["permute"], b["lds"],
                b["vmem"], b["wait"], b["salu"]))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

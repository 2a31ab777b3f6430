#!/usr/bin/env python3
"""Disassembles one kernel of librmhip.so and lists its loops with instruction counts per issue class and the cycles those
classes cost on this chip (scripts/ubench/valu_rates.hip: cycles per wave64 instruction at >= 2 waves per SIMD).

    python scripts/loop_isa.py 'rm_queue_kernel4_plain<rot=0, scl=0, 1>' [--lib path/to/librmhip.so] [--listing] [--min-valu 40]

A loop is a backward branch: the instructions from its target to the branch.  Only innermost loops with at least --min-valu
VALU instructions are printed (the two march loops, the normal-tap loop); --listing prints their instructions too.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, short_name  # noqa: E402

# measured issue cost per class, cycles per wave64 instruction
COST = {"valu_plain": 2.5, "valu_sgpr_or_literal": 4.2, "valu_minmax_cmp_cndmask": 4.2, "valu_packed": 4.5, "valu_trans": 8.2,
        "valu_dpp_readlane": 4.2}
TRANS = ("v_rsq_", "v_rcp_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
SLOW = ("v_min", "v_max", "v_med3", "v_cmp", "v_cndmask", "v_add3", "v_bfi", "v_mad_u64", "v_mad_i64", "v_lshl_add", "v_and_or", "v_or3")


def code_objects(lib, d):
    fat = os.path.join(d, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True, capture_output=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for k, a in enumerate(starts):
        part, co = os.path.join(d, f"part{k}.bin"), os.path.join(d, f"dev{k}.co")
        with open(part, "wb") as fh:
            fh.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        yield co


def find_kernel(lib, want, d):
    for co in code_objects(lib, d):
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", co], check=True, capture_output=True, text=True).stdout
        for line in syms.splitlines():
            if " FUNC " in line:
                name = line.split()[-1]
                if short_name(name) == want:
                    return co, name
    return None, None


def classify(mnem, ops):
    if not mnem.startswith("v_"):
        if mnem.startswith("s_"):
            return "salu_or_branch"
        if mnem.startswith(("ds_",)):
            return "lds"
        if mnem.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        return "other"
    if mnem.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in ops or "row_" in ops or "quad_perm" in ops:
        return "valu_dpp_readlane"
    if mnem.startswith("v_pk_"):
        return "valu_packed"
    if mnem.startswith(TRANS):
        return "valu_trans"
    if mnem.startswith(SLOW):
        return "valu_minmax_cmp_cndmask"
    # an SGPR, vcc/exec or literal source (inline constants 0, 0.5, 1.0, 2.0, small integers are free)
    srcs = ops.split(",")[1:] if "," in ops else []
    for s_ in srcs:
        s_ = s_.strip().split(" ")[0]
        if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp)", s_):
            return "valu_sgpr_or_literal"
        if re.match(r"^0x[0-9a-f]+$", s_):
            return "valu_sgpr_or_literal"
    return "valu_plain"


def disassemble(co, sym):
    # (--disassemble-symbols stops at the first local label, so the whole code object is disassembled and cut here)
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--symbolize-operands", co], check=True, capture_output=True, text=True).stdout
    ins, labels, inside = [], {}, False
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if re.match(r"^L\d+$", m.group(1)):
                if inside:
                    labels[m.group(1)] = len(ins)
            else:
                inside = m.group(1) == sym
            continue
        if not inside:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            ins.append((m.group(1), m.group(2), m.group(3)))
    return ins, labels


def loops(ins, labels):
    out = []
    for i, (mn, ops, _) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"\b(L\d+)\b", ops)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                out.append((labels[m.group(1)], i))
    # innermost: loops that contain no other loop
    inner = [l for l in out if not any(o != l and o[0] >= l[0] and o[1] <= l[1] for o in out)]
    return sorted(set(inner))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raymarching_amd", "librmhip.so"))
    ap.add_argument("--listing", action="store_true")
    ap.add_argument("--min-valu", type=int, default=40)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        co, sym = find_kernel(args.lib, args.kernel, d)
        if co is None:
            raise SystemExit(f"no kernel {args.kernel!r} in {args.lib}")
        ins, labels = disassemble(co, sym)
    print(f"kernel {args.kernel} ({sym}): {len(ins)} instructions")
    for a, b in loops(ins, labels):
        body = ins[a:b + 1]
        cls = {}
        for mn, ops, _ in body:
            c = classify(mn, ops)
            cls[c] = cls.get(c, 0) + 1
        n_valu = sum(v for k, v in cls.items() if k.startswith("valu"))
        if n_valu < args.min_valu:
            continue
        cyc = sum(COST[k] * v for k, v in cls.items() if k in COST)
        print(f"\nloop at {body[0][2]}..{body[-1][2]}: {len(body)} instructions, {n_valu} VALU")
        for k in sorted(cls):
            print(f"    {k:28s} {cls[k]:4d}" + (f"   x {COST[k]:.1f} = {COST[k] * cls[k]:7.1f} cycles" if k in COST else ""))
        print(f"    modelled VALU issue: {cyc:.0f} cycles per iteration and wave = {cyc / max(n_valu, 1):.2f} per VALU instruction")
        mn_hist = {}
        for mn, _, _ in body:
            mn_hist[mn] = mn_hist.get(mn, 0) + 1
        print("    mnemonics: " + ", ".join(f"{k} {v}" for k, v in sorted(mn_hist.items(), key=lambda kv: -kv[1])))
        if args.listing:
            for mn, ops, addr in body:
                print(f"        {addr}  {mn:24s} {ops}   [{classify(mn, ops)}]")


if __name__ == "__main__":
    main()

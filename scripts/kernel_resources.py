#!/usr/bin/env python3
"""Lists VGPR/SGPR counts and scratch bytes of every kernel in librmhip.so (from the code object's
metadata), so a spill in some variant is seen at build time rather than as a slow frame.

    python scripts/kernel_resources.py [path/to/librmhip.so]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib: str):
    notes = ""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"            # one bundle per translation unit, concatenated in the section
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, a in enumerate(starts):
            part, co = os.path.join(d, f"part{k}.bin"), os.path.join(d, f"dev{k}.co")
            with open(part, "wb") as fh:
                fh.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.split("\n"):
        m = re.match(r"\s+\.(name|private_segment_fixed_size|sgpr_count|vgpr_count|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "name":
            # .name comes after the fields of its kernel in some layouts; start a record at the first field instead
            pass
        if cur is None or k in cur:
            cur = {}
            out.append(cur)
        cur[k] = v
    for r in out:
        r["demangled"] = short_name(r.get("name", "?"))
    return out


def short_name(mangled: str) -> str:
    """rm_queue_kernel4<rot=1, scl=1> from _ZN12_GLOBAL__N_116rm_queue_kernel4ILb1ELi1EEEv8RmParams; further template
    arguments are appended as they are (rm_coop_kernelN<rot=1, scl=1, 0>)"""
    m = re.search(r"\d+(rm_[a-z_]+[4NF]?(?:_fx|_fast|_cull|_coop|_plain|_skip)?)(?:I((?:L[bi]\d+E)+)E)?", mangled)
    if not m:
        return mangled
    if m.group(2) is None:
        return m.group(1)
    args = re.findall(r"L[bi](\d+)E", m.group(2))
    names = ["rot=" + args[0]] + (["scl=" + args[1]] if len(args) > 1 else []) + args[2:]
    return f"{m.group(1)}<{', '.join(names)}>"


def short_name_from_demangled(name: str) -> str:
    """rm_queue_kernel4<rot=0, scl=0> from 'void (anonymous namespace)::rm_queue_kernel4<false, 0>(RmParams)'"""
    m = re.search(r"(rm_[a-z_]+[4NF]?(?:_fx|_fast|_cull|_coop|_plain|_skip)?)(?:<([^>]*)>)?", name)
    if not m:
        return name
    if m.group(2) is None:
        return m.group(1)
    args = [{"true": "1", "false": "0"}.get(a.strip(), a.strip()) for a in m.group(2).split(",")]
    names = ["rot=" + args[0]] + (["scl=" + args[1]] if len(args) > 1 else []) + args[2:]
    return f"{m.group(1)}<{', '.join(names)}>"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raymarching_amd", "librmhip.so")
    for r in kernel_resources(lib):
        nm = r["demangled"]
        print(f"{nm:44s} vgpr {r.get('vgpr_count', '?'):>4s} sgpr {r.get('sgpr_count', '?'):>4s} scratch {r.get('private_segment_fixed_size', '?'):>5s} B")


if __name__ == "__main__":
    main()

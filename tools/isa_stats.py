#!/usr/bin/env python3
"""Per-kernel instruction / register summary of libcama_hip's gfx950 ISA (no GPU needed).

    python tools/isa_stats.py [substring-of-kernel-name ...] [-D MACRO ...]

Compiles cama_amd/csrc/cama_hip.hip with the product's flags to assembly (--cuda-device-only -S) and prints, for
every kernel whose demangled name contains one of the substrings: VGPRs, SGPRs, LDS bytes, scratch, and counts of
the instruction classes that decide what a kernel is bound by (VMEM, LDS reads/writes/atomics, scalar loads,
fp64 VALU, waitcnts, barriers)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = [("s_load", r"\ts_load"), ("vmem_ld", r"\t(global|buffer|flat)_load"), ("vmem_st", r"\t(global|buffer|flat)_store"),
           ("atom_g", r"\tglobal_atomic"), ("ds_rd", r"\tds_read"), ("ds_wr", r"\tds_write"),
           ("ds_atom", r"\tds_(add|max|min|or|and|cmpst)"), ("f64", r"\tv_(fma|mul|add|div_\w+|rcp|trig|cmp\w*)_f64"),
           ("dot4", r"\tv_dot4"), ("valu", r"\tv_"), ("salu", r"\ts_(?!load|waitcnt|barrier|nop)"), ("waitcnt", r"\ts_waitcnt"),
           ("barrier", r"\ts_barrier")]


def main():
    args = sys.argv[1:]
    defs, pats = [], []
    while args:
        a = args.pop(0)
        if a == "-D":
            defs.append("-D" + args.pop(0))
        elif a.startswith("-D"):
            defs.append(a)
        else:
            pats.append(a)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "cama.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
               "-ffp-contract=off", "-I" + os.path.join(REPO, "include"), "--cuda-device-only", "-S", "-o", out,
               os.path.join(REPO, "cama_amd", "csrc", "cama_hip.hip")] + defs
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    names = re.findall(r"^\t\.amdhsa_kernel (\S+)", text, re.M)
    demangled = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':58s} vgpr sgpr   lds scr " + " ".join(f"{c[0]:>7s}" for c in CLASSES))
    for mangled, nice in zip(names, demangled):
        short = re.sub(r"\(anonymous namespace\)::", "", nice)
        short = re.sub(r"\(.*$", "", short).replace("void ", "")
        if pats and not any(p in short for p in pats):
            continue
        m = re.search(r"^" + re.escape(mangled) + r":.*?\n(.*?)\n\t\.amdhsa_kernel " + re.escape(mangled) + r"\n(.*?)\.end_amdhsa_kernel",
                      text, re.S | re.M)
        if not m:
            continue
        body, desc = m.group(1), m.group(2)
        g = lambda key: int(re.search(r"\." + key + r" (\d+)", desc).group(1)) if re.search(r"\." + key + r" (\d+)", desc) else -1
        print(f"{short[:58]:58s} {g('amdhsa_next_free_vgpr'):4d} {g('amdhsa_next_free_sgpr'):4d} "
              f"{g('amdhsa_group_segment_fixed_size'):5d} {g('amdhsa_private_segment_fixed_size'):3d} " +
              " ".join(f"{len(re.findall(p, body)):7d}" for _, p in CLASSES))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 code the library is built from: VGPRs, AGPRs, SGPRs, scratch (spills), static LDS, code size
and the occupancy (waves per SIMD) the compiler derives from them -- for EVERY kernel instantiation in the three device translation
units, whether a benchmark happens to launch it or not (rocprofv3's kernel statistics under profiles/ only list what ran).

    python tools/kernel_resources.py > profiles/<round>_kernel_resources.txt

Runs on the CPU: hipcc with the library's own flags (csrc/build.py: FLAGS) + `--cuda-device-only -S`, then the "; Kernel info" comment
block the AMDGPU backend prints behind each `.amdhsa_kernel`.  Dynamic LDS (the `extern __shared__` arrays the launches size on the host)
is not in the static figure; the kernels that use it are marked.  The first lines carry the source stamp (bench.py: source_stamp)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "elevation_mapping_cupy_amd", "csrc")


def demangle(names):
    filt = "c++filt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def short(sig):
    """k_name<template args> without the parameter list"""
    depth, cut = 0, len(sig)
    for i, ch in enumerate(sig):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return sig[:cut].replace("void ", "", 1).strip()


def kernels_of(src):
    from elevation_mapping_cupy_amd.csrc import build as hb
    flags = [f for f in hb.FLAGS if not f.startswith("-W")]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "dev.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-w", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    recs = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?); Occupancy: (\d+)", txt, flags=re.S):
        name, body, occ = m.group(1), m.group(2), int(m.group(3))
        g = lambda pat, d=0: int((re.search(pat, body) or [None, d])[1])      # noqa: E731
        # the kernel's instruction stream: from its label to the descriptor directive (static counts, every path once)
        i0 = txt.find("\n%s:" % name)
        code = txt[i0:m.start()] if i0 >= 0 else ""
        mn = re.findall(r"^\t([a-z][a-z0-9_]+)(?:\s|$)", code, flags=re.M)
        mix = {"valu": sum(x.startswith("v_") for x in mn), "salu": sum(x.startswith("s_") and x != "s_nop" and not x.startswith("s_waitcnt") for x in mn),
               "ldsi": sum(x.startswith("ds_") for x in mn), "vmem": sum(x.startswith(("global_", "buffer_", "flat_", "scratch_")) for x in mn),
               "nop": sum(x == "s_nop" for x in mn), "wait": sum(x.startswith("s_waitcnt") for x in mn)}
        recs.append({"mangled": name, "vgpr": g(r"; NumVgprs: (\d+)"), "agpr": g(r"; NumAgprs: (\d+)"), "sgpr": g(r"; TotalNumSgprs: (\d+)"),
                     "scratch": g(r"; ScratchSize: (\d+)"), "lds": g(r"; LDSByteSize: (\d+)"), "code": g(r"; codeLenInByte = (\d+)"),
                     "occupancy": occ, **mix})
    return recs


def main():
    import bench
    print("# per-kernel resources of the gfx950 code objects (tools/kernel_resources.py: hipcc %s --cuda-device-only -S)" % "-O3 -ffp-contract=off")
    print("# source_stamp: %s" % bench.source_stamp())
    print("# vgpr / agpr / sgpr: registers allocated; scratch: bytes per lane (spills -- 0 = none); lds: STATIC bytes per workgroup (dynamic LDS")
    print("# sized by the host launch is not included); code: bytes of ISA; occ: waves per SIMD the register / static-LDS budget allows (max 8)")
    print("# right of the bar: STATIC instruction counts of the kernel's code (every path once, not what a wave executes): VALU, SALU (without s_nop /")
    print("# s_waitcnt), LDS, vector memory, s_nop (hazard wait states the compiler inserted), s_waitcnt")
    print("# (the vgpr column of profiles/*_kernel_stats.txt is rocprofv3's VGPR_Count, which reads half of NumVgprs on this stack: k_post<32, 0> 24 there, 48 here)")
    total, spilled = 0, []
    for src in ("emap_kernels.hip", "emap_binned.hip", "emap_semantic.hip"):
        recs = kernels_of(src)
        names = demangle([r["mangled"] for r in recs])
        print("\n## %s: %d kernel instantiations" % (src, len(recs)))
        print("%-78s %5s %5s %5s %8s %7s %7s %4s | %5s %5s %5s %5s %5s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "code", "occ", "valu", "salu", "lds", "vmem", "nop", "wait"))
        for r in sorted(recs, key=lambda r: short(names[r["mangled"]])):
            n = short(names[r["mangled"]])
            print("%-78s %5d %5d %5d %8d %7d %7d %4d | %5d %5d %5d %5d %5d %5d" % (n[:78], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["code"], r["occupancy"],
                                                                                      r["valu"], r["salu"], r["ldsi"], r["vmem"], r["nop"], r["wait"]))
            total += 1
            if r["scratch"]:
                spilled.append((n, r["scratch"]))
    print("\n# %d kernel instantiations; with scratch (spills): %d" % (total, len(spilled)))
    for n, s in spilled:
        print("#   %s: %d bytes per lane" % (n, s))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The ISA of one four-step group of the visibility march (k_rays, the variant the 1024^2 benchmarks run: reference_fp16 index formula,
LDS bitmap, 1024 threads) as hipcc emits it for gfx950 with the library's flags -- the evidence behind DESIGN.md's per-step instruction
counts (section 5b / 5f).  Runs on the CPU.

    python tools/ray_step_isa.py > profiles/<round>_k_rays_step_isa.txt

Lines marked (asm) come from the kernel's inline assembly, the others are the compiler's; `s_nop` = a wait state the gfx950 hazard
recognizer inserted (a VOP3P result read by the next VALU needs one; behind an inline-asm producer it must assume so)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = "_Z6k_raysILi0ELb0ELi2ELb0ELi1024ELb1ELi1EE"


def main():
    import bench
    from elevation_mapping_cupy_amd.csrc import build as hb
    flags = [f for f in hb.FLAGS if not f.startswith("-W")]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-w", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "elevation_mapping_cupy_amd", "csrc", "emap_kernels.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    i0 = txt.index("\n" + KERNEL)
    body = txt[i0:txt.index(".amdhsa_kernel " + KERNEL)].splitlines()
    # the group: four bitmap-word reads in a row, each behind its own sample / index arithmetic (the first such run is the loop body's)
    rd = [i for i, ln in enumerate(body) if ln.strip().startswith("ds_read_b32")]
    g = next(k for k in range(len(rd) - 3) if all(20 < rd[k + j + 1] - rd[k + j] < 90 for j in range(3))
             and any("v_cvt_flr_i32_f32" in ln for ln in body[rd[k] - 60:rd[k]]))
    start = max(i for i in range(rd[g]) if body[i].strip().startswith("v_pk_mul_f32"))
    out, reads, in_asm = [], 0, False
    for ln in body[start:]:
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not s or s.startswith((";", ".")):
            continue
        out.append("%-7s %s" % ("(asm)" if in_asm else "", s))
        if s.startswith("ds_read_b32"):
            reads += 1
            out.append("")
            if reads == 4:
                break
    mn = [re.split(r"\s+", o.strip().replace("(asm)", "").strip())[0] for o in out if o.strip()]
    print("# one four-step group of the march in k_rays<0, false, 2, false, 1024, true, 1> (tools/ray_step_isa.py; hipcc -O3 -ffp-contract=off, gfx950)")
    print("# source_stamp: %s" % bench.source_stamp())
    print("# per group of four steps up to the bitmap words: %d VALU, %d LDS reads, %d s_nop (wait states); per step: %.1f / 1 / %.1f"
          % (sum(m.startswith("v_") for m in mn), sum(m.startswith("ds_") for m in mn), sum(m == "s_nop" for m in mn),
             sum(m.startswith("v_") for m in mn) / 4.0, sum(m == "s_nop" for m in mn) / 4.0))
    print("# (the rest of a step -- clamp of the step value, the test of the four words and the branch per group -- follows the group)")
    print()
    print("\n".join(out))


if __name__ == "__main__":
    main()

"""Builds libemap_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: five translation units, compiled in parallel into
csrc/_obj/*.o (only the stale ones) and linked."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "libemap_hip.so")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["emap_kernels.hip", "emap_binned.hip", "emap_semantic.hip", "emap_api.hip", "emap_inpaint_host.hip", "emap_inpaint_ns.cpp"]      # (.cpp: host-only C++)
HEADERS = ["emap_device.h", os.path.join("..", "..", "include", "emap_hip.h")]
DEPS = SOURCES + HEADERS
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS)


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in [src] + HEADERS + ["build.py"])


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]

    def compile_one(pair):
        src, obj = pair
        if not force and not _obj_stale(src, obj):
            return
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(compile_one, zip(SOURCES, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

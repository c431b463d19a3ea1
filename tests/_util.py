"""Test helpers: build matching (HIP ElevationMap, OracleMap) pairs from one config dict."""
import numpy as np

from oracle import emap_oracle as eo

PLANE_NAMES = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]


def make_parameter(cfg, cell_n, mode="reference_fp16", weights=None):
    from elevation_mapping_cupy_amd.configs import parameter_from
    full = dict(eo.DEFAULTS)
    full.update(cfg)
    return parameter_from(full, cell_n, mode, weights)


def make_pair(cfg, cell_n, mode="reference_fp16", weights=None):
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    hip = ElevationMap(make_parameter(cfg, cell_n, mode, weights))
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=cell_n, mode=mode, weights=weights))
    return hip, orc


def assert_planes_close(a, b, atol=1e-5, rtol=1e-5, names=PLANE_NAMES, max_bad=0, what=""):
    """fused height/variance within 1e-5 (north_star); flags exact."""
    for k in range(a.shape[0]):
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        bad = ~(np.abs(x - y) <= atol + rtol * np.abs(y))
        bad &= ~(np.isnan(x) & np.isnan(y))
        n = int(bad.sum())
        name = names[k] if k < len(names) else str(k)
        assert n <= max_bad, "%s plane %s: %d cells differ, max |d| = %g (first at %s: %r vs %r)" % (
            what, name, n, np.nanmax(np.abs(x - y)), tuple(np.argwhere(bad)[0]), x[bad][0], y[bad][0])


def assert_planes_equal(a, b, names=PLANE_NAMES, what=""):
    """BIT-FOR-BIT equality of whole planes.  The oracle accumulates the same integers / fixed-point sums as the HIP kernels
    (oracle/emap_oracle.c) and both use the same correctly rounded float operations, so any difference is a defect, not rounding."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    for k in range(a.shape[0]):
        x, y = a[k].view(np.uint32), b[k].view(np.uint32)
        if not np.array_equal(x, y):
            bad = x != y
            name = names[k] if k < len(names) else str(k)
            i = tuple(np.argwhere(bad)[0])
            raise AssertionError("%s plane %s: %d cells differ bitwise, max |d| = %g (first at %s: %r vs %r)" % (
                what, name, int(bad.sum()), float(np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))), i, a[k][i], b[k][i]))


def rccl_stand_in(kind="blocking"):
    """Builds one of the in-process stand-ins for the ten RCCL entry points (tests/fake_rccl/) and returns the path to hand to
    emap_comm_init: "blocking" = fake_rccl.cpp (host-synchronised copies, g++), "stream" = stream_rccl.hip (stream-ordered events +
    a reduction kernel, hipcc).  Ranks are threads of the calling process on one GPU.  Prebuilt by __graft_entry__.build() into
    tests/fake_rccl/_build/ (travels to the GPU box); rebuilt here when stale."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl")
    src = os.path.join(here, "fake_rccl.cpp" if kind == "blocking" else "stream_rccl.hip")
    out_dir = os.path.join(here, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "lib%s_rccl.so" % ("fake" if kind == "blocking" else "stream"))
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        tmp = out + ".%d.tmp" % os.getpid()
        if kind == "blocking":
            cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-w", src, "-o", tmp,
                   "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
        else:
            cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", src, "-o", tmp]
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    return out


def cu_hog():
    """Builds tests/foreign/cu_hog.hip (a foreign kernel that holds compute units: test infrastructure for the grid barriers of
    k_small_frame) and returns the loaded library.  Prebuilt by __graft_entry__.build() into tests/foreign/_build/."""
    import ctypes as ct
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "foreign")
    src = os.path.join(here, "cu_hog.hip")
    out_dir = os.path.join(here, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libcu_hog.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", src, "-o", tmp])
        os.replace(tmp, out)
    lib = ct.CDLL(out)
    lib.hog_start.argtypes = [ct.c_int, ct.c_int, ct.c_double]
    return lib

"""CPU: the two in-process RCCL stand-ins of the multi-rank GPU tests (tests/fake_rccl/) decode the data-type and reduction-operator
arguments as plain integers.  Those integers must be the values of the REAL rccl.h the library is compiled against (csrc/emap_api.hip
passes ncclFloat64 / ncclFloat32 / ncclInt64 / ncclUint32 / ncclChar, ncclSum / ncclMax): otherwise the stand-ins would test another
protocol than the one RCCL sees on a multi-GPU node."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

RCCL_H = "/opt/rocm/include/rccl/rccl.h"


@pytest.mark.skipif(not os.path.exists(RCCL_H), reason="no rccl.h on this machine")
def test_stand_in_constants_are_rccl_s(tmp_path):
    src = tmp_path / "rc.cpp"
    src.write_text('#include <rccl/rccl.h>\n#include <cstdio>\nint main() { printf("%d %d %d %d %d %d %d %d %d\\n", (int)ncclChar, (int)ncclUint32, '
                   '(int)ncclInt64, (int)ncclFloat32, (int)ncclFloat64, (int)ncclSum, (int)ncclMax, (int)ncclSuccess, (int)sizeof(ncclUniqueId)); }\n')
    exe = tmp_path / "rc"
    subprocess.check_call(["g++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o", str(exe)])
    char_, u32, i64, f32, f64, sum_, max_, ok, uid = map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert (char_, u32, i64, f32, f64, sum_, max_, ok, uid) == (0, 3, 4, 7, 8, 0, 2, 0, 128)
    for f in ("fake_rccl.cpp", "stream_rccl.hip"):
        txt = open(os.path.join(ROOT, "tests", "fake_rccl", f)).read()
        # 8-byte element types: float64 and int64; float32; byte counts for send / recv; sum is op 0 (anything else is reduced as max)
        assert re.search(r"dtype == %d \|\| dtype == %d\) \? 8 : 4" % (f64, i64), txt), f
        assert re.search(r"dtype == %d\) \w+<double>" % f64, txt) and re.search(r"dtype == %d\) \w+<float>" % f32, txt) and re.search(r"dtype == %d\) \w+<long long>" % i64, txt), f
        assert "op == %d" % sum_ in txt, f
        assert "char internal[128]" in txt, f
    # ... and the library asks only for what the stand-ins implement
    api = open(os.path.join(ROOT, "elevation_mapping_cupy_amd", "csrc", "emap_api.hip")).read()
    used_types = set(re.findall(r"\b(ncclFloat64|ncclFloat32|ncclFloat16|ncclInt64|ncclUint64|ncclInt32|ncclUint32|ncclInt8|ncclUint8|ncclChar|ncclBfloat16)\b", api))
    used_ops = set(re.findall(r"\b(ncclSum|ncclProd|ncclMax|ncclMin|ncclAvg)\b", api))
    assert used_types <= {"ncclFloat64", "ncclFloat32", "ncclInt64", "ncclUint32", "ncclChar"}, used_types
    assert used_ops <= {"ncclSum", "ncclMax"}, used_ops

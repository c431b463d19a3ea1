"""CPU: measurement hygiene.  Every file of the current round under profiles/ (bench.PROFILE_ROUND) was taken with the kernel sources
in the tree: its `source_stamp` equals bench.source_stamp().  A kernel edit without a re-take of the profiles fails here (and bench.py
then stops quoting the stale rocprofv3 / PMC figures next to its live ones, bench.py: rocprof_kernel_us / roofline)."""
import glob
import json
import os
import re

from conftest import ROOT


def test_every_profile_of_the_round_carries_the_trees_stamp():
    import bench
    now = bench.source_stamp()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND + "_*")))
    assert len(files) >= 10, files
    for f in files:
        txt = open(f).read()
        if f.endswith(".json"):
            d = json.loads(txt)
            stamp = d.get("source_stamp") or (d.get("roofline") or {}).get("source_stamp")
        else:
            m = re.search(r"^# source_stamp: ([0-9a-f]{16})", txt, flags=re.M)
            stamp = m.group(1) if m else None
        assert stamp == now, "%s: stamp %s, sources %s" % (os.path.basename(f), stamp, now)


def test_the_quoted_profile_files_exist_and_parse():
    """what bench.py reads next to its live numbers: the kernel statistics and the PMC summary of the default workloads"""
    import bench
    for wl, kern in (("cfg2", "k_post"), ("cfg3", "k_rays<")):
        us, src = bench.rocprof_kernel_us(wl, kern)
        assert us and us > 1.0 and "profiles/" in src, (wl, us, src)
        pj = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (bench.PROFILE_ROUND, wl))))
        assert any(k.startswith(kern) and v["hbm_bytes"] > 0 for k, v in pj["kernels"].items())

import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import _fixtures as fx
from _util import cu_hog, make_pair
from oracle import emap_oracle as eo
hog = cu_hog()
t0 = time.perf_counter(); print("start", hog.hog_start(0, 500, 60.0)); print("wait", hog.hog_wait(), "ms", (time.perf_counter() - t0) * 1e3)
cfg = dict(eo.YAML, enable_visibility_cleanup=False)
one, _ = make_pair(cfg, 202, "reference_fp16")
R, t = fx.POSES["rotated"]
p = fx.cloud(202, 50000, 0)
one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0); one.sync()
for groups in (256, 480, 500, 508, 512, 1024):
    for spin in (2000, 200):
        os.environ["EMAP_SF_SPIN_LIMIT"] = str(spin)
        hog.hog_start(0, groups, 40.0)
        time.sleep(0.002)
        t0 = time.perf_counter()
        one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        one.sync()
        dt = (time.perf_counter() - t0) * 1e3
        hog.hog_wait()
        print("groups", groups, "spin", spin, "frame ms %.2f" % dt, "aborts", one.small_frame_aborts(), one.last_update_path())

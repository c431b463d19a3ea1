"""debug: binned vs atomic scatter vs oracle on warm 1024^2 frames; prints per-plane mismatch counts"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo
eo.set_threads(16)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz")); w = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
C, N = 1024, 1_000_000
cfg = dict(eo.YAML, enable_visibility_cleanup=False)
hb, orc = make_pair(cfg, C, "reference_fp16", w)
ha, _ = make_pair(cfg, C, "reference_fp16", w)
hb.set_scatter_mode("binned"); ha.set_scatter_mode("atomic")
R, t = fx.POSES["rotated"]
for f, dz in enumerate((0.0, -0.03, -0.07, 0.02)):
    p = fx.cloud(C, N, f, dz=dz)
    sb = hb.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
    sa = ha.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
    orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
    print("frame", f, "err_cnt b/a/o", sb.err_cnt, sa.err_cnt, orc.last["err_cnt"], "err_sum", sb.err_sum, sa.err_sum, orc.last["err_sum"], "shift", sb.shift, sa.shift, orc.last["shift"])
    for _ in range(5):
        hb.update_time(); ha.update_time(); orc.update_time()
    mb, ma, mo = hb.elevation_map, ha.elevation_map, orc.elevation_map
    for k in range(7):
        dba = int((mb[k].view(np.uint32) != ma[k].view(np.uint32)).sum()); dbo = int((mb[k].view(np.uint32) != mo[k].view(np.uint32)).sum()); dao = int((ma[k].view(np.uint32) != mo[k].view(np.uint32)).sum())
        if dba or dbo or dao:
            bad = np.argwhere(mb[k].view(np.uint32) != mo[k].view(np.uint32))
            print("  plane", k, "binned!=atomic", dba, "binned!=oracle", dbo, "atomic!=oracle", dao, "max|b-o|", float(np.nanmax(np.abs(mb[k] - mo[k]))), "first", bad[:3].tolist())
    ti_b, ti_o = hb.traversability_input, orc.traversability_input
    print("  trav_input diff cells:", int((ti_b != ti_o).sum()), "max", float(np.abs(ti_b - ti_o).max()))

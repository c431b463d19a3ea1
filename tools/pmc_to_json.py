#!/usr/bin/env python3
"""rocprofv3 PMC databases (separate FETCH_SIZE / WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes) -> per-kernel
HBM traffic per launch (median over the launches).  Correction (gfx950, calibrated on k_var_time which reads and writes exactly 32 B/cell of a
1024^2 map: FETCH_SIZE 16 394 KB vs 32 768 KB read, WRITE_SIZE 32 768 KB vs 32 768 KB written):
bytes_read = 2 * FETCH_SIZE KB * 1024, bytes_written = WRITE_SIZE KB * 1024.
usage: pmc_to_json.py <fetch.db> <write.db> [source stamp] > profiles/rNN_pmc_<workload>.json"""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    """kernel -> (MEDIAN over its dispatches, number of dispatches): one row per dispatch in the database; the median is the steady
    state (the first frames after clear() -- bench.py's cold-start block -- move several times the bytes of a warm frame)"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where counter_name=? order by kernel_name, dispatch_id", (counter,)).fetchall()
    by = {}
    for name, v in rows:
        by.setdefault(re.sub(r"\(.*", "", name).replace("void ", "").strip(), []).append(v)
    out = {}
    for k, vals in by.items():
        vals.sort()
        n = len(vals)
        out[k] = ((vals[n // 2] if n % 2 else 0.5 * (vals[n // 2 - 1] + vals[n // 2])), n)
    return out


def main(fetch_db, write_db, stamp=None):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]
        out[k] = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "hbm_read_bytes": int(2 * fk * 1024),
                  "hbm_write_bytes": int(wk * 1024), "hbm_bytes": int(2 * fk * 1024 + wk * 1024), "launches": f.get(k, (0, 0))[1]}
    print(json.dumps({"correction": "read = 2 x FETCH_SIZE (gfx950, calibrated on k_var_time), write = WRITE_SIZE", "source_stamp": stamp, "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)

#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel count / avg / min / max duration and share.
usage: python tools/rocprof_summary.py <results.db> [> profiles/rNN_<what>.txt]"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                       "from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    # median: the first frames after clear() are not steady state (e.g. the ray pass's atomic storm on an unknown map) and pull the mean
    med = {}
    for name, dur in cur.execute("select name, end-start from kernels order by name, end-start").fetchall():
        med.setdefault(name, []).append(dur)
    med = {k: v[len(v) // 2] for k, v in med.items()}
    print("%-86s %6s %10s %10s %10s %10s %6s %5s %5s %7s %7s" % ("kernel", "calls", "avg_us", "median_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
    for r in rows:
        print("%-86s %6d %10.2f %10.2f %10.2f %10.2f %6.1f %5s %5s %7s %7s" % (r[0][:86], r[1], r[2] / 1e3, med[r[0]] / 1e3, r[3] / 1e3, r[4] / 1e3, 100 * r[5] / tot, r[6], r[7], r[8], r[9]))
    try:
        pm = cur.execute("select k.name, p.name, avg(e.value), count(*) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id "
                         "join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol k on d.kernel_id=k.id "
                         "group by k.name, p.name order by k.name").fetchall()
        if pm:
            print("\nPMC counters (average per dispatch)")
            for r in pm:
                print("%-70s %-28s %16.1f  (n=%d)" % (r[0][:70], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        print("(no PMC tables: %s)" % e)


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (sqlite) as text:
per-kernel calls / total / average / min / max duration. Usage: prof_summary.py results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size, scratch_size, end-start from kernels"))
    agg = {}
    for name, gx, wx, vg, sg, lds, scr, dur in rows:
        a = agg.setdefault(name, dict(calls=0, total=0, mn=1 << 62, mx=0, wg=wx, vgpr=vg, sgpr=sg, lds=lds, scratch=scr))
        a["calls"] += 1; a["total"] += dur; a["mn"] = min(a["mn"], dur); a["mx"] = max(a["mx"], dur)
    tot = sum(a["total"] for a in agg.values()) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s}  wg vgpr sgpr lds scratch")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["total"]):
        print(f"{name[:70]:70s} {a['calls']:6d} {a['total']/1e6:10.3f} {a['total']/a['calls']/1e6:10.3f} {a['mn']/1e6:10.3f} {a['mx']/1e6:10.3f} "
              f"{100*a['total']/tot:6.2f}  {a['wg']} {a['vgpr']} {a['sgpr']} {a['lds']} {a['scratch']}")


if __name__ == "__main__":
    main(sys.argv[1])

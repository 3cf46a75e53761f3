#!/usr/bin/env python3
"""Per-stream timeline of a rocprofv3 --kernel-trace database: for every stream, first start / last
end and busy time; plus the lms kernels' launches in start order. Usage: prof_timeline.py db [t_from_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id, grid_x, workgroup_x from kernels order by start"))
t0 = rows[0][1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.
streams = {}
for name, s, e, st, q, gx, wx in rows:
    if (s - t0) / 1e6 < skip: continue
    a = streams.setdefault((st, q), [s, e, 0, 0])
    a[0] = min(a[0], s); a[1] = max(a[1], e); a[2] += e - s; a[3] += 1
for k, a in sorted(streams.items(), key=lambda kv: kv[1][0]):
    print(f"stream {k[0]} queue {k[1]}: {(a[0]-t0)/1e6:9.2f} .. {(a[1]-t0)/1e6:9.2f} ms  busy {a[2]/1e6:9.2f} ms  {a[3]} kernels")
for name, s, e, st, q, gx, wx in rows:
    if (s - t0) / 1e6 < skip: continue
    if "k_lms" in name or "k_cost" in name:
        print(f"{(s-t0)/1e6:9.2f} {(e-s)/1e6:8.2f} ms  st{st} wgs {gx//max(wx,1):4d}  {name[:48]}")

#!/usr/bin/env python3
"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (rocpd sqlite). usage: sq_summary.py results.db [kernel substring]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "lms"
agg = collections.defaultdict(dict)
for name, cn, cnt, s in db.execute("select kernel_name,counter_name,count(*),sum(value) from counters_collection group by kernel_name,counter_name"):
    if pat in name:
        agg[name.split("(")[0].replace("void ", "")][cn] = (cnt, s)
for k, v in agg.items():
    print(k)
    for cn, (cnt, s) in sorted(v.items()):
        print(f"   {cn:28s} dispatches {cnt:4d}  sum {s:16.0f}  per dispatch {s/cnt:14.0f}")

"""Idle gaps of the whole GPU (no kernel running) in a rocprofv3 --kernel-trace database, within the last burst of kernels (a burst
= kernels separated by less than 20 ms): count, total, the largest. usage: idle_gaps.py db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, name from kernels order by start"))
# last burst
k = len(rows) - 1
while k > 0 and rows[k][0] - max(r[1] for r in rows[max(0, k - 50):k]) < 20e6: k -= 1
rows = rows[k:]
t0 = rows[0][0]; cur = rows[0][1]; gaps = []
for s, e, n in rows[1:]:
    if s > cur: gaps.append((s - cur, (cur - t0) / 1e6, n))
    cur = max(cur, e)
tot = sum(g[0] for g in gaps)
print(f"burst {(cur - t0) / 1e6:.1f} ms, {len(rows)} kernels, {len(gaps)} idle gaps, {tot / 1e6:.2f} ms idle in all")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0] / 1e3:8.1f} us at {g[1]:8.2f} ms, before {g[2][:50]}")

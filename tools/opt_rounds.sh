#!/bin/bash
# per-round spans of the optimizer alone (15 frames): rocprofv3 kernel trace of tools/opt_only.py -> tools/rounds_summary.py
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/optr; mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 tools/opt_only.py ${1:-15} > $O/run.log 2>&1 || exit 1
python3 tools/rounds_summary.py $O/kt/kt_results.db
python3 - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/optr/kt/kt_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
fins = [i for i, r in enumerate(rows) if "k_cost_fin" in r[0]][-9:]
# last call: kernels between the first and the last k_cost_fin; gaps with NO kernel running
lo = rows[fins[0] - 200][1] if fins[0] > 200 else rows[0][1]
seg = [r for r in rows if r[1] >= rows[fins[0]][1] and r[2] <= rows[fins[-1]][2] + 1]
ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
idle = 0; cur = 0; last = None
for t, d in ev:
    if cur == 0 and last is not None: idle += t - last
    cur += d; last = t
print(f"between the first and last cost kernel of the last call: {(seg[-1][2]-seg[0][1])/1e6:.1f} ms, of which no kernel running {idle/1e6:.2f} ms")
import collections
tot = collections.Counter()
for r in seg: tot[r[0].split('(')[0][:40]] += r[2] - r[1]
for k, v in tot.most_common(12): print(f"   {k:42s} {v/1e6:8.2f} ms summed")
PY
rm -rf $O/kt

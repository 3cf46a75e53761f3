#!/bin/bash
# the first kernels of the optimizer's first round (frames = $1): how long is the OLS stage alone on the chip before the lead stage starts?
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/r0h; mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 tools/opt_only.py ${1:-75} > $O/run.log 2>&1 || exit 1
python3 - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/r0h/kt/kt_results.db")
rows = list(db.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
fins = [i for i, r in enumerate(rows) if "k_cost_fin" in r[0]]
# the last call: 8 cost rounds -> its first round starts after the k_cost_fin that is 9th from the end (end of the call before)
i0 = fins[-9] + 1 if len(fins) >= 9 else 0
t0 = rows[i0][1]
for r in rows[i0:i0 + 14]:
    print(f"{(r[1]-t0)/1e6:8.2f} .. {(r[2]-t0)/1e6:8.2f} ms  {r[0].split('(')[0][:44]:44s} grid {r[3]}x{r[4]}")
PY
rm -rf $O/kt

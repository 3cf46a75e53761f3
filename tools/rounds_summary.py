"""Per-round times of the optimizer from a rocprofv3 --kernel-trace database of tools/opt_only.py: a round ends with its cost
kernels (k_cost_fin); prints, for the last call in the trace, each round's span and the busy time of its lead-stage launches."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
fins = [i for i, r in enumerate(rows) if "k_cost_fin" in r[0]]
fins = fins[-9:]  # baseline + 8 rounds of the last call
prev_end = rows[fins[0] - 1][1] if fins else 0
start = None
for n, i in enumerate(fins):
    lo = fins[n - 1] + 1 if n else max(0, i - 400)
    seg = rows[lo:i + 1]
    if n == 0:
        # the call's first kernels: find the gap before them
        k = len(seg) - 1
        while k > 0 and seg[k][1] - seg[k - 1][2] < 20e6: k -= 1
        seg = seg[k:]
    t0 = seg[0][1]; t1 = seg[-1][2]
    lead = [r for r in seg if "k_lms<1280" in r[0] or "k_lms1c" in r[0] or "k_lms2" in r[0]]
    h2 = [r for r in seg if "k_lms<256" in r[0]]
    print(f"round {n}: {(t1 - t0) / 1e6:7.1f} ms  kernels {len(seg):4d}  lead launches {len(lead):3d} busy {sum(r[2]-r[1] for r in lead)/1e6:7.1f} ms"
          f"  (256,128) launches {len(h2):3d} busy {sum(r[2]-r[1] for r in h2)/1e6:7.1f} ms   gap before {(t0 - prev_end) / 1e6:5.2f} ms")
    prev_end = t1

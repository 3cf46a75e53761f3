"""How many significant bits do this CPU's rsqrtss table entries carry? (for packing the table in LDS)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
tab, bits = sacopt.probe_rsqrt()
u = tab.view(np.uint32)
print("bits", bits, "entries", len(u))
for k in (4, 8, 9, 10, 11, 12):
    print(f" low {k} bits zero in all entries:", bool(np.all((u & ((1 << k) - 1)) == 0)))
print(" distinct exponent fields:", sorted(set((u >> 23).tolist())))
print(" distinct values:", len(set(u.tolist())))

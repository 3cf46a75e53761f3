#!/usr/bin/env python3
"""Per-kernel sums of the 75-frame rocprofv3 --pmc passes (tools/profile_round.sh): pmc75_summary.py <out.json> <results.db>...
HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024."""
import collections, json, sqlite3, sys
short = lambda name: name.split("(")[0].replace("void ", "")
agg = collections.defaultdict(dict)
for f in sys.argv[2:]:
    db = sqlite3.connect(f)
    for name, cn, cnt, s in db.execute("select kernel_name,counter_name,count(*),sum(value) from counters_collection group by kernel_name,counter_name"):
        agg[short(name)][cn] = s; agg[short(name)]["dispatches"] = cnt
for v in agg.values():
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v: v["hbm_bytes"] = v.get("FETCH_SIZE", 0) * 2048 + v.get("WRITE_SIZE", 0) * 1024
tot_wave = sum(v.get("SQ_WAVE_CYCLES", 0) for v in agg.values()) or 1
tot_valu = sum(v.get("SQ_ACTIVE_INST_VALU", 0) for v in agg.values()) or 1
for v in agg.values():
    v["share_of_wave_cycles"] = v.get("SQ_WAVE_CYCLES", 0) / tot_wave
    v["share_of_valu_active"] = v.get("SQ_ACTIVE_INST_VALU", 0) / tot_valu
out = {"command": "rocprofv3 --pmc <one pass each: FETCH_SIZE | WRITE_SIZE | SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT> -- python3 bench.py --no-extras --frames 75 --steps 1 --warmup 0",
       "note": "sums over all dispatches of ONE 75-frame call (counter collection serialises the kernels: the sums are per kernel, not a timeline); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU as reported (4-cycle units)",
       "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)))}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in list(out["kernels"].items())[:8]:
    print(f"  {k[:46]:46s} wave-cycles {100 * v['share_of_wave_cycles']:5.1f} %  VALU-active {100 * v['share_of_valu_active']:5.1f} %  HBM {v.get('hbm_bytes', 0) / 1e9:7.2f} GB  waves {v.get('SQ_WAVES', 0):.0f}")

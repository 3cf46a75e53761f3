#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of the bench command into the JSON files bench.py / DESIGN.md cite.
usage: pmc_summary.py <tag> [outdir]  (reads gpurun_out/pmc_{fetch,write,sq}/*_results.db, writes <outdir>/<tag>_pmc_{hbm,sq}.json;
outdir defaults to profiles)"""
import collections, json, sqlite3, sys
tag = sys.argv[1]
outdir = sys.argv[2] if len(sys.argv) > 2 else "profiles"
short = lambda name: name.split("(")[0].replace("void ", "")
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-extras (two separate passes)",
       "units": "KiB per dispatch as reported; hbm_bytes_per_dispatch = FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
       "kernels": {}}
for f, c in (("gpurun_out/pmc_fetch/fetch_results.db", "FETCH_SIZE"), ("gpurun_out/pmc_write/write_results.db", "WRITE_SIZE")):
    db = sqlite3.connect(f)
    for name, cnt, s, a in db.execute("select kernel_name,count(*),sum(value),avg(value) from counters_collection where counter_name=? group by kernel_name", (c,)):
        k = out["kernels"].setdefault(short(name), {})
        k["dispatches"] = cnt; k[c + "_KiB_avg"] = a; k[c + "_KiB_sum"] = s
for v in out["kernels"].values():
    v["hbm_bytes_per_dispatch"] = v.get("FETCH_SIZE_KiB_avg", 0) * 2048 + v.get("WRITE_SIZE_KiB_avg", 0) * 1024
json.dump(out, open(f"{outdir}/{tag}_pmc_hbm.json", "w"), indent=1)
db = sqlite3.connect("gpurun_out/pmc_sq/sq_results.db")
agg = collections.defaultdict(dict)
for name, cn, cnt, s in db.execute("select kernel_name,counter_name,count(*),sum(value) from counters_collection group by kernel_name,counter_name"):
    agg[short(name)][cn] = s; agg[short(name)]["dispatches"] = cnt
for v in agg.values():
    v["valu_f64_flop"] = 64 * (v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0))
sq = {"command": "rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --no-extras",
      "note": "sums over all dispatches of one bench step (per-wave instruction counts; valu_f64_flop = 64 lanes x (ADD + MUL + 2 FMA + TRANS)); MFMA_F64 are v_mfma_f64_4x4x4 used as exact adders (16 chains x 4 adds per instruction)",
      "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1]["valu_f64_flop"]))}
json.dump(sq, open(f"{outdir}/{tag}_pmc_sq.json", "w"), indent=1)
tot = sum(v["valu_f64_flop"] for v in agg.values())
print("HBM MB/dispatch:", {k: round(v["hbm_bytes_per_dispatch"] / 1e6, 2) for k, v in out["kernels"].items() if "lms" in k})
print("total VALU F64 GFLOP/step", tot / 1e9, " MFMA f64 instr (M)", sum(v.get("SQ_INSTS_VALU_MFMA_F64", 0) for v in agg.values()) / 1e6)
for k, v in list(sq["kernels"].items())[:6]:
    print(f"  {k[:44]:44s} bank_conflict/idx_active = {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}")

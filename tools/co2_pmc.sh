#!/bin/bash
# SQ counters of the lead (1280,640) kernel at one and at two workgroups per CU (tools/co2.py with 256 / 512 chains), one rocprofv3 pass each.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; export CO2_HOST=1
O=gpurun_out/co2_pmc; mkdir -p $O
for N in 256 512; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $O/p$N -o p -- python3 tools/co2.py $N > $O/run$N.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $O/q$N -o q -- python3 tools/co2.py $N > $O/runq$N.log 2>&1 || exit 1
done
python3 - <<'PY'
import sqlite3, glob
for N in (256, 512):
    row = {}
    for f in glob.glob(f"gpurun_out/co2_pmc/p{N}/*_results.db") + glob.glob(f"gpurun_out/co2_pmc/q{N}/*_results.db"):
        db = sqlite3.connect(f)
        for cn, cnt, s in db.execute("select counter_name,count(*),sum(value) from counters_collection where kernel_name like '%k_lms2%' group by counter_name"):
            row[cn] = (s, cnt)
    print(f"N={N}:")
    for k, (s, cnt) in sorted(row.items()): print(f"   {k:28s} {s:16.0f}  ({cnt} dispatches)")
PY
rm -rf $O/p256 $O/p512 $O/q256 $O/q512

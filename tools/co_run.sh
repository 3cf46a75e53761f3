set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_co
mkdir -p $O
A=tools/_ablate/libsacopt.so; B=tools/_nt512/libsacopt.so
PMC="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
SACOPT_LIB=$A SACOPT_DBG_ALL=512 timeout -k 10 120 python3 -u tools/coresident.py 56 112 224 >> $O/timing.log 2>&1
SACOPT_LIB=$B SACOPT_DBG_ALL=512 timeout -k 10 120 python3 -u tools/coresident.py 56 112 224 >> $O/timing.log 2>&1
SACOPT_LIB=$A SACOPT_DBG_ALL=256 timeout -k 10 120 python3 -u tools/coresident.py 112 224 448 >> $O/timing.log 2>&1
cat $O/timing.log
export TMPDIR=/tmp
run_pmc() { # name lib dbg N
  SACOPT_LIB=$2 SACOPT_DBG_ALL=$3 timeout -s KILL 120 rocprofv3 --pmc $PMC -d $O/$1 -o sq -- python3 tools/coresident.py $4 > $O/$1.log 2>&1
  python3 tools/sq_summary.py $O/$1/sq_results.db lms1c > $O/$1.txt; python3 tools/sq_summary.py $O/$1/sq_results.db 1280 >> $O/$1.txt
}
run_pmc a112 $A 512 112
run_pmc a224 $A 512 224
run_pmc b112 $B 512 112
run_pmc b224 $B 512 224
run_pmc c224 $A 256 224
run_pmc c448 $A 256 448
rm -rf $O/*/ ; tail -n 12 $O/*.txt

#!/bin/bash
# Profiles of one round, on the GPU box: rocprofv3 kernel stats of the bench command, HBM (FETCH_SIZE / WRITE_SIZE, separate
# passes) and SQ counter passes, summarised into gpurun_out/prof_<tag>/ (copy what is cited into profiles/). usage: profile_round.sh <tag>
set -e
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/${TAG}_bench_nocpu.json 2> $O/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python3 bench.py --no-extras --steps 3 --warmup 1 > $O/${TAG}_bench_under_rocprof.json 2> $O/kt.log
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --steps 3 --warmup 1   (4 steps of the bench workload incl. warmup, nothing else; summarised by tools/prof_summary.py)" > $O/${TAG}_kernel_stats.txt
python3 tools/prof_summary.py gpurun_out/kt/kt_results.db >> $O/${TAG}_kernel_stats.txt
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2> $O/pmc_f.log
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o write -- python3 bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2> $O/pmc_w.log
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_sq -o sq -- python3 bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2> $O/pmc_s.log
python3 tools/pmc_summary.py $TAG $O > $O/pmc_summary.txt
# the full-chip regime (what every rank runs at N > 1): one 75-frame call per step
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt75 -o kt -- python3 bench.py --no-extras --frames 75 --steps 2 --warmup 1 > $O/${TAG}_bench_75frames_under_rocprof.json 2> $O/kt75.log
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --frames 75 --steps 2 --warmup 1   (3 calls of 75 frames x 176400; summarised by tools/prof_summary.py)" > $O/${TAG}_kernel_stats_75frames.txt
python3 tools/prof_summary.py gpurun_out/kt75/kt_results.db >> $O/${TAG}_kernel_stats_75frames.txt
rm -rf gpurun_out/kt75
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/p75f -o p -- python3 bench.py --no-extras --frames 75 --steps 1 --warmup 0 > /dev/null 2> $O/pmc75_f.log
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/p75w -o p -- python3 bench.py --no-extras --frames 75 --steps 1 --warmup 0 > /dev/null 2> $O/pmc75_w.log
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d gpurun_out/p75s -o p -- python3 bench.py --no-extras --frames 75 --steps 1 --warmup 0 > /dev/null 2> $O/pmc75_s.log
python3 tools/pmc75_summary.py $O/${TAG}_pmc_75frames.json gpurun_out/p75f/p_results.db gpurun_out/p75w/p_results.db gpurun_out/p75s/p_results.db > $O/pmc75_summary.txt
rm -rf gpurun_out/p75f gpurun_out/p75w gpurun_out/p75s
rm -rf gpurun_out/kt gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
head -12 $O/${TAG}_kernel_stats.txt | cut -c1-150; cat $O/pmc_summary.txt; head -10 $O/${TAG}_kernel_stats_75frames.txt | cut -c1-150; cat $O/pmc75_summary.txt

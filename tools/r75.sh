#!/bin/bash
# 75-frame call: time, kernel trace, per-round spans. usage: r75.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/r75_$1; mkdir -p $O
timeout -k 10 200 python3 bench.py --no-extras --frames 75 --steps 2 --warmup 1 > $O/bench75.json 2> $O/err.log || exit 1
timeout -k 10 200 python3 bench.py --no-extras --steps 3 --warmup 1 > $O/bench15.json 2>> $O/err.log || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 bench.py --no-extras --frames 75 --steps 1 --warmup 1 > /dev/null 2> $O/kt.log || exit 1
python3 tools/rounds_summary.py $O/kt/kt_results.db > $O/rounds.txt
python3 tools/prof_summary.py $O/kt/kt_results.db > $O/stats.txt
rm -rf $O/kt
python3 -c "
import json
for f in ('bench75','bench15'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['ms_per_step'],1),'ms', round(d['value']/1e6,3),'M samples/s')
"
cat $O/rounds.txt; head -8 $O/stats.txt | cut -c1-150

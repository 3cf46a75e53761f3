#!/bin/bash
# idle gaps of the GPU inside a 75-frame call (host time between rounds shows up here). usage: r75gaps.sh [frames]
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/r75gaps; mkdir -p $O; NF=${1:-75}
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 bench.py --no-extras --frames $NF --steps 1 --warmup 1 > $O/bench.json 2> $O/kt.log || exit 1
python3 tools/idle_gaps.py $O/kt/kt_results.db
python3 tools/rounds_summary.py $O/kt/kt_results.db
rm -rf $O/kt

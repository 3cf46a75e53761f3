#!/bin/bash
# Does the library's number of CU-masked streams (hardware queues) leave headroom? The 15-frame step plain, under rocprofv3 (its
# own queue), and with 1..4 extra CU-masked streams created by the caller first (tools/queue_cliff.py).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
python3 tools/queue_cliff.py 0 2 4 5 6 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace -d gpurun_out/qc -o qc -- python3 bench.py --no-extras --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('under rocprofv3 --kernel-trace:', round(d['ms_per_step'],1), 'ms')"
rm -rf gpurun_out/qc

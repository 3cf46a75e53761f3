#!/bin/bash
# Same-run A/B of library variants on the whole call: tools/ab_bench.sh <variant dirs under tools/>...
# per variant: bench.py timed steps only (60 s clip), then the 75-frame call and the decoder (tools/unp_bench.py)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for v in "$@"; do
  echo -n "$v: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 200 python3 -u bench.py --no-extras --steps 4 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))" || exit 1
done; done
for v in "$@"; do
  echo -n "$v 75 frames: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 200 python3 -u bench.py --no-extras --frames 75 --steps 2 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))" || exit 1
  echo "$v decode: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 200 python3 -u tools/unp_bench.py 15 2>&1 | tail -3 || exit 1
done

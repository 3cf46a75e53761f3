#!/bin/bash
# Same-run A/B of library variants where the two-channel (1280,640) kernel matters: optimize-only on the 60 s clip, the bench's
# timed steps, the 75-frame call. usage: ab_opt.sh <variant dirs under tools/>...
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for v in "$@"; do
  echo -n "$v: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 120 python3 tools/opt_only.py 2>&1 | tail -1 | tr "\n" " " || exit 1
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 200 python3 -u bench.py --no-extras --steps 3 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', round(d['ms_per_step'],1))" || exit 1
done; done
for i in 1 2; do for v in "$@"; do
  echo -n "$v 75 frames: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 200 python3 -u bench.py --no-extras --frames 75 --steps 2 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))" || exit 1
done; done

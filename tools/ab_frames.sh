#!/bin/bash
# Same-run A/B of library variants on calls of a few frames: tools/ab_frames.sh "<frame counts>" <variant dirs under tools/>...
cd ${GRAFT_REPO_ROOT:-.}
FR=$1; shift
for fr in $FR; do for v in "$@"; do
  echo -n "frames $fr $v: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 250 python3 -u bench.py --no-extras --frames $fr --steps 3 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))" || exit 1
done; done

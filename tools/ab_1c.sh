#!/bin/bash
# Same-run A/B: the (1280,640) stage's kernel choice forced to k_lms1c for every launch (ablate build, switch 512) against the
# policy (k_lms1c for <= 20 chains), on the 60 s clip and the 75-frame track.
cd ${GRAFT_REPO_ROOT:-.}
export SACOPT_LIB=tools/_ablate/libsacopt.so
for fr in 15 75; do for sw in 0 512; do
  echo -n "frames $fr switch $sw: "
  SACOPT_DBG_ALL=$sw timeout -k 10 250 python3 -u bench.py --no-extras --frames $fr --steps 2 --warmup 1 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))" || exit 1
done; done

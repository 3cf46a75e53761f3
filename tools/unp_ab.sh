#!/bin/bash
# decoder timing of several library builds in one call: unp_ab.sh <lib>...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for lib in "$@"; do
  echo -n "$lib: "; SACOPT_LIB=$lib timeout -k 10 120 python3 tools/unp_bench.py 15 176400 2>&1 | grep "profile 2" | sed 's/.*kernel/kernel/' || exit 1
done; done

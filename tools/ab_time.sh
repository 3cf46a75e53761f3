#!/bin/bash
# Same-run A/B of library variants (boxes differ by 2-3 %): tools/ab_time.sh <variant dirs under tools/>...
# per variant and repetition: k_lms1c cycles per sample (tools/l1_time.py) and the two-channel kernel (tools/coresident.py, 224 chains)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for v in "$@"; do
  echo -n "$v: "
  SACOPT_LIB=tools/$v/libsacopt.so timeout -k 10 120 python3 -u tools/l1_time.py 2>&1 | grep -E "k_lms1c" | sed "s/cycles.*//; s/seed [0-9]*: k_lms1c//" | tr "\n" " "
  echo
done; done

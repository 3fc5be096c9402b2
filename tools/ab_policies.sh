#!/bin/bash
# tools/ab_policies.sh "POLICY IDS" LIB... — tools/time_policies.py (B = 65536, 100 steps on the loaded network) for several builds
ids=$1; shift
for lib in "$@"; do
  echo "== $(basename $lib)"
  ONGYM_HIP_LIB=$PWD/$lib python3 tools/time_policies.py 65536 100 $ids 2>&1 | grep policy
done

#!/bin/bash
# tools/build_p0_variant.sh NAME "FLAGS" — experiment build that recompiles ONLY the narrow first-fit unit (ongym_fast.hip, policy 0 or $POLICY, WIDE = 0) with
# FLAGS and links it with the other objects of csrc/build (run `python __graft_entry__.py` first): csrc/variants/lib_NAME.so
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$REPO/optical-networking-gym_amd/csrc
mkdir -p $CSRC/variants $CSRC/build/p0v
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DONGYM_FAST_POLICY=${POLICY:-0} -DONGYM_FAST_WIDE=0 $2 -c -o $CSRC/build/p0v/$1.o $CSRC/ongym_fast.hip
objs=""
for o in ongym_hip.o ongym_fast_p0.o ongym_fast_p1.o ongym_fast_p2.o ongym_fast_p10.o ongym_fast_p0w.o ongym_fast_p1w.o ongym_fast_p2w.o ongym_fast_p10w.o; do
  if [ "$o" == "ongym_fast_p${POLICY:-0}.o" ]; then objs="$objs $CSRC/build/p0v/$1.o"; else objs="$objs $CSRC/build/$o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CSRC/variants/lib_$1.so $objs

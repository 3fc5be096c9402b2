#!/bin/bash
# tools/build_variants.sh NAME[:FLAGS] ... — experiment builds of libongym_hip.so under csrc/variants/ (for tools/ab_bench.py).
# NAME "head" builds the committed sources (git HEAD) instead of the working tree.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$REPO/optical-networking-gym_amd/csrc
mkdir -p $CSRC/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=""; [[ "$spec" == *:* ]] && flags=${spec#*:}
  if [ "$name" == "head" ]; then
    T=$(mktemp -d); mkdir -p $T/optical-networking-gym_amd/csrc $T/include
    for f in ongym_hip.hip $(cd $CSRC && ls *.hpp); do git -C $REPO show HEAD:optical-networking-gym_amd/csrc/$f > $T/optical-networking-gym_amd/csrc/$f; done
    for f in ongym.h ongym_traffic.h; do git -C $REPO show HEAD:include/$f > $T/include/$f; done
    SRC=$T/optical-networking-gym_amd/csrc/ongym_hip.hip
  else
    SRC=$CSRC/ongym_hip.hip
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w $flags -o $CSRC/variants/lib_$name.so $SRC &
done
wait
ls -la $CSRC/variants/

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
    for f in $(git -C $REPO ls-tree --name-only HEAD optical-networking-gym_amd/csrc/ | xargs -n1 basename | grep -E '\.(hip|hpp)$'); do git -C $REPO show HEAD:optical-networking-gym_amd/csrc/$f > $T/optical-networking-gym_amd/csrc/$f; done
    for f in ongym.h ongym_traffic.h; do git -C $REPO show HEAD:include/$f > $T/include/$f; done
    # (round-2 sources are one translation unit; later ones carry their own build recipe)
    if git -C $REPO cat-file -e HEAD:optical-networking-gym_amd/csrc/ongym_fast.hip 2>/dev/null; then
      git -C $REPO show HEAD:__graft_entry__.py > $T/__graft_entry__.py
      (cd $T && python3 __graft_entry__.py --variant head -w $flags && cp $T/optical-networking-gym_amd/csrc/variants/lib_head.so $CSRC/variants/) &
    else
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w $flags -o $CSRC/variants/lib_$name.so $T/optical-networking-gym_amd/csrc/ongym_hip.hip &
    fi
  else
    (cd $REPO && python3 __graft_entry__.py --variant $name -w $flags) &
  fi
done
wait
ls -la $CSRC/variants/

#!/bin/bash
# tools/resource_usage.sh [policy ids...] — registers, spills, scratch and LDS of every k_fast instantiation (compile only, no GPU)
REPO=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$REPO/optical-networking-gym_amd/csrc
for p in "${@:-0 1 2 10}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DONGYM_FAST_POLICY=$p -DONGYM_FAST_WIDE=${WIDE:-0} ${ONGYM_HIP_EXTRA_FLAGS:--mllvm -disable-machine-licm} -c -o /dev/null \
    -Rpass-analysis=kernel-resource-usage $CSRC/ongym_fast.hip 2>&1 |
  python3 -c '
import re, sys
name = None; row = {}
for l in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", l)
    if m:
        if name: print(name, row)
        name = m.group(1); row = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", l)
    if m: row[m.group(1).split()[0] + ("Spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
if name: print(name, row)
' | sed -e 's/_ZN5ongym6k_fastI//' -e 's/EEEvPKNS_6ParamsEiP14ongym_step_rec//' -e 's/Lb1/T/g' -e 's/Lb0/F/g' -e 's/Li//g' -e 's/E/ /g' | while read -r line; do echo "p$p $line"; done
done

#!/bin/bash
# tools/profile_bench.sh TAG [bench.py args...] — rocprofv3 evidence for one bench.py workload (run on the GPU box):
#   gpurun_out/prof_TAG/bench.json        the un-profiled bench line
#   gpurun_out/prof_TAG/kt/               --kernel-trace --stats
#   gpurun_out/prof_TAG/pmc_*/            one --pmc pass per counter group (never combined with tracing)
# then: python tools/pmc_summarise.py TAG  (here or in the build container) writes profiles/TAG_*.csv/json
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o p -- python3 bench.py --no-cpu-baseline --steps 8 "$@" > $OUT/pmc_$i.log 2>&1 || echo "pmc group $i failed"
  echo "pmc $i done"
done

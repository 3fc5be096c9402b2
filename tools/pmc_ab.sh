#!/bin/bash
# tools/pmc_ab.sh LIB... — instruction-count PMC groups for several builds of the library (same box, back to back)
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  OUT=gpurun_out/pmcab_$name
  mkdir -p $OUT
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INSTS_FLAT"; do
    i=$((i+1))
    ONGYM_HIP_LIB=$PWD/$lib rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o p -- python3 bench.py --no-cpu-baseline --steps 8 > $OUT/pmc_$i.log 2>&1 || echo "$name group $i failed"
  done
  python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(out+'/pmc_*/**/*counter_collection.csv',recursive=True):
    per=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'k_fast' not in r['Kernel_Name'] and 'k_run' not in r['Kernel_Name']: continue
        per[r['Counter_Name']][int(r['Dispatch_Id'])]=float(r['Counter_Value'])
    for c,d in per.items():
        ids=sorted(d); keep=ids[len(ids)//2:]
        tot[c]+=sum(d[i] for i in keep); n[c]+=len(keep)
print(out, {c: round(tot[c]/n[c]/16384000,2) for c in sorted(tot)})
PY
done

#!/bin/bash
# tools/profile_observe.sh TAG — rocprofv3 evidence for k_observe (tools/time_observe.py at B = 16384)
set -e
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/time_observe.py 16384 > $OUT/time.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 tools/time_observe.py 16384 > $OUT/kt.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/time_observe.py 16384 > $OUT/pmc_$i.log 2>&1 || echo "pmc group $i failed"
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(out+'/pmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_observe' not in r['Kernel_Name']: continue
        tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
with open(out+'/pmc_summary.csv','w') as f:
    f.write('# k_observe at B=16384 (NSFNET-320 after 600 first-fit steps), rocprofv3 --pmc separate passes; per launch and per observation\ncounter,per_launch,per_observation\n')
    for c in sorted(tot): f.write(f"{c},{tot[c]/n[c]:.6g},{tot[c]/n[c]/16384:.6g}\n")
print(open(out+'/pmc_summary.csv').read()); print(open(out+'/time.log').read())
PY

set -e
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -x -q > gpurun_out/r2d/tests.log 2>&1 || { tail -40 gpurun_out/r2d/tests.log; exit 1; }
tail -2 gpurun_out/r2d/tests.log
bash tools/profile_bench.sh r02b_nsfnet320 > gpurun_out/r2d/prof1.log 2>&1
bash tools/profile_bench.sh r02b_cost239_320 --workload cost239_320 --batch 16384 > gpurun_out/r2d/prof2.log 2>&1
bash tools/profile_bench.sh r02b_nobeleu768 --workload nobeleu768 > gpurun_out/r2d/prof3.log 2>&1
python bench.py --workload nsfnet320 --batch 4096 --no-cpu-baseline > gpurun_out/r2d/bench_nsfnet_4k.json 2> /dev/null
for f in gpurun_out/prof_r02b_*/bench.json gpurun_out/r2d/bench_nsfnet_4k.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.4e'%d['value'], d['occupancy'], 'launch ms %.3f'%d['roofline']['avg_launch_ms'])"; done

set -e
mkdir -p gpurun_out/r2i
python -m pytest tests -m gpu -x -q > gpurun_out/r2i/tests.log 2>&1 || { tail -60 gpurun_out/r2i/tests.log; exit 1; }
tail -2 gpurun_out/r2i/tests.log
bash tools/profile_observe.sh r02c_observe > gpurun_out/r2i/prof_obs.log 2>&1; tail -3 gpurun_out/r2i/prof_obs.log
python bench.py --workload cost239_320 --batch 16384 --no-cpu-baseline > gpurun_out/r2i/bench_cost239.json 2>/dev/null
python bench.py --workload cost239_320 --batch 65536 --no-cpu-baseline > gpurun_out/r2i/bench_cost239_65k.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r2i/bench_nsfnet.json 2>/dev/null
for f in bench_cost239 bench_cost239_65k bench_nsfnet; do python -c "
import json
d=json.loads(open('gpurun_out/r2i/$f.json').read().strip().splitlines()[-1]); print('$f', '%.4e'%d['value'], d['occupancy'], 'launch ms %.3f'%d['roofline']['avg_launch_ms'], d.get('cpu_baseline',{}).get('value'))"; done

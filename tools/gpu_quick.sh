# quick loop: parity subset, then bench variants given as env-var prefixes in $VARIANTS (";"-separated)
set -e
OUT=gpurun_out/quick
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "random_traffic or sharded or bench_size" > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick/bench.json').read().strip().splitlines()[-1])
print('default: %.4e env-steps/s  launch %.3f ms'%(d['value'], d['roofline']['avg_launch_ms']))
PY
ONGYM_FAST_WAVES=4 python bench.py --no-cpu-baseline "$@" > $OUT/bench4.json 2>> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick/bench4.json').read().strip().splitlines()[-1])
print('4 waves: %.4e env-steps/s  launch %.3f ms'%(d['value'], d['roofline']['avg_launch_ms']))
PY

set -e
mkdir -p gpurun_out/r2e
python -m pytest tests -m gpu -x -q > gpurun_out/r2e/tests.log 2>&1 || { tail -60 gpurun_out/r2e/tests.log; exit 1; }
tail -2 gpurun_out/r2e/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2e/bench1.json 2> gpurun_out/r2e/bench1.err
ONGYM_BENCH_REHEARSE=1 python bench.py --gpus 2 --scaling strong --batch 65536 --no-cpu-baseline --steps 8 > gpurun_out/r2e/bench2.json 2> gpurun_out/r2e/bench2.err || { tail -20 gpurun_out/r2e/bench2.err; echo REHEARSAL FAILED; }
python - <<'PY'
import json
for f in ("bench1","bench2"):
    try:
        d=json.loads(open(f"gpurun_out/r2e/{f}.json").read().strip().splitlines()[-1])
        print(f, "%.4e"%d["value"], d["n_gpus"], d["scaling"], d["config"]["batch_per_gpu"], d["config"]["global_batch"], d.get("rehearsal","")[:20], "block %.5f"%d["blocking_rate"])
    except Exception as e: print(f, "ERR", e)
PY

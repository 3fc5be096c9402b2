set -e
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q -k "random_traffic or sharded or bench_size or full_batch or long_wide" > gpurun_out/r2b/tests.log 2>&1 || { tail -40 gpurun_out/r2b/tests.log; exit 1; }
tail -3 gpurun_out/r2b/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2b/bench_nsfnet.json 2> gpurun_out/r2b/bench.err
cat gpurun_out/r2b/bench_nsfnet.json

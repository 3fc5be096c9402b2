set -e
mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1 || { tail -40 gpurun_out/r2c/tests.log; exit 1; }
tail -3 gpurun_out/r2c/tests.log
bash tools/profile_bench.sh r02a_nsfnet320 > gpurun_out/r2c/prof.log 2>&1
tail -3 gpurun_out/r2c/prof.log

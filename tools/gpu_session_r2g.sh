set -e
mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -x -q > gpurun_out/r2g/tests.log 2>&1 || { tail -60 gpurun_out/r2g/tests.log; exit 1; }
tail -2 gpurun_out/r2g/tests.log
python tools/time_observe.py 16384 2>&1 | tail -2

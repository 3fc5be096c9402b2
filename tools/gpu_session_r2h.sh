set -e
mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -x -q -k "observation or modulations_to_consider or compat or highest or other_fused or vec_env or masked" > gpurun_out/r2h/tests.log 2>&1 || { tail -60 gpurun_out/r2h/tests.log; exit 1; }
tail -2 gpurun_out/r2h/tests.log
python tools/time_observe.py 16384 2>&1 | tail -2

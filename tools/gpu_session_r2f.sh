set -e
mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -x -q -k "modulations_to_consider or observation or compat" > gpurun_out/r2f/tests.log 2>&1 || { tail -60 gpurun_out/r2f/tests.log; exit 1; }
tail -2 gpurun_out/r2f/tests.log
bash tools/profile_observe.sh r02a_observe > gpurun_out/r2f/prof.log 2>&1; tail -30 gpurun_out/r2f/prof.log

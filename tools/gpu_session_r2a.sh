set -e
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q -k "sharded or continuous_bit or bench_size or random_traffic" > gpurun_out/r2a/tests.log 2>&1 || { tail -30 gpurun_out/r2a/tests.log; exit 1; }
tail -3 gpurun_out/r2a/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a/bench_nsfnet.json 2> gpurun_out/r2a/bench_nsfnet.err
cat gpurun_out/r2a/bench_nsfnet.json
python bench.py --workload cost239_320 --batch 16384 --no-cpu-baseline > gpurun_out/r2a/bench_cost239_16k.json 2>> gpurun_out/r2a/bench.err
python bench.py --workload nobeleu768 --no-cpu-baseline > gpurun_out/r2a/bench_nobeleu.json 2>> gpurun_out/r2a/bench.err
python bench.py --workload nsfnet320 --batch 4096 --no-cpu-baseline > gpurun_out/r2a/bench_nsfnet_4k.json 2>> gpurun_out/r2a/bench.err
python bench.py --gpus 2 --scaling strong --batch 8192 --no-cpu-baseline --steps 4 > gpurun_out/r2a/bench_spawn2.json 2> gpurun_out/r2a/bench_spawn2.err || echo "spawn2 failed (expected on 1 GPU?)"
python tools/diag_stamps.py > gpurun_out/r2a/stamps_nsfnet.log 2>&1
python tools/diag_stamps.py --workload nobeleu768 > gpurun_out/r2a/stamps_nobeleu.log 2>&1
cat gpurun_out/r2a/stamps_nsfnet.log

set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/e10_tests.log 2>&1 && tail -3 gpurun_out/e10_tests.log &&
for c in 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e10_c$c.json 2> gpurun_out/e10_c$c.err &&
  SBM_NO_TILE_64x128=1 timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e10_c${c}_old.json 2>> gpurun_out/e10_c$c.err
done
tail -2 gpurun_out/e10_c*.json

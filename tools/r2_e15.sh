set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_alt.py tests/test_ddpm_refiner.py -x -q -m gpu > gpurun_out/e15_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e15_tests.log; [ $rc -eq 0 ] || exit $rc
for c in 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e15_c$c.json 2> gpurun_out/e15_c$c.err || exit 1
done
SBM_GN32_ONE_PASS=1 timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/e15_c5_onepass.json 2>> gpurun_out/e15_c5.err || exit 1
SBM_NAME_TILES=1 timeout -k 10 300 python tools/profile_ops.py --config 5 > gpurun_out/e15_ops_c5.txt 2>&1
